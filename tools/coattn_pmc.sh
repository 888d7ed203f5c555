#!/bin/bash
# MFMA utilisation of the BertBiAttention forward kernel (fused QK^T / softmax / dropout / PV) at N=56, h=8, d=128, T=80, R=288.
# Counters in their own pass, kernel trace only (no hip/hsa tracing).  Output: gpurun_out/coattn_pmc.json
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp
rm -rf /tmp/copmc
CASES=co FWD_ONLY=1 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d /tmp/copmc -o p -- python $R/tools/attn_bench.py > /tmp/copmc.log 2>&1
tail -3 /tmp/copmc.log
python - <<PYEOF
import csv, glob, collections, json
f = glob.glob("/tmp/copmc/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "attn_fwd" in r["Kernel_Name"]:
        if "pair" in r["Kernel_Name"]:
            acc["pair_launch"][r["Counter_Name"]].append(float(r["Counter_Value"])); continue
        acc[r["Grid_Size"] if "Grid_Size" in r else "all"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for grid, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMD groups in units that make 128 x GRBM_GUI_ACTIVE = 100 % busy
    # (same normalisation as profiles/round1_pmc_summary.json; cross-checked there against the GEMM's flop rate)
    m["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * m["GRBM_GUI_ACTIVE"])
    m["launches"] = len(d["GRBM_GUI_ACTIVE"])
    out["grid_" + str(grid)] = m
print(json.dumps(out, indent=1))
open("$R/gpurun_out/coattn_pmc.json", "w").write(json.dumps(out, indent=1))
PYEOF
