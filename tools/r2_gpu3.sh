#!/bin/bash
# round-2 GPU pass 3: where does the attention forward's time go?  timing probes + hardware counters
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
{
for P in 0 1 2 3 4 7 8 16 24 32 35 39 31; do
  echo "== PROBE=$P"; YTVLN_LIB=scratch/lib_probe.so YTVLN_ATTN_PROBE=$P CASES=img FWD_ONLY=1 timeout 120 python tools/attn_bench.py 2>&1 | grep "img self"
done
for W in 1 4; do
for P in 0 3 7 39; do
  echo "== WAVES=$W PROBE=$P"; YTVLN_ATTN_WAVES=$W YTVLN_LIB=scratch/lib_probe.so YTVLN_ATTN_PROBE=$P CASES=img FWD_ONLY=1 timeout 120 python tools/attn_bench.py 2>&1 | grep "img self"
done; done
echo "== co pair (remap inside each problem)"; CASES=co timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_attn_probe.log 2>&1
cat gpurun_out/r2_attn_probe.log
cd /tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/apmc
  CASES=img FWD_ONLY=1 timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/apmc -o p -- python $R/tools/attn_bench.py > /tmp/apmc.log 2>&1
  python - <<PYEOF
import csv, glob, collections, json
f = glob.glob("/tmp/apmc/*counter_collection.csv")
if not f:
    print("no output for $C:", open("/tmp/apmc.log").read()[-300:])
else:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "attn_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(json.dumps({c: sum(v) / len(v) for c, v in acc.items()}))
PYEOF
done > $R/gpurun_out/r2_attn_fwd_pmc.log 2>&1
cat $R/gpurun_out/r2_attn_fwd_pmc.log
