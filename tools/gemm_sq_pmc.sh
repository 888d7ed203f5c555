#!/bin/bash
# SQ counters of the GEMM kernel per shape (MFMA busy, LDS conflicts, wait fractions): rocprofv3 --pmc, kernel trace only, one pass per group
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp
python - <<PYEOF
import json; json.dump({}, open("/tmp/gsq_acc.json", "w"))
PYEOF
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  rm -rf /tmp/gsq
  SHAPES=${SH:-probe} timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/gsq -o p -- python $R/tools/gemm_shapes_bench.py > /tmp/gsq.log 2>&1
  python - <<PYEOF
import csv, glob, collections, json
f = glob.glob("/tmp/gsq/*counter_collection.csv")
acc = json.load(open("/tmp/gsq_acc.json"))
if not f:
    print("no output for $C:", open("/tmp/gsq.log").read()[-300:])
else:
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        d = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": {}})
        d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    groups, cur = [], []
    for i in sorted(per):
        nm = per[i]["name"]
        if "distribution" in nm or "normal" in nm:
            if cur: groups.append(cur); cur = []
        elif "gemm_dma_kernel" in nm or "gemm_f32_kernel" in nm:
            cur.append(per[i])
    if cur: groups.append(cur)
    shapes = [l.split() for l in open("/tmp/gsq.log") if " tA" in l]
    for s, grp in zip(shapes, groups):
        key = " ".join(s[:5]); grp = grp[3:]
        for c in grp[0]["c"]:
            acc.setdefault(key, {})[c] = sum(g["c"][c] for g in grp) / len(grp)
        acc[key]["kernel"] = grp[-1]["name"][25:60]
json.dump(acc, open("/tmp/gsq_acc.json", "w"))
PYEOF
done
python - <<PYEOF
import json
acc = json.load(open("/tmp/gsq_acc.json"))
for k, m in acc.items():
    o = {"kernel": m.get("kernel")}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: o["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * m["GRBM_GUI_ACTIVE"]), 3)
    if m.get("SQ_LDS_IDX_ACTIVE"): o["lds_conflict"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3); o["lds_active_of_wave"] = round(m["SQ_LDS_IDX_ACTIVE"] / m["SQ_WAVE_CYCLES"], 3)
    if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in m: o["wait_inst_any"] = round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3); o["wait_any"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3); o["wait_lds"] = round(m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"], 3)
    if m.get("SQ_INSTS_MFMA"): o["valu_per_mfma"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 2); o["lds_per_mfma"] = round(m["SQ_INSTS_LDS"] / m["SQ_INSTS_MFMA"], 3); o["salu_per_mfma"] = round(m["SQ_INSTS_SALU"] / m["SQ_INSTS_MFMA"], 2)
    print(k, o)
json.dump(acc, open("$R/gpurun_out/r2_gemm_sq_pmc.json", "w"), indent=1)
PYEOF
