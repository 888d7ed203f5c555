#!/bin/bash
# Hardware counters of the attention kernels at the cfg-2 shapes (tools/attn_bench.py), one rocprofv3 --pmc pass per counter group
# (kernel trace only).  usage: tools/attn_pmc.sh <tag> [CASES]   -> gpurun_out/<tag>_attn_pmc.json
export TMPDIR=/tmp; R=$PWD; TAG=$1; CASES_=${2:-img}; mkdir -p $R/gpurun_out; cd /tmp
python - <<PYEOF
import json; json.dump({}, open("/tmp/attn_pmc_acc.json", "w"))
PYEOF
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  rm -rf /tmp/apmc
  CASES=$CASES_ timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/apmc -o p -- python $R/tools/attn_bench.py > /tmp/apmc.log 2>&1
  python - <<PYEOF
import csv, glob, collections, json, re
f = glob.glob("/tmp/apmc/*counter_collection.csv")
acc = json.load(open("/tmp/attn_pmc_acc.json"))
if not f:
    print("no output for $C:", open("/tmp/apmc.log").read()[-300:])
else:
    tmp = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"(attn_\w+)_kernel<(\d+)", r["Kernel_Name"])
        if m and "delta" not in m.group(1):
            tmp[m.group(1) + "<" + m.group(2) + "> grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in tmp.items():
        acc.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
        acc[k]["launches"] = len(next(iter(d.values())))
json.dump(acc, open("/tmp/attn_pmc_acc.json", "w"))
PYEOF
done
python - <<PYEOF
import json
acc = json.load(open("/tmp/attn_pmc_acc.json"))
for k, m in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # same normalisation as profiles/round1_pmc_summary.json: 128 x GRBM_GUI_ACTIVE = every matrix core busy for the whole launch
        m["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * m["GRBM_GUI_ACTIVE"])
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_INST_ANY" in m and m.get("SQ_WAVE_CYCLES"):
        m["wait_inst_any_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]; m["wait_any_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in m and m.get("SQ_INSTS_MFMA"):
        m["valu_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"]
json.dump(acc, open("$R/gpurun_out/${TAG}_attn_pmc.json", "w"), indent=1)
for k, m in acc.items():
    print(k, {c: (round(v, 4) if isinstance(v, float) and v < 100 else v) for c, v in m.items() if c.endswith("frac") or c in ("valu_per_mfma", "launches")})
PYEOF
