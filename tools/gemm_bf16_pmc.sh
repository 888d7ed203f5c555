#!/bin/bash
# SQ / GRBM counters and kernel durations of the bf16 forward GEMM per main-loop form and for the vendor library: effective clock
# (GRBM_GUI_ACTIVE / duration), matrix-pipe busy fraction, wait fractions, LDS activity.  rocprofv3 --pmc, kernel trace only, one pass per group.
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp
echo '{}' > /tmp/bpmc_acc.json
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU"; do
  rm -rf /tmp/bpmc
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/bpmc -o p -- python $R/tools/gemm_bf16_pmc_driver.py > /tmp/bpmc.log 2>&1
  python - <<PYEOF
import csv, glob, collections, json
acc = json.load(open("/tmp/bpmc_acc.json"))
fc = glob.glob("/tmp/bpmc/**/*counter_collection.csv", recursive=True); ft = glob.glob("/tmp/bpmc/**/*kernel_trace.csv", recursive=True)
if not fc:
    print("no output for $C:", open("/tmp/bpmc.log").read()[-400:])
else:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fc[0])):
        n = r["Kernel_Name"]
        if "gemm_bf16" in n or "Cijk" in n or "gemm" in n.lower():
            per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, cs in per.items():
        for c, v in cs.items():
            v = v[2:] if len(v) > 4 else v
            acc.setdefault(n, {})[c] = sum(v) / len(v)
    if ft:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(ft[0])):
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for n, v in dur.items():
            if n in acc:
                v = v[2:] if len(v) > 4 else v
                acc[n].setdefault("ns", []).append(sum(v) / len(v))
json.dump(acc, open("/tmp/bpmc_acc.json", "w"))
PYEOF
done
python - <<PYEOF
import json
acc = json.load(open("/tmp/bpmc_acc.json"))
for n, m in acc.items():
    o = {"kernel": n[:70]}
    ns = m.get("ns", [0])
    o["us_by_pass"] = [round(x / 1e3, 1) for x in ns]
    if "GRBM_GUI_ACTIVE" in m and ns[0]: o["clock_GHz"] = round(m["GRBM_GUI_ACTIVE"] / ns[0], 3); o["gui_cycles"] = int(m["GRBM_GUI_ACTIVE"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: o["mfma_busy_of_gui"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"]), 3)
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]; o["wave_cycles"] = int(w)
        for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if k in m: o[k.lower()] = round(m[k] / w, 3)
    if m.get("SQ_INSTS_MFMA"):
        o["lds_per_mfma"] = round(m["SQ_INSTS_LDS"] / m["SQ_INSTS_MFMA"], 3); o["valu_per_mfma"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 2); o["salu_per_mfma"] = round(m["SQ_INSTS_SALU"] / m["SQ_INSTS_MFMA"], 2)
        o["lds_conflict"] = round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 3); o["insts_mfma"] = int(m["SQ_INSTS_MFMA"])
    print(o)
json.dump(acc, open("$R/gpurun_out/r6_gemm_bf16_pmc.json", "w"), indent=1)
PYEOF
