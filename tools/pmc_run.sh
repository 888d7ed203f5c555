#!/bin/bash
# tools/pmc_run.sh <tag> "<counter set 1>;<counter set 2>;..." <command...>
# One rocprofv3 --pmc pass per counter set (own passes, kernel trace only: the combination gpurun allows); per-kernel means -> gpurun_out/pmc_<tag>.json
tag=$1; sets=$2; shift 2
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp
rm -f /tmp/pmc_acc.jsonl
IFS=';' read -ra SETS <<< "$sets"
for C in "${SETS[@]}"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc -o p -- "$@" > /tmp/pmc.log 2>&1
  python - <<PYEOF
import csv,glob,collections,json
f=glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
if not f:
    print("no output:", open("/tmp/pmc.log").read()[-800:]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if "ytvln" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("/tmp/pmc_acc.jsonl","a") as o:
    for k,d in acc.items():
        o.write(json.dumps({"kernel":k, "counters":{c:{"launches":len(v),"mean":sum(v)/len(v)} for c,v in d.items()}})+"\n")
PYEOF
done
python - <<PYEOF
import json,collections
out=collections.defaultdict(dict)
for l in open("/tmp/pmc_acc.jsonl"):
    d=json.loads(l); out[d["kernel"]].update(d["counters"])
json.dump(out, open("$R/gpurun_out/pmc_$tag.json","w"), indent=1)
for k,cs in out.items():
    m={c:v["mean"] for c,v in cs.items()}
    line=k[-70:]
    if "SQ_WAVE_CYCLES" in m:
        wc=m["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_MISC","SQ_ACTIVE_INST_SCA"):
            if c in m: line+=f" {c[3:]}={m[c]/wc:.3f}"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m: line+=f" mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES']/(128.0*m['GRBM_GUI_ACTIVE']):.3f}"
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]>0: line+=f" lds_conflict_frac={m['SQ_LDS_BANK_CONFLICT']/m['SQ_LDS_IDX_ACTIVE']:.3f}"
    if "SQ_INSTS_VALU" in m and "SQ_INSTS_VALU_MFMA_MOPS_BF16" in m: pass
    print(line)
PYEOF
