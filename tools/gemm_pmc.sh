#!/bin/bash
# L2 / fabric traffic of the GEMM kernel per cfg-2 shape: rocprofv3 --pmc (kernel trace only), one pass per counter group, over
# tools/gemm_shapes_bench.py (23 launches per shape, in SETS order).  usage: tools/gemm_pmc.sh <tag>  -> gpurun_out/<tag>_gemm_pmc.json
export TMPDIR=/tmp; R=$PWD; TAG=$1; mkdir -p $R/gpurun_out; cd /tmp
python - <<PYEOF
import json; json.dump({}, open("/tmp/gemm_pmc_acc.json", "w"))
PYEOF
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rm -rf /tmp/gpmc
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/gpmc -o p -- python $R/tools/gemm_shapes_bench.py > /tmp/gpmc.log 2>&1
  python - <<PYEOF
import csv, glob, collections, json
f = glob.glob("/tmp/gpmc/*counter_collection.csv")
acc = json.load(open("/tmp/gemm_pmc_acc.json"))
if not f:
    print("no output for $C:", open("/tmp/gpmc.log").read()[-300:])
else:
    per = collections.OrderedDict()          # dispatch id -> (name, counters)
    for r in csv.DictReader(open(f[0])):
        d = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "?"), "c": {}})
        d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    groups, cur = [], []
    for i in sorted(per):                    # a torch.randn launch (operand creation) opens the next shape
        nm = per[i]["name"]
        if "distribution" in nm or "normal" in nm:
            if cur: groups.append(cur); cur = []
        elif "gemm_dma_kernel" in nm or "gemm_f32_kernel" in nm:
            cur.append(per[i])
    if cur: groups.append(cur)
    shapes = [l.split() for l in open("/tmp/gpmc.log") if " tA" in l]
    assert len(groups) == len(shapes), (len(groups), len(shapes))
    for s, grp in zip(shapes, groups):
        key = " ".join(s[:5])
        grp = grp[3 * (len(grp) // 23):]       # skip the warm-up launches (a shape may take more than one GEMM launch per call)
        per_call = len(grp) // 20
        for c in grp[0]["c"]:
            acc.setdefault(key, {})[c] = sum(g["c"][c] for g in grp) / 20.0
        acc[key]["us"] = float(s[5]); acc[key]["launches_per_call"] = per_call; acc[key]["grid"] = grp[-1]["grid"]
json.dump(acc, open("/tmp/gemm_pmc_acc.json", "w"))
PYEOF
done
python - <<PYEOF
import json
acc = json.load(open("/tmp/gemm_pmc_acc.json"))
for k, m in acc.items():
    M, N, K = [int(x) for x in k.split()[:3]]
    m["algorithmic_MB"] = 4e-6 * (M * K + K * N + M * N)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["fetch_MB_x2"] = 2 * m["FETCH_SIZE"] * 1024e-6; m["write_MB"] = m["WRITE_SIZE"] * 1024e-6
        m["traffic_over_algorithmic"] = (m["fetch_MB_x2"] + m["write_MB"]) / m["algorithmic_MB"]
    if "TCC_EA0_RDREQ_sum" in m and "TCC_EA0_WRREQ_sum" in m:
        m["ea_read_MB_128B"] = m["TCC_EA0_RDREQ_sum"] * 128e-6; m["ea_write_MB_64B"] = m["TCC_EA0_WRREQ_sum"] * 64e-6
        m["ea_over_algorithmic"] = (m["ea_read_MB_128B"] + m["ea_write_MB_64B"]) / m["algorithmic_MB"]
    if "TCC_HIT_sum" in m:
        m["l2_hit"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in m.items() if not a.startswith("TCC_") and a not in ("FETCH_SIZE", "WRITE_SIZE")})
json.dump(acc, open("$R/gpurun_out/${TAG}_gemm_pmc.json", "w"), indent=1)
PYEOF
