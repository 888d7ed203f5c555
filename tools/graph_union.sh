#!/bin/bash
# GPU busy time (union of kernel intervals) inside the replayed hipGraph steps, per stream count: how much of a step has NO kernel running?
export TMPDIR=/tmp; export R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; rm -rf /tmp/gun
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gun -o g -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-variants "$@" > /tmp/gun.log 2>&1 < /dev/null
tail -1 /tmp/gun.log | cut -c1-160
python - <<'PYEOF'
import csv, glob, json, os
fs = glob.glob("/tmp/gun/**/*kernel_trace.csv", recursive=True)
if not fs:
    print("no trace"); raise SystemExit
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(fs[0]))]
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
ends = ad[1::2]
for a, b in zip(ends[-5:-1], ends[-4:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], max(e for _, e, _ in seg)
    busy, cur_s, cur_e = 0, None, None
    two = 0
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            two += min(e, cur_e) - s if e > s else 0
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _ in seg)
    gaps = []
    ce = seg[0][1]
    for s, e, n in seg[1:]:
        if s > ce: gaps.append((s - ce, n))
        ce = max(ce, e)
    gaps.sort(reverse=True)
    print({"kernels": len(seg), "wall_ms": round((t1 - t0) / 1e6, 3), "busy_union_ms": round(busy / 1e6, 3), "idle_ms": round((t1 - t0 - busy) / 1e6, 3),
           "sum_kernel_ms": round(tot / 1e6, 3), "n_gaps": len(gaps), "gaps_gt_3us": sum(1 for g, _ in gaps if g > 3000)})
    print("  largest gaps (us, next kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:8]])
PYEOF
