#!/bin/bash
# Idle time between kernels inside the replayed hipGraph: rocprofv3 --kernel-trace over the default bench run (graph on), then per-kernel
# start/end timestamps of the timed replays.  usage: tools/graph_gaps.sh [bench args] -> gpurun_out/round2_graph_gaps.json
export TMPDIR=/tmp; export R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; rm -rf /tmp/ggap
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ggap -o g -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-variants "$@" > /tmp/ggap.log 2>&1
tail -1 /tmp/ggap.log | cut -c1-200
python - <<'PYEOF'
import csv, glob, json, os
f = glob.glob("/tmp/ggap/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region = the last 4 replays: find them as the last 4 occurrences of the AdamW kernel (one per step; 2 launches each)
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
per_step = 2
ends = ad[per_step - 1::per_step]           # index of the last AdamW launch of every step
steps = []
for a, b in zip(ends[-5:-1], ends[-4:]):
    seg = rows[a + 1:b + 1]
    busy = sum(e - s for s, e, _ in seg)
    wall = seg[-1][1] - seg[0][0]
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    steps.append({"kernels": len(seg), "wall_ms": wall / 1e6, "sum_of_kernel_ms": busy / 1e6, "idle_ms": sum(pos) / 1e6,
                  "overlap_ms": -sum(g for g in gaps if g < 0) / 1e6, "median_gap_us": sorted(pos)[len(pos) // 2] / 1e3 if pos else 0.0,
                  "gaps_over_5us": sum(1 for g in pos if g > 5000)})
out = {"source": "rocprofv3 --kernel-trace over python bench.py --steps 4 --warmup 2 (hipGraph replay), the four timed replays", "steps": steps}
json.dump(out, open(os.path.join(os.environ.get("R", "."), "gpurun_out", "round2_graph_gaps.json"), "w"), indent=1)
for s in steps: print({k: round(v, 3) if isinstance(v, float) else v for k, v in s.items()})
PYEOF
