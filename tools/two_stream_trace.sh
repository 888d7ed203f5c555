#!/bin/bash
# Concurrency inside the replayed hipGraph of the two-stream step: rocprofv3 --kernel-trace over the bench run, then for each timed replay the
# wall time, the sum of kernel durations, and the time during which two or more kernels were in flight.
# usage: tools/two_stream_trace.sh <tag> [bench args] -> gpurun_out/<tag>_two_stream_trace.json
tag=$1; shift
export TMPDIR=/tmp; export R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; rm -rf /tmp/ts2
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ts2 -o g -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-variants --host-probe 0 "$@" > /tmp/ts2.log 2>&1
tail -1 /tmp/ts2.log | cut -c1-160
TAG=$tag python - <<'PYEOF'
import csv, glob, json, os
f = glob.glob("/tmp/ts2/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
ends = ad[1::2]                              # last AdamW launch of every step (2 launches per step)
steps = []
for a, b in zip(ends[-5:-1], ends[-4:]):
    seg = rows[a + 1:b + 1]
    ev = sorted([(s, 1) for s, e, _ in seg] + [(e, -1) for s, e, _ in seg])
    depth, last, t = 0, ev[0][0], {0: 0, 1: 0, 2: 0}
    for ts, d in ev:
        t[min(depth, 2)] += ts - last
        last, depth = ts, depth + d
    fam = {}
    for s, e, n in seg:
        k = "gemm" if "gemm" in n else "attention" if "attn" in n else "layernorm" if "ln_" in n else "other"
        fam[k] = fam.get(k, 0) + (e - s)
    steps.append({"kernels": len(seg), "wall_ms": (max(e for _, e, _ in seg) - seg[0][0]) / 1e6, "sum_of_kernel_ms": sum(e - s for s, e, _ in seg) / 1e6,
                  "idle_ms": t[0] / 1e6, "one_kernel_ms": t[1] / 1e6, "two_or_more_ms": t[2] / 1e6,
                  "sum_by_family_ms": {k: round(v / 1e6, 3) for k, v in fam.items()}})
out = {"source": "rocprofv3 --kernel-trace over python bench.py --steps 4 --warmup 2 " + os.environ.get("ARGS", "") + " (hipGraph replay), the four timed replays",
       "bench_line": open("/tmp/ts2.log").read().strip().splitlines()[-1][:400], "steps": steps}
json.dump(out, open(os.path.join(os.environ["R"], "gpurun_out", os.environ["TAG"] + "_two_stream_trace.json"), "w"), indent=1)
for s in steps: print({k: round(v, 3) if isinstance(v, float) else v for k, v in s.items()})
PYEOF
