#!/bin/bash
# rocprofv3 kernel trace of `python bench.py --graph off ...` (eager launches so every kernel is visible) -> per-kernel totals.
# usage: tools/kernel_stats.sh <tag> [bench args...]   writes gpurun_out/<tag>_kernel_stats.csv and prints the top kernels
export TMPDIR=/tmp; R=$PWD; TAG=$1; shift; mkdir -p $R/gpurun_out; cd /tmp; rm -rf /tmp/kst
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/bench.py --graph off --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-variants "$@" > /tmp/kst.log 2>&1
tail -1 /tmp/kst.log | cut -c1-200
F=$(ls /tmp/kst/*kernel_stats.csv | head -1)
cp $F $R/gpurun_out/${TAG}_kernel_stats.csv
python - $F <<'PYEOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over the traced run (6 steps)")
for r in rows[:int(__import__("os").environ.get("TOPN", "22"))]:
    print(f'{float(r["TotalDurationNs"])/6e6:8.2f} ms/step {float(r["Percentage"]):5.1f}% calls/step {int(r["Calls"])/6:7.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:90]}')
PYEOF
