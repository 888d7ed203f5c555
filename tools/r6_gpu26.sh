#!/bin/bash
# the 16-row last tile (contracting matmuls run 8 of their 16 steps): attention tests, then the cfg-2 attention sites at T = 80 (compare with
# profiles/round6_attn_tail_bound.log, same tool, the build before)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "attn or attention or coatt" 2>&1 | tail -2
for rep in 1 2 3; do
echo "== TEXT=80 short-tile build (pass $rep)"; TEXT=80 timeout 600 python tools/attn_bench.py 2>/dev/null | grep -v "img self"
done > gpurun_out/r6_attn_tail_after.log 2>&1
cat gpurun_out/r6_attn_tail_after.log
