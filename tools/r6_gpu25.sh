#!/bin/bash
# what a 16-row tail of the fp32 attention could save at most: the cfg-2 attention sites at T = 64 / 80 / 96 text tokens (80 runs as three
# 32-row tiles today; a perfect 16-row tail would sit half-way between 64 and 96)
export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do for t in 64 80 96; do
echo "== TEXT=$t (pass $rep)"; TEXT=$t timeout 600 python tools/attn_bench.py 2>/dev/null | grep -v "img self"
done; done > gpurun_out/r6_attn_tail_bound.log 2>&1
cat gpurun_out/r6_attn_tail_bound.log
