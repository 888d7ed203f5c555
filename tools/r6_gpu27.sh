#!/bin/bash
# the 16-row last tile against a build without it (libytvln_noshort.so), ABAB on one box: the attention sites, then the headline
export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do for v in short noshort; do
if [ $v = short ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_noshort.so; fi
echo "== $v (pass $rep)"; TEXT=80 timeout 600 python tools/attn_bench.py 2>/dev/null | grep -v "img self"
done; done > gpurun_out/r6_attn_short_tile_ab.log 2>&1
cat gpurun_out/r6_attn_short_tile_ab.log
timeout 900 python -m pytest tests -m gpu -x -q -k "attn or attention or coatt" 2>&1 | tail -1
