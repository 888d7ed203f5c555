#!/bin/bash
# rocprofv3 kernel trace + stats of a bench.py command -> gpurun_out/<name>_kernel_stats.csv   usage: tools/kernel_stats.sh <name> <bench args...>
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o $name -- python $GRAFT_REPO_ROOT/bench.py "$@" > $out/${name}_bench.json 2> $out/${name}_bench.err
f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
rm -rf $out/prof_$name
head -c 400 $out/${name}_bench.json; echo
