#!/bin/bash
# kernel names (and average times) the vendor library behind torch.matmul runs for the cfg-2 fp32 shapes -> gpurun_out/r5_lib_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out/prof_libk
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_libk -o libk -- python $GRAFT_REPO_ROOT/tools/lib_kernel_names.py > $out/r5_libk.log 2>&1
f=$(find $out/prof_libk -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $out/r5_lib_kernel_stats.csv; cut -c1-330 $out/r5_lib_kernel_stats.csv | head -24; else echo "no stats file"; tail -5 $out/r5_libk.log; fi
rm -rf $out/prof_libk
