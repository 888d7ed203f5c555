#!/bin/bash
export TMPDIR=/tmp; R=$PWD; cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/bench.py --graph off --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants --host-probe 0 $PMC_BENCH_ARGS > /tmp/pmc.log 2>&1
  python - <<PYEOF
import csv,glob,collections,json
f=glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
if not f:
    print("no output:", open("/tmp/pmc.log").read()[-500:]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    k = "gemm_dma" if "gemm_dma" in n else "gemm_bf16" if "gemm_bf16_kernel" in n else "gemm_generic" if ("gemm_f32_kernel" in n or "gemm_bf16_generic" in n) else "attn" if "attn_" in n and "delta" not in n else None
    if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:{"launches":len(v),"mean":sum(v)/len(v),"sum":sum(v)} for c,v in d.items()} for k,d in acc.items()}
print(json.dumps(out))
open("$R/gpurun_out/pmc_${PMC_TAG}"+"$C".split()[0]+".json","w").write(json.dumps(out))
PYEOF
done
