#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_sw10.log
: > $L
for V in "YTVLN_GEMM_SW=1" "YTVLN_GEMM_TILE=4 YTVLN_GEMM_SW=1" "YTVLN_GEMM_TILE=3 YTVLN_GEMM_SW=1" "YTVLN_GEMM_TILE=0 YTVLN_GEMM_SW=1"; do
  echo "== tests $V" >> $L
  env $V timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm and not x3 and not bf16" 2>&1 | tail -3 >> $L
done
for V in "X=default" "YTVLN_GEMM_SW=1"; do
  T=$(echo $V | tr -c 'A-Za-z0-9' '_')
  env $V timeout 900 python bench.py --no-variants --no-cpu-baseline --kernel-table --graph off > gpurun_out/r3_kt_$T.json 2> gpurun_out/r3_kt_$T.txt
done
python - >> $L <<'PY'
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)==8 and p[0].isdigit(): d[tuple(p[:5])]=(int(p[5]),float(p[6]),float(p[7]))
    return d
a=load("gpurun_out/r3_kt_X_default_.txt"); b=load("gpurun_out/r3_kt_YTVLN_GEMM_PP_13_.txt")
ta=tb=0
for k in sorted(a,key=lambda k:-a[k][1]):
    if k in b:
        print("%7s %6s %6s %s %s calls %4d  default %8.3f ms %6.1f TF | sw %8.3f ms %6.1f TF  %+5.1f%%"%(*k,a[k][0],a[k][1],a[k][2],b[k][1],b[k][2],100*(a[k][1]/b[k][1]-1)))
        ta+=a[k][1]; tb+=b[k][1]
print("total",ta,tb)
PY
cat $L
