#!/bin/bash
# ABBA order: the chip runs slower once it is warm, so an A-then-B comparison is biased towards A
mkdir -p gpurun_out
L=gpurun_out/r3_sw11.log
: > $L
i=0
for V in "YTVLN_GEMM_SW=1" "X=default" "X=default" "YTVLN_GEMM_SW=1"; do
  i=$((i+1))
  env $V timeout 900 python bench.py --no-variants --no-cpu-baseline --kernel-table --graph off > gpurun_out/r3_kt11_$i.json 2> gpurun_out/r3_kt11_$i.txt
  echo "== run $i $V" >> $L
  python - >> $L <<PY
import json
d=json.loads(open("gpurun_out/r3_kt11_$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
done
python - >> $L <<'PY'
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)==8 and p[0].isdigit(): d[tuple(p[:5])]=(int(p[5]),float(p[6]),float(p[7]))
    return d
s1=load("gpurun_out/r3_kt11_1.txt"); d1=load("gpurun_out/r3_kt11_2.txt"); d2=load("gpurun_out/r3_kt11_3.txt"); s2=load("gpurun_out/r3_kt11_4.txt")
ta=tb=0
for k in sorted(d1,key=lambda k:-d1[k][1]):
    a=(d1[k][1]+d2[k][1])/2; b=(s1[k][1]+s2[k][1])/2
    print("%7s %6s %6s %s %s calls %4d  default %8.3f ms | sw %8.3f ms  %+5.1f%%   (d %.3f %.3f  s %.3f %.3f)"%(*k,d1[k][0],a,b,100*(a/b-1),d1[k][1],d2[k][1],s1[k][1],s2[k][1]))
    ta+=a; tb+=b
print("total",ta,tb)
PY
cat $L
