#!/bin/bash
# what the chip reports while the fp32 GEMM runs on random vs constant operands: power, cap, clocks (rocm-smi sampled beside the loop)
mkdir -p gpurun_out
L=gpurun_out/r3_power.log
: > $L
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -40 >> $L
for D in randn zeros; do
  echo "== DATA=$D (16128x1024x1024 fwd GEMM in a loop)" >> $L
  DATA=$D ITERS=40000 SHAPES=img1 python tools/gemm_shapes_bench.py > /tmp/gb_$D.log 2>&1 &
  PID=$!
  sleep 25
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)\|hotspot" | head -12 >> $L
    echo "--" >> $L
    sleep 2
  done
  wait $PID
  cat /tmp/gb_$D.log | grep -v amdgpu >> $L
done
cat $L
