#!/bin/bash
# Sample board power / clocks while a conv-dominated loop runs (is the chip at its power cap?)
python tools/conv_shapes_bench.py --shapes "l4.conv2 d4" --only fwd --reps 4000 $1 > gpurun_out/power_probe_bench.txt 2>&1 &
pid=$!
sleep 12
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|Max" | tr '\n' ' '; echo
  sleep 1
done
wait $pid
cat gpurun_out/power_probe_bench.txt | grep -E "fwd"
