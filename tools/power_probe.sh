#!/bin/bash
# Samples GPU power and clocks (rocm-smi) while the scale-space runs in a loop.
# usage: tools/power_probe.sh <out_dir>
out=${1:-gpurun_out/power}
mkdir -p "$out"
python tools/kernel_bench.py --batch 128 --iters 700 --scalespace-only > "$out/kernel_bench.txt" 2>&1 &
pid=$!
sleep 5
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" >> "$out/smi.txt"
  cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input 2>/dev/null | tr '\n' ' ' >> "$out/smi.txt"
  echo "---" >> "$out/smi.txt"
  sleep 0.5
done
wait $pid
