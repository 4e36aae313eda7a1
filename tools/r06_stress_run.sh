#!/bin/bash
# two INDEPENDENT single-GPU processes side by side, three ways of restoring the parameters between repetitions
mkdir -p gpurun_out/r06g
for R in h2d d2d lr0; do
  NSC_STRESS_RESTORE=$R NSC_STRESS_SOLO=1 timeout 900 python tools/dp_race_stress.py 300 > gpurun_out/r06g/solo_${R}_a.txt 2>&1 &
  PA=$!
  NSC_STRESS_RESTORE=$R NSC_STRESS_SOLO=1 timeout 900 python tools/dp_race_stress.py 300 > gpurun_out/r06g/solo_${R}_b.txt 2>&1 &
  PB=$!
  wait $PA $PB
done
grep -h "^restore" gpurun_out/r06g/*.txt | cut -c1-300
