#!/bin/bash
mkdir -p gpurun_out/r06t
run2() {  # two independent processes side by side; $1 = tag, rest = env assignments
  tag=$1; shift
  env "$@" NSC_STRESS_RESTORE=lr0 NSC_STRESS_SOLO=1 timeout 900 python tools/dp_race_stress.py $REPS > gpurun_out/r06t/${tag}_a.txt 2>&1 &
  PA=$!
  env "$@" NSC_STRESS_RESTORE=lr0 NSC_STRESS_SOLO=1 timeout 900 python tools/dp_race_stress.py $REPS > gpurun_out/r06t/${tag}_b.txt 2>&1 &
  PB=$!
  wait $PA $PB
  echo "== $tag"; grep -h "^restore" gpurun_out/r06t/${tag}_a.txt gpurun_out/r06t/${tag}_b.txt | cut -c1-200
}
NP=$PWD/nsc_amd/libnsc_hip_nopk.so
REPS=300 run2 b4_nopk NSC_LIB_PATH=$NP
REPS=300 run2 b4_shipped
REPS=80 run2 b128_nopk NSC_LIB_PATH=$NP NSC_STRESS_B=128
REPS=80 run2 b128_shipped NSC_STRESS_B=128
B="python bench.py --no-cpu-baseline --no-infer --no-op-surface --steps 20 --warmup 5 --passes 3"
for t in 1 2; do
NSC_LIB_PATH=$NP $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nopk', d['ms_per_step'])"
$B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('shipped', d['ms_per_step'])"
done
