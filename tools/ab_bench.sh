#!/bin/bash
# whole-step A/B of two library builds on one box (baseline: tools/ab/libriders_hip_base.so)
for i in 1 2; do
  RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | sed 's/^/base /' | cut -c1-230
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | sed 's/^/new  /' | cut -c1-230
done
RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so python3 bench.py --workload sml --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | sed 's/^/base /' | cut -c1-230
python3 bench.py --workload sml --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | sed 's/^/new  /' | cut -c1-230
