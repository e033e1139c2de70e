#!/bin/bash
run() { python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s %.1f  ms/step %.4f  kernels(us) %s' % (d['value'], d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['kernel_ms_per_step'].items()}))"; }
for v in "" 1; do
  if [ -n "$v" ]; then export SGPMP_NO_SMALL_SAMPLER=1; echo "standard sampler"; else unset SGPMP_NO_SMALL_SAMPLER; echo "small-launch sampler"; fi
  run --workload planar --steps 500 --warmup 50
  run --workload planar --particles 4 --samples 16 --traj-len 64 --dtype f64 --steps 500 --warmup 50
  run --workload panda --particles 16 --samples 32 --steps 300 --warmup 30
done
