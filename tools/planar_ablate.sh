#!/bin/bash
for v in BASE PLANAR_NO_A1 PLANAR_NO_A2 PLANAR_NO_GRID BASE; do
  echo -n "$v: "
  SGPMP_LIB_PATH=$(pwd)/tools/ab/libsgpmp_$v.so python3 bench.py --workload planar --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s' % d['value'], {k:round(v*1e3,1) for k,v in d['kernel_ms_per_step'].items()})"
done
