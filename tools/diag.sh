#!/bin/bash
for d in 0 1 2 3; do
  echo -n "SGPMP_K3_DIAG=$d  "
  SGPMP_K3_DIAG=$d python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('K2 %.1f  K3 %.1f' % (k['sample']*1e3, k['cost_sweep']*1e3))"
done
