#!/bin/bash
# A/B on ONE box: tools/ab.sh "ENVVAR=1" [reps]   -- alternates baseline / variant bench runs
V="$1"; R=${2:-3}
for i in $(seq $R); do
  for mode in base var; do
    if [ $mode = var ]; then export $V; else unset ${V%%=*}; fi
    echo -n "$mode  "
    python3 bench.py --steps 150 --warmup 15 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('it/s %.1f  K2 %.1f  K3 %.1f' % (d['value'], k['sample']*1e3, k['cost_sweep']*1e3))"
  done
done
