#!/bin/bash
# K3 grid-size sweep at config 3 (bench kernel timings); run on the GPU box.
for b in 1024 2048 3072 4096 6144 8192 16384; do
  echo -n "SGPMP_K3_BLOCKS=$b  "
  SGPMP_K3_BLOCKS=$b python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s %.1f' % d['value'], d['kernel_ms_per_step'])"
done
