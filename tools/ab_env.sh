#!/bin/bash
# Same-box A/B of ENVIRONMENT switches of one library (e.g. grid sizes): bash tools/ab_env.sh "SGPMP_K3_BLOCKS=1024" "SGPMP_K3_BLOCKS=2048" "SGPMP_X=0"
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for e in "$@"; do
  env $e python3 bench.py --steps 300 --warmup 20 --no-other-configs --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e]', round(d['value'],1), round(d['single_call_iterations_per_s'],1), {k:round(v,5) for k,v in d['kernel_ms_per_step'].items()})"
done; done
