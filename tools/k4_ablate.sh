#!/bin/bash
# K4 (update kernel) by parts, event-timed inside bench.py (config 3): build variants of update.hip with
# -DK4_EMPTY / -DK4_NO_TAIL / -DK4_NO_STATS / -DK4_NO_MAIN guards into ab_libs/<variant>/ first (see git history of
# this file's commit message).  Measured: empty launch 6.2 us (launch + events), softmax 2.7, gather + mean update
# 3.6, importance-sampling tail 4.6, statistics atomics 0.4 -> 17.0 us.  Preloading the tail's Q^-1 entries and
# removing its integer division changed nothing (17.2 vs 17.2): the tail's cost is its barrier + store chain.
for rep in 1 2; do
for v in BASE K4_EMPTY K4_NO_TAIL K4_NO_STATS K4_NO_MAIN; do
  SGPMP_LIB_PATH=$PWD/ab_libs/$v/stoch_gpmp_amd/libsgpmp.so python3 bench.py --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --single-iteration-calls 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'update', round(d['kernel_ms_per_step']['update']*1e3,2), 'us  fused', round(d['kernel_ms_per_step']['cost_sweep']*1e3,1), ' iter', round(d['ms_per_step']*1e3,1))"
done; done
