#!/bin/bash
# Timing experiments on the in-launch update (fused_tail.inc); SGPMP_TAIL_DEBUG modes give WRONG results, timing only.
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs"
for rep in 1; do
for mode in ${MODES:-notail 0 2 4 8 16 32 64 124}; do
  if [ $mode = notail ]; then out=$($B 2>/dev/null); else out=$(SGPMP_TAIL_UPDATE=1 SGPMP_TAIL_DEBUG=$mode $B 2>/dev/null); fi
  echo "$mode $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["single_iteration_calls"]["iterations_per_s"]), d["kernel_ms_per_step"]["cost_sweep"], d["kernel_ms_per_step"]["update"])')"
done; done
