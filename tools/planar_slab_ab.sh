#!/bin/bash
# fused_planar_kernel (SGPMP_PLANAR_SLABS=1) against the slab-parallel launch with 2 and 4 time slabs, config 2, same box.
for rep in 1 2; do for m in 1 2 4; do
  echo "slabs=$m $(SGPMP_PLANAR_SLABS=$m python bench.py --workload planar --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["single_iteration_calls"]["iterations_per_s"]), d["kernel_ms_per_step"]["cost_sweep"], d["roofline"]["kernel"])')"
done; done
