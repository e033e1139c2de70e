#!/bin/bash
# Same-box A/B of several builds of libsgpmp.so (boxes differ by +-5 % in clock; only a same-box comparison
# says whether a kernel change helped).  Build the other versions out of tree first, e.g.
#   c=<commit>; mkdir -p ab_libs/$c; git archive $c stoch_gpmp_amd/csrc stoch_gpmp_amd/robots include | tar -x -C ab_libs/$c
#   make -C ab_libs/$c/stoch_gpmp_amd/csrc
# (ab_libs/ is git-ignored but travels with gpurun), then on the GPU box:
#   bash tools/ab_libs.sh ab_libs/<commit>/stoch_gpmp_amd/libsgpmp.so stoch_gpmp_amd/libsgpmp.so
LIBS=${@:-stoch_gpmp_amd/libsgpmp.so}
for rep in 1 2; do
for lib in $LIBS; do
  SGPMP_LIB_PATH=$PWD/$lib python3 bench.py --steps 300 --warmup 20 --no-other-configs --no-cpu-baseline --no-parity --no-sweep-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'storing', round(d['value'],1), round(d['ms_per_step'],5), 'launch', d['roofline']['launch_ms'], 'store-free', (d.get('store_free') or {}).get('iterations_per_s'), 'single calls', (d.get('single_iteration_calls') or {}).get('iterations_per_s'))"
done; done
