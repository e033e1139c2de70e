#!/bin/bash
# Same-box A/B of builds of libsgpmp.so on the fp64 shapes: config 3's shape in fp64 (all-fp64 step and with the link fields in
# fp32) and config 2's shape in fp64 -- it/s, ms/step, event-pass launch ms.   usage: bash tools/ab_f64.sh lib1.so lib2.so ...
LIBS=${@:-stoch_gpmp_amd/libsgpmp.so}
for rep in 1 2 3; do
for lib in $LIBS; do
  for opt in 0 1; do
  SGPMP_F64_FIELDS_F32=$opt SGPMP_LIB_PATH=$PWD/$lib python3 bench.py --dtype f64 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-parity --no-sweep-alone --no-store-free 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'fields-f32' if $opt else 'all-fp64', round(d['value'],1), round(d['ms_per_step'],5), d['roofline']['launch_ms'])"
  done
  SGPMP_LIB_PATH=$PWD/$lib python3 bench.py --workload planar --dtype f64 --steps 100 --no-other-configs --no-cpu-baseline --no-parity --no-sweep-alone --no-store-free 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'planar-256x64x128', round(d['value'],1), round(d['ms_per_step'],5), d['roofline']['launch_ms'])"
done; done
