#!/bin/bash
# Same-box A/B of builds of libsgpmp.so on the STAND-ALONE sweep and sampler of config 3 (bench.py's sweep_alone leg: launches back to
# back between two events) -- launch ms and fraction of the HBM roofline.   usage: bash tools/ab_sweep.sh lib1.so lib2.so ...
LIBS=${@:-stoch_gpmp_amd/libsgpmp.so}
for rep in 1 2 3; do
for lib in $LIBS; do
  SGPMP_LIB_PATH=$PWD/$lib python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-parity --no-store-free 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['sweep_alone']; print('$lib', 'sweep', a['launch_ms'], a['frac'], 'sampler', a['sampler_launch_ms'], 'step it/s', round(d['value'],1))"
done; done
