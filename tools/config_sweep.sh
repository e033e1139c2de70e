#!/bin/bash
# bench lines of the other BASELINE configurations (single-GPU shares) -- run on the GPU box
run() { echo "== $*"; python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s %.1f  ms/step %.4f  kernels(us) %s' % (d['value'], d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['kernel_ms_per_step'].items()}))"; }
run --workload planar --steps 500 --warmup 50                                   # config 2
run --workload planar --particles 4 --samples 16 --traj-len 64 --dtype f64 --steps 500 --warmup 50   # config 1
run --workload panda --particles 512 --samples 256 --traj-len 128 --steps 60 --warmup 10             # config 5 share
run --workload panda --field sdf --steps 100 --warmup 10                        # config 3, sphere-SDF
run --workload panda --dtype f64 --steps 30 --warmup 5                          # config 3 in fp64
run --workload panda --field occupancy --steps 100 --warmup 10                  # config 3, occupancy count
run --workload panda --spheres 64 --steps 60 --warmup 10                        # config 3, O = 64 stress variant
run --workload panda --spheres 64 --field sdf --steps 60 --warmup 10
