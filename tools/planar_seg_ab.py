"""Planar one-launch step as fused_planar_seg_kernel (lane = sample, wave = time segment) against fused_planar_kernel
(8 samples per wave through an LDS tile; no_planar_seg): event-timed launch, iterations/s as single calls and inside
optimize(opt_iters=K); interleaved rounds in one process on one box.  usage: planar_seg_ab.py [nppg S T]  (4 goals)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
om = synthetic_obstacle_map(seed=0, tensor_args=ta)
P, S, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 64, 128)))
pls = {}
for name in ("tile", "seg"):
    pl = W.hip_planar_planner(W.PLANAR, T, goals, P, S, om, ta, seed=0)
    pl._engine.set_option("no_planar_seg", 1 if name == "tile" else 0)
    for _ in range(3):
        pl.optimize(opt_iters=100)
    pls[name] = pl
res = {c: [] for c in pls}
for rnd in range(5):
    for name, pl in pls.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300):
            pl.optimize()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pl.optimize(opt_iters=300)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        pl._engine.profile_enable(True)
        for _ in range(50):
            pl.optimize()
        torch.cuda.synchronize()
        kms, launches = pl._engine.profile_read()
        pl._engine.profile_enable(False)
        res[name].append((300 / (t1 - t0), 300 / (t2 - t1), {k: round(1e3 * v / launches, 2) for k, v in kms.items()}))
print(f"planar {4 * P} particles x {S} samples x {T} waypoints")
for name, v in res.items():
    best = max(v, key=lambda r: r[1])
    print(f"{name:8s} {pls[name]._engine.last_cost_kernel():26s} single calls {max(r[0] for r in v):8.0f} it/s  "
          f"in one call {max(r[1] for r in v):8.0f} it/s  kernels (us) {best[2]}")
