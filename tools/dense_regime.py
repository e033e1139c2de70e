"""Iteration rate when the softmax weights are SPREAD (planner.py:263-275 at a soft temperature): the update through the fused
launch's partials (default) against the row-reading update (no_dense_partials).  usage: dense_regime.py [temperature]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
temps = [float(v) for v in sys.argv[1:]] or [1.0, 1e14, 1e17]
# (a weak sampling prior: under the reference's stiff one the importance-sampling term keeps the softmax one-hot at any temperature)
SOFT = dict(sigma_start_sample=1.0, sigma_goal_sample=1.0, sigma_gp_sample=30.0)
sph = torch.as_tensor(W.panda_spheres(num=5)).to(**ta)
for temp in temps:
    c = dict(W.PANDA, temperature=temp, **(SOFT if temp > 1 else {}))
    row = {}
    for name, opt in (("partials", 0), ("row reads", 1)):
        pl = W.hip_panda_planner(c, 64, 1024, 128, ta, seed=0)
        pl._engine.set_option("no_dense_partials", opt)
        pl.optimize(opt_iters=150, obstacle_spheres=sph)
        res = {}
        for mode, call in (("in one call", lambda: pl.optimize(opt_iters=200, obstacle_spheres=sph)),
                           ("single calls", lambda: [pl.optimize(opt_iters=1, obstacle_spheres=sph) for _ in range(200)])):
            torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / 200 * 1e6
        kms = None
        pl._engine.profile_enable(True)
        for _ in range(50):
            pl.optimize(opt_iters=1, obstacle_spheres=sph)
        torch.cuda.synchronize()
        k, nl = pl._engine.profile_read()
        pl._engine.profile_enable(False)
        nnz = int((pl._weights_buf != 0).sum()) / 1024
        row[name] = (res, {kk: round(v / nl * 1e3, 1) for kk, v in k.items()}, pl._engine.dense_particles(), nnz)
        del pl
    for name, (res, k, dp, nnz) in row.items():
        print(f"temperature {temp:g}  {name:10s}: {res['in one call']:7.1f} us/iteration in one call, {res['single calls']:7.1f} as single calls; "
              f"kernels (us) {k}; dense particles {dp}; rows with weight per particle {nnz:.1f}")
