import sys, os
sys.path.insert(0, os.getcwd())
import torch
from stoch_gpmp_amd import workloads as W
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
sph = torch.as_tensor(W.panda_spheres(num=5)).to(**ta)
def mk(opt=None, T=64, P=40, S=32):
    pl = W.hip_panda_planner(W.PANDA, T, P, S, ta, seed=27)
    if opt: pl._engine.set_option(opt, 1)
    return pl
for shape in ((64, 40, 32), (64, 1024, 128), (32, 8, 8)):
    a, b, c = mk(None, *shape), mk("fused_pipe", *shape), mk("no_fused_step", *shape)
    for it in range(2):
        for pl in (a, b, c): pl.optimize(opt_iters=1, obstacle_spheres=sph)
        da = (a.state_samples - b.state_samples).abs()
        dc = (a.state_samples - c.state_samples).abs()
        bad = (da > 0).nonzero()
        print(shape, it, "default vs pipe: max", float(da.max()), "count", int((da > 0).sum()), "first", bad[:3].tolist(), "| default vs sampler:", float(dc.max()),
              "| costs equal", bool(torch.equal(a._costs, b._costs)), a._engine.last_cost_kernel(), b._engine.last_cost_kernel())
