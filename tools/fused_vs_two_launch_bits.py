"""Are the fused launch's samples the stand-alone samplers' bit for bit at small and odd shapes too?  (rng.h scan_step: one
evaluation order for every sampler kernel.)  python3 tools/fused_vs_two_launch_bits.py on the GPU box."""
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
for shape in ((64, 40, 32), (64, 1024, 128), (32, 8, 8), (16, 3, 8), (48, 5, 24), (128, 6, 16)):
    a, c = mk(None, *shape), mk("no_fused_step", *shape)
    for it in range(2):
        for pl in (a, c): pl.optimize(opt_iters=1, obstacle_spheres=sph)
        dc = (a.state_samples - c.state_samples).abs()
        print(shape, it, "fused vs two launches: samples max diff", float(dc.max()), "costs equal", bool(torch.equal(a._costs, c._costs)),
              "means equal", bool(torch.equal(a.particle_means, c.particle_means)), a._engine.last_cost_kernel(), "|", c._engine.last_cost_kernel())
