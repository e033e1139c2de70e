"""Ablation of the cost sweep (K3) at config 3: time per launch for sub-sets of the cost list."""
import sys, time; sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.costs.cost_functions import CostComposite
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
P, S, T = 1024, 128, 64
sph = torch.as_tensor(W.panda_spheres()).to(**ta).reshape(-1, 4).contiguous()
pl = W.hip_panda_planner(W.PANDA, T, P, S, ta, seed=0)
pl.optimize(obstacle_spheres=sph)
full = pl.cost
names = ["gp", "goal", "self", "spheres"]
def timeit(cost, isw):
    eng = cost._engine(ta["dtype"], ta["device"])
    w = pl._engine.is_weights(pl.particle_means, 1.0) if isw else None
    out = torch.empty(P * S, **ta)
    for _ in range(5): eng.cost_eval(pl.state_samples, spheres=sph, is_weights=w, rows_per_particle=S, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): eng.cost_eval(pl.state_samples, spheres=sph, is_weights=w, rows_per_particle=S, out=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 50 * 1e6
for sel in ([0,1,2,3], [0,1], [2], [3], [2,3], [0,1,3], [0,1,2]):
    c = CostComposite(7, T, [full.cost_list[i] for i in sel], FK=full.FK, tensor_args=ta)
    print([names[i] for i in sel], "%.1f us (with IS %.1f us)" % (timeit(c, False), timeit(c, True)))
