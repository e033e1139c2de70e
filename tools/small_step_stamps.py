"""Per-phase cycles of small_step_kernel (diagnostic build -DSMALL_STAMPS loaded through SGPMP_LIB_PATH), config 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
ta = {"device": torch.device("cuda:0"), "dtype": torch.float64}
goals = [[9., 6., 0., 0.], [9., -3., 0., 0.]]
om = synthetic_obstacle_map(seed=0, tensor_args=ta)
pl = W.hip_planar_planner(W.PLANAR, 64, goals, 2, 16, om, ta, seed=0)
for _ in range(20):
    pl.optimize()
c = pl._costs[:, :6].double().cpu()
names = ["noise + coefficients", "scan", "x = mu + y, stores", "costs", "update", "statistics hand-over"]
print(pl._engine.last_cost_kernel(), "total", float(c.sum(1).mean()), "cycles =", float(c.sum(1).mean()) / 2.3e3, "us at 2.3 GHz")
for n, v in zip(names, c.mean(0)):
    print(f"  {n:24s} {float(v):9.0f} cycles")
