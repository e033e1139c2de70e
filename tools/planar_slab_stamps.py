"""Per-phase cycles of the slab-parallel planar launch (diagnostic build -DPLANAR_STAMPS loaded through SGPMP_LIB_PATH):
the costs buffer holds, per item (its slab-0 wave), the cycles of (prologue, phase 1, staging wait, barrier, hand-off,
phase 2, barrier)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
om = synthetic_obstacle_map(seed=0, tensor_args=ta)
pl = W.hip_planar_planner(W.PLANAR, 128, goals, 64, 64, om, ta, seed=0)
for _ in range(20):
    pl.optimize()
c = pl._costs.reshape(-1, 8).double().cpu()
names = ["prologue", "phase 1 (noise + scan)", "staging wait", "barrier 1", "hand-off", "phase 2 (x, costs, stores)", "barrier 2", "-"]
tot = c.sum(1).mean()
print(f"{pl._engine.last_cost_kernel()}: mean cycles per workgroup (slab-0 wave): {tot:.0f} (= {tot / 2.3e3:.1f} us at 2.3 GHz)")
for n, v, mx in zip(names, c.mean(0), c.max(0)[0]):
    print(f"  {n:28s} {v:9.0f} cycles  {100 * v / tot:5.1f} %   max {mx:9.0f}")
