"""Per-phase cycle shares of the planar fused kernel (diagnostic build -DPLANAR_STAMPS loaded through
SGPMP_LIB_PATH): the costs buffer holds, per item, the cycles of (top, A1, A2, wait, pend+B1, C, B2, carry+end)."""
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
names = ["top(DMA issue)", "A1 noise", "A2 recurrence", "vmcnt wait", "pend + B1", "C", "B2 stores", "carry / end"]
tot = c.sum(1).mean()
print(f"mean cycles per item: {tot:.0f} (= {tot / 2.3e3:.1f} us at 2.3 GHz), {pl._engine.last_cost_kernel()}")
for n, v in zip(names, c.mean(0)):
    print(f"  {n:16s} {v:9.0f} cycles  {100 * v / tot:5.1f} %")
