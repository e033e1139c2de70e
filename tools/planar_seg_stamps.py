"""Where a launch of fused_planar_seg_kernel spends its time (diagnostic build -DSEG_STAMPS loaded through
SGPMP_LIB_PATH): s_memtime stamps of the even waves of every workgroup at (start, end of phase 1, behind barrier 1, end
of the phase-2 arithmetic, stores issued, behind barrier 2, end of phase 3, behind barrier 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
om = synthetic_obstacle_map(seed=0, tensor_args=ta)
P, S, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 64, 128)))
pl = W.hip_planar_planner(W.PLANAR, T, goals, P, S, om, ta, seed=0)
for _ in range(20):
    pl.optimize()
torch.cuda.synchronize()
assert pl._engine.last_cost_kernel() == "fused_planar_seg_kernel"
c = pl._costs.reshape(-1, 8, 8).double().cpu()          # [workgroup][recorded wave][stamp]
start = c[:, :, 0]
k0 = start.min()
rel = (start - k0) % (1 << 23)
names = ["phase 1 (noise + scan)", "barrier 1", "phase 2 arithmetic", "gathers issued", "barrier 2", "stores + costs",
         "barrier 3"]
dur = c[:, :, 1:].clone()
dur[:, :, 1:] -= c[:, :, 1:-1]
print(f"{c.shape[0]} workgroups; wave start after the first wave's: median {rel.median():.0f}, max {rel.max():.0f} cycles")
end = rel + c[:, :, 7]
print(f"last recorded stamp: median {end.median():.0f}, max {end.max():.0f} cycles after the first start "
      f"(= {end.max() / 2.3e3:.1f} us at 2.3 GHz)")
for i, n in enumerate(names):
    print(f"  {n:26s} median {dur[:, :, i].median():8.0f}  mean {dur[:, :, i].mean():8.0f}  max {dur[:, :, i].max():8.0f} cycles")
