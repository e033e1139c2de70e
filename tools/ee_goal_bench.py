"""Cost of the end-effector goal term (CostGoal) on top of the config-3 cost list."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.costs.cost_functions import CostGoal
from stoch_gpmp_amd.costs.fields import EESE3DistanceField
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
P, S, T = 1024, 128, 64
sph = torch.as_tensor(W.panda_spheres()).to(**ta)
for with_ee in (False, True):
    pl = W.hip_panda_planner(W.PANDA, T, P, S, ta, seed=0)
    if with_ee:
        H = torch.eye(4, **ta); H[:3, 3] = torch.tensor([0.3, 0.3, 0.3])
        pl.cost.cost_list.append(CostGoal(7, T, field=EESE3DistanceField(H, tensor_args=ta), sigma_goal=7e-5, tensor_args=ta))
        pl.cost.compile_into(pl._engine)
    for _ in range(20): pl.optimize(obstacle_spheres=sph)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): pl.optimize(obstacle_spheres=sph)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"CostGoal {'on ' if with_ee else 'off'}: {dt*1e6:.1f} us per iteration")
