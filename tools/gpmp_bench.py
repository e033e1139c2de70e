"""Gauss-Newton planner (GPMP) step time on the Panda problem: P particles x T waypoints."""
import sys, time; sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.planner import GPMP
c, n = W.PANDA, 7
for dtype in (torch.float32, torch.float64):
    ta = {"device": torch.device("cuda:0"), "dtype": dtype}
    for P, T in ((1024, 64), (128, 64)):
        goals = torch.tensor([c["goal_q"] + [0.] * n], **ta)
        cost = W.hip_panda_cost(c, T, P, 1, ta, goals=goals)
        pl = GPMP(num_particles_per_goal=P, traj_len=T, opt_iters=1, dt=c["dt"], n_dof=n, step_size=0.5,
                  start_state=torch.tensor(c["start_q"] + [0.] * n, **ta), multi_goal_states=goals, cost=cost,
                  sigma_start_init=c["sigma_start_init"], sigma_start_sample=c["sigma_start_sample"],
                  sigma_goal_init=c["sigma_goal_init"], sigma_goal_sample=c["sigma_goal_sample"],
                  sigma_gp_init=c["sigma_gp_init"], sigma_gp_sample=c["sigma_gp_sample"], seed=0,
                  solver_params=dict(delta=1e-2, trust_region=True, method='cholesky'), tensor_args=ta)
        sph = torch.as_tensor(W.panda_spheres()).to(**ta)
        for kernel in ("registers (block Thomas)", "LDS block Cholesky (round 3)"):
            pl._engine.set_option("gpmp_cholesky", 0 if kernel.startswith("reg") else 1)
            for _ in range(3): pl.optimize(obstacle_spheres=sph)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): _, _, costs = pl.optimize(obstacle_spheres=sph)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(f"{str(dtype):14s} P={P:5d} T={T} {kernel:30s}: {dt*1e3:7.3f} ms per Gauss-Newton step (N = {T*14} unknowns per particle), mean cost {float(costs.mean()):.4g}")
