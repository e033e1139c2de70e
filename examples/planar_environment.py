#!/usr/bin/env python3
"""2-D point-mass planning on a random occupancy grid with stoch_gpmp_amd -- the scenario of the
reference's `examples/planar_environment.py` (same scene generator call, cost list, sigmas and
planner parameters, reference lines 12-111), headless: prints the cost statistics the reference
prints and optionally saves one figure instead of animating.

    python examples/planar_environment.py [--iters 500] [--seed 0] [--plot planar.png]
"""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from stoch_gpmp_amd.costs.cost_functions import CostCollision, CostComposite, CostGP, CostGoalPrior  # noqa: E402
from stoch_gpmp_amd.envs.map_generator import generate_obstacle_map  # noqa: E402
from stoch_gpmp_amd.planner import StochGPMP, print_info  # noqa: E402


def main(opt_iters=500, seed=None, num_particles_per_goal=5, num_samples=128, traj_len=64, plot=None,
         dtype=torch.float64, verbose=True):
    tensor_args = {'device': torch.device('cuda:0'), 'dtype': dtype}
    n_dof, dt = 2, 0.02
    seed = int(time.time()) if seed is None else seed
    start_state = torch.tensor([-9., -9., 0., 0.], **tensor_args)
    multi_goal_states = torch.tensor([[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.]], **tensor_args)

    random.seed(seed)                                   # obstacle positions (reference line 49)
    obst_map = generate_obstacle_map(map_dim=[20, 20], obst_list=[], cell_size=0.1, random_gen=True,
                                     num_obst=15, rand_limits=[[-7.5, 7.5], [-7.5, 7.5]],
                                     rand_rect_shape=[2, 2], tensor_args=tensor_args)[0]

    cost = CostComposite(n_dof, traj_len, [
        CostGP(n_dof, traj_len, start_state, dt, dict(sigma_start=0.001, sigma_gp=0.1), tensor_args),
        CostGoalPrior(n_dof, traj_len, multi_goal_states=multi_goal_states,
                      num_particles_per_goal=num_particles_per_goal, num_samples=num_samples,
                      sigma_goal_prior=0.001, tensor_args=tensor_args),
        CostCollision(n_dof, traj_len, field=obst_map, sigma_coll=1e-5),
    ])
    planner = StochGPMP(
        num_particles_per_goal=num_particles_per_goal, num_samples=num_samples, traj_len=traj_len, dt=dt,
        n_dof=n_dof, opt_iters=1, temperature=1., start_state=start_state,
        multi_goal_states=multi_goal_states, cost=cost, step_size=0.5,
        sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=20.,
        sigma_start_sample=1e-3, sigma_goal_sample=1e-3, sigma_gp_sample=3, seed=seed,
        tensor_args=tensor_args)

    start_time = time.time()
    costs = None
    for i in range(opt_iters + 1):
        t_iter = time.time()
        _, _, _, _, costs, _ = planner.optimize()
        if verbose and (i == 1 or i % 50 == 0):
            print_info(i, opt_iters, t_iter, start_time, costs)
    torch.cuda.synchronize()
    if verbose:
        print(f"{opt_iters + 1} iterations in {time.time() - start_time:.3f} s")
    trajectories, _ = planner.get_recent_samples()
    if plot:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        import numpy as np
        xs = np.linspace(-10, 10, obst_map.map.shape[1])
        ys = np.linspace(-10, 10, obst_map.map.shape[0])
        fig, ax = plt.subplots()
        ax.contourf(xs, ys, obst_map.map, 20)
        tr = trajectories.cpu().numpy()
        for p in range(tr.shape[0]):
            for s in range(0, tr.shape[1], 8):
                ax.plot(tr[p, s, :, 0], tr[p, s, :, 1], 'r', alpha=0.15)
            ax.plot(tr[p].mean(0)[:, 0], tr[p].mean(0)[:, 1], 'b')
        fig.savefig(plot, dpi=120)
    return planner, costs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--plot", default=None)
    a = ap.parse_args()
    main(opt_iters=a.iters, seed=a.seed, plot=a.plot)
