#!/usr/bin/env python3
"""Franka Panda reaching among sphere obstacles with stoch_gpmp_amd -- the scenario of the reference's
`examples/panda_environment.py` (reference lines 23-147: cost list GP + goal prior + self collision +
sphere collision + end-effector goal, same sigmas and planner parameters, same sphere spawner),
headless.  Two things the reference gets from libraries that are not available here are replaced by
fixed data: the goal joint configuration (PyBullet IK in the reference, lines 55-62) is a constant
inside the joint limits, and the end-effector target frame is the FK of that configuration.

    python examples/panda_environment.py [--iters 500] [--seed 0]
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from stoch_gpmp_amd.costs.cost_functions import (CostCollision, CostComposite, CostGoal, CostGP,  # noqa: E402
                                                 CostGoalPrior)
from stoch_gpmp_amd.costs.fields import (EESE3DistanceField, LinkDistanceField,  # noqa: E402
                                         LinkSelfDistanceField)
from stoch_gpmp_amd.envs.spheres import random_init_static_sphere  # noqa: E402
from stoch_gpmp_amd.planner import StochGPMP, print_info  # noqa: E402
from stoch_gpmp_amd.robots.panda import DifferentiableFrankaPanda  # noqa: E402


def main(opt_iters=500, seed=None, num_particles_per_goal=5, num_samples=32, num_obst=5, traj_len=64,
         dtype=torch.float32, verbose=True):
    device = torch.device('cuda:0')
    tensor_args = {'device': device, 'dtype': dtype}
    dt = 0.05
    seed = int(time.time()) if seed is None else seed
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)

    panda_fk = DifferentiableFrankaPanda(gripper=False, device=device)
    n_dof = panda_fk._n_dofs
    start_q = torch.tensor([0.012, -0.57, 0., -2.81, 0., 3.037, 0.741], **tensor_args)
    start_state = torch.cat((start_q, torch.zeros_like(start_q)))
    q_goal = torch.tensor([0.5, 0.2, 0.3, -1.5, 0.1, 2.0, 0.3], **tensor_args)       # stands in for the IK solution
    multi_goal_states = torch.cat([q_goal, torch.zeros_like(q_goal)]).unsqueeze(0)
    target_H = panda_fk.compute_forward_kinematics_all_links(q_goal.unsqueeze(0))[0, -1]   # end-effector frame

    cost = CostComposite(n_dof, traj_len, [
        CostGP(n_dof, traj_len, start_state, dt, dict(sigma_start=0.0001, sigma_gp=0.0007), tensor_args),
        CostGoalPrior(n_dof, traj_len, multi_goal_states=multi_goal_states,
                      num_particles_per_goal=num_particles_per_goal, num_samples=num_samples,
                      sigma_goal_prior=20., tensor_args=tensor_args),
        CostCollision(n_dof, traj_len, field=LinkSelfDistanceField(margin=0.03, tensor_args=tensor_args),
                      sigma_coll=0.01),
        CostCollision(n_dof, traj_len, field=LinkDistanceField(tensor_args=tensor_args), sigma_coll=0.01),
        CostGoal(n_dof, traj_len, field=EESE3DistanceField(target_H, tensor_args=tensor_args),
                 sigma_goal=0.00007),
    ], FK=panda_fk.compute_forward_kinematics_all_links)

    planner = StochGPMP(
        num_particles_per_goal=num_particles_per_goal, num_samples=num_samples, traj_len=traj_len, dt=dt,
        n_dof=n_dof, opt_iters=1, temperature=1., start_state=start_state,
        multi_goal_states=multi_goal_states, cost=cost, step_size=0.1,
        sigma_start_init=0.0001, sigma_goal_init=0.1, sigma_gp_init=0.8,
        sigma_start_sample=0.001, sigma_goal_sample=0.07, sigma_gp_sample=0.1, seed=seed,
        tensor_args=tensor_args)

    # spawn obstacles (reference lines 124-133)
    obstacle_spheres = np.zeros((1, num_obst, 4))
    for i in range(num_obst):
        r, pos = random_init_static_sphere(0.1, 0.2, np.array([0.6, -0.2, 0.6]), np.array([1., 0.2, 1]), 0.01)
        obstacle_spheres[0, i, :3], obstacle_spheres[0, i, 3] = pos, r
    obs = {'obstacle_spheres': torch.from_numpy(obstacle_spheres).to(**tensor_args)}

    start_time = time.time()
    costs = None
    for i in range(opt_iters + 1):
        t_iter = time.time()
        _, _, _, _, costs, _ = planner.optimize(**obs)
        if verbose and (i == 1 or i % 50 == 0):
            print_info(i, opt_iters, t_iter, start_time, costs)
    torch.cuda.synchronize()
    if verbose:
        print(f"{opt_iters + 1} iterations in {time.time() - start_time:.3f} s")
        ee = panda_fk.compute_forward_kinematics_all_links(planner.particle_means[:, -1, :n_dof].contiguous())[:, -1, :3, 3]
        print("end-effector distance to target per particle [m]:",
              [round(float(v), 4) for v in (ee - target_H[:3, 3]).norm(dim=-1)])
    return planner, costs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    main(opt_iters=a.iters, seed=a.seed)
