"""The workloads of BASELINE.json as data + builders, written against the public reference-shaped
API of stoch_gpmp_amd (the way reference examples/planar_environment.py:62-97 and
examples/panda_environment.py:83-122 build their planners).  Hyper-parameters: SURVEY.md 8(d)."""
import numpy as np
import torch

from stoch_gpmp_amd.costs.cost_functions import (CostCollision, CostComposite, CostGP,
                                                 CostGoalPrior)
from stoch_gpmp_amd.costs.fields import LinkDistanceField, LinkSelfDistanceField
from stoch_gpmp_amd.planner import StochGPMP
from stoch_gpmp_amd.robots.panda import DifferentiableFrankaPanda, URDFChain


PLANAR = dict(n_dof=2, dt=0.02, start=[-9., -9., 0., 0.],
              cost_sigma_start=1e-3, cost_sigma_gp=0.1, sigma_coll=1e-5, sigma_goal_prior=1e-3,
              sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=20.,
              sigma_start_sample=1e-3, sigma_goal_sample=1e-3, sigma_gp_sample=3.,
              step_size=0.5, temperature=1.)

PANDA = dict(n_dof=7, dt=0.05,
             start_q=[0.012, -0.57, 0., -2.81, 0., 3.037, 0.741],
             goal_q=[0.5, 0.2, 0.3, -1.5, 0.1, 2.0, 0.3],
             cost_sigma_start=1e-4, cost_sigma_gp=7e-4, sigma_self=0.01, sigma_coll=0.01,
             sigma_goal_prior=20., self_margin=0.03,
             sigma_start_init=1e-4, sigma_goal_init=0.1, sigma_gp_init=0.8,
             sigma_start_sample=1e-3, sigma_goal_sample=0.07, sigma_gp_sample=0.1,
             step_size=0.1, temperature=1.)


def panda_spheres(num=5, seed=0):
    """Synthetic sphere obstacles [1,O,4]."""
    rng = np.random.default_rng(seed)
    sph = np.zeros((1, num, 4))
    sph[0, :, :3] = rng.uniform([0.2, -0.5, 0.2], [1.0, 0.5, 1.0], size=(num, 3))
    sph[0, :, 3] = rng.uniform(0.1, 0.2, size=num)
    return sph


def hip_planar_cost(c, T, goals, nppg, S, obst_map, ta):
    n = c["n_dof"]
    start = torch.tensor(c["start"], **ta)
    goals_t = torch.as_tensor(goals).to(**ta)
    return CostComposite(n, T, [
        CostGP(n, T, start, c["dt"], dict(sigma_start=c["cost_sigma_start"],
                                          sigma_gp=c["cost_sigma_gp"]), ta),
        CostGoalPrior(n, T, multi_goal_states=goals_t, num_particles_per_goal=nppg, num_samples=S,
                      sigma_goal_prior=c["sigma_goal_prior"], tensor_args=ta),
        CostCollision(n, T, field=obst_map, sigma_coll=c["sigma_coll"], tensor_args=ta),
    ], tensor_args=ta)


def hip_planar_planner(c, T, goals, nppg, S, obst_map, ta, initial_particle_means=None, seed=None,
                       noise='philox', temperature=None, **kw):
    cost = hip_planar_cost(c, T, goals, nppg, S, obst_map, ta)
    return StochGPMP(
        num_particles_per_goal=nppg, num_samples=S, traj_len=T, opt_iters=1, dt=c["dt"],
        n_dof=c["n_dof"], step_size=c["step_size"],
        temperature=c["temperature"] if temperature is None else temperature,
        start_state=torch.tensor(c["start"], **ta), multi_goal_states=torch.as_tensor(goals).to(**ta),
        initial_particle_means=initial_particle_means, cost=cost,
        sigma_start_init=c["sigma_start_init"], sigma_start_sample=c["sigma_start_sample"],
        sigma_goal_init=c["sigma_goal_init"], sigma_goal_sample=c["sigma_goal_sample"],
        sigma_gp_init=c["sigma_gp_init"], sigma_gp_sample=c["sigma_gp_sample"], seed=seed,
        tensor_args=ta, noise=noise, **kw)


def hip_panda_cost(c, T, nppg, S, ta, field_type='rbf', goals=None, with_self=True,
                   with_spheres=True, clamp_sdf=False, chain=None):
    n = c["n_dof"]
    start = torch.tensor(c["start_q"] + [0.] * n, **ta)
    goals_t = torch.tensor([c["goal_q"] + [0.] * n], **ta) if goals is None \
        else torch.as_tensor(goals).to(**ta)
    fk = DifferentiableFrankaPanda(gripper=False, device=ta["device"]) if chain is None \
        else URDFChain(chain, device=ta["device"])             # (any serial chain: c["n_dof"] revolute joints)
    terms = [
        CostGP(n, T, start, c["dt"], dict(sigma_start=c["cost_sigma_start"],
                                          sigma_gp=c["cost_sigma_gp"]), ta),
        CostGoalPrior(n, T, multi_goal_states=goals_t, num_particles_per_goal=nppg, num_samples=S,
                      sigma_goal_prior=c["sigma_goal_prior"], tensor_args=ta),
    ]
    if with_self:
        terms.append(CostCollision(n, T, field=LinkSelfDistanceField(margin=c["self_margin"],
                                                                     tensor_args=ta),
                                   sigma_coll=c["sigma_self"], tensor_args=ta))
    if with_spheres:
        terms.append(CostCollision(n, T, field=LinkDistanceField(field_type=field_type, clamp_sdf=clamp_sdf,
                                                                 tensor_args=ta),
                                   sigma_coll=c["sigma_coll"], tensor_args=ta))
    return CostComposite(n, T, terms, FK=fk.compute_forward_kinematics_all_links, tensor_args=ta)


def hip_panda_planner(c, T, nppg, S, ta, field_type='rbf', seed=None, noise='philox', goals=None,
                      initial_particle_means=None, chain=None, **kw):
    n = c["n_dof"]
    goals_t = torch.tensor([c["goal_q"] + [0.] * n], **ta) if goals is None \
        else torch.as_tensor(goals).to(**ta)
    cost = hip_panda_cost(c, T, nppg, S, ta, field_type=field_type, goals=goals_t, chain=chain)
    return StochGPMP(
        num_particles_per_goal=nppg, num_samples=S, traj_len=T, opt_iters=1, dt=c["dt"], n_dof=n,
        step_size=c["step_size"], temperature=c["temperature"],
        start_state=torch.tensor(c["start_q"] + [0.] * n, **ta), multi_goal_states=goals_t,
        initial_particle_means=initial_particle_means, cost=cost,
        sigma_start_init=c["sigma_start_init"], sigma_start_sample=c["sigma_start_sample"],
        sigma_goal_init=c["sigma_goal_init"], sigma_goal_sample=c["sigma_goal_sample"],
        sigma_gp_init=c["sigma_gp_init"], sigma_gp_sample=c["sigma_gp_sample"], seed=seed,
        tensor_args=ta, noise=noise, **kw)
