"""Problem definitions shared by the oracle tests, the HIP parity tests and bench.py.

Hyper-parameters are those of the reference examples (reference
examples/planar_environment.py:16-96, examples/panda_environment.py:52-121).  This module builds
ORACLE objects only; the HIP-side twins are built in the tests from the same dicts.
"""
import numpy as np
import torch

from oracle import ref_equiv as R

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


def oracle_planar_cost(c, T, goals, nppg, S, grid, cell_size, c_offset, dtype):
    n = c["n_dof"]
    start = torch.tensor(c["start"], dtype=dtype)
    goals_t = torch.as_tensor(goals, dtype=dtype)
    grid_t = torch.as_tensor(np.asarray(grid, dtype=np.float64)).to(dtype)
    off = torch.as_tensor(c_offset, dtype=dtype)
    terms = [
        lambda tr, xt, **o: R.cost_gp(tr, start, n, c["dt"], c["cost_sigma_start"],
                                      c["cost_sigma_gp"]),
        lambda tr, xt, **o: R.cost_goal_prior(tr, goals_t, nppg, S, n, c["sigma_goal_prior"]),
        lambda tr, xt, **o: R.cost_collision_grid(tr, n, grid_t, cell_size, off, c["sigma_coll"]),
    ]
    return R.CompositeCost(n, T, terms)


def oracle_planar_planner(c, T, goals, nppg, S, grid, cell_size, c_offset, dtype=torch.float64,
                          initial_particle_means=None, seed=None, eps_init=None,
                          temperature=None):
    cost = oracle_planar_cost(c, T, goals, nppg, S, grid, cell_size, c_offset, dtype)
    return R.OraclePlanner(
        nppg, S, T, c["dt"], c["n_dof"], torch.tensor(c["start"], dtype=dtype),
        torch.as_tensor(goals, dtype=dtype), cost, c["step_size"],
        c["temperature"] if temperature is None else temperature,
        c["sigma_start_init"], c["sigma_start_sample"], c["sigma_goal_init"],
        c["sigma_goal_sample"], c["sigma_gp_init"], c["sigma_gp_sample"],
        initial_particle_means=initial_particle_means, seed=seed, dtype=dtype, eps_init=eps_init)


def oracle_panda_cost(c, T, nppg, S, dtype, field_type='rbf', goals=None, with_self=True,
                      with_spheres=True):
    from oracle.fk import fk_all_links
    n = c["n_dof"]
    start = torch.tensor(c["start_q"] + [0.] * n, dtype=dtype)
    goals_t = torch.tensor([c["goal_q"] + [0.] * n], dtype=dtype) if goals is None \
        else torch.as_tensor(goals, dtype=dtype)
    terms = [
        lambda tr, xt, **o: R.cost_gp(tr, start, n, c["dt"], c["cost_sigma_start"],
                                      c["cost_sigma_gp"]),
        lambda tr, xt, **o: R.cost_goal_prior(tr, goals_t, nppg, S, n, c["sigma_goal_prior"]),
    ]
    if with_self:
        terms.append(lambda tr, xt, **o: R.cost_collision_links(
            xt, lambda f: R.field_self(f, margin=c["self_margin"]), c["sigma_self"]))
    if with_spheres:
        terms.append(lambda tr, xt, **o: R.cost_collision_links(
            xt, lambda f: R.field_spheres(f, o["obstacle_spheres"], field_type=field_type),
            c["sigma_coll"]))
    return R.CompositeCost(n, T, terms, FK=fk_all_links)


def oracle_panda_planner(c, T, nppg, S, dtype=torch.float64, field_type='rbf', seed=None,
                         eps_init=None, goals=None):
    n = c["n_dof"]
    goals_t = torch.tensor([c["goal_q"] + [0.] * n], dtype=dtype) if goals is None \
        else torch.as_tensor(goals, dtype=dtype)
    cost = oracle_panda_cost(c, T, nppg, S, dtype, field_type=field_type, goals=goals_t)
    return R.OraclePlanner(
        nppg, S, T, c["dt"], n, torch.tensor(c["start_q"] + [0.] * n, dtype=dtype), goals_t,
        cost, c["step_size"], c["temperature"],
        c["sigma_start_init"], c["sigma_start_sample"], c["sigma_goal_init"],
        c["sigma_goal_sample"], c["sigma_gp_init"], c["sigma_gp_sample"],
        seed=seed, dtype=dtype, eps_init=eps_init)
