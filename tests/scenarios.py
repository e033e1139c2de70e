"""Problem definitions shared by the oracle tests, the HIP parity tests and bench.py.

Hyper-parameters are those of the reference examples (reference
examples/planar_environment.py:16-96, examples/panda_environment.py:52-121).  This module builds
ORACLE objects only; the HIP-side twins are built in the tests from the same dicts.
"""
import numpy as np
import torch

from oracle import ref_equiv as R

from stoch_gpmp_amd.workloads import PANDA, PLANAR, panda_spheres  # noqa: E402,F401


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
                      with_spheres=True, chain=None):
    from oracle.fk import fk_all_links
    if chain is not None:                                # any serial chain (same URDF semantics, oracle/fk.py)
        import functools
        fk_all_links = functools.partial(fk_all_links, chain=chain)
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
                         eps_init=None, goals=None, chain=None):
    n = c["n_dof"]
    goals_t = torch.tensor([c["goal_q"] + [0.] * n], dtype=dtype) if goals is None \
        else torch.as_tensor(goals, dtype=dtype)
    cost = oracle_panda_cost(c, T, nppg, S, dtype, field_type=field_type, goals=goals_t, chain=chain)
    return R.OraclePlanner(
        nppg, S, T, c["dt"], n, torch.tensor(c["start_q"] + [0.] * n, dtype=dtype), goals_t,
        cost, c["step_size"], c["temperature"],
        c["sigma_start_init"], c["sigma_start_sample"], c["sigma_goal_init"],
        c["sigma_goal_sample"], c["sigma_gp_init"], c["sigma_gp_sample"],
        seed=seed, dtype=dtype, eps_init=eps_init)
