"""StochGPMP restated the way a careful CPU implementation would run it -- TEST / BENCH INFRASTRUCTURE ONLY.

`oracle.ref_equiv.OraclePlanner` follows the reference's dense algorithm op for op (replicated
[P,M,M] precision, torch MultivariateNormal rebuilt on every iteration, dense [M,M] sampling and
importance-sampling matmuls): that is the reference's CPU path and `bench.py`'s `cpu_baseline`.
This file is the SAME mathematics with the structure SURVEY.md 8(a) points out exploited, so that
the GPU number also has a *fair* CPU figure beside it (last row):

  * the sampling prior is factored ONCE (its precision never changes: planner.py:273 only moves
    the mean) -- mp_priors_multi.py:100-110,120-123;
  * every prior the public API can build is isotropic over the DOFs (unary_factor.py:19,
    gp_factor.py:25-27), so scale_tril is n identical 2T x 2T blocks under the DOF-major
    permutation: x = mu + L_2T eps_k per DOF, one [S P n, 2T] x [2T, 2T] GEMM;
  * the importance-sampling term x^T Sigma^-1 mu (planner.py:233-236) is a dot product with the
    vector Sigma^-1 mu, computed once per particle;
  * the cost terms are the oracle's own functions (same file:line citations), evaluated in particle
    chunks to bound memory.

It is pinned by tests/test_oracle_golden.py::test_banded_restatement_equals_dense_oracle (same noise
-> same costs and means as OraclePlanner to 1e-9) and never imported by the product.
"""
import torch
from torch.distributions import MultivariateNormal

from . import ref_equiv as R


class BandedPlanner:
    def __init__(self, nppg, S, T, dt, n, start, goals, cost, step_size, temperature,
                 sigma_start_sample, sigma_goal_sample, sigma_gp_sample, particle_means,
                 dtype=torch.float64, chunk=32):
        self.nppg, self.S, self.T, self.dt, self.n, self.d = nppg, S, T, dt, n, 2 * n
        self.G = 1 if goals is None else goals.shape[0]
        self.P = nppg * self.G
        self.cost, self.step_size, self.temperature = cost, step_size, temperature
        self.dtype, self.chunk = dtype, chunk
        self.particle_means = particle_means.detach().clone().to(dtype)          # [P,T,d]
        # per-DOF 2T x 2T problem (state order (t, pos), (t, vel)): same blocks as the full prior with n = 1
        Ks = R.unary_K(2, sigma_start_sample, torch.float64)
        Kg = None if goals is None else R.unary_K(2, sigma_goal_sample, torch.float64)
        Q = R.q_inv_matrix(1, dt, sigma_gp_sample, torch.float64)
        self.Sinv1 = R.dense_prior_precision(T, 1, dt, Ks, Q, Kg, torch.float64)            # [2T,2T]
        mvn = MultivariateNormal(torch.zeros(2 * T, dtype=torch.float64), precision_matrix=self.Sinv1)
        self.L1 = mvn._unbroadcasted_scale_tril.to(dtype)                                   # [2T,2T], once
        self.Sinv1 = self.Sinv1.to(dtype)
        self.state_samples = None
        self.weights = None

    def _per_dof(self, x):
        """[..., T, d] -> [..., n, 2T] with the (t, pos_k), (t, vel_k) interleave of the per-DOF problem."""
        n, T = self.n, self.T
        return torch.stack((x[..., :n], x[..., n:]), dim=-1).transpose(-3, -2).reshape(*x.shape[:-2], n, 2 * T)

    def _from_dof(self, y):
        """inverse of _per_dof: [..., n, 2T] -> [..., T, d]."""
        n, T = self.n, self.T
        y = y.reshape(*y.shape[:-1], T, 2).transpose(-3, -2)                 # [..., T, n, 2]
        return torch.cat((y[..., 0], y[..., 1]), dim=-1)

    def sample(self, eps):
        """eps [S,P,M] in torch.randn layout (multivariate_normal.py:250-253) -> samples [P,S,T,d]."""
        S, P, T, d = self.S, self.P, self.T, self.d
        e = self._per_dof(eps.to(self.dtype).view(S, P, T, d))               # [S,P,n,2T]
        y = self._from_dof(e @ self.L1.t())                                  # [S,P,T,d]
        return (self.particle_means.unsqueeze(0) + y).transpose(0, 1)

    def get_costs(self, **obs):
        P, S, T, d = self.P, self.S, self.T, self.d
        costs = torch.empty(P, S, dtype=self.dtype)
        for p0 in range(0, P, self.chunk):
            p1 = min(P, p0 + self.chunk)
            costs[p0:p1] = self._chunk_costs(p0, p1, **obs)
        # importance-sampling term: temperature * x . (Sigma^-1 mu), Sigma^-1 block-diagonal per DOF
        g = self._from_dof(self._per_dof(self.particle_means) @ self.Sinv1)  # Sinv1 symmetric -> [P,T,d]
        costs += self.temperature * (self.state_samples * g.unsqueeze(1)).sum((-1, -2))
        return costs

    def _chunk_costs(self, p0, p1, **obs):
        """The composite cost of particles [p0,p1): rows keep their GLOBAL index for the goal lookup
        (cost_functions.py:379-386), which the oracle's cost_goal_prior derives from the row number --
        so evaluate goal by goal."""
        x = self.state_samples[p0:p1]
        out = torch.empty(p1 - p0, self.S, dtype=self.dtype)
        # rows of one goal at a time so that the oracle's `row // (nppg*S)` goal lookup stays right
        g0, g1 = p0 // self.nppg, (p1 - 1) // self.nppg
        for g in range(g0, g1 + 1):
            a, b = max(p0, g * self.nppg), min(p1, (g + 1) * self.nppg)
            out[a - p0:b - p0] = self.cost(x[a - p0:b - p0], g, **obs).reshape(b - a, self.S)
        return out

    def step(self, eps, **obs):
        self.state_samples = self.sample(eps)
        costs = self.get_costs(**obs)
        self.weights = torch.softmax(-costs / self.temperature, dim=1).reshape(-1, self.S, 1, 1)
        grad = (self.weights * (self.state_samples - self.particle_means.unsqueeze(1))).sum(1)
        self.particle_means = self.particle_means + self.step_size * grad
        return costs, grad


def panda_chunk_cost(c, T, S, goals, field_type='rbf', dtype=torch.float64):
    """cost(x [p,S,T,d] of goal g, g, obstacle_spheres=...) -> [p*S] with the oracle's term functions
    (CostGP, CostGoalPrior for goal g, self and sphere link fields through oracle.fk)."""
    from .fk import fk_all_links
    n = c["n_dof"]
    start = torch.tensor(c["start_q"] + [0.] * n, dtype=dtype)
    goals = torch.as_tensor(goals, dtype=dtype)

    def cost(x, g, **obs):
        p = x.shape[0]
        trajs = x.reshape(-1, T, 2 * n)
        xt = fk_all_links(trajs.reshape(-1, 2 * n)[:, :n]).reshape(trajs.shape[0], T, -1, 4, 4)
        out = R.cost_gp(trajs, start, n, c["dt"], c["cost_sigma_start"], c["cost_sigma_gp"])
        out = out + R.cost_goal_prior(trajs, goals[g:g + 1], p, S, n, c["sigma_goal_prior"])
        out = out + R.cost_collision_links(xt, lambda f: R.field_self(f, margin=c["self_margin"]), c["sigma_self"])
        out = out + R.cost_collision_links(
            xt, lambda f: R.field_spheres(f, obs["obstacle_spheres"], field_type=field_type), c["sigma_coll"])
        return out
    return cost


def planar_chunk_cost(c, T, S, goals, grid, cell_size, c_offset, dtype=torch.float64):
    """The planar twin: cost(x [p,S,T,d] of goal g, g) -> [p*S] with the oracle's CostGP, CostGoalPrior (goal g) and
    occupancy-grid terms (cost_functions.py:128-146, 376-388, 247-261 + obst_map.py:164-182)."""
    import numpy as np
    n = c["n_dof"]
    start = torch.tensor(c["start"], dtype=dtype)
    goals = torch.as_tensor(goals, dtype=dtype)
    grid_t = torch.as_tensor(np.asarray(grid, dtype=np.float64)).to(dtype)
    off = torch.as_tensor(c_offset, dtype=dtype)

    def cost(x, g, **obs):
        p = x.shape[0]
        trajs = x.reshape(-1, T, 2 * n)
        out = R.cost_gp(trajs, start, n, c["dt"], c["cost_sigma_start"], c["cost_sigma_gp"])
        out = out + R.cost_goal_prior(trajs, goals[g:g + 1], p, S, n, c["sigma_goal_prior"])
        out = out + R.cost_collision_grid(trajs, n, grid_t, cell_size, off, c["sigma_coll"])
        return out
    return cost
