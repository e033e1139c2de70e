"""Reference-equivalent torch-CPU restatement of the StochGPMP hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference
file:line (relative to /root/reference) whose arithmetic it restates.  The *algorithm* is the
reference's dense one on purpose -- replicated [P,M,M] precision, a fresh
`torch.distributions.MultivariateNormal` (flip-Cholesky + triangular solve) every iteration,
dense `scale_tril @ eps` sampling, dense importance-sampling matmul -- so that timing this module
on the GPU box's host cores is a fair stand-in for timing the reference there
(`bench.py`'s `cpu_baseline`, kind "port").

Differences from the reference that do not change results:
  * block-diagonal operators are assembled with index assignment instead of T-long
    `torch.block_diag` loops (same entries, so `A.t() @ Q @ A` is bit-identical);
  * noise may be injected (`eps=`) instead of drawn; when drawn it comes from
    `torch.randn(shape)` on the global generator, which is the same stream the reference's
    `MultivariateNormal.sample` consumes (torch `multivariate_normal.py:250-253`), verified
    bit-for-bit by tests/test_oracle_golden.py against captured reference runs.
"""
import math

import torch
from torch.distributions import MultivariateNormal


# ----------------------------------------------------------------------------- factors
def phi_matrix(n, dt, dtype):
    """Constant-velocity transition [[I, dt I],[0, I]] -- costs/factors/gp_factor.py:36-42."""
    phi = torch.eye(2 * n, dtype=dtype)
    phi[:n, n:] = torch.eye(n, dtype=dtype) * dt
    return phi


def q_inv_matrix(n, dt, sigma, dtype, Q_c_inv=None):
    """One-step GP precision -- gp_factor.py:25-27 (Q_c_inv = I/sigma^2) and :44-52."""
    if Q_c_inv is None:
        Q_c_inv = torch.eye(n, dtype=dtype) / sigma ** 2
    m1 = 12. * (dt ** -3.) * Q_c_inv
    m2 = -6. * (dt ** -2.) * Q_c_inv
    m3 = 4. * (dt ** -1.) * Q_c_inv
    return torch.cat((torch.cat((m1, m2), dim=-1), torch.cat((m2, m3), dim=-1)), dim=-2)


def unary_K(d, sigma, dtype):
    """Unary-factor weight I/sigma^2 -- costs/factors/unary_factor.py:19."""
    return torch.eye(d, dtype=dtype) / sigma ** 2


# ----------------------------------------------------------------------------- prior
def dense_prior_precision(T, n, dt, K_s, Q_inv, K_g, dtype):
    """Sigma^-1 = A^T blkdiag(K_s, Q^-1 x (T-1), [K_g]) A -- mp_priors_multi.py:170-202."""
    d = 2 * n
    M = T * d
    phi = phi_matrix(n, dt, dtype)
    A = torch.eye(M, dtype=dtype)
    for i in range(T - 1):                       # A[d:, :-d] += -blkdiag(Phi...)  (:185-186)
        A[(i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] += -1. * phi
    rows = M
    if K_g is not None:                          # goal row-block (:187-190)
        b = torch.zeros(d, M, dtype=dtype)
        b[:, -d:] = torch.eye(d, dtype=dtype)
        A = torch.cat((A, b))
        rows += d
    Q = torch.zeros(rows, rows, dtype=dtype)     # (:192-196)
    Q[:d, :d] = K_s
    for i in range(T - 1):
        Q[(i + 1) * d:(i + 2) * d, (i + 1) * d:(i + 2) * d] = Q_inv
    if K_g is not None:
        Q[M:, M:] = K_g
    return A.t() @ Q @ A                         # (:198)


def const_vel_mean(start, goal, dt, T, n, dtype):
    """Straight line start->goal, velocity (goal-start)/((T-1) dt) -- mp_priors_multi.py:130-144."""
    steps = T - 1
    traj = torch.zeros(T, 2 * n, dtype=dtype)
    vel = (goal[:n] - start[:n]) / (steps * dt)
    for i in range(T):
        traj[i, :n] = start[:n] * (steps - i) * 1. / steps + goal[:n] * i * 1. / steps
    traj[:, n:] = vel.unsqueeze(0)
    return traj


def planner_const_vel(start, goals, nppg, T, n, dt, dtype):
    """`initial_particle_means='const_vel'`: velocity uses /(T dt) -- planner.py:142-155."""
    G = goals.shape[0]
    out = torch.zeros(G, nppg, T, 2 * n, dtype=dtype)
    vel = (goals[:, :n] - start[:n]) / (T * dt)
    for i in range(T):
        interp = start[:n] * (T - i - 1) / (T - 1) + goals[:, :n] * i / (T - 1)
        out[:, :, i, :n] = interp.unsqueeze(1)
    out[:, :, :, n:] = vel.unsqueeze(1).unsqueeze(1)
    return out


class TrajPrior:
    """MultiMPPrior restated -- mp_priors_multi.py:14-128,204-207."""

    def __init__(self, T, n, dt, K_s, Q_inv, start, means=None, K_g=None, goals=None,
                 dtype=torch.float64):
        self.T, self.n, self.d, self.M = T, n, 2 * n, T * 2 * n
        self.dtype = dtype
        goal_directed = goals is not None
        if means is None:                         # (:72-79, :146-168)
            if goal_directed:
                means = torch.stack([const_vel_mean(start, goals[g], dt, T, n, dtype)
                                     for g in range(goals.shape[0])], dim=0)
            else:
                means = start.repeat(T, 1).unsqueeze(0)
        self.modes = means.shape[0]
        self.means = means.reshape(self.modes, -1)
        self.Sigma_inv = dense_prior_precision(T, n, dt, K_s, Q_inv,
                                               K_g if goal_directed else None, dtype)
        self.Sigma_invs = self.Sigma_inv.repeat(self.modes, 1, 1)     # (:97)
        self._rebuild()

    def _rebuild(self):                           # (:100-110) -> torch MVN: validation +
        self.dist = MultivariateNormal(self.means, precision_matrix=self.Sigma_invs)

    def set_mean(self, means_new):                # (:120-123)
        assert means_new.shape == self.means.shape
        self.means = means_new.clone().detach()
        self._rebuild()

    def set_Sigma_invs(self, Sigma_invs_new):     # (:125-128)
        assert Sigma_invs_new.shape == self.Sigma_invs.shape
        self.Sigma_invs = Sigma_invs_new.clone().detach()
        self._rebuild()

    def scale_tril(self):
        return self.dist._unbroadcasted_scale_tril

    def sample(self, num, eps=None):
        """-> [modes, num, T, d] (a transposed view, like the reference) (:204-207)."""
        if eps is None:
            x = self.dist.sample((num,))
        else:                                     # torch multivariate_normal.py:250-253
            assert eps.shape == (num, self.modes, self.M)
            L = self.dist._unbroadcasted_scale_tril
            x = self.dist.loc + torch.matmul(L, eps.unsqueeze(-1)).squeeze(-1)
        return x.view(num, self.modes, self.T, self.d).transpose(1, 0)


# ----------------------------------------------------------------------------- costs
def cost_gp(trajs, start, n, dt, sigma_start, sigma_gp, with_start=True):
    """CostGP.eval / CostGPTrajectory.eval -- costs/cost_functions.py:128-146, 202-215.

    trajs [B,T,d].  start term (start - x0)^T K_s (start - x0); GP term sum_i e_i^T Q^-1 e_i with
    e_i = x_{i+1} - Phi x_i (gp_factor.py:54-58)."""
    dtype = trajs.dtype
    d = 2 * n
    phi = phi_matrix(n, dt, dtype)
    Qi = q_inv_matrix(n, dt, sigma_gp, dtype)
    e = trajs[:, 1:, :] - trajs[:, :-1, :] @ phi.t()
    c = torch.einsum('bti,ij,btj->b', e, Qi, e)
    if with_start:
        e0 = start.to(dtype) - trajs[:, 0, :]
        c = c + torch.einsum('bi,ij,bj->b', e0, unary_K(d, sigma_start, dtype), e0)
    return c


def cost_goal_prior(trajs, goals, nppg, S, n, sigma):
    """CostGoalPrior.eval -- cost_functions.py:376-388; b = (g*nppg + k)*S + s."""
    dtype = trajs.dtype
    G = goals.shape[0]
    B, T, d = trajs.shape
    x = trajs.reshape(G, nppg * S, T, d)
    out = torch.zeros(G, nppg * S, dtype=dtype)
    K = unary_K(d, sigma, dtype)
    for g in range(G):
        err = goals[g].to(dtype) - x[g, :, -1, :]
        out[g] = torch.einsum('bi,ij,bj->b', err, K, err)
    return out.flatten()


def grid_lookup(X, grid, cell_size, c_offset):
    """ObstacleMap.get_collisions -- envs/obst_map.py:164-182.

    X [...,2]; grid [ny, nx] float; index = floor(X * (1/cell) + c_offset); x clamped by
    grid.shape[0]-1, y by grid.shape[1]-1 (as the reference does); value grid[iy, ix]."""
    occ = X * (1 / cell_size) + c_offset
    occ = occ.floor().int()
    ix = occ[..., 0].clamp(0, grid.shape[0] - 1)
    iy = occ[..., 1].clamp(0, grid.shape[1] - 1)
    return grid[iy.long(), ix.long()]


def cost_collision_grid(trajs, n, grid, cell_size, c_offset, sigma):
    """CostCollision.eval with an ObstacleMap field and no FK --
    cost_functions.py:241-261, field_factor.py:18-40 (waypoints 1..T-1, K = 1/sigma^2)."""
    B, T, d = trajs.shape
    pos = trajs[:, 1:T, :n].reshape(-1, n)
    err = grid_lookup(pos, grid, cell_size, c_offset).reshape(B, T - 1)
    return (1. / sigma ** 2) * err.sum(1)


def _link_points(link_tensor, num_interpolate, link_interpolate_range):
    """Positions of the link frames plus optional interpolated points -- fields.py:66-74."""
    pts = link_tensor[..., :3, -1]
    if num_interpolate > 0:
        alpha = torch.linspace(0, 1, num_interpolate + 2).type_as(pts)[1:num_interpolate + 1]
        alpha = alpha.view(tuple([1] * (pts.dim() - 2) + [-1, 1]))
        for i in range(link_interpolate_range[0], link_interpolate_range[1]):
            a, b = pts[..., i, :].unsqueeze(-2), pts[..., i + 1, :].unsqueeze(-2)
            pts = torch.cat([pts, a + (b - a) * alpha], dim=-2)
    return pts


def field_spheres(link_tensor, spheres, field_type='rbf', clamp_sdf=False, num_interpolate=0,
                  link_interpolate_range=(5, 7)):
    """LinkDistanceField.compute_cost -- costs/fields.py:63-86.  link_tensor [...,L,4,4];
    spheres [1,O,4] or [O,4] (cx,cy,cz,r)."""
    pts = _link_points(link_tensor, num_interpolate, link_interpolate_range).unsqueeze(-2)
    sph = spheres.unsqueeze(0)
    c, r = sph[..., :3], sph[..., 3]
    if field_type == 'rbf':
        return torch.exp(-0.5 * torch.square(pts - c).sum(-1) / torch.square(r)).sum((-1, -2))
    dist = torch.linalg.norm(pts - c, dim=-1)
    if field_type == 'sdf':
        sdf = -dist + r
        if clamp_sdf:
            sdf = sdf.clamp(max=0.)
        return sdf.max(-1)[0].max(-1)[0]
    if field_type == 'occupancy':
        return (dist < r).sum((-1, -2))
    raise ValueError(field_type)


def link_sphere_distances(link_tensor, spheres):
    """LinkDistanceField.distances -- fields.py:40-46: |p_l - c_o| - r_o, [..,L,O] (no interpolated points)."""
    pos = link_tensor[..., :3, -1].unsqueeze(-2)
    sph = spheres.reshape(-1, 4)
    return torch.linalg.norm(pos - sph[:, :3], dim=-1) - sph[:, 3]


def link_sphere_collision(link_tensor, spheres, buffer=0.02):
    """LinkDistanceField.compute_collision -- fields.py:48-54."""
    return (link_sphere_distances(link_tensor, spheres) < buffer).any(-1).any(-1)


def link_sphere_distance_sum(link_tensor, spheres):
    """LinkDistanceField.compute_distance -- fields.py:56-61."""
    return link_sphere_distances(link_tensor, spheres).sum((-1, -2))


def link_self_distances(link_tensor):
    """LinkSelfDistanceField.distances -- fields.py:100-102: |p_i - p_j|, [..,L,L]."""
    pos = link_tensor[..., :3, -1]
    return torch.linalg.norm(pos.unsqueeze(-2) - pos.unsqueeze(-3), dim=-1)


def link_self_collision(link_tensor, buffer=0.05):
    """LinkSelfDistanceField.compute_collision -- fields.py:104-108: pairs at least two links apart (tril, diagonal -2)."""
    return torch.tril(link_self_distances(link_tensor) < buffer, diagonal=-2).any(-1).any(-1)


def link_self_distance_sum(link_tensor):
    """LinkSelfDistanceField.compute_distance -- fields.py:110-112."""
    return link_self_distances(link_tensor).sum((-1, -2))


def xy_grid(xlim, ylim, x_dim, y_dim):
    """ObstacleMap.get_xy_grid -- obst_map.py:158-162."""
    xv, yv = torch.meshgrid([torch.linspace(xlim[0], xlim[1], x_dim), torch.linspace(ylim[0], ylim[1], y_dim)],
                            indexing="ij")
    return torch.stack((xv, yv), dim=2)


def field_self(link_tensor, margin=0.03, num_interpolate=0, link_interpolate_range=(5, 7)):
    """LinkSelfDistanceField.compute_cost -- fields.py:114-124 (full LxL sum, diagonal included)."""
    pts = _link_points(link_tensor, num_interpolate, link_interpolate_range)
    d2 = torch.square(pts.unsqueeze(-2) - pts.unsqueeze(-3)).sum(-1)
    return torch.exp(d2 / (-margin ** 2 * 2)).sum((-1, -2))


def cost_collision_links(x_trajs, field_fn, sigma):
    """CostCollision.eval with FK link frames -- cost_functions.py:247-261, field_factor.py:28-32.
    x_trajs [B,T,L,4,4]; field_fn maps [B,T-1,L,4,4] -> [B,T-1]."""
    B, T = x_trajs.shape[:2]
    err = field_fn(x_trajs[:, 1:T]).reshape(B, T - 1)
    return (1. / sigma ** 2) * err.sum(1)


def se3_distance(H, H_target, w_pos=1., w_rot=1.):
    """w_pos |p - p*| + w_rot angle(R*^T R).  PARITY UNPINNED: the reference calls
    torch_robotics' SE3_distance (fields.py:4,143-144), a dependency that is neither vendored nor
    version-pinned (setup.py) -- this is the definition this build documents."""
    dp = (H[..., :3, 3] - H_target[..., :3, 3]).norm(dim=-1)
    Rr = H_target[..., :3, :3].transpose(-1, -2) @ H[..., :3, :3]
    c = ((Rr.diagonal(dim1=-2, dim2=-1).sum(-1) - 1.) * 0.5).clamp(-1., 1.)
    return w_pos * dp + w_rot * torch.acos(c)


def field_ee_se3(link_tensor, target_H, w_pos=1., w_rot=1., square=True):
    """EESE3DistanceField.compute_cost -- fields.py:141-149 (last link frame only)."""
    dist = se3_distance(link_tensor[..., -1, :, :], target_H, w_pos, w_rot)
    return torch.square(dist) if square else dist


def cost_goal_ee(x_trajs, field_fn, sigma):
    """CostGoal.eval -- cost_functions.py:304-321: the field on the LAST waypoint's frames only."""
    B, T = x_trajs.shape[:2]
    err = field_fn(x_trajs[:, T - 1:T]).reshape(B, 1)
    return (1. / sigma ** 2) * err.sum(1)


def field_error_and_jacobian(q_trajs, n, traj_range, FK, field_fn):
    """FieldFactor.get_error(calc_jacobian=True) -- field_factor.py:28-38: error [B,len] and
    H = -d(error.sum())/d q restricted to the waypoint range and the position columns, by autograd
    through the FK callable exactly as the reference does."""
    a, b = traj_range
    q_trajs = q_trajs.detach().clone().requires_grad_(True)
    B, T, d = q_trajs.shape
    x_trajs = FK(q_trajs.reshape(-1, d)[:, :n]).reshape(B, T, -1, 4, 4)
    error = field_fn(x_trajs[:, a:b]).reshape(B, b - a)
    H = -torch.autograd.grad(error.sum(), q_trajs)[0][:, a:b, :n]
    return error.detach(), H


def collision_linear_system(q_trajs, n, FK, field_fn, sigma):
    """CostCollision.get_linear_system -- cost_functions.py:263-279."""
    B, T, d = q_trajs.shape
    err, H = field_error_and_jacobian(q_trajs, n, (1, T), FK, field_fn)
    A = torch.zeros(B, T - 1, d * T, dtype=q_trajs.dtype)
    for i in range(T - 1):
        A[:, i, (i + 1) * d:(i + 1) * d + n] = H[:, i]
    K = (1. / sigma ** 2) * torch.eye(T - 1, dtype=q_trajs.dtype).repeat(B, 1, 1)
    return A, err.unsqueeze(-1), K


def goal_ee_linear_system(q_trajs, n, FK, field_fn, sigma):
    """CostGoal.get_linear_system -- cost_functions.py:323-337: one row per trajectory, the field on the last
    waypoint; Jacobian by autograd through FK as FieldFactor.get_error does (field_factor.py:34-38)."""
    B, T, d = q_trajs.shape
    err, H = field_error_and_jacobian(q_trajs, n, (T - 1, T), FK, field_fn)
    A = torch.zeros(B, 1, d * T, dtype=q_trajs.dtype)
    A[:, :, (T - 1) * d:(T - 1) * d + n] = H
    K = (1. / sigma ** 2) * torch.ones(B, 1, 1, dtype=q_trajs.dtype)
    return A, err.unsqueeze(-1), K


class CompositeCost:
    """CostComposite.eval -- cost_functions.py:47-58.  `terms` is a list of callables
    term(trajs[B,T,d], x_trajs or None, **obs) -> [B], summed in list order."""

    def __init__(self, n, T, terms, FK=None):
        self.n, self.T, self.d, self.terms, self.FK = n, T, 2 * n, terms, FK

    def eval(self, trajs, **obs):
        trajs = trajs.reshape(-1, self.T, self.d)
        B = trajs.shape[0]
        x_trajs = None
        if self.FK is not None:
            x_trajs = self.FK(trajs.view(-1, self.d)[:, :self.n]).reshape(B, self.T, -1, 4, 4)
        costs = 0
        for term in self.terms:
            costs = costs + term(trajs, x_trajs, **obs)
        return costs


# ----------------------------------------------------------------------------- planner
class OraclePlanner:
    """StochGPMP restated -- planner.py:18-348.  `cost` is any object with .eval(trajs, **obs)."""

    def __init__(self, nppg, S, T, dt, n, start, goals, cost, step_size, temperature,
                 sigma_start_init, sigma_start_sample, sigma_goal_init, sigma_goal_sample,
                 sigma_gp_init, sigma_gp_sample, initial_particle_means=None, seed=None,
                 dtype=torch.float64, eps_init=None):
        if seed is not None:
            torch.manual_seed(seed)               # planner.py:48-49 (global generator)
        self.nppg, self.S, self.T, self.dt, self.n, self.d = nppg, S, T, dt, n, 2 * n
        self.dtype = dtype
        self.start = start.detach().clone().to(dtype)
        self.goals = None if goals is None else goals.detach().clone().to(dtype)
        self.G = 1 if goals is None else goals.shape[0]
        self.P = nppg * self.G
        self.cost, self.step_size, self.temperature = cost, step_size, temperature
        d = self.d
        Ks_i, Ks_s = unary_K(d, sigma_start_init, dtype), unary_K(d, sigma_start_sample, dtype)
        Kg_i = None if goals is None else unary_K(d, sigma_goal_init, dtype)
        Kg_s = None if goals is None else unary_K(d, sigma_goal_sample, dtype)
        Q_i = q_inv_matrix(n, dt, sigma_gp_init, dtype)
        Q_s = q_inv_matrix(n, dt, sigma_gp_sample, dtype)
        # reset(): planner.py:181-227
        if initial_particle_means is None:
            init = TrajPrior(T, n, dt, Ks_i, Q_i, self.start, K_g=Kg_i, goals=self.goals,
                             dtype=dtype)
            self.init_means = init.means.clone()
            self.init_Sigma_inv = init.Sigma_inv
            self.init_scale_tril = init.scale_tril()[0].clone()
            pm = init.sample(nppg, eps=eps_init).to(dtype)       # [G,nppg,T,d]
        elif isinstance(initial_particle_means, str) and initial_particle_means == 'const_vel':
            pm = planner_const_vel(self.start, self.goals, nppg, T, n, dt, dtype)
        else:
            pm = initial_particle_means
        self.particle_means = pm.flatten(0, 1).clone()           # [P,T,d] (:215); own storage
        self.prior = TrajPrior(T, n, dt, Ks_s, Q_s, self.start, means=self.particle_means,
                               K_g=Kg_s, goals=self.goals, dtype=dtype)
        self.Sigma_inv = self.prior.Sigma_inv
        self.state_samples = None
        self.weights = None

    def draw_discarded(self, eps=None):
        """The throw-away draw at the end of reset() -- planner.py:227."""
        self.state_samples = self.prior.sample(self.S, eps=eps)

    def get_costs(self, **obs):                   # planner.py:229-237
        P, S, M = self.P, self.S, self.T * self.d
        costs = self.cost.eval(self.state_samples, **obs).reshape(P, S)
        V = self.state_samples.reshape(P, S, M)
        U = self.particle_means.view(P, 1, M)
        costs = costs + self.temperature * (V @ self.Sigma_inv @ U.transpose(1, 2)).squeeze(2)
        return costs

    def sample_and_eval(self, eps=None, **obs):   # planner.py:239-261
        self.state_samples = self.prior.sample(self.S, eps=eps)
        return self.get_costs(**obs)

    def update(self, costs, samples):             # planner.py:263-275
        self.weights = torch.softmax(-costs / self.temperature, dim=1).reshape(-1, self.S, 1, 1)
        grad = (self.weights * (samples - self.particle_means.unsqueeze(1))).sum(1)
        self.particle_means.add_(self.step_size * grad)
        self.prior.set_mean(self.particle_means.view(self.P, -1))
        return grad

    def step(self, eps=None, **obs):              # one body of planner.py:289-299
        costs = self.sample_and_eval(eps=eps, **obs)
        grad = self.update(costs, self.state_samples)
        return costs, grad
