"""Dense CPU restatement of the reference's Gauss-Newton planner `GPMP` (planner.py:352-661) and of the
`get_linear_system` methods it calls -- TEST INFRASTRUCTURE ONLY.

Every cost contributes rows (A, b, K): b the factor errors, A = -d b / d theta, K the factor weights;
the step solves (A^T K A + damping) d_theta = A^T K b and moves the particle means by
step_size * d_theta.  Pinned against the reference itself by tests/golden/g7_gpmp.npz
(oracle/gen_golden.py g7: the unmodified reference GPMP driven with oracle.fk as its FK callable).
"""
import torch

from . import ref_equiv as R


def linear_system_gp(trajs, start, n, dt, sigma_start, sigma_gp):
    """CostGP.get_linear_system -- cost_functions.py:148-168 with unary_factor.py:22-29 and
    gp_factor.py:54-67 (H1 = Phi, H2 = -I)."""
    B, T, d = trajs.shape
    dtype = trajs.dtype
    A = torch.zeros(B, d * T, d * T, dtype=dtype)
    b = torch.zeros(B, d * T, 1, dtype=dtype)
    K = torch.zeros(B, d * T, d * T, dtype=dtype)
    A[:, :d, :d] = torch.eye(d, dtype=dtype)
    b[:, :d, 0] = start - trajs[:, 0]
    K[:, :d, :d] = R.unary_K(d, sigma_start, dtype)
    Phi, Q_inv = R.phi_matrix(n, dt, dtype), R.q_inv_matrix(n, dt, sigma_gp, dtype)
    for i in range(T - 1):
        A[:, (i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] = Phi
        A[:, (i + 1) * d:(i + 2) * d, (i + 1) * d:(i + 2) * d] = -torch.eye(d, dtype=dtype)
        b[:, (i + 1) * d:(i + 2) * d, 0] = trajs[:, i + 1] - trajs[:, i] @ Phi.t()
        K[:, (i + 1) * d:(i + 2) * d, (i + 1) * d:(i + 2) * d] = Q_inv
    return A, b, K


def linear_system_goal_prior(trajs, goals, nppg, n, sigma):
    """CostGoalPrior.get_linear_system -- cost_functions.py:390-405."""
    B, T, d = trajs.shape
    dtype = trajs.dtype
    A = torch.zeros(B, d, d * T, dtype=dtype)
    A[:, :, -d:] = torch.eye(d, dtype=dtype)
    g = goals.repeat_interleave(nppg, dim=0)                       # particle p -> goal p // nppg
    b = (g - trajs[:, -1]).unsqueeze(-1)
    K = R.unary_K(d, sigma, dtype).repeat(B, 1, 1)
    return A, b, K


def composite_linear_system(trajs, systems):
    """CostComposite.get_linear_system -- cost_functions.py:60-85: rows stacked, K block-diagonal."""
    As, bs, Ks = zip(*systems)
    A, b = torch.cat(As, dim=1), torch.cat(bs, dim=1)
    m = A.shape[1]
    K = torch.zeros(trajs.shape[0], m, m, dtype=trajs.dtype)
    o = 0
    for Ki in Ks:
        K[:, o:o + Ki.shape[1], o:o + Ki.shape[1]] = Ki
        o += Ki.shape[1]
    return A, b, K


def grad_terms(A, b, K, delta, trust_region):
    """GPMP._get_grad_terms -- planner.py:607-624 (note: with trust_region the damping is delta times
    the diagonal of the PARTICLE-MEAN normal matrix)."""
    N = A.shape[2]
    I = torch.eye(N, dtype=A.dtype)
    AtK = A.transpose(1, 2) @ K
    AtA = AtK @ A
    if not trust_region:
        JtJ = AtA + delta * I
    else:
        JtJ = AtA + delta * (AtA.mean(0) * I)
    return JtJ, AtK @ b


def solve(JtJ, g, method="cholesky"):
    """GPMP.get_torch_solve -- planner.py:626-640.

    'inverse' and 'cholesky' return the solution of JtJ x = g.  The reference's 'cholesky' branch
    (planner.py:634-636) passes `upper=False` together with `l.mT` to its second triangular solve, so
    torch reads only the DIAGONAL of that factor and the branch returns diag(L)^-1 L^-1 g, not the
    solution; 'cholesky_reference_quirk' reproduces exactly that, for pinning this restatement against
    the reference run in tests/golden/g7_gpmp.npz.  The HIP planner solves the system properly
    (documented divergence)."""
    if method == "inverse":
        return torch.linalg.solve(JtJ, g)
    L = torch.linalg.cholesky(JtJ)
    z = torch.linalg.solve_triangular(L, g, upper=False)
    if method == "cholesky_reference_quirk":
        return torch.linalg.solve_triangular(L.mT, z, upper=False)
    return torch.linalg.solve_triangular(L.mT, z, upper=True)


class OracleGPMP:
    """One `_step` (planner.py:580-605) per call of step(); `systems_fn(means, **obs)` returns the list
    of (A, b, K) of the cost list."""

    def __init__(self, particle_means, systems_fn, step_size, delta, trust_region, method="cholesky"):
        self.particle_means = particle_means.clone()
        self.systems_fn, self.step_size = systems_fn, step_size
        self.delta, self.trust_region, self.method = delta, trust_region, method

    def step(self, **obs):
        P, T, d = self.particle_means.shape
        A, b, K = composite_linear_system(self.particle_means, self.systems_fn(self.particle_means, **obs))
        JtJ, g = grad_terms(A, b, K, self.delta, self.trust_region)
        d_theta = solve(JtJ, g, self.method).view(P, T, d)
        costs = (b.transpose(1, 2) @ K @ b).reshape(P)          # GPMP._get_costs, planner.py:642-644
        self.particle_means = self.particle_means + self.step_size * d_theta
        return d_theta, costs


def panda_systems_fn(c, T, nppg, goals, FK, sphere_field="rbf"):
    """The Panda cost list of tests/scenarios.py as linear systems (GP, goal prior, self, spheres)."""
    n = c["n_dof"]

    def fn(means, obstacle_spheres=None):
        start = torch.tensor(c["start_q"] + [0.] * n, dtype=means.dtype)
        out = [linear_system_gp(means, start, n, c["dt"], c["cost_sigma_start"], c["cost_sigma_gp"]),
               linear_system_goal_prior(means, goals, nppg, n, c["sigma_goal_prior"]),
               R.collision_linear_system(means, n, FK, lambda fr: R.field_self(fr, margin=c["self_margin"]),
                                         c["sigma_self"])]
        if obstacle_spheres is not None:
            out.append(R.collision_linear_system(
                means, n, FK, lambda fr: R.field_spheres(fr, obstacle_spheres, field_type=sphere_field),
                c["sigma_coll"]))
        return out
    return fn
