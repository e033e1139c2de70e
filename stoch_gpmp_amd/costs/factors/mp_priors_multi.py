"""MultiMPPrior -- same interface as reference costs/factors/mp_priors_multi.py, backed by the HIP
prior factor (K1) and sampler (K2).

The reference materialises Sigma^-1 [M,M], replicates it per mode and lets torch's
MultivariateNormal re-factor it on every `set_mean`; here the factor is the T pairs of d x d scan
blocks K1 emits once, `set_mean` only swaps the mean pointer, and `sample` is the O(T d) scan.
`Sigma_inv` is assembled densely from K1's four distinct blocks on first access, for API parity.

`PlannerPrior` (below) is the same object bound to a planner's own context: `StochGPMP._sample_dist` / `_init_dist`
(planner.py:206-227).  `planner._sample_dist.set_Sigma_invs(new)` then makes the planner's loop sample particle p from
precision p while its importance-sampling term keeps `planner.Sigma_inv`, as in the reference.
"""
import torch

from ... import _lib as L
from ...engine import Engine


class MultiMPPrior:

    _which = L.PRIOR_SAMPLE

    def __init__(self, num_steps, dt, state_dim, dof, K_s_inv, K_gp_inv, start_state, means=None,
                 K_g_inv=None, goal_states=None, use_numpy=False, tensor_args=None, seed=0):
        self.state_dim, self.dof, self.num_steps = state_dim, dof, num_steps
        self.use_numpy = use_numpy
        self.M = state_dim * (num_steps + 1)
        self.tensor_args = tensor_args
        self.dt = dt
        self.goal_directed = goal_states is not None
        T = num_steps + 1
        if means is None:                                            # mp_priors_multi.py:72-79
            self.num_modes = goal_states.shape[0] if self.goal_directed else 1
            means = self.get_const_vel_mean(start_state, goal_states, dt, num_steps, dof)
        else:
            self.num_modes = means.shape[0]
        self.means = means.reshape(self.num_modes, -1).contiguous()
        # K_s, K_g are I/sigma^2 in the reference's API (unary_factor.py:19); recover the sigmas
        sigma_start = float(K_s_inv[0, 0]) ** -0.5
        sigma_goal = float(K_g_inv[0, 0]) ** -0.5 if (self.goal_directed and K_g_inv is not None) else None
        # Q^-1 = [[12 dt^-3 Qc, .], ..] (gp_factor.py:44-52)  ->  Qc^-1 = Q^-1[:n,:n] dt^3 / 12
        qc = (K_gp_inv[:dof, :dof].detach().double().cpu() / (12. * dt ** -3.))
        iso = torch.allclose(qc, torch.eye(dof, dtype=torch.float64) * qc[0, 0], rtol=1e-12, atol=0)
        self._engine = Engine(dof, T, self.num_modes, 1, tensor_args=tensor_args)
        if iso:
            self._engine.set_prior(L.PRIOR_SAMPLE, dt, sigma_start, float(qc[0, 0]) ** -0.5, sigma_goal)
        else:
            self._engine.set_prior(L.PRIOR_SAMPLE, dt, sigma_start, None, sigma_goal, Q_c_inv=qc)
        self._seed, self._draw = seed, 0
        self._Sigma_inv = None
        self._Sigma_invs = None          # per-mode precisions after set_Sigma_invs
        self._logdet = None              # log det Sigma_m^-1 per mode (from K1's factor)

    # ---- reference API
    @property
    def Sigma_inv(self):
        if self._Sigma_inv is None:
            blocks, _, _ = self._engine.get_prior(self._which, blocks_only=True)
            d, T = self.state_dim, self.num_steps + 1
            S = torch.zeros(self.M, self.M, dtype=torch.float64)
            for t in range(T):
                S[t * d:(t + 1) * d, t * d:(t + 1) * d] = blocks[0 if t == 0 else (2 if t == T - 1 else 1)]
                if t + 1 < T:
                    S[(t + 1) * d:(t + 2) * d, t * d:(t + 1) * d] = blocks[3]
                    S[t * d:(t + 1) * d, (t + 1) * d:(t + 2) * d] = blocks[3].t()
            self._Sigma_inv = S.to(**self.tensor_args)
        return self._Sigma_inv

    @property
    def Sigma_invs(self):
        if self._Sigma_invs is not None:
            return self._Sigma_invs
        return self.Sigma_inv.unsqueeze(0).expand(self.num_modes, -1, -1)

    def get_mean(self, reshape=True):
        if reshape:
            return self.means.clone().detach().reshape(self.num_modes, self.num_steps + 1, self.state_dim)
        return self.means.clone().detach()

    def set_mean(self, means_new):
        assert means_new.shape == self.means.shape
        self.means = means_new.clone().detach().contiguous()

    def set_Sigma_invs(self, Sigma_invs_new):
        """mp_priors_multi.py:125-128: one precision matrix per mode.  The reference hands the dense
        [modes,M,M] tensor to torch's MultivariateNormal (a dense Cholesky per mode); here K1 factors each
        matrix on its block-tridiagonal structure (d x d blocks), which is what every trajectory prior of
        this family has -- a matrix with weight outside that band is refused with ValueError, and so is
        one that is not symmetric positive definite (torch raises ValueError there too)."""
        assert Sigma_invs_new.shape == self.Sigma_invs.shape
        new = Sigma_invs_new.clone().detach()
        d, T, modes = self.state_dim, self.num_steps + 1, self.num_modes
        S = new.double().cpu().reshape(modes, T, d, T, d).permute(0, 1, 3, 2, 4)        # [m, ti, tj, d, d]
        idx = torch.arange(T)
        D = S[:, idx, idx]                                                               # [m, T, d, d]
        E = S[:, idx[1:], idx[:-1]]                                                      # [m, T-1, d, d]
        band = torch.zeros(T, T, dtype=torch.bool)
        band[idx, idx] = True
        band[idx[1:], idx[:-1]] = True
        band[idx[:-1], idx[1:]] = True
        scale = float(S.abs().max())
        if float(S[:, ~band].abs().max() if T > 2 else 0.) > 1e-12 * scale:
            raise ValueError("set_Sigma_invs: precision matrices must be block tridiagonal in the d x d waypoint blocks")
        if float((S - S.permute(0, 2, 1, 4, 3)).abs().max()) > 1e-9 * scale:
            raise ValueError("set_Sigma_invs: precision matrices must be symmetric")
        self._engine.set_prior_blocks(self._which, D, E)                                 # ValueError if not PD
        self._Sigma_invs = new
        self._logdet = None

    def update_dist(self, means, Sigma_invs):
        """mp_priors_multi.py:100-110.  The reference rebuilds a torch MultivariateNormal here (a dense Cholesky of every
        mode's precision, on every set_mean); this object's "distribution" is (means, K1's factor): the means are taken
        over, and the factor is rebuilt only when `Sigma_invs` is not what it was built from."""
        self.means = means.reshape(self.num_modes, -1).contiguous()
        current = self._Sigma_invs
        if Sigma_invs is current or (current is None and Sigma_invs is self.Sigma_invs):
            return
        if current is None and Sigma_invs.shape == self.Sigma_invs.shape and bool((Sigma_invs == self.Sigma_inv.unsqueeze(0)).all()):
            return                                           # the shared precision, replicated (what __init__ passes)
        self.set_Sigma_invs(Sigma_invs)

    def get_const_vel_covariance(self, dt, K_s_inv, K_gp_inv, K_g_inv, precision_matrix=True):
        """mp_priors_multi.py:170-202: Sigma^-1 = A^T blkdiag(K_s, Q^-1 x (T - 1), K_g) A as a dense [M, M] matrix (or its
        inverse).  The reference forms A and the block-diagonal by T-long loops of torch.block_diag and two dense products;
        the result is block tridiagonal with D_0 = K_s + Phi^T Q^-1 Phi, D_i = Q^-1 + Phi^T Q^-1 Phi, D_last = Q^-1 (+ K_g),
        sub-diagonal blocks -Q^-1 Phi (a3) -- assembled here from those four blocks, the same ones K1 factors."""
        d, T, n = self.state_dim, self.num_steps + 1, self.dof
        ta = self.tensor_args
        Phi = torch.eye(d, **ta)
        Phi[:n, n:] = torch.eye(n, **ta) * dt
        Q = K_gp_inv.to(**ta)
        PtQ = Phi.t() @ Q
        PtQP = PtQ @ Phi
        E = -(Q @ Phi)                                       # block (i + 1, i)
        S = torch.zeros(self.M, self.M, **ta)
        for t in range(T):
            blk = (K_s_inv.to(**ta) if t == 0 else Q) + (PtQP if t < T - 1 else 0.)
            if t == T - 1 and self.goal_directed and K_g_inv is not None:
                blk = blk + K_g_inv.to(**ta)
            S[t * d:(t + 1) * d, t * d:(t + 1) * d] = blk
            if t + 1 < T:
                S[(t + 1) * d:(t + 2) * d, t * d:(t + 1) * d] = E
                S[t * d:(t + 1) * d, (t + 1) * d:(t + 2) * d] = E.t()
        return S if precision_matrix else torch.inverse(S)

    def const_vel_trajectory(self, start_state, goal_state, dt, num_steps, dof):
        traj = torch.zeros(num_steps + 1, 2 * dof, **self.tensor_args)
        mean_vel = (goal_state[:dof] - start_state[:dof]) / (num_steps * dt)
        for i in range(num_steps + 1):
            traj[i, :dof] = start_state[:dof] * (num_steps - i) * 1. / num_steps + goal_state[:dof] * i * 1. / num_steps
        traj[:, dof:] = mean_vel.unsqueeze(0)
        return traj

    def get_const_vel_mean(self, start_state, goal_states, dt, num_steps, dof):
        if self.goal_directed:
            return torch.stack([self.const_vel_trajectory(start_state, goal_states[i], dt, num_steps, dof)
                                for i in range(self.num_modes)], dim=0)
        return start_state.repeat(num_steps + 1, 1).unsqueeze(0)

    def sample(self, num_samples, eps=None):
        """-> [num_modes, num_samples, T, state_dim] (contiguous; the reference returns a
        transposed view of [num_samples, num_modes, ...] with the same logical layout)."""
        out = self._engine.sample(self._which, self._seed, self._draw, self.means.view(
            self.num_modes, self.num_steps + 1, self.state_dim), num_samples, eps=eps)
        self._draw += 1
        return out

    def _log_dets(self):
        """log det Sigma_m^-1 per mode from K1's factor: Sigma^-1 = L_inv^T L_inv with lower-triangular
        diagonal blocks B_t, and K1 emits G_t = B_t^-1."""
        if self._logdet is None:
            per_mode = self._Sigma_invs is not None
            _, G, _ = self._engine.get_prior(self._which, n_modes=self.num_modes if per_mode else None)
            G = G.reshape(-1, self.num_steps + 1, self.state_dim, self.state_dim)
            ld = -2. * torch.log(torch.diagonal(G, dim1=-2, dim2=-1)).sum((-1, -2))       # [modes] or [1]
            self._logdet = ld.expand(self.num_modes).clone()
        return self._logdet

    def log_prob(self, x):
        """mp_priors_multi.py:209-210 -> torch MultivariateNormal.log_prob: x [..., modes, M] -> [..., modes]
        = -1/2 (x - mu)^T Sigma^-1 (x - mu) - M/2 log(2 pi) + 1/2 log det Sigma^-1, with the quadratic form
        evaluated on the block-tridiagonal precision by `prior_quadform_kernel` (fp64 accumulation)."""
        import math
        assert x.shape[-1] == self.M and x.shape[-2] == self.num_modes
        lead = x.shape[:-1]
        xr = x.reshape(-1, self.M).to(**self.tensor_args).contiguous()
        q = self._engine.prior_quadform(self._which, xr, self.means)
        ld = self._log_dets().to(q.device)
        out = -0.5 * q.reshape(-1, self.num_modes) - 0.5 * self.M * math.log(2. * math.pi) + 0.5 * ld
        return out.reshape(lead).to(self.tensor_args['dtype'])


class PlannerPrior(MultiMPPrior):
    """A planner's live trajectory distribution: `StochGPMP._sample_dist` (and, during reset(), `_init_dist`), reference
    planner.py:206-227 -- the MultiMPPrior interface over the PLANNER'S OWN context and tensors instead of a context of its
    own: `means` are the planner's particle means (this rank's shard: one mode per local particle), the factor is the
    planner's K1 output, `sample` draws with the planner's noise key and draw counter.

    `set_Sigma_invs(new)` (mp_priors_multi.py:125-128) gives every particle its own precision: the planner's steps then
    sample particle p from factor p (csrc/sampler.hip: sample_dense_kernel on the matrix cores; such steps run sampler, sweep
    and update as separate launches) while the importance-sampling term keeps `planner.Sigma_inv`, which the reference
    captures at reset and never updates (planner.py:226,233-236).  reset() returns to the shared prior."""

    def __init__(self, planner, which, means=None):
        self._planner = planner
        self._which = which
        self._engine = planner._engine
        self.state_dim, self.dof, self.num_steps = planner.d_state_opt, planner.n_dof, planner.traj_len - 1
        self.M = self.state_dim * planner.traj_len
        self.tensor_args = planner.tensor_args
        self.dt = planner.dt
        self.use_numpy = False
        self.goal_directed = planner.goal_directed
        self._fixed_means = None if means is None else means.reshape(means.shape[0], -1).contiguous()
        self._Sigma_inv = None
        self._Sigma_invs = None
        self._logdet = None

    # (the planner's means ARE this distribution's means: no copy to keep in step)
    @property
    def means(self):
        if self._fixed_means is not None:
            return self._fixed_means
        pm = self._planner.particle_means
        return pm.view(pm.shape[0], -1)

    @means.setter
    def means(self, new):
        if self._fixed_means is not None:
            self._fixed_means = new.reshape(self._fixed_means.shape).contiguous()
            return
        pm = self._planner.particle_means
        if new.data_ptr() != pm.data_ptr():
            pm.copy_(new.reshape(pm.shape))              # (in place: torch's version counter tells the planner its prepared IS weights are stale)

    @property
    def num_modes(self):
        return self.means.shape[0]

    def set_mean(self, means_new):
        assert means_new.shape == self.means.shape
        self.means = means_new

    def set_Sigma_invs(self, Sigma_invs_new):
        super().set_Sigma_invs(Sigma_invs_new)
        self._planner._Sigma_invs_set = True

    def sample(self, num_samples, eps=None):
        """-> [num_modes, num_samples, T, state_dim] drawn with the planner's noise key (seed, draw, GLOBAL particle)."""
        pl = self._planner
        T, d = pl.traj_len, pl.d_state_opt
        local = self._fixed_means is None
        out = self._engine.sample(self._which, pl.seed, pl._draw, self.means.view(self.num_modes, T, d), num_samples, eps=eps,
                                  eps_mode_offset=pl.p0 if (eps is not None and local) else 0, mode_offset=pl.p0 if local else 0)
        pl._draw += 1
        return out
