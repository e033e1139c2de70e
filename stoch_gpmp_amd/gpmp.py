"""`GPMP` -- the Gauss-Newton planner of reference `stoch_gpmp/planner.py:352-661`, same constructor,
`reset`, `optimize` and return values.

The reference stacks every factor into dense A, b, K, forms the [P, N, N] normal matrix (N = T d) and
solves it densely.  Here one call pair does a step on the GPU: `sgpmp_gpmp_linearize` (analytic link-
field Jacobians at every waypoint) and `sgpmp_gpmp_solve` (per particle, block-tridiagonal assembly
and block Cholesky on the fp64 matrix cores, means updated in place) -- see csrc/gpmp.hip.

Divergence, on purpose: `solver_params['method'] = 'cholesky'` SOLVES the system here.  The reference's
branch passes `upper=False` with `l.mT` to its second triangular solve (planner.py:634-636), which makes
torch read only the diagonal of the factor, so it returns diag(L)^-1 L^-1 g instead of the solution;
its 'inverse' branch is correct and is what both methods reproduce here (oracle/gpmp_equiv.py pins
both behaviours against a reference run).
"""
import time

import torch

from . import dist as D
from .planner import StochGPMP, print_info


class GPMP(StochGPMP):
    _discard_draw_at_reset = False

    def __init__(self, num_particles_per_goal, traj_len, opt_iters, dt=None, n_dof=None, step_size=1.,
                 temperature=1., start_state=None, multi_goal_states=None, initial_particle_means=None,
                 cost=None, sigma_start_init=None, sigma_start_sample=None, sigma_goal_init=None,
                 sigma_goal_sample=None, sigma_goal=None, sigma_gp_init=None, sigma_gp_sample=None,
                 seed=None, solver_params=None, tensor_args=None, **kwargs):
        self.sigma_goal = sigma_goal
        self.solver_params = dict(delta=0., trust_region=False, method='cholesky')
        self.solver_params.update(solver_params or {})
        if self.solver_params['method'] not in ('inverse', 'cholesky'):
            raise NotImplementedError                       # planner.py:639-640
        self.N = 2 * n_dof * traj_len
        self.costs = None
        if cost is None or not hasattr(cost, "compile_into"):
            raise TypeError("GPMP needs a stoch_gpmp_amd CostComposite (its factors are linearised on the GPU)")
        super().__init__(num_particles_per_goal, 1, traj_len, opt_iters, dt=dt, n_dof=n_dof,
                         step_size=step_size, temperature=temperature, start_state=start_state,
                         multi_goal_states=multi_goal_states, initial_particle_means=initial_particle_means,
                         cost=cost, sigma_start_init=sigma_start_init, sigma_start_sample=sigma_start_sample,
                         sigma_goal_init=sigma_goal_init, sigma_goal_sample=sigma_goal_sample,
                         sigma_gp_init=sigma_gp_init, sigma_gp_sample=sigma_gp_sample, seed=seed,
                         tensor_args=tensor_args, **kwargs)

    def get_dist(self, start_K, gp_K, goal_K, state_init, particle_means=None, goal_states=None):
        """planner.py:479-501: a stand-alone MultiMPPrior of this problem (the same object StochGPMP.get_prior_dist builds)."""
        return self.get_prior_dist(start_K, gp_K, goal_K, state_init, particle_means=particle_means, goal_states=goal_states)

    def get_torch_solve(self, A, b, method):
        """planner.py:619-633 for callers that hold a dense system: 'inverse' = torch.linalg.solve; 'cholesky' SOLVES it too
        (the reference's second triangular solve reads only the factor's diagonal, see the module docstring).  The planner's
        own step never forms the dense [P, N, N] system: it solves the block-tridiagonal one on the GPU (csrc/gpmp.hip)."""
        if method == 'inverse':
            return torch.linalg.solve(A, b)
        if method == 'cholesky':
            return torch.cholesky_solve(b, torch.linalg.cholesky(A))
        raise NotImplementedError

    def reset(self, start_state=None, multi_goal_states=None, initial_particle_means=None):
        super().reset(start_state, multi_goal_states, initial_particle_means=initial_particle_means)
        T, d = self.traj_len, self.d_state_opt
        dev = self.tensor_args['device']
        self._d_theta = torch.empty(self.num_particles_local, T, d, **self.tensor_args)
        self._gn_costs = torch.empty(self.num_particles_local, **self.tensor_args)
        self._diag_sum = torch.zeros(T * d, device=dev, dtype=torch.float64)

    # ------------------------------------------------------------------------------- the loop
    def _step(self, **observation):
        """planner.py:580-605: linearise the cost list at the particle means, solve the damped normal
        equations per particle, move the means by step_size * d_theta.  Returns (d_theta, costs) where
        costs = b^T K b at the linearisation point (what the reference's `_get_costs(b, K)` gives)."""
        eng = self._engine
        cv = self.cost.version()
        if cv != self._cost_version:                     # a field / cost was edited (the reference reads them live)
            self.cost.compile_into(eng)
            self._cost_version = cv
        trust = bool(self.solver_params['trust_region'])
        diag = self._diag_sum if trust else None
        if self.num_particles_local > 0:
            eng.gpmp_linearize(self.particle_means, spheres=self._spheres(observation), diag_sum=diag)
        elif trust:
            self._diag_sum.zero_()
        if trust and self.world_size > 1 and torch.distributed.is_initialized():
            # the damping is delta * diag(mean over ALL particles of A^T K A) (planner.py:618-622)
            D.dist.all_reduce(self._diag_sum, op=D.dist.ReduceOp.SUM, group=self.process_group)
        if self.num_particles_local > 0:
            eng.gpmp_solve(self.particle_means, self.solver_params['delta'], self.step_size, diag_sum=diag,
                           d_theta=self._d_theta, costs=self._gn_costs)
        return self._d_theta, self._gn_costs

    def step(self, **observation):
        return self._step(**observation)

    def optimize(self, opt_iters=None, debug=False, **observation):
        """planner.py:547-578 -> (velocity means [P,T,n], position means [P,T,n], costs [P])."""
        if opt_iters is None:
            opt_iters = self.opt_iters
        start_time = time.time()
        costs = self._gn_costs
        for opt_step in range(opt_iters):
            t_iter = time.time()
            _, costs = self._step(**observation)
            if debug and opt_step % 50 == 0:
                print_info(opt_step, opt_iters, t_iter, start_time, costs)
        self.costs = costs
        n = self.n_dof
        position_seq_mean = self.particle_means[..., :n].clone()
        velocity_seq_mean = self.particle_means[..., -n:].clone()
        self._recent_control_particles = velocity_seq_mean
        self._recent_state_trajectories = position_seq_mean
        return velocity_seq_mean, position_seq_mean, costs.clone()

    def _get_costs(self, errors, w_mat):
        """planner.py:642-644 on explicit (b, K) as produced by cost.get_linear_system."""
        costs = errors.transpose(1, 2) @ w_mat @ errors
        return costs.reshape(-1)

    def get_recent_samples(self):
        n = self.n_dof
        return (self.particle_means[..., :n].detach().clone(), self.particle_means[..., -n:].detach().clone())
