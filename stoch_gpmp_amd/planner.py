"""StochGPMP -- the sampling planner of reference `stoch_gpmp/planner.py:18-348` with the same
constructor, `reset`, `optimize`, `sample_and_eval`, `_update_distribution`, `get_recent_samples`
and `sample_trajectories`, running on the HIP kernels of libsgpmp.so.

Per iteration (the loop body at reference planner.py:289-299) the reference rebuilds a
MultivariateNormal over a replicated [P,M,M] precision, samples through a dense M x M factor,
evaluates the costs with a dozen torch ops and a dense importance-sampling matmul.  Here one call per
iteration, `sgpmp_step`, enqueues K5 (IS weights) -> K2 (scan sampler) -> K3 (cost sweep; K2 + K3 as ONE launch
where the step qualifies) -> K4 (reweight + mean update) on the current HIP stream, and one call per optimize(),
`sgpmp_optimize`, runs the loop of those steps on the C side; Python only passes pointers.

Additions over the reference API (all optional keyword arguments):
  noise='philox' | 'torch'   'philox' (default) draws counter-based noise inside K2, keyed on
                             (seed, draw, global particle, sample, element) -- independent of how
                             particles are sharded.  'torch' replays the reference's noise stream:
                             eps = torch.randn(S, P, M) from the global CPU generator in the
                             reference's call order (planner.py:48-49,213,227,243), fed to K2 --
                             identical seeds then give the reference's trajectories (parity mode).
  rank / world_size          shard the particles over one process per GPU (contiguous ranges);
                             statistics are all-reduced over RCCL once per iteration, inside
                             `sgpmp_step` (C ABI, side stream; no Python in the loop).  The RCCL
                             id travels through torch.distributed, which must be initialised.
                             collective='torch' keeps the torch.distributed all-reduce instead
                             (what the gloo CPU tests exercise).
  step()                     one loop body, public (the reference only has it inline).
  mode_stats=False           True: every step also leaves the per-goal statistics of the particle means (sum and sum
                             of squares per goal, all-reduced over the ranks on the side stream -- the "weighted-mean /
                             covariance statistics" of the modes); global_mode_stats() reads them.  False: the same
                             numbers are computed (and all-reduced) when global_mode_stats() is called.
  clone_outputs=True         optimize() returns the pre-update means in a tensor of the call's own, as the reference does with its
                             clones (planner.py:252-253) -- the call's last step writes them there, so there is no copy;
                             False hands out views of the persistent buffer.
  state_dict() / load_state_dict()   checkpoint / resume: means, seed, draw counter (the noise is counter-based, so a
                             resumed run -- also under another sharding -- continues bit for bit).
  pipeline_steps=True        optimize(opt_iters >= 2) lets the context run the iterations of the call as two
                             particle-half chains on streams of its own (one half's update kernel under the
                             other half's sampler + sweep launch); same results, bit for bit.
  c_loop=True                optimize() is ONE call into the library (include/sgpmp.h: sgpmp_optimize runs the K-loop);
                             False: one sgpmp_step call per iteration from Python, as in rounds 1-5 (same results, bit for bit).
  f64_fields_f32=False       fp64 planners only, opt-in: the one-launch step evaluates the LINK fields (forward kinematics, self
                             distance, sphere fields) on the fp32 launches' packed code from the fp64 waypoint rounded to fp32;
                             noise, recurrence, samples, means, GP / goal-prior / importance-sampling terms stay fp64.  Costs
                             within ~1e-9 of the all-fp64 step's, 1.7 x faster.
  store_free=True            optimize(opt_iters = K) returns the LAST iteration's tensors only (planner.py:289-317), so
                             iterations 1 .. K - 1 do not write their samples (470 MB per iteration at 1024 x 128 x 64):
                             the update regenerates the rows that carry weight from their noise keys, bit for bit
                             (include/sgpmp.h: SGPMP_STEP_NO_SAMPLES).  The last iteration of every call -- and so every
                             optimize(opt_iters=1) -- stores as always; all returned tensors, `particle_means` and
                             `state_samples` are identical either way.  The library takes the permission where it
                             pays (problems of >= 2.8 MB of samples per waypoint, or planar ones whose update runs
                             inside the launch); smaller problems store.  False: every iteration stores.
"""
import itertools
import time

import torch

from . import _lib as L
from .costs.factors.gp_factor import GPFactor
from .costs.factors.mp_priors_multi import MultiMPPrior, PlannerPrior
from .costs.factors.unary_factor import UnaryFactor
from .dist import allgather_means, allreduce_mode_sums, allreduce_stats_async, mode_moments, shard_range
from .engine import Engine


_UNSEEDED = itertools.count()


class StochGPMP:
    _discard_draw_at_reset = True

    def __init__(
            self,
            num_particles_per_goal,
            num_samples,
            traj_len,
            opt_iters,
            dt=None,
            n_dof=None,
            step_size=1.,
            temperature=1.,
            start_state=None,
            multi_goal_states=None,
            initial_particle_means=None,
            cost=None,
            sigma_start_init=None,
            sigma_start_sample=None,
            sigma_goal_init=None,
            sigma_goal_sample=None,
            sigma_gp_init=None,
            sigma_gp_sample=None,
            seed=None,
            tensor_args=None,
            noise='philox',
            rank=None,
            world_size=None,
            process_group=None,
            **kwargs
    ):
        if tensor_args is None:
            tensor_args = {'device': torch.device('cuda:0'), 'dtype': torch.float32}
        self.tensor_args = tensor_args
        L.require_cuda(tensor_args)
        if noise not in ('philox', 'torch'):
            raise ValueError("noise must be 'philox' or 'torch'")
        self.noise = noise

        if seed is not None:
            torch.manual_seed(seed)              # same global side effect as planner.py:48-49
        # unseeded planners must not all replay one stream: like the reference (which keeps drawing
        # from torch's global generator) they take their key from that generator's current seed, and a
        # process-wide counter keeps two unseeded planners of one process apart
        if seed is None:
            self.seed = (int(torch.initial_seed()) + 0x9E3779B97F4A7C15 * next(_UNSEEDED)) & ((1 << 63) - 1)
        else:
            self.seed = int(seed)

        self.n_dof = n_dof
        self.d_state_opt = 2 * self.n_dof
        self.dt = dt
        self.traj_len = traj_len
        self.goal_directed = (multi_goal_states is not None)
        if not self.goal_directed:
            self.num_goals = 1
        else:
            assert multi_goal_states.dim() == 2
            self.num_goals = multi_goal_states.shape[0]
        self.num_particles_per_goal = num_particles_per_goal
        self.num_particles = num_particles_per_goal * self.num_goals        # global P
        self.num_samples = num_samples
        self.opt_iters = opt_iters
        self.step_size = step_size
        self.temperature = temperature
        self.sigma_start_init = sigma_start_init
        self.sigma_start_sample = sigma_start_sample
        self.sigma_goal_init = sigma_goal_init
        self.sigma_goal_sample = sigma_goal_sample
        self.sigma_gp_init = sigma_gp_init
        self.sigma_gp_sample = sigma_gp_sample
        self.start_states = start_state
        self.multi_goal_states = multi_goal_states
        self.cost = cost

        # particle sharding (one process per GPU)
        if world_size is None:
            if torch.distributed.is_available() and torch.distributed.is_initialized() \
                    and kwargs.get('distributed', False):
                world_size = torch.distributed.get_world_size(process_group)
                rank = torch.distributed.get_rank(process_group)
            else:
                world_size, rank = 1, 0
        self.world_size, self.rank, self.process_group = world_size, rank or 0, process_group
        self.p0, self.p1 = shard_range(self.num_particles, self.rank, self.world_size)
        self.num_particles_local = self.p1 - self.p0

        self._mean = None
        self._weights = None
        self._sample_dist = None
        self._engine = None
        self._draw = 0
        self._pending_reduce = []
        self._force_reduce = bool(kwargs.get('force_stats_allreduce', False))   # 1-rank RCCL smoke test
        self._collective = kwargs.get('collective', 'rccl')
        if self._collective not in ('rccl', 'torch'):
            raise ValueError("collective must be 'rccl' or 'torch'")
        self._comm_attached = False
        # optimize(opt_iters >= 2) runs its iterations as two particle-half chains (sgpmp_pipeline_begin)
        self.pipeline_steps = bool(kwargs.get('pipeline_steps', True))
        self.mode_stats_every_step = bool(kwargs.get('mode_stats', False))
        self.clone_outputs = bool(kwargs.get('clone_outputs', True))
        self.store_free = bool(kwargs.get('store_free', True))
        self.c_loop = bool(kwargs.get('c_loop', True))
        self.f64_fields_f32 = bool(kwargs.get('f64_fields_f32', False))
        self._mode_buf = None

        self.reset(start_state, multi_goal_states, initial_particle_means=initial_particle_means)

    # ------------------------------------------------------------------------------- factors
    def set_prior_factors(self):
        """Descriptor objects with the reference's attribute names (planner.py:84-140)."""
        ta, d, n, T = self.tensor_args, self.d_state_opt, self.n_dof, self.traj_len
        self.start_prior_init = UnaryFactor(d, self.sigma_start_init, self.start_states, ta)
        self.gp_prior_init = GPFactor(n, self.sigma_gp_init, self.dt, T - 1, ta)
        self.start_prior_sample = UnaryFactor(d, self.sigma_start_sample, self.start_states, ta)
        self.gp_prior_sample = GPFactor(n, self.sigma_gp_sample, self.dt, T - 1, ta)
        self.multi_goal_prior_init, self.multi_goal_prior_sample = [], []
        if self.goal_directed:
            for i in range(self.num_goals):
                self.multi_goal_prior_init.append(
                    UnaryFactor(d, self.sigma_goal_init, self.multi_goal_states[i], ta))
                self.multi_goal_prior_sample.append(
                    UnaryFactor(d, self.sigma_goal_sample, self.multi_goal_states[i], ta))

    def const_vel_trajectories(self, start_state, multi_goal_states):
        """planner.py:142-155 (velocity uses /(T dt)); setup-time, tiny."""
        T, n = self.traj_len, self.n_dof
        traj_dim = (multi_goal_states.shape[0], self.num_particles_per_goal, T, self.d_state_opt)
        state_traj = torch.zeros(traj_dim, **self.tensor_args)
        mean_vel = (multi_goal_states[:, :n] - start_state[:n]) / (T * self.dt)
        # (the reference's T-long loop as one expression with the same elementwise operations in the same order:
        # 64 waypoints x 5 tiny launches were 3 ms of every reset())
        i = torch.arange(T, **self.tensor_args).reshape(1, T, 1)
        interp = start_state[:n] * (T - i - 1) / (T - 1) + multi_goal_states[:, None, :n] * i / (T - 1)   # [G,T,n]
        state_traj[:, :, :, :n] = interp.unsqueeze(1)
        state_traj[:, :, :, n:] = mean_vel.unsqueeze(1).unsqueeze(1)
        return state_traj

    def get_prior_dist(self, start_K, gp_K, goal_K, state_init, particle_means=None, goal_states=None):
        """planner.py:157-179: a stand-alone MultiMPPrior of this problem's shape (a context of its own; the planner's OWN
        distributions, `_sample_dist` / `_init_dist`, are PlannerPrior views of the planner's context)."""
        return MultiMPPrior(self.traj_len - 1, self.dt, 2 * self.n_dof, self.n_dof, start_K, gp_K, state_init,
                            K_g_inv=goal_K, means=particle_means, goal_states=goal_states, tensor_args=self.tensor_args)

    def _const_vel_prior_means(self):
        """Means of the initialisation prior (mp_priors_multi.py:130-168): [G,T,d]."""
        T, n = self.traj_len, self.n_dof
        steps = T - 1
        if not self.goal_directed:
            return self.start_state.repeat(T, 1).unsqueeze(0).contiguous()
        means = torch.zeros(self.num_goals, T, self.d_state_opt, **self.tensor_args)
        vel = (self.multi_goal_states[:, :n] - self.start_state[:n]) / (steps * self.dt)
        i = torch.arange(T, **self.tensor_args).reshape(1, T, 1)        # (one expression instead of a T-long loop)
        means[:, :, :n] = self.start_state[:n] * (steps - i) * 1. / steps \
            + self.multi_goal_states[:, None, :n] * i * 1. / steps
        means[:, :, n:] = vel.unsqueeze(1)
        return means

    # ------------------------------------------------------------------------------- reset
    def reset(self, start_state=None, multi_goal_states=None, initial_particle_means=None):
        if start_state is not None:
            self.start_state = start_state.detach().clone()
        if multi_goal_states is not None:
            self.multi_goal_states = multi_goal_states.detach().clone()
        self.set_prior_factors()

        ta = self.tensor_args
        T, d, n, S = self.traj_len, self.d_state_opt, self.n_dof, self.num_samples
        G, nppg, P, Pl = self.num_goals, self.num_particles_per_goal, self.num_particles, \
            self.num_particles_local
        M = T * d

        # The context (factor buffers, cost program, RCCL communicator) and the iteration buffers
        # survive reset(): a receding-horizon loop resets every control cycle, and K1's output only
        # depends on (dt, sigmas), which Engine.set_prior remembers.
        fresh = self._engine is None
        if fresh:
            self._engine = Engine(n, T, Pl, S, G, nppg, self.p0, P, tensor_args=ta)
            if self.f64_fields_f32:
                self._engine.set_option("f64_fields_f32", 1)
            self._attach_comm()
            if self.mode_stats_every_step:
                self._mode_buf = torch.zeros(G, M + 1, 2, device=ta['device'], dtype=torch.float64)
                self._engine.set_step_mode_stats(self._mode_buf)
        eng = self._engine
        goal_init = self.sigma_goal_init if self.goal_directed else None
        goal_sample = self.sigma_goal_sample if self.goal_directed else None
        # K1 twice (init + sampling priors): planner.py:206-212, 218-225 -- side by side when both are needed
        if initial_particle_means is None:
            eng.set_priors(self.dt, (self.sigma_start_init, self.sigma_gp_init, goal_init),
                           (self.sigma_start_sample, self.sigma_gp_sample, goal_sample))
        else:
            eng.set_prior(L.PRIOR_SAMPLE, self.dt, self.sigma_start_sample, self.sigma_gp_sample, goal_sample)
        # (the draw counter keeps running across reset(), as the reference's generator does: a
        # replanning loop must not see the same noise after every reset)

        if initial_particle_means is not None:
            if isinstance(initial_particle_means, str) and initial_particle_means == 'const_vel':
                pm = self.const_vel_trajectories(self.start_state, self.multi_goal_states)
            else:
                pm = initial_particle_means
            pm = pm.to(**ta)
        else:
            init_means = self._const_vel_prior_means().contiguous()
            eps = None
            if self.noise == 'torch':                      # reference draw #1: randn(nppg, G, M)
                eps = torch.randn(nppg, G, M, dtype=ta['dtype']).to(ta['device'])
            # planner.py:206-214: the initialisation distribution lives for this one draw
            self._init_dist = PlannerPrior(self, L.PRIOR_INIT, means=init_means)
            pm = self._init_dist.sample(nppg, eps=eps)
            del self._init_dist                            # free memory (planner.py:214)
            self._draw -= 1                                # (counted below, for both ways of initialising)
        self._draw += 1
        # flatten(0,1): p = g * nppg + k (planner.py:215); keep this rank's shard
        self.particle_means = pm.reshape(P, T, d)[self.p0:self.p1].contiguous().clone()

        # persistent buffers of the iteration.  optimize() hands out VIEWS of the sample buffer (the reference
        # returns views of its sample tensor too, planner.py:246-249) and, like the reference (planner.py:252-253),
        # CLONES of the pre-update means unless clone_outputs=False.
        if fresh:
            self._samples_buf = torch.empty(Pl, S, T, d, **ta)      # iteration buffer (pointers are pre-bound)
            self._costs = torch.empty(Pl, S, **ta)
            self._costs64 = torch.empty(Pl, S, device=ta['device'], dtype=torch.float64)
            self._weights_buf = torch.empty(Pl, S, **ta)
            self._grad = torch.empty(Pl, T, d, **ta)
            self._means_prev_buf = torch.empty(Pl, T, d, **ta)     # pre-update means of a step (scratch of the iterations)
            self._means_prev = self._means_prev_buf                # ... of the LAST step: the buffer, or the tensor optimize() handed out
            self._stats = torch.zeros(2, L.STAT_SHARDS, 4, device=ta['device'], dtype=torch.float64)
            # views handed back by optimize(): created once, the buffers are persistent
            self._weights = self._weights_buf.view(-1, S, 1, 1)
            self._views = (self._means_prev_buf[..., :n], self._means_prev_buf[..., -n:],
                           self._samples_buf[..., :n], self._samples_buf[..., -n:])
        else:
            self._engine.stats_wait(None)                # a side-stream all-reduce may still use _stats
            self._stats.zero_()
        self.state_samples = self._samples_buf
        self._costs64_fresh = False
        self._stats_slot = 0
        # per-particle row counts of the last update (they steer the next step's launch and update per particle): a fresh
        # problem starts without them, whatever ran on this context before
        eng.clear_row_counts()                          # (stream-ordered: a replanning loop resets every control cycle)
        self._step_calls = {}
        self._opt_calls = {}
        self._pm_obj, self._pm_version = None, -1       # means tensor / version after our last fused step
        self._mode_fresh = False                        # _mode_buf holds the statistics of the current means
        self._Sigma_inv = None
        # planner.py:217-226: the sampling distribution -- a live object over this planner's context and means
        self._sample_dist = PlannerPrior(self, L.PRIOR_SAMPLE)
        self._Sigma_invs_set = False
        self._obs_src = None        # strong reference to the caller's obstacle tensor (see _spheres)
        self._obs_ver = -1
        self._obs_dev = None

        # cost program: our CostComposite is compiled into the engine; anything else with .eval is
        # called as user code on the samples tensor (planner.py:76,231)
        # (a composite around a foreign FK callable evaluates its link fields outside the sweep)
        self._native_cost = hasattr(self.cost, "compile_into") and not getattr(self.cost, "foreign_fk", False)
        if self._native_cost:
            self.cost.compile_into(eng)
            self._cost_version = self.cost.version()
        else:
            eng.set_costs([])

        if not self._discard_draw_at_reset:              # GPMP.reset (planner.py:508-545) draws nothing more
            return
        # the reference draws one throw-away batch here (planner.py:227); keep the stream aligned
        eps = None
        if self.noise == 'torch':
            eps = torch.randn(S, P, M, dtype=ta['dtype']).to(ta['device'])
        if Pl > 0:
            eng.sample(L.PRIOR_SAMPLE, self.seed, self._draw, self.particle_means, S,
                       out=self.state_samples, eps=eps, eps_mode_offset=self.p0 if eps is not None else 0,
                       mode_offset=self.p0)
        self._draw += 1

    @property
    def Sigma_inv(self):
        """Dense [M,M] precision of the sampling prior (planner.py:226), assembled on demand from
        K1's blocks -- the kernels never materialise it."""
        if self._Sigma_inv is None:
            blocks, _, _ = self._engine.get_prior(L.PRIOR_SAMPLE, blocks_only=True)
            d, T = self.d_state_opt, self.traj_len
            S = torch.zeros(T * d, T * d, dtype=torch.float64)
            for t in range(T):
                S[t * d:(t + 1) * d, t * d:(t + 1) * d] = blocks[0 if t == 0 else (2 if t == T - 1 else 1)]
                if t + 1 < T:
                    S[(t + 1) * d:(t + 2) * d, t * d:(t + 1) * d] = blocks[3]
                    S[t * d:(t + 1) * d, (t + 1) * d:(t + 2) * d] = blocks[3].t()
            self._Sigma_inv = S.to(**self.tensor_args)
        return self._Sigma_inv

    # ------------------------------------------------------------------------------- helpers
    def _spheres(self, observation):
        """observation['obstacle_spheres'] as a contiguous [O,4] device tensor of the planner's dtype.

        A tensor that already is one is used in place (its pointer goes to the kernels, so in-place
        edits by the caller are seen).  Anything else is converted into a planner-owned buffer that
        keeps its address while the sphere count stays the same.  The source tensor is held by a
        strong reference and re-converted whenever it is another object or its version counter
        moved -- never keyed on id(), which CPython reuses as soon as a tensor is freed."""
        sph = observation.get('obstacle_spheres', None)
        if sph is None:
            self._obs_src = None
            return None
        ta = self.tensor_args
        direct = (sph.is_cuda and sph.device == torch.device(ta['device']) and sph.dtype == ta['dtype']
                  and sph.is_contiguous() and sph.numel() % 4 == 0)
        if direct:
            self._obs_src = sph
            return sph.view(-1, 4)
        if sph is self._obs_src and sph._version == self._obs_ver:
            return self._obs_dev
        conv = sph.detach().to(**ta).reshape(-1, 4)
        if self._obs_dev is None or self._obs_dev.shape != conv.shape:
            self._obs_dev = conv.contiguous().clone()
        else:
            self._obs_dev.copy_(conv)
        self._obs_src, self._obs_ver = sph, sph._version
        return self._obs_dev

    def _draw_eps(self):
        if self.noise != 'torch':
            return None
        ta = self.tensor_args
        return torch.randn(self.num_samples, self.num_particles, self.traj_len * self.d_state_opt,
                           dtype=ta['dtype']).to(ta['device'])

    def _attach_comm(self):
        """Give the engine an RCCL communicator over the ranks of the process group (multi-GPU runs)."""
        import torch.distributed as dist
        self._comm_attached = False
        if not ((self.world_size > 1 or self._force_reduce) and self._collective == 'rccl'
                and dist.is_available() and dist.is_initialized()):
            return
        group = self.process_group
        if dist.get_world_size(group) != self.world_size:
            return                                       # a shard run stand-alone: no peers to reduce with
        box = [self._engine.comm_unique_id() if dist.get_rank(group) == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        # (the id travels through whatever backend the group has: device tensors for nccl, host tensors for gloo)
        on_host = dist.get_backend(group) == 'gloo'
        dist.broadcast_object_list(box, src=src, group=group,
                                   device=torch.device('cpu') if on_host else torch.device(self.tensor_args['device']))
        self._engine.comm_init(box[0], self.world_size, dist.get_rank(group))
        self._comm_attached = True

    def _reduce_stats(self, slot):
        if self._comm_attached:
            return                                       # sgpmp_step enqueued the RCCL all-reduce itself
        # torch.distributed path (collective='torch'; a shard run stand-alone has no peers to reduce with)
        if (self.world_size > 1 or self._force_reduce) and torch.distributed.is_initialized() \
                and self._collective == 'torch':
            self._pending_reduce.append(allreduce_stats_async(self._stats[slot], self.process_group))
            if len(self._pending_reduce) > 1:           # never gate the next iteration's kernels
                self._pending_reduce.pop(0).wait()

    def global_stats(self):
        """(mean over particles of sum_s cost, mean over particles of min_s cost) of the last
        iteration, over ALL ranks (reference print_info statistic, planner.py:668-672)."""
        for w in self._pending_reduce:
            w.wait()
        self._pending_reduce = []
        if self._comm_attached:
            self._engine.stats_wait(None)                # stream-side wait for the side-stream all-reduce
        s = self._stats[self._stats_slot ^ 1].sum(0).cpu()      # sum the shards
        cnt = max(float(s[2]), 1.0)
        return float(s[0]) / cnt, float(s[1]) / cnt

    def global_mode_stats(self):
        """(mean [G,T,d], variance [G,T,d], particles [G]) of the particle means of every goal over ALL ranks: the first
        two moments of each mode of the trajectory distribution (a mode = the particles of one goal, planner.py:215).
        The reference keeps its particles on one device and has no counterpart; the sums behind these numbers are what
        the multi-GPU all-reduce carries (north_star; include/sgpmp.h sgpmp_mode_stats) and what a covariance
        adaptation through MultiMPPrior.set_Sigma_invs (mp_priors_multi.py:125-128) would consume.  With
        mode_stats=True they are produced by every step on the side stream; else here, on demand."""
        eng = self._engine
        torch_dist = (self.world_size > 1 or self._force_reduce) and not self._comm_attached \
            and self._collective == 'torch' and torch.distributed.is_initialized()
        # the per-step buffer describes the means the LAST step wrote: before the first step, after reset() or after an
        # edit of particle_means (torch's version counter moved) it is stale, and the on-demand path below answers
        pm = self.particle_means
        fresh = self._mode_fresh and pm is self._pm_obj and pm._version == self._pm_version
        if self.mode_stats_every_step and self._native_cost and fresh:     # (a foreign cost object steps through sgpmp_update)
            eng.mode_stats_wait()
            buf = self._mode_buf.clone() if torch_dist else self._mode_buf
        else:
            buf = eng.mode_stats(self.particle_means)
            if self._comm_attached:
                eng.allreduce_f64(buf)
                eng.stats_wait(buf)
        if torch_dist:
            allreduce_mode_sums(buf, self.process_group)
        return mode_moments(buf, self.traj_len, self.d_state_opt)

    # ------------------------------------------------------------------------------- the loop
    def step(self, _means_prev_out=None, _samples_unread=False, **observation):
        """One body of the loop at planner.py:289-299 on this rank's particle shard.  (_means_prev_out: where this step leaves
        its pre-update means -- optimize() passes a fresh tensor for the step whose means it returns.  _samples_unread:
        optimize() vouches that nobody reads this step's samples -- it is not the call's last -- so the step need not
        write them, include/sgpmp.h: SGPMP_STEP_NO_SAMPLES.)"""
        self.state_samples = self._samples_buf           # (sample_trajectories may have re-pointed it)
        prev_out = self._means_prev_buf if _means_prev_out is None else _means_prev_out
        self._means_prev = prev_out
        if not self._native_cost:
            return self._step_foreign_cost(**observation)
        slot = self._stats_slot
        cv = self.cost.version()
        if cv != self._cost_version:                     # a field / cost was edited (e.g. update_target)
            self.cost.compile_into(self._engine)
            self._cost_version = cv
        if self.num_particles_local > 0:
            if self.noise == 'philox':
                # hot path: all arguments except the draw counter are pre-bound (one ctypes call)
                sph = self._spheres(observation)
                key = (slot, 0 if sph is None else sph.data_ptr(), 0 if sph is None else sph.shape[0],
                       self.temperature, self.step_size, self.particle_means.data_ptr(),
                       self.state_samples.data_ptr())
                call = self._step_calls.get(key)
                if call is None:
                    if len(self._step_calls) > 8:
                        self._step_calls.clear()
                    call = self._engine.prepare_step(
                        self.seed, self.particle_means, self.state_samples, self.temperature,
                        self.step_size, costs=self._costs, weights=self._weights_buf, grad=self._grad,
                        means_prev=self._means_prev_buf, spheres=sph, stats=self._stats[slot])
                    self._step_calls[key] = call
                # torch bumps a tensor's version counter on every in-place edit; the kernels do not: if the
                # means are the same tensor at the version recorded after our last step, only we wrote them
                pm = self.particle_means
                kept = pm is self._pm_obj and pm._version == self._pm_version
                flags = (L.STEP_MEANS_KEPT if kept else 0) | (L.STEP_NO_SAMPLES if _samples_unread and self.store_free else 0)
                call(self._draw, flags, None if _means_prev_out is None else L.ptr(prev_out))
                self._pm_obj, self._pm_version = pm, pm._version
                self._mode_fresh = self.mode_stats_every_step
            else:
                self._engine.step(self.seed, self._draw, self.particle_means, self.state_samples,
                                  self.temperature, self.step_size, costs=self._costs,
                                  weights=self._weights_buf, grad=self._grad, means_prev=prev_out,
                                  spheres=self._spheres(observation), eps=self._draw_eps(),
                                  eps_mode_offset=self.p0, stats=self._stats[slot])
                pm = self.particle_means
                self._pm_obj, self._pm_version = pm, pm._version
                self._mode_fresh = self.mode_stats_every_step
        elif self._comm_attached:
            # an empty shard (more ranks than particles) still joins the step's statistics all-reduce: it is a
            # collective, and the other ranks' side streams would wait for this rank for ever
            if self.noise == 'torch':
                self._draw_eps()                         # (keep the global generator aligned with the other ranks)
            self._engine.step(self.seed, self._draw, self.particle_means, self.state_samples, self.temperature,
                              self.step_size, stats=self._stats[slot])
            # (the step issued the per-goal statistics all-reduce too: this rank must read the per-step buffer like the
            # others do, not start an on-demand collective of its own)
            pm = self.particle_means
            self._pm_obj, self._pm_version = pm, pm._version
            self._mode_fresh = self.mode_stats_every_step
        self._draw += 1
        self._reduce_stats(slot)
        self._stats_slot ^= 1
        return self._costs, self._grad

    def _step_foreign_cost(self, **observation):
        """`cost` is user code with .eval(trajs, **obs) (e.g. a learned EBM, reference README.md:3):
        sample with K2, hand the samples tensor to it, add the IS term (K5 + K3 with an empty
        program) and update with K4."""
        eng = self._engine
        eps = self._draw_eps()
        eng.sample(L.PRIOR_SAMPLE, self.seed, self._draw, self.particle_means, self.num_samples,
                   out=self.state_samples, eps=eps, eps_mode_offset=self.p0 if eps is not None else 0,
                   mode_offset=self.p0)
        self._draw += 1
        costs = self._get_costs(**observation)
        grad = self._update_distribution(costs, self.state_samples)
        return costs, grad

    def _get_costs(self, **observation):
        """planner.py:229-237 as separate calls (cost.eval + importance-sampling term)."""
        eng = self._engine
        Pl, S = self.num_particles_local, self.num_samples
        isw = eng.is_weights(self.particle_means, self.temperature)
        if self._native_cost:
            eng.cost_eval(self.state_samples, batch_offset=self.p0 * S,
                          spheres=self._spheres(observation), is_weights=isw, rows_per_particle=S,
                          out=self._costs, out64=self._costs64)
            self._costs64_fresh = True       # fp64 twin of the costs just returned
            return self._costs
        user = self.cost.eval(self.state_samples, **observation).reshape(Pl, S)
        eng.cost_eval(self.state_samples, batch_offset=self.p0 * S, is_weights=isw,
                      rows_per_particle=S, out=self._costs)
        return user.to(self._costs.dtype) + self._costs

    def sample_and_eval(self, **observation):
        """planner.py:239-261."""
        self.state_samples = self._samples_buf
        eps = self._draw_eps()
        self._engine.sample(L.PRIOR_SAMPLE, self.seed, self._draw, self.particle_means,
                            self.num_samples, out=self.state_samples, eps=eps,
                            eps_mode_offset=self.p0 if eps is not None else 0, mode_offset=self.p0)
        self._draw += 1
        costs = self._get_costs(**observation)
        n = self.n_dof
        return (self.state_samples[..., -n:], self.state_samples[..., :n],
                self.particle_means[..., -n:].clone(), self.particle_means[..., :n].clone(), costs)

    def _update_distribution(self, costs, traj_samples):
        """planner.py:263-275 (K4)."""
        if costs is self._costs and self._costs64_fresh:
            costs = self._costs64            # our own costs handed back: keep their fp64 accumulators
        self._costs64_fresh = False
        costs = costs.contiguous()
        self._engine.update(costs, traj_samples.contiguous(), self.particle_means, self.temperature,
                            self.step_size, weights=self._weights_buf, grad=self._grad,
                            means_prev=self._means_prev)
        return self._grad

    def optimize(self, opt_iters=None, debug=False, **observation):
        """planner.py:277-317.  Returns (state_particles, control_particles, state_trajectories,
        control_samples, costs, approx_grad) of the LAST iteration; the particle tensors are the
        PRE-update means, as in the reference (planner.py:252-253)."""
        if opt_iters is None:
            opt_iters = self.opt_iters
        start_time = time.time()
        costs = approx_grad = fresh_prev = None
        # Nobody looks at the buffers between the iterations of one call (the reference returns the last
        # iteration's tensors only), so the context may run them as two particle-half chains on streams of its
        # own -- one half's update kernel under the other half's sampler + sweep launch (include/sgpmp.h:
        # sgpmp_pipeline_begin).  `debug` prints costs in between and therefore keeps the single chain.
        torch_reduce = (self.world_size > 1 or self._force_reduce) and not self._comm_attached \
            and self._collective == 'torch' and torch.distributed.is_initialized()   # (reads the statistics per step)
        piped = (opt_iters >= 2 and not debug and self._native_cost and self.noise == 'philox'
                 and self.num_particles_local > 0 and self.pipeline_steps and not torch_reduce
                 and not self.mode_stats_every_step)
        if (not debug and self._native_cost and self.noise == 'philox' and self.num_particles_local > 0
                and not torch_reduce and opt_iters >= 1 and self.c_loop):
            return self._optimize_one_call(opt_iters, piped, observation)
        if piped:
            self._spheres(observation)                   # (a first use copies on THIS stream: before the chains fork)
            self._engine.pipeline_begin()
        try:
            for opt_step in range(opt_iters):
                start_time_iter = time.time()
                if opt_step == opt_iters - 1 and self.clone_outputs and self.num_particles_local > 0:
                    fresh_prev = torch.empty_like(self._means_prev_buf)      # (caching allocator: no launch)
                costs, approx_grad = self.step(_means_prev_out=fresh_prev, _samples_unread=opt_step < opt_iters - 1,
                                               **observation)
                if debug and opt_step % 50 == 0:
                    print_info(opt_step, opt_iters, start_time_iter, start_time, costs)
        finally:
            if piped:
                self._engine.pipeline_end()
        state_particles, control_particles, state_trajectories, control_samples = self._views
        if fresh_prev is not None:                       # planner.py:252-253: the reference hands out clones of the means
            n = self.n_dof                               # (here: a tensor of this call's own, written by its last step -- no copy)
            state_particles, control_particles = fresh_prev[..., :n], fresh_prev[..., -n:]
        self._recent_control_samples = control_samples
        self._recent_control_particles = control_particles
        self._recent_state_trajectories = state_trajectories
        self._recent_state_particles = state_particles
        self._recent_weights = self._weights
        return (state_particles, control_particles, state_trajectories, control_samples, costs,
                approx_grad)

    def _optimize_one_call(self, opt_iters, piped, observation):
        """optimize() as ONE call into the library (include/sgpmp.h: sgpmp_optimize -- the K-loop, the draw counters, the
        alternating statistics slot, the flags of every step and the two-chain bracket on the C side): what the per-iteration
        step() calls of the loop above do, bit for bit (tests/test_gpu_planner.py::test_pipelined_optimize_*)."""
        self.state_samples = self._samples_buf
        cv = self.cost.version()
        if cv != self._cost_version:                     # a field / cost was edited (e.g. update_target)
            self.cost.compile_into(self._engine)
            self._cost_version = cv
        sph = self._spheres(observation)
        pm = self.particle_means
        key = (0 if sph is None else sph.data_ptr(), 0 if sph is None else sph.shape[0], self.temperature, self.step_size,
               pm.data_ptr(), self.state_samples.data_ptr())
        call = self._opt_calls.get(key)
        if call is None:
            if len(self._opt_calls) > 8:
                self._opt_calls.clear()
            call = self._engine.prepare_optimize(
                self.seed, pm, self.state_samples, self.temperature, self.step_size, costs=self._costs,
                weights=self._weights_buf, grad=self._grad, means_prev=self._means_prev_buf, spheres=sph,
                stats_pair=self._stats)
            self._opt_calls[key] = call
        fresh_prev = torch.empty_like(self._means_prev_buf) if self.clone_outputs else None   # (caching allocator: no launch)
        kept = pm is self._pm_obj and pm._version == self._pm_version
        flags = (L.STEP_MEANS_KEPT if kept else 0) | (L.OPT_PIPELINE if piped else 0) \
            | (L.OPT_STORE_FREE if self.store_free else 0)
        try:
            call(opt_iters, self._draw, self._stats_slot, flags, None if fresh_prev is None else fresh_prev.data_ptr())
        finally:
            # (a failing step leaves the counters where a Python loop would have left them at worst: past the call)
            self._draw += opt_iters
            self._stats_slot ^= opt_iters & 1
        self._pm_obj, self._pm_version = pm, pm._version
        self._mode_fresh = self.mode_stats_every_step
        n = self.n_dof
        state_particles, control_particles, state_trajectories, control_samples = self._views
        if fresh_prev is not None:
            self._means_prev = fresh_prev
            state_particles, control_particles = fresh_prev[..., :n], fresh_prev[..., -n:]
        else:
            self._means_prev = self._means_prev_buf
        self._recent_control_samples = control_samples
        self._recent_control_particles = control_particles
        self._recent_state_trajectories = state_trajectories
        self._recent_state_particles = state_particles
        self._recent_weights = self._weights
        return (state_particles, control_particles, state_trajectories, control_samples, self._costs, self._grad)

    def _get_traj(self, mode='best'):
        if mode == 'best':
            particle_ind = self._weights.argmax()        # same (flat) indexing as planner.py:321-323
            return self.state_samples[particle_ind].clone()
        elif mode == 'mean':
            return self._mean.clone()
        raise ValueError('Unidentified sampling mode in get_next_action')

    def get_recent_samples(self):
        return (self._recent_state_trajectories.detach().clone(),
                self._recent_control_samples.detach().clone())

    def sample_trajectories(self, num_samples_per_particle):
        """planner.py:339-348: fresh draws about the current means."""
        Pl, T, d = self.num_particles_local, self.traj_len, self.d_state_opt
        eps = None
        if self.noise == 'torch':
            eps = torch.randn(num_samples_per_particle, self.num_particles, T * d,
                              dtype=self.tensor_args['dtype']).to(self.tensor_args['device'])
        # (a fresh tensor: the iteration buffers keep their size and their pre-bound pointers)
        fresh = self._engine.sample(
            L.PRIOR_SAMPLE, self.seed, self._draw, self.particle_means, num_samples_per_particle,
            eps=eps, eps_mode_offset=self.p0 if eps is not None else 0, mode_offset=self.p0)
        self._draw += 1
        self.state_samples = fresh                       # as the reference does (planner.py:341)
        return fresh[..., :self.n_dof], fresh[..., -self.n_dof:]

    # ------------------------------------------------------------------------------- checkpoint / resume
    def state_dict(self):
        """Everything a run needs to continue where it stands (the reference keeps the same state in
        `particle_means` and torch's global generator, planner.py:215,243,270): this rank's particle means with their
        global particle range, the noise key (seed) and the draw counter.  The in-kernel noise is a pure function of
        (seed, draw, GLOBAL particle, sample, element), so `load_state_dict` into a planner with the same problem
        continues bit for bit -- also under another sharding (a state saved at world_size 1 feeds every shard).
        noise='torch' planners also carry torch's CPU generator state."""
        self._engine.stats_wait(None)
        sd = {
            'version': 1,
            'particle_means': self.particle_means.detach().clone(),
            'particle_range': (self.p0, self.p1), 'num_particles': self.num_particles,
            'shape': (self.num_goals, self.num_particles_per_goal, self.num_samples, self.traj_len, self.n_dof),
            'seed': self.seed, 'draw': self._draw, 'stats_slot': self._stats_slot,
            'stats': self._stats.detach().clone(),
            'temperature': self.temperature, 'step_size': self.step_size, 'noise': self.noise,
            'cost_version': self.cost.version() if self._native_cost else None,
            # rows that carried weight in each particle's last update: the next step's launch / update decide on them per
            # particle (partials or rows: equal to 1e-6 only), so a bit-for-bit continuation needs them
            'row_counts': torch.from_numpy(self._engine.row_counts().astype('int64')),
        }
        if self.noise == 'torch':
            sd['torch_rng_state'] = torch.get_rng_state()
        return sd

    def load_state_dict(self, sd):
        """Continue the run `sd` was taken from (see state_dict).  The saved particle range must cover this planner's
        shard; problem shape and noise mode must match (ValueError otherwise)."""
        if sd.get('version') != 1:
            raise ValueError("load_state_dict: unknown state version")
        shape = (self.num_goals, self.num_particles_per_goal, self.num_samples, self.traj_len, self.n_dof)
        if tuple(sd['shape']) != shape or sd['num_particles'] != self.num_particles or sd['noise'] != self.noise:
            raise ValueError(f"load_state_dict: state of problem {tuple(sd['shape'])} / noise {sd['noise']!r}, "
                             f"planner is {shape} / {self.noise!r}")
        q0, q1 = sd['particle_range']
        if q0 > self.p0 or q1 < self.p1:
            raise ValueError(f"load_state_dict: state holds particles [{q0}, {q1}), this shard needs [{self.p0}, {self.p1})")
        self._engine.stats_wait(None)
        pm = sd['particle_means'][self.p0 - q0:self.p1 - q0].to(**self.tensor_args)
        self.particle_means.copy_(pm)                    # (in place: bumps the version counter -> no prepared IS weights)
        self.seed, self._draw = int(sd['seed']), int(sd['draw'])
        self._step_calls = {}                            # (the seed is pre-bound in the prepared calls)
        self._opt_calls = {}
        self._pm_obj, self._pm_version = None, -1
        self._mode_fresh = False
        self._stats_slot = int(sd['stats_slot'])
        if (q0, q1) == (self.p0, self.p1) and sd['stats'].shape == self._stats.shape:
            self._stats.copy_(sd['stats'])               # (the last iteration's statistics: global_stats() answers as before)
        self.temperature, self.step_size = sd['temperature'], sd['step_size']
        if self._native_cost and sd.get('cost_version') is not None and sd['cost_version'] != self.cost.version():
            import warnings
            warnings.warn("load_state_dict: the state was taken under another version of the cost program "
                          f"({sd['cost_version']} != {self.cost.version()}); the run continues on the current one")
        rc = sd.get('row_counts')
        self._engine.set_row_counts(None if rc is None else rc[self.p0 - q0:self.p1 - q0].numpy())
        if self.noise == 'torch' and 'torch_rng_state' in sd:
            torch.set_rng_state(sd['torch_rng_state'])

    def gather_particle_means(self):
        """All ranks' particle means [P,T,d] (RCCL all-gather over xGMI); identity on one GPU."""
        if self.world_size == 1:
            return self.particle_means
        if self._comm_attached and self.num_particles_local * self.world_size == self.num_particles:
            return self._engine.allgather_means(self.particle_means, self.world_size)
        return allgather_means(self.particle_means, self.num_particles, self.world_size,
                               self.process_group)


def print_info(iteration, max_iterations, start_time_iter, start_time, costs):
    """Same line format as reference planner.py:664-672."""
    now = time.time()
    fields = ['Iteration: %5d/%5d ' % (iteration, max_iterations),
              ' Iter Time: %.3f' % (now - start_time_iter),
              ' Total Time: %.3f ' % (now - start_time),
              ' Cost: %.6f' % float(costs.sum(-1).mean())]
    print('|'.join(fields))


from .gpmp import GPMP  # noqa: E402,F401  (reference planner.py defines both planners in one module)
