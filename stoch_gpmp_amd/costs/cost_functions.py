"""Cost classes with the names, constructor signatures and `eval` semantics of reference
`stoch_gpmp/costs/cost_functions.py`, compiled to the device cost program that the HIP cost sweep
(csrc/cost_sweep.hip, K3) executes in ONE pass over the trajectory batch.

A cost object here is a descriptor: `CostComposite` turns its `cost_list` into
`sgpmp_cost_desc[]` (include/sgpmp.h).  `eval(trajs, **observation)` runs K3 and returns [B] costs,
like the reference (cost_functions.py:47-58).  `get_linear_system` returns the reference's dense
(A, b, K) for user code; the GPMP planner itself consumes the factors in block-tridiagonal form.
"""
from abc import ABC, abstractmethod

import torch

from .. import _lib as L
from ..engine import Engine
from .factors.field_factor import FieldFactor
from .factors.gp_factor import GPFactor
from .factors.unary_factor import UnaryFactor


def _host_list(t):
    return [float(v) for v in torch.as_tensor(t).detach().cpu().double().flatten()]


class Cost(ABC):
    def __init__(self, n_dof, traj_len):
        self.n_dof = n_dof
        self.dim = 2 * n_dof
        self.traj_len = traj_len
        self._solo = None
        self._version = 0

    def set_cost_factors(self):
        pass

    def touch(self):
        """Mark this cost as edited: a planner that has compiled it re-compiles before its next step
        (the reference reads cost attributes live on every eval)."""
        self._version += 1

    def version(self):
        """Edit counter of this cost and of the field it wraps."""
        f = getattr(self, "field", None)
        return self._version + (getattr(f, "_version", 0) if f is not None else 0)

    def __call__(self, trajs, **observation):
        return self.eval(trajs, **observation)

    @abstractmethod
    def descriptors(self):
        """-> list of cost-term dicts for Engine.set_costs."""

    def eval(self, trajs, x_trajs=None, **observation):
        """Stand-alone evaluation of this single cost: [B,T,d] -> [B] (runs K3 with one term).
        `x_trajs` is ignored: link frames are recomputed inside the sweep from the FK chain."""
        if self._solo is None:
            self._solo = CostComposite(self.n_dof, self.traj_len, [self], FK=getattr(self, "_fk", None),
                                       tensor_args=getattr(self, "tensor_args", None))
        return self._solo.eval(trajs, **observation)

    def get_linear_system(self, trajs, **observation):
        """Dense (A, b, K) of this cost for inspection / user code.  `GPMP` itself never builds these:
        it consumes the same factors in block-tridiagonal form on the GPU (csrc/gpmp.hip)."""
        raise NotImplementedError(f"{type(self).__name__} has no linear system")


class CostComposite(Cost):
    """reference cost_functions.py:32-58."""

    def __init__(self, n_dof, traj_len, cost_list, FK=None, tensor_args=None):
        super().__init__(n_dof, traj_len)
        self.cost_list = cost_list
        self.FK = FK
        self.tensor_args = tensor_args
        self._engines = {}
        self.chain = self._resolve_chain(FK)
        # An FK callable that is not one of this package's URDF chains cannot run inside the HIP
        # sweep: the composite then calls it (user torch code, as reference cost_functions.py:51-52
        # does) and evaluates the link fields on the frames it returns (`sgpmp_field_eval`) --
        # same results, one extra pass over the frames.
        self.foreign_fk = FK is not None and self.chain is None
        for cost in cost_list:                          # children evaluate stand-alone through the same chain
            if getattr(cost, "_fk", None) is None:
                cost._fk = FK
                cost._chain = self.chain

    @staticmethod
    def _resolve_chain(FK):
        if FK is None:
            return None
        owner = getattr(FK, "__self__", FK)          # bound method of a URDFChain, or the chain itself
        return getattr(owner, "chain", None)

    @staticmethod
    def _is_link_field_cost(cost):
        return isinstance(cost, (CostCollision, CostGoal)) and cost.field is not None and \
            getattr(cost.field, "works_on_frames", False)

    def version(self):
        v = self._version
        for cost in self.cost_list:
            v += cost.version() if hasattr(cost, "version") else 0
        return v

    def descriptors(self):
        out = []
        for cost in self.cost_list:
            if not hasattr(cost, "descriptors"):
                raise TypeError(f"{type(cost).__name__} is not a stoch_gpmp_amd cost; wrap foreign costs "
                                "at the planner level (StochGPMP accepts any object with .eval)")
            if self.foreign_fk and self._is_link_field_cost(cost):
                continue                                # evaluated on the foreign FK's frames in eval()
            out.extend(cost.descriptors())
        return out

    def get_linear_system(self, trajs, **observation):
        """cost_functions.py:60-85: rows of all children stacked, K block-diagonal.  Assembly is plain
        tensor indexing (as in the reference); the field Jacobians inside come from the HIP kernel."""
        trajs = trajs.reshape(-1, self.traj_len, self.dim)
        As, bs, Ks = [], [], []
        for cost in self.cost_list:
            A, b, K = cost.get_linear_system(trajs, **observation)
            if A is None or b is None or K is None:
                continue
            As.append(A.detach()); bs.append(b.detach()); Ks.append(K.detach())
        A, b = torch.cat(As, dim=1), torch.cat(bs, dim=1)
        K = torch.zeros(trajs.shape[0], A.shape[1], A.shape[1], device=trajs.device, dtype=trajs.dtype)
        o = 0
        for Ki in Ks:
            K[:, o:o + Ki.shape[1], o:o + Ki.shape[1]] = Ki
            o += Ki.shape[1]
        return A, b, K

    def compile_into(self, engine):
        """Load this composite (and its FK chain) into an Engine's cost program."""
        if self.chain is not None:
            engine.set_fk(self.chain)
        engine.set_costs(self.descriptors())

    def needs_spheres(self):
        return any(dsc["kind"] == L.COST_SPHERES for cost in self.cost_list if hasattr(cost, "descriptors")
                   for dsc in cost.descriptors())

    def _engine(self, dtype, device):
        key = (dtype, str(device))
        v = self.version()
        ent = self._engines.get(key)
        if ent is None:
            eng = Engine(self.n_dof, self.traj_len, 0, 1, tensor_args={"device": device, "dtype": dtype})
            self.compile_into(eng)
            self._engines[key] = ent = [eng, v]
        elif ent[1] != v:                                # a child cost / field was edited since
            self.compile_into(ent[0])
            ent[1] = v
        return ent[0]

    def eval(self, trajs, **observation):
        trajs = trajs.reshape(-1, self.traj_len, self.dim)
        if not trajs.is_contiguous():
            trajs = trajs.contiguous()
        spheres = observation.get('obstacle_spheres', None)
        if spheres is None and self.needs_spheres():
            # the reference fails here too: LinkDistanceField returns int 0 and FieldFactor calls
            # .reshape on it (fields.py:64-65, field_factor.py:32)
            raise AttributeError("obstacle_spheres observation is required by LinkDistanceField costs")
        if spheres is not None:
            spheres = spheres.to(device=trajs.device, dtype=trajs.dtype).reshape(-1, 4).contiguous()
        costs = self._engine(trajs.dtype, trajs.device).cost_eval(trajs, spheres=spheres)
        if self.foreign_fk:
            # cost_functions.py:51-52: x_trajs = FK(q).reshape(B, T, -1, 4, 4); then every link-field
            # child as in CostCollision.eval / CostGoal.eval (cost_functions.py:247-261, 308-321)
            B = trajs.shape[0]
            x_trajs = self.FK(trajs.view(-1, self.dim)[:, :self.n_dof]).reshape(B, self.traj_len, -1, 4, 4)
            for cost in self.cost_list:
                if self._is_link_field_cost(cost):
                    factor = cost.obst_factor if isinstance(cost, CostCollision) else cost.goal_factor
                    err = factor.get_error(trajs, cost.field, x_trajs=x_trajs, calc_jacobian=False,
                                           obstacle_spheres=spheres)
                    costs = costs + factor.K * err.sum(1)
        return costs


class CostGP(Cost):
    """reference cost_functions.py:88-146: start-state unary factor + GP transition factors."""

    def __init__(self, n_dof, traj_len, start_state, dt, sigma_params, tensor_args, **kwargs):
        super().__init__(n_dof, traj_len)
        self.start_state = start_state
        self.dt = dt
        self.sigma_start = sigma_params['sigma_start']
        self.sigma_gp = sigma_params['sigma_gp']
        self.tensor_args = tensor_args
        self.set_cost_factors()

    def set_cost_factors(self):
        self.start_prior = UnaryFactor(self.dim, self.sigma_start, self.start_state, self.tensor_args)
        self._version = getattr(self, '_version', 0) + 1
        self.gp_prior = GPFactor(self.n_dof, self.sigma_gp, self.dt, self.traj_len - 1, self.tensor_args)

    def descriptors(self):
        return [dict(kind=L.COST_GP, flags=L.FLAG_GP_START, sigma=self.sigma_gp, sigma2=self.sigma_start,
                     dt=self.dt, host_data=_host_list(self.start_state))]

    def get_linear_system(self, trajs, x_trajs=None, **observation):
        """cost_functions.py:148-168: start factor (H = I) in the first block row, GP factor i
        (H1 = Phi on waypoint i, H2 = -I on waypoint i+1) in block row i+1."""
        trajs = trajs.reshape(-1, self.traj_len, self.dim)
        B, T, d = trajs.shape[0], self.traj_len, self.dim
        kw = dict(device=trajs.device, dtype=trajs.dtype)
        A = torch.zeros(B, d * T, d * T, **kw)
        b = torch.zeros(B, d * T, 1, **kw)
        K = torch.zeros(B, d * T, d * T, **kw)
        eye = torch.eye(d, **kw)
        A[:, :d, :d] = eye
        b[:, :d, 0] = self.start_state.to(**kw) - trajs[:, 0]
        K[:, :d, :d] = self.start_prior.K.to(**kw)
        phi, Q_inv = self.gp_prior.phi.to(**kw), self.gp_prior.Q_inv[0].to(**kw)
        for i in range(T - 1):
            A[:, (i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] = phi
            A[:, (i + 1) * d:(i + 2) * d, (i + 1) * d:(i + 2) * d] = -eye
            b[:, (i + 1) * d:(i + 2) * d, 0] = trajs[:, i + 1] - trajs[:, i] @ phi.t()
            K[:, (i + 1) * d:(i + 2) * d, (i + 1) * d:(i + 2) * d] = Q_inv
        return A, b, K


class CostGPTrajectory(Cost):
    """reference cost_functions.py:171-218: GP transition factors only."""

    def __init__(self, n_dof, traj_len, start_state, dt, sigma_params, tensor_args, **kwargs):
        super().__init__(n_dof, traj_len)
        self.start_state = start_state
        self.dt = dt
        self.sigma_gp = sigma_params['sigma_gp']
        self.tensor_args = tensor_args
        self.set_cost_factors()

    def set_cost_factors(self):
        self.gp_prior = GPFactor(self.n_dof, self.sigma_gp, self.dt, self.traj_len - 1, self.tensor_args)
        self._version = getattr(self, '_version', 0) + 1

    def descriptors(self):
        return [dict(kind=L.COST_GP, flags=0, sigma=self.sigma_gp, dt=self.dt)]


class CostCollision(Cost):
    """reference cost_functions.py:221-261: K * sum_{t=1}^{T-1} field(state_t)."""

    def __init__(self, n_dof, traj_len, field=None, sigma_coll=None, tensor_args=None):
        super().__init__(n_dof, traj_len)
        self.field = field
        self.sigma_coll = sigma_coll
        self.tensor_args = tensor_args
        self.set_cost_factors()

    def set_cost_factors(self):
        self.obst_factor = FieldFactor(self.n_dof, self.sigma_coll, [1, self.traj_len])
        self._version = getattr(self, '_version', 0) + 1

    def descriptors(self):
        if self.field is None:
            return []                                   # cost_functions.py:248-249: contributes 0
        if not hasattr(self.field, "descriptor"):
            raise TypeError(f"field {type(self.field).__name__} is not a stoch_gpmp_amd field")
        return [self.field.descriptor(self.sigma_coll)]

    def get_linear_system(self, trajs, x_trajs=None, **observation):
        """cost_functions.py:263-279: A [B,T-1,T*d] holds the field Jacobian H_i = -d field / d q of
        waypoint i+1 in the columns of that waypoint's positions, b = field values, K = I / sigma^2.
        The Jacobian comes from `field_grad_kernel` (analytic) instead of autograd."""
        if self.field is None:
            return None, None, None
        chain = getattr(self, "_chain", None)
        trajs = trajs.reshape(-1, self.traj_len, self.dim)
        B, T = trajs.shape[0], self.traj_len
        err, H = self.obst_factor.get_error(trajs, self.field, calc_jacobian=True, fk_chain=chain,
                                            obstacle_spheres=observation.get('obstacle_spheres', None))
        A = torch.zeros(B, T - 1, self.dim * T, device=trajs.device, dtype=trajs.dtype)
        rows = torch.arange(T - 1, device=trajs.device)
        cols = ((rows + 1) * self.dim).unsqueeze(1) + torch.arange(self.n_dof, device=trajs.device)
        A[:, rows.unsqueeze(1), cols] = H
        K = self.obst_factor.K * torch.eye(T - 1, device=trajs.device, dtype=trajs.dtype).repeat(B, 1, 1)
        return A, err.unsqueeze(-1), K


class CostGoalPrior(Cost):
    """reference cost_functions.py:340-388: per-goal unary factor on the last waypoint;
    row b of the batch belongs to goal b // (num_particles_per_goal * num_samples)."""

    def __init__(self, n_dof, traj_len, multi_goal_states=None, num_particles_per_goal=None,
                 num_samples=None, sigma_goal_prior=None, tensor_args=None):
        super().__init__(n_dof, traj_len)
        self.multi_goal_states = multi_goal_states
        self.num_goals = multi_goal_states.shape[0]
        self.num_particles_per_goal = num_particles_per_goal
        self.num_particles = num_particles_per_goal * self.num_goals
        self.num_samples = num_samples
        self.sigma_goal_prior = sigma_goal_prior
        self.tensor_args = tensor_args
        self.set_cost_factors()

    def set_cost_factors(self):
        self.multi_goal_prior = [UnaryFactor(self.dim, self.sigma_goal_prior, self.multi_goal_states[i],
                                             self.tensor_args) for i in range(self.num_goals)]
        self._version = getattr(self, '_version', 0) + 1

    def descriptors(self):
        return [dict(kind=L.COST_GOAL_PRIOR, sigma=self.sigma_goal_prior, dim0=self.num_goals,
                     dim1=self.num_particles_per_goal * self.num_samples,
                     host_data=_host_list(self.multi_goal_states))]

    def get_linear_system(self, trajs, x_trajs=None, **observation):
        """cost_functions.py:390-405: one unary factor on the last waypoint, goal of particle p is
        p // num_particles_per_goal."""
        trajs = trajs.reshape(-1, self.traj_len, self.dim)
        B, T, d = trajs.shape[0], self.traj_len, self.dim
        kw = dict(device=trajs.device, dtype=trajs.dtype)
        A = torch.zeros(B, d, d * T, **kw)
        A[:, :, -d:] = torch.eye(d, **kw)
        goals = self.multi_goal_states.to(**kw).repeat_interleave(self.num_particles_per_goal, dim=0)
        b = (goals - trajs[:, -1]).unsqueeze(-1)
        K = (torch.eye(d, **kw) / self.sigma_goal_prior ** 2).repeat(B, 1, 1)
        return A, b, K


class CostGoal(Cost):
    """reference cost_functions.py:282-321: K * field(last waypoint), field = EESE3DistanceField."""

    def __init__(self, n_dof, traj_len, field=None, sigma_goal=None, tensor_args=None):
        super().__init__(n_dof, traj_len)
        self.field = field
        self.sigma_goal = sigma_goal
        self.tensor_args = tensor_args
        self.set_cost_factors()

    def set_cost_factors(self):
        self.goal_factor = FieldFactor(self.n_dof, self.sigma_goal, [self.traj_len - 1, self.traj_len])
        self._version = getattr(self, '_version', 0) + 1

    def descriptors(self):
        if self.field is None:
            return []
        if not hasattr(self.field, "descriptor"):
            raise TypeError(f"field {type(self.field).__name__} is not a stoch_gpmp_amd field")
        return [self.field.descriptor(self.sigma_goal)]

    def get_linear_system(self, trajs, x_trajs=None, **observation):
        """cost_functions.py:323-337: A [B,1,T*d] holds the field Jacobian H = -d field / d q of the LAST waypoint
        in that waypoint's position columns, b = field value, K = 1 / sigma^2.  The Jacobian is analytic
        (`ee_grad_kernel`: position part u . (z_j x (p - o_j)), rotation part z_j . axis) instead of autograd."""
        if self.field is None:
            return None, None, None
        chain = getattr(self, "_chain", None)
        trajs = trajs.reshape(-1, self.traj_len, self.dim)
        B, T = trajs.shape[0], self.traj_len
        err, H = self.goal_factor.get_error(trajs, self.field, calc_jacobian=True, fk_chain=chain)
        A = torch.zeros(B, 1, self.dim * T, device=trajs.device, dtype=trajs.dtype)
        A[:, :, (T - 1) * self.dim:(T - 1) * self.dim + self.n_dof] = H
        K = self.goal_factor.K * torch.ones(B, 1, 1, device=trajs.device, dtype=trajs.dtype)
        return A, err.unsqueeze(-1), K
