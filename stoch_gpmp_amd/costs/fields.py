"""Link distance fields -- same classes and signatures as reference `stoch_gpmp/costs/fields.py`,
evaluated by the HIP kernels in csrc/cost_sweep.hip.

Inside a `CostComposite` these objects are only *descriptors* (the cost sweep fuses FK and the field
into one pass); `compute_cost(link_tensor, ...)` on explicit frames is kept for API parity and runs
`field_eval_kernel`.
"""
from abc import ABC, abstractmethod

import torch

from .. import _lib as L
from ..engine import Engine


def _linspace_alpha(num_interpolate):
    # fields.py:69 -- torch.linspace(0, 1, K + 2)[1:K + 1] evaluated in fp32 then cast
    if num_interpolate <= 0:
        return []
    return [float(v) for v in torch.linspace(0, 1, num_interpolate + 2)[1:num_interpolate + 1]]


class DistanceField(ABC):
    def __init__(self, tensor_args=None):
        self.tensor_args = tensor_args
        self._engines = {}
        self._version = 0           # edit counter: planners re-compile their cost program when it moves

    @abstractmethod
    def descriptor(self, sigma):
        """-> dict understood by Engine.set_costs (one cost term with weight 1/sigma^2)."""

    def distances(self, *args, **kwargs):
        """fields.py:11-12: the base class declares it and does nothing (the link fields override it)."""

    def compute_collision(self, *args, **kwargs):
        """fields.py:14-15: likewise."""

    @abstractmethod
    def compute_cost(self, *args, **kwargs):
        pass

    def zero_grad(self):
        pass

    def _engine(self, dtype, device):
        key = (dtype, str(device))
        if key not in self._engines:
            eng = Engine(1, 2, 0, 1, tensor_args={"device": device, "dtype": dtype})
            eng.set_costs([self.descriptor(1.0)])
            self._engines[key] = eng
        return self._engines[key]

    def compute_cost_and_grad(self, q, chain, **observations):
        """Field value [B] and its gradient [B,n] with respect to the joint positions q [B,n], through
        the FK chain `chain` (a stoch_gpmp_amd URDF chain) -- what the reference obtains with
        torch.autograd.grad in FieldFactor.get_error(calc_jacobian=True) (field_factor.py:34-38),
        here analytically in `field_grad_kernel`."""
        n = sum(1 for j in chain if j[1] == "revolute")
        key = ("grad", q.dtype, str(q.device), id(chain))
        if key not in self._engines:
            eng = Engine(n, 2, 0, 1, tensor_args={"device": q.device, "dtype": q.dtype})
            eng.set_fk(chain)
            eng.set_costs([self.descriptor(1.0)])
            self._engines[key] = eng
        sph = observations.get("obstacle_spheres", None)
        if sph is not None:
            sph = sph.to(device=q.device, dtype=q.dtype).reshape(-1, 4).contiguous()
        return self._engines[key].field_grad(0, q.reshape(-1, n).contiguous(), sph)


class LinkDistanceField(DistanceField):
    """reference fields.py:30-89 (rbf / sdf / occupancy against sphere obstacles)."""
    works_on_frames = True          # compute_cost takes link frames [.., L, 4, 4]
    _TYPES = {"rbf": L.FIELD_RBF, "sdf": L.FIELD_SDF, "occupancy": L.FIELD_OCCUPANCY}

    def __init__(self, field_type='rbf', clamp_sdf=False, num_interpolate=0,
                 link_interpolate_range=[5, 7], **kwargs):
        super().__init__(**kwargs)
        if field_type not in self._TYPES:
            raise ValueError(f"unknown field_type {field_type!r}")
        self.field_type = field_type
        self.clamp_sdf = clamp_sdf
        self.num_interpolate = num_interpolate
        self.link_interpolate_range = link_interpolate_range

    def descriptor(self, sigma):
        return dict(kind=L.COST_SPHERES,
                    flags=self._TYPES[self.field_type] | (L.FLAG_SDF_CLAMP if self.clamp_sdf else 0),
                    sigma=sigma, num_interpolate=self.num_interpolate,
                    interp_lo=self.link_interpolate_range[0], interp_hi=self.link_interpolate_range[1],
                    alpha=_linspace_alpha(self.num_interpolate))

    def _dist(self, link_tensor, obstacle_spheres, mode, buffer=0.0):
        frames = link_tensor.contiguous()
        sph = obstacle_spheres.to(device=frames.device, dtype=frames.dtype).reshape(-1, 4).contiguous()
        return self._engine(frames.dtype, frames.device).link_distances(frames, sph, mode, buffer), sph.shape[0]

    def distances(self, link_tensor, obstacle_spheres):
        """fields.py:40-46: |p_l - c_o| - r_o for every link frame and sphere -> [.., L, O]."""
        out, n_sph = self._dist(link_tensor, obstacle_spheres, 0)
        return out.reshape(tuple(link_tensor.shape[:-2]) + (n_sph,))

    def compute_collision(self, link_tensor, obstacle_spheres=None, buffer=0.02):
        """fields.py:48-54: does any link come closer than `buffer` to any sphere? -> bool [..]"""
        if obstacle_spheres is None:
            return torch.zeros(link_tensor.shape[:2]).to(**self.tensor_args)
        out, _ = self._dist(link_tensor, obstacle_spheres, 1, buffer)
        return out.reshape(link_tensor.shape[:-3]) != 0

    def compute_distance(self, link_tensor, obstacle_spheres=None, **kwargs):
        """fields.py:56-61: the signed link-sphere distances summed over links and spheres."""
        if obstacle_spheres is None:
            return 1e10
        out, _ = self._dist(link_tensor, obstacle_spheres, 2)
        return out.reshape(link_tensor.shape[:-3])

    def compute_cost(self, link_tensor, obstacle_spheres=None, **kwargs):
        if obstacle_spheres is None:
            return 0                                        # fields.py:64-65
        shape = link_tensor.shape[:-3]
        frames = link_tensor.contiguous()
        sph = obstacle_spheres.to(frames.dtype).contiguous()
        out = self._engine(frames.dtype, frames.device).field_eval(0, frames, sph)
        return out.reshape(shape)


class LinkSelfDistanceField(DistanceField):
    """reference fields.py:92-127 (pairwise rbf over all link points, diagonal included)."""
    works_on_frames = True

    def __init__(self, margin=0.03, num_interpolate=0, link_interpolate_range=[5, 7], **kwargs):
        super().__init__(**kwargs)
        self.num_interpolate = num_interpolate
        self.link_interpolate_range = link_interpolate_range
        self.margin = margin

    def descriptor(self, sigma):
        return dict(kind=L.COST_SELF, sigma=sigma, sigma2=self.margin,
                    num_interpolate=self.num_interpolate,
                    interp_lo=self.link_interpolate_range[0], interp_hi=self.link_interpolate_range[1],
                    alpha=_linspace_alpha(self.num_interpolate))

    def _dist(self, link_tensor, mode, buffer=0.0):
        frames = link_tensor.contiguous()
        return self._engine(frames.dtype, frames.device).link_distances(frames, None, mode, buffer)

    def distances(self, link_tensor):
        """fields.py:100-102: pairwise link distances -> [.., L, L]."""
        return self._dist(link_tensor, 0).reshape(tuple(link_tensor.shape[:-2]) + (link_tensor.shape[-3],))

    def compute_collision(self, link_tensor, buffer=0.05):
        """fields.py:104-108: any pair of links at least two apart in the chain closer than `buffer`? -> bool [..]"""
        return self._dist(link_tensor, 1, buffer).reshape(link_tensor.shape[:-3]) != 0

    def compute_distance(self, link_tensor):
        """fields.py:110-112: sum of all pairwise link distances."""
        return self._dist(link_tensor, 2).reshape(link_tensor.shape[:-3])

    def compute_cost(self, link_tensor, **kwargs):
        shape = link_tensor.shape[:-3]
        frames = link_tensor.contiguous()
        return self._engine(frames.dtype, frames.device).field_eval(0, frames).reshape(shape)


def SE3_distance(H1, H2, w_pos=1., w_rot=1.):
    """The SE(3) distance this build uses:  w_pos |p1 - p2| + w_rot angle(R2^T R1).

    The reference imports `SE3_distance` from the un-vendored, un-versioned `torch_robotics`
    (reference fields.py:4,143-144); its exact formula cannot be checked here, so this is a documented
    definition (PARITY UNPINNED), evaluated by the HIP kernels `ee_goal_kernel` /
    `ee_field_kernel`.  This helper exists for API parity and runs the same kernel."""
    field = EESE3DistanceField(H2, w_pos=w_pos, w_rot=w_rot, square=False,
                               tensor_args={"device": H1.device, "dtype": H1.dtype})
    return field.compute_distance(H1.unsqueeze(-3))


class EESE3DistanceField(DistanceField):
    """reference fields.py:130-153: distance of the LAST link frame to a target frame."""
    works_on_frames = True

    def __init__(self, target_H, w_pos=1., w_rot=1., square=True, **kwargs):
        super().__init__(**kwargs)
        self.target_H = target_H
        self.square = square
        self.w_pos = w_pos
        self.w_rot = w_rot

    def update_target(self, target_H):
        """fields.py:140-141.  The reference reads `target_H` on every eval; here the target is part
        of the compiled cost program, so the edit counter tells every planner / composite holding
        this field to re-compile before its next evaluation."""
        self.target_H = target_H
        self._engines = {}
        self._version += 1

    def descriptor(self, sigma, square=None):
        H = torch.as_tensor(self.target_H).detach().cpu().double().reshape(-1, 4, 4)[0]
        sq = self.square if square is None else square
        return dict(kind=L.COST_EE_GOAL, flags=L.FLAG_EE_SQUARE if sq else 0, sigma=sigma,
                    host_data=[float(v) for v in H.flatten()], p0=self.w_pos, p1=self.w_rot)

    def _field(self, link_tensor, square):
        shape = link_tensor.shape[:-3]
        frames = link_tensor.contiguous()
        key = (frames.dtype, str(frames.device), bool(square))
        if key not in self._engines:
            eng = Engine(1, 2, 0, 1, tensor_args={"device": frames.device, "dtype": frames.dtype})
            eng.set_costs([self.descriptor(1.0, square=square)])
            self._engines[key] = eng
        return self._engines[key].field_eval(0, frames).reshape(shape)

    def compute_distance(self, link_tensor):
        return self._field(link_tensor, False)

    def compute_cost(self, link_tensor, **kwargs):
        return self._field(link_tensor, self.square)
