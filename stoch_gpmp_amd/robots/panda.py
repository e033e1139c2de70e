"""Franka Panda (no gripper) kinematic chain as plain URDF constants.

The reference gets FK from the third-party `torch_robotics.DifferentiableFrankaPanda`
(reference examples/panda_environment.py:13,47,98), which is neither vendored nor versioned; this
class provides the same call -- `compute_forward_kinematics_all_links(q) -> [B, L, 4, 4]` -- backed by
the HIP FK kernel, with the joint origins typed from
assets/franka_description/robots/panda_arm_no_gripper.urdf:41-47,66-72,91-97,116-122,141-147,
166-172,191-197 (revolute, axis z) and :200-204,206-210,230-235 (fixed).  Link table (documented
choice, the third-party one cannot be verified): panda_link0..8, panda_hand, ee_link (L = 11).
"""
import torch

from ..engine import Engine

from .panda_chain import PANDA_CHAIN, PANDA_LINK_NAMES, PANDA_Q_LOWER, PANDA_Q_UPPER  # noqa: F401


class URDFChain:
    """A serial chain of revolute(z)/fixed joints; FK runs in the HIP kernel `fk_frames_kernel`."""

    def __init__(self, chain, link_names=None, device=None, tensor_args=None):
        self.chain = list(chain)
        self._n_dofs = sum(1 for j in self.chain if j[1] == "revolute")
        self.link_names = link_names or [f"link{i}" for i in range(len(self.chain) + 1)]
        if tensor_args is None:
            tensor_args = {"device": device if device is not None else torch.device("cuda:0"),
                           "dtype": torch.float32}
        self.tensor_args = tensor_args
        self._engines = {}

    def print_link_names(self):
        for i, name in enumerate(self.link_names):
            print(i, name)

    def _engine(self, dtype, device):
        key = (dtype, str(device))
        if key not in self._engines:
            eng = Engine(self._n_dofs, 2, 0, 1, tensor_args={"device": device, "dtype": dtype})
            eng.set_fk(self.chain, codegen=False)           # (frames only: no sweep kernels to compile)
            self._engines[key] = eng
        return self._engines[key]

    def compute_forward_kinematics_all_links(self, q):
        """q [B, n_dofs] -> homogeneous link frames [B, L, 4, 4]."""
        q = q.contiguous()
        return self._engine(q.dtype, q.device).fk(q)


class DifferentiableFrankaPanda(URDFChain):
    """Name-compatible stand-in for torch_robotics' class (forward pass only)."""

    def __init__(self, gripper=False, device=None, tensor_args=None):
        if gripper:
            raise NotImplementedError("only the gripper-less chain of the reference example is built")
        super().__init__(PANDA_CHAIN, PANDA_LINK_NAMES, device=device, tensor_args=tensor_args)
