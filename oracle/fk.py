"""Forward kinematics oracle for a serial URDF chain -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference takes FK as an injected callable from the un-vendored
`torch_robotics` package (reference costs/cost_functions.py:39,51-52;
examples/panda_environment.py:47,98).  This file restates standard URDF semantics

    H_child = H_parent . Trans(xyz) . Rz(yaw) Ry(pitch) Rx(roll) . Rz(q)      (axis = +z)

for the chain in assets/franka_description/robots/panda_arm_no_gripper.urdf:41-47,66-72,91-97,
116-122,141-147,166-172,191-197 (revolute) and :200-204,206-210,230-235 (fixed).  The link table
returned is the build's documented choice: panda_link0..8, panda_hand, ee_link (L = 11).
"""
import math

import torch

# (name, type, rpy, xyz) in chain order; constants typed from the URDF lines cited above and held to the URDF itself
# by tests/test_oracle_golden.py::test_panda_chain_tables_equal_the_reference_urdf (fixture g9: the file, parsed).
PANDA_CHAIN = [
    ("panda_joint1", "revolute", (0.0, 0.0, 0.0), (0.0, 0.0, 0.333)),
    ("panda_joint2", "revolute", (-1.57079632679, 0.0, 0.0), (0.0, 0.0, 0.0)),
    ("panda_joint3", "revolute", (1.57079632679, 0.0, 0.0), (0.0, -0.316, 0.0)),
    ("panda_joint4", "revolute", (1.57079632679, 0.0, 0.0), (0.0825, 0.0, 0.0)),
    ("panda_joint5", "revolute", (-1.57079632679, 0.0, 0.0), (-0.0825, 0.384, 0.0)),
    ("panda_joint6", "revolute", (1.57079632679, 0.0, 0.0), (0.0, 0.0, 0.0)),
    ("panda_joint7", "revolute", (1.57079632679, 0.0, 0.0), (0.088, 0.0, 0.0)),
    ("panda_joint8", "fixed", (0.0, 0.0, 0.0), (0.0, 0.0, 0.107)),
    ("panda_hand_joint", "fixed", (0.0, 0.0, -0.785398163397), (0.0, 0.0, 0.0)),
    ("ee_fixed_joint", "fixed", (0.0, 0.0, -1.57), (0.0, 0.0, 0.1)),
]


def _origin(rpy, xyz, dtype):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = (math.cos(r), math.sin(r), math.cos(p), math.sin(p),
                              math.cos(y), math.sin(y))
    H = torch.eye(4, dtype=dtype)
    H[:3, :3] = torch.tensor([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr]], dtype=dtype)
    H[:3, 3] = torch.tensor(xyz, dtype=dtype)
    return H


def fk_all_links(q, chain=PANDA_CHAIN):
    """q [B, n_revolute] -> link frames [B, 1 + len(chain), 4, 4] (base frame first)."""
    B, dtype = q.shape[0], q.dtype
    H = torch.eye(4, dtype=dtype).expand(B, 4, 4).clone()
    out = [H]
    k = 0
    for _, kind, rpy, xyz in chain:
        H = H @ _origin(rpy, xyz, dtype)
        if kind == "revolute":
            c, s = torch.cos(q[:, k]), torch.sin(q[:, k])
            R = torch.zeros(B, 4, 4, dtype=dtype)
            R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1] = c, -s, s, c
            R[:, 2, 2] = 1.
            R[:, 3, 3] = 1.
            H = H @ R
            k += 1
        out.append(H)
    return torch.stack(out, dim=1)
