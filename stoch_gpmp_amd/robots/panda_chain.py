"""Franka Panda (no gripper) joint table as plain data: (name, type, rpy, xyz) per joint, typed from
assets/franka_description/robots/panda_arm_no_gripper.urdf:41-47,66-72,91-97,116-122,141-147,166-172,
191-197 (revolute, axis z) and :200-204,206-210,230-235 (fixed) of the reference.  No imports, so the
build-time chain code generator (csrc/gen/chain_codegen.py) can load it without torch."""

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
PANDA_LINK_NAMES = ["panda_link0", "panda_link1", "panda_link2", "panda_link3", "panda_link4",
                    "panda_link5", "panda_link6", "panda_link7", "panda_link8", "panda_hand",
                    "ee_link"]
# joint limits (URDF :47,72,97,122,147,172,197)
PANDA_Q_LOWER = [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973]
PANDA_Q_UPPER = [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973]
