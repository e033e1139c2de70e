"""Random sphere obstacles for the Panda scenes -- `random_init_static_sphere` with the signature and
random-number consumption of reference `stoch_gpmp/envs/panda.py:42-66` (without the PyBullet
environment around it), plus the spawning loop of `examples/panda_environment.py:124-133`.
Setup-time host code (numpy global generator), SURVEY.md 8f rank 4.
"""
import numpy as np
import torch


def random_init_static_sphere(scale_min, scale_max, base_position_min, base_position_max, base_offset):
    """-> (radius, centre[3]).  Draw order: uniform() radius blend; permutation picking ONE axis;
    rand(3) centre; rand(1) blend that places the picked axis inside [min, max]; randint signs for
    x and y; finally |centre| is clipped into [base_offset, base_position_max] per axis."""
    lo = np.asarray(base_position_min, dtype=float)
    hi = np.asarray(base_position_max, dtype=float)
    a = np.random.uniform()
    radius = a * scale_min + (1 - a) * scale_max
    picked = np.random.permutation([1, 0, 0]) == 1
    centre = np.random.rand(3)
    blend = np.random.rand(1)
    centre[picked] = blend * lo[picked] + (1 - blend) * hi[picked]
    centre[:-1] *= np.random.randint(2, size=2) * 2 - 1
    centre = np.sign(centre) * np.clip(np.abs(centre), a_min=base_offset, a_max=hi)
    return radius, centre


def spawn_obstacle_spheres(num_obst, obst_r=(0.1, 0.2), obst_range_lower=(0.6, -0.2, 0.6),
                           obst_range_upper=(1., 0.2, 1.), base_offset=0.01, tensor_args=None):
    """The reference example's obstacle set: [1, num_obst, 4] rows (x, y, z, r)
    (examples/panda_environment.py:124-133), as the `obstacle_spheres` observation."""
    out = np.zeros((1, num_obst, 4))
    for i in range(num_obst):
        r, pos = random_init_static_sphere(obst_r[0], obst_r[1], np.asarray(obst_range_lower),
                                           np.asarray(obst_range_upper), base_offset)
        out[0, i, :3] = pos
        out[0, i, 3] = r
    t = torch.from_numpy(out)
    return t if tensor_args is None else t.to(**tensor_args)
