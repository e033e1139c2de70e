"""`generate_obstacle_map` with the signature and random-number consumption of reference
`stoch_gpmp/envs/map_generator.py:9-92` (setup-time scene tooling, host side).

Fixed obstacles are painted first; then, until `num_obst` obstacles exist, a coin
(`np.random.choice(2)`: 1 -> rectangle, 0 -> circle) picks the shape, `random.uniform` draws its
centre, and the obstacle is kept only if it overlaps nothing already on the map (at most 26 tries
per obstacle).  Same draws in the same order as the reference => same scene for the same seeds.
"""
import copy

import numpy as np

from .obst_map import ObstacleMap
from .obst_utils import random_circle, random_rect

MAX_ATTEMPTS = 25


def _overlaps(obst, obst_map):
    """Would painting `obst` create a cell covered twice? (reference Obstacle._obstacle_collision_check)"""
    trial = copy.deepcopy(obst_map)
    obst._add_to_map(trial)
    return bool(np.any(trial.map > 1))


def generate_obstacle_map(map_dim=(10, 10), obst_list=[], cell_size=1., random_gen=False, num_obst=0,
                          rand_limits=None, rand_rect_shape=[2, 2], rand_circle_radius=1,
                          tensor_args=None):
    """-> (ObstacleMap, list of obstacles); map origin at the grid centre, dims even (asserted)."""
    obst_map = ObstacleMap(map_dim, cell_size, tensor_args=tensor_args)
    n_fixed = len(obst_list)
    for obst in obst_list:
        obst._add_to_map(obst_map)
    obstacles = copy.deepcopy(obst_list)
    if random_gen:
        assert n_fixed <= num_obst, \
            "Total number of obstacles must be greater than or equal to number specified in obst_list"
        (xlim, ylim), (width, height) = rand_limits, rand_rect_shape
        for _ in range(num_obst - n_fixed):
            for attempt in range(MAX_ATTEMPTS + 1):
                if np.random.choice(2):
                    obst = random_rect(xlim, ylim, width, height)
                else:
                    obst = random_circle(xlim, ylim, rand_circle_radius)
                if not _overlaps(obst, obst_map):
                    obst._add_to_map(obst_map)
                    obstacles.append(obst)
                    break
                if attempt == MAX_ATTEMPTS:
                    print("Obstacle generation: Max. number of attempts reached. ")
                    print("Total num. obstacles: {}.  Num. random obstacles: {}.\n"
                          .format(len(obstacles), len(obstacles) - n_fixed))
    obst_map.convert_map()
    return obst_map, obstacles
