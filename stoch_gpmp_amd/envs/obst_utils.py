"""Random obstacle helpers with the names of reference `stoch_gpmp/envs/obst_utils.py`.

Positions are drawn from Python's global `random` module (two `random.uniform` calls per obstacle,
x then y), like the reference (obst_utils.py:11-27), so that a script seeding `random` gets the same
scene from either package (tests/test_cpu_host.py checks this against the committed reference grid).
"""
import random
from math import ceil

from .obst_map import ObstacleCircle, ObstacleRectangle


def round_up(n, decimals=0):
    scale = 10 ** decimals
    return ceil(n * scale) / scale


def random_rect(xlim=(0, 0), ylim=(0, 0), width=2, height=2):
    cx = random.uniform(xlim[0], xlim[1])
    cy = random.uniform(ylim[0], ylim[1])
    return ObstacleRectangle(cx, cy, width, height)


def random_circle(xlim=(0, 0), ylim=(0, 0), radius=2):
    cx = random.uniform(xlim[0], xlim[1])
    cy = random.uniform(ylim[0], ylim[1])
    return ObstacleCircle(cx, cy, radius)
