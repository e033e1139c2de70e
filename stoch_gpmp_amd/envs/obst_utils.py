"""Random obstacle helpers (names of reference `stoch_gpmp/envs/obst_utils.py`).

Every obstacle centre is drawn by ONE helper, `_draw_centre`, from Python's global `random` module:
x first, then y, two `random.uniform` calls.  That is the reference's consumption of the stream
(obst_utils.py:15-16, 24-25), so a script that seeds `random` gets the same scene from either
package; tests/test_cpu_host.py checks it against the committed reference grid.
"""
import math
import random

from .obst_map import ObstacleCircle, ObstacleRectangle


def _draw_centre(xlim, ylim):
    lo_x, hi_x = xlim
    lo_y, hi_y = ylim
    x = random.uniform(lo_x, hi_x)
    return x, random.uniform(lo_y, hi_y)


def round_up(n, decimals=0):
    """Smallest multiple of 10**-decimals that is >= n."""
    step = 10.0 ** decimals
    return math.ceil(n * step) / step


def random_rect(xlim=(0, 0), ylim=(0, 0), width=2, height=2):
    """Axis-aligned rectangle of the given size at a uniformly random centre."""
    return ObstacleRectangle(*_draw_centre(xlim, ylim), width, height)


def random_circle(xlim=(0, 0), ylim=(0, 0), radius=2):
    """Circle of the given radius at a uniformly random centre."""
    return ObstacleCircle(*_draw_centre(xlim, ylim), radius)
