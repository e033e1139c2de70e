"""2-D occupancy-grid obstacle field -- the `ObstacleMap` interface of reference
`stoch_gpmp/envs/obst_map.py` (`compute_cost` / `get_collisions` / `__call__`), with the lookup
(obst_map.py:164-182) running in the HIP kernels (`grid_value` in csrc/cost_sweep.hip).

The rasteriser below is this package's own (setup-time, host side, numpy): obstacles are painted
into `map` with the same cell conventions as the reference (`map[y, x]`, origin in the grid
centre) so that grids produced by either side are interchangeable.
"""
from math import ceil

import numpy as np
import torch

from .. import _lib as L
from ..engine import Engine


class ObstacleRectangle:
    def __init__(self, center_x=0, center_y=0, width=None, height=None):
        self.center_x, self.center_y, self.width, self.height = center_x, center_y, width, height

    def _add_to_map(self, obst_map):
        cs = obst_map.cell_size
        w, h = ceil(self.width / cs), ceil(self.height / cs)
        cx, cy = ceil(self.center_x / cs), ceil(self.center_y / cs)
        y0, y1 = cy - ceil(h / 2.) + obst_map.origin_yi, cy + ceil(h / 2.) + obst_map.origin_yi
        x0, x1 = cx - ceil(w / 2.) + obst_map.origin_xi, cx + ceil(w / 2.) + obst_map.origin_xi
        obst_map.map[max(y0, 0):max(y1, 0), max(x0, 0):max(x1, 0)] += 1
        return obst_map


class ObstacleCircle:
    def __init__(self, center_x=0, center_y=0, radius=1.):
        self.center_x, self.center_y, self.radius = center_x, center_y, radius

    def _add_to_map(self, obst_map):
        cs = obst_map.cell_size
        ny, nx = obst_map.map.shape
        ys = (np.arange(ny) - obst_map.origin_yi) * cs
        xs = (np.arange(nx) - obst_map.origin_xi) * cs
        inside = np.sqrt((xs[None, :] - self.center_x) ** 2 + (ys[:, None] - self.center_y) ** 2) \
            <= self.radius
        obst_map.map[inside] += 1
        return obst_map


class ObstacleMap:
    """Occupancy grid; field object accepted by CostCollision(field=...)."""

    def __init__(self, map_dim, cell_size, tensor_args=None):
        assert map_dim[0] % 2 == 0
        assert map_dim[1] % 2 == 0
        if tensor_args is None:
            tensor_args = {'device': torch.device('cuda:0'), 'dtype': torch.float32}
        self.tensor_args = tensor_args
        cmap_dim = [ceil(map_dim[0] / cell_size), ceil(map_dim[1] / cell_size)]
        self.map = np.zeros(cmap_dim)
        self.cell_size = cell_size
        self.origin_xi = int(cmap_dim[0] / 2)
        self.origin_yi = int(cmap_dim[1] / 2)
        self.x_dim, self.y_dim = self.map.shape
        self.xlim = [-self.cell_size * self.x_dim / 2, self.cell_size * self.x_dim / 2]
        self.ylim = [-self.cell_size * self.y_dim / 2, self.cell_size * self.y_dim / 2]
        self.map_torch = None
        self._engines = {}

    @property
    def c_offset(self):
        return torch.tensor([self.origin_xi, self.origin_yi], **self.tensor_args)

    @classmethod
    def from_grid(cls, grid, cell_size, tensor_args=None):
        """Wrap an existing [ny, nx] occupancy array (e.g. one produced by the reference)."""
        grid = np.asarray(grid, dtype=np.float64)
        om = cls([int(round(grid.shape[0] * cell_size)), int(round(grid.shape[1] * cell_size))],
                 cell_size, tensor_args=tensor_args)
        assert om.map.shape == grid.shape, (om.map.shape, grid.shape)
        om.map = grid.copy()
        om.convert_map()
        return om

    def convert_map(self):
        self.map_torch = torch.tensor(self.map, **self.tensor_args).contiguous()
        self._engines = {}
        return self.map_torch

    def get_xy_grid(self, device):
        """World coordinates of the grid nodes, [x_dim, y_dim, 2] (obst_map.py:158-162; a plotting / sampling helper
        with no caller in the reference): x_dim points across xlim along axis 0, y_dim across ylim along axis 1."""
        xs = torch.linspace(self.xlim[0], self.xlim[1], self.x_dim)
        ys = torch.linspace(self.ylim[0], self.ylim[1], self.y_dim)
        return torch.stack((xs[:, None].expand(self.x_dim, self.y_dim), ys[None, :].expand(self.x_dim, self.y_dim)),
                           dim=2).to(device)

    def descriptor(self, sigma):
        if self.map_torch is None:
            self.convert_map()
        return dict(kind=L.COST_GRID, sigma=sigma, device_tensor=self.map_torch,
                    dim0=self.map.shape[0], dim1=self.map.shape[1], p0=self.cell_size,
                    p1=float(self.origin_xi), p2=float(self.origin_yi))

    def _engine(self, dtype, device):
        key = (dtype, str(device))
        if key not in self._engines:
            eng = Engine(2, 2, 0, 1, tensor_args={"device": device, "dtype": dtype})
            eng.set_costs([self.descriptor(1.0)])
            self._engines[key] = eng
        return self._engines[key]

    def plot(self, save_dir=None, filename="obst_map.png"):
        """obst_map.py:149-156: the occupancy grid as an image (matplotlib, host side; not part of the planning path)."""
        import os.path as osp
        import matplotlib.pyplot as plt
        fig = plt.figure()
        plt.imshow(self.map)
        plt.gca().invert_yaxis()
        if save_dir is not None:
            plt.savefig(osp.join(save_dir, filename))
        return fig

    def get_collisions(self, X, **kwargs):
        """X [..., 2] -> occupancy value at each point (reference obst_map.py:164-182)."""
        shape = X.shape[:-1]
        xy = X[..., :2].contiguous()
        return self._engine(xy.dtype, xy.device).grid_lookup(0, xy).reshape(shape)

    def compute_cost(self, X, **kwargs):
        return self.get_collisions(X, **kwargs)

    def __call__(self, X, **kwargs):
        return self.compute_cost(X, **kwargs)

    def zero_grad(self):
        pass


def synthetic_obstacle_map(seed=0, map_dim=(20, 20), cell_size=0.1, num_obst=15,
                           rand_limits=((-7.5, 7.5), (-7.5, 7.5)), rect_shape=(2, 2),
                           circle_radius=1., tensor_args=None):
    """Seeded random scene of non-overlapping 2x2 rectangles / r=1 circles in the spirit of the
    reference example (examples/planar_environment.py:37-49); this package's own generator
    (numpy Generator), used by bench.py and the tests where the reference is unavailable."""
    rng = np.random.default_rng(seed)
    om = ObstacleMap(list(map_dim), cell_size, tensor_args=tensor_args)
    placed = 0
    for _ in range(num_obst * 26):
        if placed == num_obst:
            break
        cx = rng.uniform(*rand_limits[0])
        cy = rng.uniform(*rand_limits[1])
        ob = ObstacleRectangle(cx, cy, *rect_shape) if rng.integers(2) else \
            ObstacleCircle(cx, cy, circle_radius)
        before = om.map.copy()
        ob._add_to_map(om)
        if np.any(om.map > 1):
            om.map = before
        else:
            placed += 1
    om.convert_map()
    return om
