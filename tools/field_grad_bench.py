"""Throughput of the analytic field Jacobian kernel (field_grad_kernel) at config-3 waypoint counts."""
import sys, time; sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from stoch_gpmp_amd import workloads as W
from stoch_gpmp_amd.costs.fields import LinkDistanceField, LinkSelfDistanceField
from stoch_gpmp_amd.robots.panda_chain import PANDA_CHAIN
for dtype in (torch.float32, torch.float64):
    ta = {"device": torch.device("cuda:0"), "dtype": dtype}
    B = 1024 * 128 * 63 // 16                      # 1/16 of config 3's waypoints
    q = (torch.rand(B, 7, **ta) * 4 - 2)
    sph = torch.as_tensor(W.panda_spheres()).to(**ta)
    for name, f, obs in (("rbf spheres (O=5)", LinkDistanceField(tensor_args=ta), {"obstacle_spheres": sph}),
                         ("self", LinkSelfDistanceField(tensor_args=ta), {})):
        for _ in range(3): f.compute_cost_and_grad(q, PANDA_CHAIN, **obs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f.compute_cost_and_grad(q, PANDA_CHAIN, **obs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{str(dtype):14s} {name:18s} B={B}: {dt*1e6:8.1f} us  {B/dt/1e9:.3f} G configurations/s")
