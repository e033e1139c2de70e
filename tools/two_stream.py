"""Experiment: one GPU, the particle set split into K shards stepped concurrently on K streams
(kernel tails of one shard overlap with the other's kernels)."""
import sys, time; sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from stoch_gpmp_amd import workloads as W
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
P, S, T = 1024, 128, 64
sph = torch.as_tensor(W.panda_spheres()).to(**ta)
for K in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(K)]
    pls = []
    for r in range(K):
        with torch.cuda.stream(streams[r]):
            pls.append(W.hip_panda_planner(W.PANDA, T, P, S, ta, seed=0, rank=r, world_size=K))
    torch.cuda.synchronize()
    def it():
        for r in range(K):
            with torch.cuda.stream(streams[r]):
                pls[r].step(obstacle_spheres=sph)
    for _ in range(20): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): it()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"{K} shard(s) on {K} stream(s): {dt*1e6:.1f} us per iteration of all {P} particles = {1/dt:.0f} it/s")
