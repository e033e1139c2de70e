"""Host-side cost of one StochGPMP.optimize(opt_iters=1) call (config 1 is host-bound: its kernels are a few us).
usage: host_profile.py [panda]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
if len(sys.argv) > 1 and sys.argv[1] == "panda":          # the reference's example size (panda_environment.py:29-32)
    pl, obs, _ = bench.build_planner(torch, "panda", 5, 32, 64, torch.float32, dev)
else:
    pl, obs, _ = bench.build_planner(torch, "planar", 4, 16, 64, torch.float64, dev, goals=2)
for _ in range(200):
    pl.optimize(opt_iters=1, **obs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    pl.optimize(opt_iters=1, **obs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6 * (t1 - t0) / 2000:.2f} us per call, with the final sync {1e6 * (t2 - t0) / 2000:.2f} us")
t0 = time.perf_counter()
pl.optimize(opt_iters=2000, **obs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"one call of 2000 iterations: host {1e6 * (t1 - t0) / 2000:.2f} us per iteration, with sync {1e6 * (t2 - t0) / 2000:.2f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    pl.optimize(opt_iters=1, **obs)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
