"""What StochGPMP.reset() costs (both priors re-factored / priors cached), and where the host time goes."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
pl, obs, _ = bench.build_planner(torch, "panda", 64, 32, 64, torch.float32, dev)
torch.cuda.synchronize()
for trial in range(3):
    pl._engine._prior_key.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter(); pl.reset(); torch.cuda.synchronize()
    print(f"reset with both priors refactored: {1e3 * (time.perf_counter() - t0):.3f} ms")
    torch.cuda.synchronize(); t0 = time.perf_counter(); pl.reset(); torch.cuda.synchronize()
    print(f"reset with cached priors: {1e3 * (time.perf_counter() - t0):.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    pl.reset()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
