"""Host cost of ONE sgpmp_step through the pre-bound ctypes call (no planner code around it), and of the planner's optimize(opt_iters=1)
around it: where do the microseconds per call of a small problem go?  usage: host_step_cost.py [panda|planar]"""
import sys
import time

import torch

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from stoch_gpmp_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
if len(sys.argv) > 1 and sys.argv[1] == "planar":
    pl, obs, _ = bench.build_planner(torch, "planar", 15, 128, 64, torch.float64, dev, goals=3)
else:
    pl, obs, _ = bench.build_planner(torch, "panda", 5, 32, 64, torch.float32, dev)
for _ in range(100):
    pl.optimize(opt_iters=1, **obs)
torch.cuda.synchronize()
call = next(iter(pl._step_calls.values()))
N = 3000
t0 = time.perf_counter()
for i in range(N):
    call(pl._draw + i, L.STEP_MEANS_KEPT, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
pl._draw += N
print(f"sgpmp_step through ctypes: {1e6 * (t1 - t0) / N:.2f} us per call (host), {1e6 * (t2 - t0) / N:.2f} us with the queue drained; "
      f"{pl._engine.last_step_launches()} launches per step, kernel {pl._engine.last_cost_kernel()}")
t0 = time.perf_counter()
for i in range(N):
    pl.optimize(opt_iters=1, **obs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"optimize(opt_iters=1):     {1e6 * (t1 - t0) / N:.2f} us per call (host), {1e6 * (t2 - t0) / N:.2f} us with the queue drained")
t0 = time.perf_counter()
pl.optimize(opt_iters=N, **obs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"optimize(opt_iters={N}):  {1e6 * (t1 - t0) / N:.2f} us per iteration (host), {1e6 * (t2 - t0) / N:.2f} us with the queue drained")
pl._engine.close() if hasattr(pl._engine, "close") else None     # (a -DSGPMP_HOST_TIMING build prints its segments when the context goes)
del pl, call
import gc
gc.collect()
