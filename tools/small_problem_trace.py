"""One small problem in a loop, for `rocprofv3 --kernel-trace` (tools/trace_gaps.py reads the trace): where do the microseconds of
an iteration of a latency-bound step go?   usage: small_problem_trace.py [robot P S T K]   (SGPMP_* switches from the environment)"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
P, S, T, K = [int(a) for a in sys.argv[2:6]] if len(sys.argv) > 5 else (64, 128, 64, 300)
dev = torch.device("cuda:0")
pl, obs, name = bench.build_planner(torch, robot, P, S, T, torch.float32, dev, goals=1 if robot == "panda" else 4)
pl.optimize(opt_iters=50, **obs)
torch.cuda.synchronize()
el = bench.time_loop(torch, pl, obs, K, 0)
print(name, "it/s", K / el, "us/it", el / K * 1e6, pl._engine.last_cost_kernel())
