"""One optimize(opt_iters=K) call under `rocprofv3 --kernel-trace`: warm-up, a marker gap, the call.  usage: trace_call.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
pl, obs, _ = bench.build_planner(torch, "panda", 1024, 128, 64, torch.float32, dev)
for _ in range(3):
    pl.optimize(opt_iters=100, **obs)
torch.cuda.synchronize()
time.sleep(0.01)
pl.optimize(opt_iters=K, **obs)
torch.cuda.synchronize()
