"""How long do the context's two chain streams take to reach full speed in a fresh process?  (dev probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
pl, obs, _ = bench.build_planner(torch, "panda", 1024, 128, 64, torch.float32, dev)
for _ in range(200): pl.optimize(opt_iters=1, **obs)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "calls"
if mode == "one_long":
    t0 = time.perf_counter(); pl.optimize(opt_iters=150, **obs); torch.cuda.synchronize()
    print(f"one call of 150: {1e3*(time.perf_counter()-t0):.3f} ms")
elif mode == "tiny":
    for _ in range(3): pl.optimize(opt_iters=2, **obs)
    torch.cuda.synchronize()
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pl.optimize(opt_iters=20, **obs)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{mode}: pipelined call {i}: host {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms", flush=True)
