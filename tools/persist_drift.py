import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench as B
dev = torch.device("cuda", 0)
pl, obs, _ = B.build_planner(torch, "planar", 256, 64, 128, torch.float32, dev, store_free=True, goals=4)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
done = 0
for K in (150, 150, 10, 60, 150, 150, 500, 500, 10, 60, 150):
    torch.cuda.synchronize()
    e0.record(); pl.optimize(opt_iters=K, **obs); e1.record(); torch.cuda.synchronize()
    done += K
    rc = pl._engine.row_counts()
    print(f"K={K:4d}  after {done:5d} iterations: {e0.elapsed_time(e1) * 1e3 / K:6.2f} us/iteration   rows with weight: mean {rc.mean():.2f} max {int(rc.max())}   mean cost {float(pl._costs.mean()):.1f}")
