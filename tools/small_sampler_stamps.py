"""Cycles of sample_iso_small_kernel's phases (diagnostic build -DSGPMP_SMALL_STAMPS via SGPMP_LIB_PATH): workgroup (0, 0), wave 0."""
import os
import sys

import torch

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import planar_environment as ex  # noqa: E402

planner, _ = ex.main(opt_iters=20, seed=0, verbose=False)
tot = torch.zeros(6, dtype=torch.float64)
for _ in range(20):
    planner.optimize()
    torch.cuda.synchronize()
    tot += planner.state_samples.reshape(-1)[:6].double().cpu()
names = ["coefficients + noise", "barrier", "recurrence", "barrier", "flush (x = mu + y, stores)", "barrier"]
for n, v in zip(names, tot / 20):
    print(f"{n:32s} {float(v):9.0f} cycles")
print(f"{'sum':32s} {float(tot.sum() / 20):9.0f} cycles")
