"""Soak of the small-step launch (fused_step_small_kernel: barriers, LDS hand-overs, LDS-DMA tables) against the one-wave-per-item
launch: several shapes, thousands of iterations, every buffer compared bit for bit every few iterations.  usage: soak_small_step.py [seconds]"""
import os
import sys
import time

os.environ["SGPMP_NO_SMALL_STEP"] = "1"
ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + "/tests")
import torch  # noqa: E402
from test_gpu_planner import hip_panda_planner, SC, F32  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.
shapes = [(64, 5, 32, {}), (64, 64, 64, {}), (50, 7, 27, {}), (176, 3, 40, {}), (32, 6, 16, dict(field_type="sdf")), (128, 16, 64, {}),
          (48, 4, 24, dict(field_type="occupancy"))]
t_end = time.time() + budget
rounds = bad = 0
while time.time() < t_end:
    for T, nppg, S, kw in shapes:
        seed = 1000 + rounds
        sph = torch.as_tensor(SC.panda_spheres(num=5, seed=seed % 17)).to(**F32)
        a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=seed, **kw)
        b = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=seed, **kw)
        a._engine.set_option("no_small_step", 0)
        for k in (1, 7, 1, 20, 3):
            a.optimize(opt_iters=k, obstacle_spheres=sph)
            b.optimize(opt_iters=k, obstacle_spheres=sph)
            ok = (torch.equal(a._costs, b._costs) and torch.equal(a.particle_means, b.particle_means)
                  and torch.equal(a.state_samples, b.state_samples) and torch.equal(a._weights_buf, b._weights_buf))
            if not ok:
                bad += 1
                print("MISMATCH", (T, nppg, S, kw), "seed", seed, "after", k, flush=True)
        assert a._engine.last_cost_kernel() == "fused_step_small_kernel", a._engine.last_cost_kernel()
    rounds += 1
print(f"{rounds} rounds x {len(shapes)} shapes x 32 iterations, {bad} mismatches")
