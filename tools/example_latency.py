"""The reference's own example sizes, run the way its examples run them -- one optimize() per loop trip, opt_iters = 1
(examples/panda_environment.py:107,141-147; planar_environment.py:82,102-108): wall time per call, with and without a
synchronisation per call, and (under rocprofv3 --kernel-trace, see tools/trace_gaps.py) the kernels of one call.
usage: example_latency.py [panda|planar] [calls]"""
import os
import sys
import time

import torch

ROOT = __file__.rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def run(which, calls):
    if which == "panda":
        import panda_environment as ex
        planner, _ = ex.main(opt_iters=20, seed=0, verbose=False)
        import numpy as np
        sph = np.zeros((1, 5, 4))
        sph[0, :, :3] = [[0.8, 0., 0.8], [0.7, -0.1, 0.7], [0.9, 0.1, 0.9], [0.65, 0.15, 0.95], [0.95, -0.15, 0.65]]
        sph[0, :, 3] = 0.12
        obs = {"obstacle_spheres": torch.from_numpy(sph).to(**planner.tensor_args)}
    else:
        import planar_environment as ex
        planner, _ = ex.main(opt_iters=20, seed=0, verbose=False)
        obs = {}
    for sync in (False, True):
        for _ in range(50):
            planner.optimize(**obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            planner.optimize(**obs)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{which}: P={planner.num_particles} S={planner.num_samples} T={planner.traj_len}  "
              f"{'sync per call' if sync else 'free-running  '}: {el / calls * 1e6:8.1f} us per optimize(opt_iters=1), "
              f"kernel {planner._engine.last_cost_kernel()}, launches {planner._engine.last_step_launches() if hasattr(planner._engine, 'last_step_launches') else '?'}")
    # the host's share: the same loop without a GPU behind it cannot be had; time the Python + ctypes path by enqueueing only
    t0 = time.perf_counter()
    for _ in range(calls):
        planner.optimize(**obs)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{which}: enqueue-only loop {t_enq / calls * 1e6:8.1f} us per call (host side; the queue drains behind it)")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    for w in (("panda", "planar") if which == "both" else (which,)):
        run(w, calls)
