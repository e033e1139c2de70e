"""Stand-alone cost sweep (sgpmp_cost_eval) at config 3 / config 5 share: chunked kernel vs the 64-lane-pass
two-trajectory kernels, interleaved rounds in one process (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W

dev = torch.device("cuda:0")
ta = {"device": dev, "dtype": torch.float32}


def bench(P, S, T, field="rbf"):
    sph = torch.as_tensor(W.panda_spheres()).to(**ta)
    pl = W.hip_panda_planner(W.PANDA, T, P, S, ta, field_type=field, seed=1)
    for _ in range(50):
        pl.optimize(obstacle_spheres=sph)
    eng = pl._engine
    w = eng.is_weights(pl.particle_means, pl.temperature)
    sphc = sph.reshape(-1, 4).contiguous()
    out = torch.empty(P * S, **ta)

    def run(n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            eng.cost_eval(pl.state_samples, spheres=sphc, is_weights=w, rows_per_particle=S, out=out)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    res = {}
    for rnd in range(4):
        for name, opt in (("chunked", 0), ("dual", 1)):
            eng.set_option("no_chunked_sweep", opt)
            run(10)
            res.setdefault(name, []).append(run())
            res[name + "_kernel"] = eng.last_cost_kernel()
    for name in ("chunked", "dual"):
        v = res[name]
        print(f"P={P} S={S} T={T} {field}: {res[name + '_kernel']:34s} min {min(v):7.1f} us  median {sorted(v)[len(v)//2]:7.1f} us")


bench(1024, 128, 64)
bench(1024, 128, 64, "sdf")
bench(512, 256, 128)
