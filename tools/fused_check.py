"""Fused K2+K3 step against the separate sampler + sweep on the same draws (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W

dev = torch.device("cuda:0")
ta = {"device": dev, "dtype": torch.float32}


def run(P, S, T, field, goals=None, iters=3, nsph=5):
    sph = torch.as_tensor(W.panda_spheres(num=nsph)).to(**ta)
    a = W.hip_panda_planner(W.PANDA, T, P, S, ta, field_type=field, seed=5, goals=goals)
    b = W.hip_panda_planner(W.PANDA, T, P, S, ta, field_type=field, seed=5, goals=goals)
    b._engine.set_option("no_fused_step", 1)
    for it in range(iters):
        a.optimize(obstacle_spheres=sph)
        b.optimize(obstacle_spheres=sph)
        ka, kb = a._engine.last_cost_kernel(), b._engine.last_cost_kernel()
        same = torch.equal(a.state_samples, b.state_samples)
        rel = float(((a._costs.double() - b._costs.double()).abs() / b._costs.double().abs()).max())
        am = float((a._costs.argmin(1) == b._costs.argmin(1)).double().mean())
        dm = float((a.particle_means - b.particle_means).abs().max())
        print(f"P={P} S={S} T={T} {field} it={it}: {ka} vs {kb}: samples equal {same}, costs rel {rel:.2e}, same argmin {am:.4f}, means diff {dm:.2e}")
        b.particle_means.copy_(a.particle_means)
    return a, b, sph


def timeit(pl, sph, n=100):
    for _ in range(10):
        pl.optimize(obstacle_spheres=sph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pl.optimize(obstacle_spheres=sph)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if __name__ == "__main__":
    run(3, 8, 16, "rbf")
    g2 = [W.PANDA["goal_q"] + [0.] * 7, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * 7]
    run(4, 16, 128, "sdf", goals=g2, nsph=9)
    run(4, 16, 32, "occupancy", nsph=3)
    a, b, sph = run(1024, 128, 64, "rbf", iters=2)
    timeit(a, sph, 300)                                     # bring the device to its working clock
    res = {}
    for rnd in range(3):                                    # interleaved rounds in one process
        for blocks in (0, 1024, 2048, 4096):
            a._engine.set_option("k3_blocks", blocks)
            res.setdefault(f"fused blocks={blocks}", []).append(timeit(a, sph))
        res.setdefault("unfused", []).append(timeit(b, sph))
    for k, v in res.items():
        print(f"{k}: min {min(v):.4f} median {sorted(v)[1]:.4f} ms/iter")
