"""One-off / occasional: random (T, P, S, field, spheres, goals) against the two-launch path -- the fused launch (both
instantiations) must give the sampler's samples bit for bit and the generic sweep's costs to fp32 rounding.
python3 tools/fuzz_fused_shapes.py [count] [seed] on the GPU box."""
import sys, os, random
sys.path.insert(0, os.getcwd())
import torch
from tests import scenarios as SC
from tests.hip_builders import hip_panda_planner
F32 = {"device": torch.device("cuda:0"), "dtype": torch.float32}
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
n = 7
bad = 0
for i in range(count):
    T = 2 * rng.randint(1, 70)
    S = rng.randint(1, 150)
    nppg = rng.randint(1, 6)
    ft = rng.choice(["rbf", "sdf", "occupancy"])
    nsph = rng.choice([1, 5, 17, 64])
    G = rng.choice([1, 1, 2])
    goals = None if G == 1 else [SC.PANDA["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n]
    sph = torch.as_tensor(SC.panda_spheres(num=nsph, seed=3 + i)).to(**F32)
    a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, field_type=ft, seed=100 + i, goals=goals)
    b = hip_panda_planner(SC.PANDA, T, nppg, S, F32, field_type=ft, seed=100 + i, goals=goals)
    b._engine.set_option("no_fused_step", 1)
    ok = True
    for it in range(2):
        a.optimize(opt_iters=1, obstacle_spheres=sph)
        b.optimize(opt_iters=1, obstacle_spheres=sph)
        ka = a._engine.last_cost_kernel()
        same = bool(torch.equal(a.state_samples, b.state_samples))
        crel = float(((a._costs - b._costs).abs() / b._costs.abs().clamp_min(1e-30)).max())
        mrel = float((a.particle_means - b.particle_means).abs().max()) / max(float(b.particle_means.abs().max()), 1e-30)
        flip = not bool(torch.equal(a._costs.argmin(1), b._costs.argmin(1)))
        if not ka.startswith("fused_step") or not same or crel > 3e-5 or (mrel > 2e-6 and not flip) or not torch.isfinite(a._costs).all():
            ok = False
            print("MISMATCH", dict(T=T, S=S, nppg=nppg, G=G, ft=ft, nsph=nsph, it=it, kernel=ka, samples_equal=same, cost_rel=crel, means_rel=mrel, argmin_flip=flip))
        if flip:
            a.particle_means.copy_(b.particle_means)
    bad += 0 if ok else 1
print(f"{count} shapes, {bad} with a mismatch")
