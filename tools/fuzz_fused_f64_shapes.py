"""One-off / occasional: random shapes of the fp64 one-launch step (fused_step_f64_kernel; Panda chain with any field / sphere count /
goals, with and without the fp32-link-fields option; planar n = 2, 3) against the two-launch path (sample_iso_kernel<double> +
cost_sweep_kernel<double>): samples within 1e-12 of the largest sample, costs 1e-11 (1e-6 with the option), same arg-mins.
python3 tools/fuzz_fused_f64_shapes.py [count] [seed]   on the GPU box."""
import sys, os, random
sys.path.insert(0, os.getcwd())
import torch
from tests import scenarios as SC
from tests.hip_builders import hip_panda_planner, hip_planar_planner
from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
F64 = {"device": torch.device("cuda:0"), "dtype": torch.float64}
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
om = synthetic_obstacle_map(seed=0, tensor_args=F64)
bad = 0
for i in range(count):
    T = rng.choice([2, 3, 5, 16, 33, 63, 64, 65, 100, 127, 128, 129, 190, 257])
    S = rng.choice([1, 2, 7, 16, 40, 128])
    nppg = rng.choice([1, 2, 5, 24])
    kind = rng.choice(["panda", "panda", "panda_mixed", "planar2", "planar3"])
    obs = {}
    if kind.startswith("panda"):
        ft = rng.choice(["rbf", "sdf", "occupancy"])
        nsph = rng.choice([1, 5, 17, 64, 130])
        G = rng.choice([1, 1, 2])
        goals = None if G == 1 else [SC.PANDA["goal_q"] + [0.] * 7, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * 7]
        mk = lambda: hip_panda_planner(SC.PANDA, T, nppg, S, F64, field_type=ft, seed=100 + i, goals=goals)   # noqa: E731
        obs = {"obstacle_spheres": torch.as_tensor(SC.panda_spheres(num=nsph, seed=3 + i)).to(**F64)}
        tag = dict(ft=ft, nsph=nsph, G=G)
    else:
        n = 2 if kind == "planar2" else 3
        c = SC.PLANAR if n == 2 else dict(SC.PLANAR, n_dof=3, start=[-9., -9., 0.5, 0., 0., 0.])
        goals = [[9., 6., 0., 0.], [9., -3., 0., 0.]] if n == 2 else [[9., 6., 1., 0., 0., 0.], [9., -3., -1., 0., 0., 0.]]
        mk = lambda: hip_planar_planner(c, T, goals, nppg, S, om, F64, seed=100 + i)   # noqa: E731
        tag = dict(n=n)
    a, b = mk(), mk()
    if kind == "panda_mixed":
        a._engine.set_option("f64_fields_f32", 1)
    b._engine.set_option("no_fused_step", 1)
    # (with the option the occupancy COUNT is taken on fp32 link positions: a point within fp32 rounding of a sphere's surface may
    # count differently -- one quantum 1 / sigma_coll^2 of a cost, ~1e-5 of a total here; the smooth fields stay within 2e-6)
    ctol = (1e-4 if tag.get("ft") == "occupancy" else 2e-6) if kind == "panda_mixed" else 1e-11
    ok = True
    for it in range(2):
        a.optimize(opt_iters=1, **obs)
        b.optimize(opt_iters=1, **obs)
        ka = a._engine.last_cost_kernel()
        scale = max(float(b.state_samples.abs().max()), 1e-30)
        srel = float((a.state_samples - b.state_samples).abs().max()) / scale
        crel = float(((a._costs - b._costs).abs() / b._costs.abs().clamp_min(1e-30)).max())
        flip = not bool(torch.equal(a._costs.argmin(1), b._costs.argmin(1)))
        if not ka.startswith("fused_step_f64") or srel > 1e-12 or crel > ctol or flip or not torch.isfinite(a._costs).all():
            ok = False
            print("MISMATCH", dict(kind=kind, T=T, S=S, nppg=nppg, it=it, kernel=ka, samples_rel=srel, cost_rel=crel, argmin_flip=flip, **tag))
        b.particle_means.copy_(a.particle_means)
    bad += 0 if ok else 1
print(f"fuzz_fused_f64_shapes: {count} shapes, {bad} with a mismatch")
sys.exit(1 if bad else 0)
