"""Occasional: random planar problems with 64 samples per particle (the shapes whose store-free step carries its update inside the
launch) through optimize(opt_iters = K) calls of random lengths -- three planners per case on the same seed:
  a  the default: all store-free iterations of a call in ONE launch where the launch has that form (fused_planar_seg.inc: PERSIST)
  b  option no_persist_planar: one launch per store-free iteration (round 5)
  c  store_free = False: every iteration stores its samples and update_kernel updates (rounds 1-4)
Everything a caller can see must be bit-identical in all three after every call.  Temperatures from one-hot to soft weights.
python3 tools/fuzz_persist_shapes.py [count] [seed]   on the GPU box."""
import sys, os, random
sys.path.insert(0, os.getcwd())
import torch
from tests import scenarios as SC
from tests.hip_builders import hip_planar_planner
from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
F32 = {"device": torch.device("cuda:0"), "dtype": torch.float32}
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
om = synthetic_obstacle_map(seed=0, tensor_args=F32)
bad = multi = iters = 0
for i in range(count):
    T = rng.choice([16, 24, 32, 64, 64, 96, 128, 128, 256])
    nppg = rng.choice([1, 2, 7, 64, 100])
    G = rng.choice([1, 2, 4])
    n = rng.choice([2, 2, 2, 3])
    temperature = rng.choice([1., 1., 20., 200.])
    c = SC.PLANAR if n == 2 else dict(SC.PLANAR, n_dof=3, start=[-9., -9., 0.5, 0., 0., 0.])
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]][:G]
    if n == 3:
        goals = [g[:2] + [0.3 * (j + 1), 0., 0., 0.] for j, g in enumerate(goals)]
    if temperature > 1.:                                 # (a small workspace: many samples carry weight)
        c = dict(c, start=[-2.9, -2.9] + c["start"][2:], dt=0.5, cost_sigma_start=0.5, cost_sigma_gp=8., sigma_coll=0.4,
                 sigma_goal_prior=2., sigma_start_sample=2., sigma_goal_sample=2., sigma_gp_sample=6.)
        goals = [[0.3 * g[0], 0.3 * g[1]] + g[2:] for g in goals]
    mk = lambda **kw: hip_planar_planner(c, T, goals, nppg, 64, om, F32, seed=200 + i, temperature=temperature, **kw)   # noqa: E731
    a, b, cc = mk(), mk(), mk(store_free=False)
    b._engine.set_option("no_persist_planar", 1)
    ok = True
    for call in range(4):
        K = rng.choice([1, 2, 3, 4, 5, 9, 30, 120])
        outs = [p.optimize(opt_iters=K) for p in (a, b, cc)]
        iters += K
        for other, name in ((b, "one launch per iteration"), (cc, "storing")):
            j = 1 if other is b else 2
            same = all(torch.equal(x, y) for x, y in zip(outs[0], outs[j]))
            for nm in ("particle_means", "state_samples", "_weights_buf", "_grad", "_costs", "_means_prev"):
                same = same and torch.equal(getattr(a, nm), getattr(other, nm))
            same = same and bool((a._engine.row_counts() == other._engine.row_counts()).all())
            sa, so = a.global_stats(), other.global_stats()
            same = same and abs(sa[0] / so[0] - 1) < 1e-12 and abs(sa[1] / so[1] - 1) < 1e-12
            if not same or not torch.isfinite(a.particle_means).all():
                ok = False
                print("MISMATCH against", name, dict(T=T, nppg=nppg, G=G, n=n, temperature=temperature, call=call, K=K))
    multi += a._engine.multi_iteration_launches()
    bad += 0 if ok else 1
    del a, b, cc
print(f"fuzz_persist_shapes: {count} problems, {iters} iterations per planner, {multi} launches of several iterations, {bad} problems with a mismatch")
sys.exit(1 if bad else 0)
