"""Soak of sgpmp_optimize (optimize() as ONE call into the library: the K-loop, draw counters, statistics slots, MEANS_KEPT /
NO_SAMPLES per step, the two-chain bracket on the C side) against the per-step calls from Python (c_loop=False) for `seconds`:
random call lengths 1 .. 9, now and then an edit of the means or other obstacles between calls, two shapes (one that splits into
two chains, one that does not) -- every buffer compared bit for bit after every call.
usage: soak_c_loop.py [seconds]"""
import os, random, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import scenarios as SC
from tests.hip_builders import hip_panda_planner
F32 = {"device": torch.device("cuda:0"), "dtype": torch.float32}
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.
rng = random.Random(7)
calls = iters = 0
t0 = time.time()
for nppg, S, T in ((129, 128, 32), (6, 40, 48)):
    a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=5)
    b = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=5, c_loop=False)
    obs = [{"obstacle_spheres": torch.as_tensor(SC.panda_spheres(num=k, seed=k)).to(**F32)} for k in (5, 9)]
    t1 = time.time()
    while time.time() - t1 < seconds / 2:
        k = rng.randint(1, 9)
        o = obs[rng.random() < 0.2]
        if rng.random() < 0.15:
            for pl in (a, b):
                pl.particle_means.mul_(0.9995)
        ra, rb = a.optimize(opt_iters=k, **o), b.optimize(opt_iters=k, **o)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (calls, k)
        for x, y in ((a.particle_means, b.particle_means), (a.state_samples, b.state_samples), (a._weights_buf, b._weights_buf),
                     (a._grad, b._grad), (a._means_prev, b._means_prev)):
            assert torch.equal(x, y), (calls, k)
        sa, sb = a.global_stats(), b.global_stats()
        assert sa == sb, (sa, sb)
        assert a._draw == b._draw and a._stats_slot == b._stats_slot
        calls += 1
        iters += k
    print(f"{nppg} x {S} x {T}: split steps {a._engine.pipeline_split_steps()} / {b._engine.pipeline_split_steps()}, "
          f"store-free steps {a._engine.store_free_steps()} / {b._engine.store_free_steps()}")
print(f"soak_c_loop: {calls} calls, {iters} iterations in {time.time() - t0:.0f} s, 0 mismatches")
