"""One launch for ALL store-free iterations of an optimize(opt_iters = K) call (fused_planar_seg.inc: PERSIST) against one launch
per iteration (option no_persist_planar), planar problems whose update runs inside the launch (S = 64; BASELINE configs[1]):
bit-identity of everything the call returns and leaves behind, and microseconds per iteration, alternating on one box.
usage: persist_ab.py [K ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
dev = torch.device("cuda", 0)
Ks = [int(x) for x in sys.argv[1:]] or [3, 10, 100, 500]
for shape in ((256, 64, 128), (64, 64, 64), (1024, 64, 128)):
    P, S, T = shape
    pls = []
    for off in (0, 1):
        pl, obs, _ = B.build_planner(torch, "planar", P, S, T, torch.float32, dev, store_free=True, goals=4)
        pl._engine.set_option("no_persist_planar", off)
        pls.append(pl)
    for K in Ks:
        outs = [pl.optimize(opt_iters=K, **obs) for pl in pls]
        same = all(torch.equal(x, y) for x, y in zip(*outs)) and torch.equal(pls[0].particle_means, pls[1].particle_means) and \
            torch.equal(pls[0].state_samples, pls[1].state_samples) and torch.equal(pls[0]._costs, pls[1]._costs)
        t = [[], []]
        for rep in range(max(5, 400 // K)):
            for i, pl in enumerate(pls):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pl.optimize(opt_iters=K, **obs)
                torch.cuda.synchronize()
                t[i].append(time.perf_counter() - t0)
            same = same and torch.equal(pls[0].particle_means, pls[1].particle_means) and torch.equal(pls[0].state_samples, pls[1].state_samples) \
                and torch.equal(pls[0]._costs, pls[1]._costs) and torch.equal(pls[0]._weights_buf, pls[1]._weights_buf)
        a, b = min(t[0]) / K * 1e6, min(t[1]) / K * 1e6
        print(f"planar {P} x {S} x {T}  K = {K:4d}   one launch {a:7.2f} us/iteration   launch per iteration {b:7.2f} us/iteration   "
              f"({b / a:.3f} x)   bit-identical: {same}   store-free steps {pls[0]._engine.store_free_steps()} / {pls[1]._engine.store_free_steps()}")
        assert same
    del pls
