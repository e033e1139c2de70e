"""Bisecting tool for the several-iterations-in-one-launch planar step (fused_planar_seg.inc: PERSIST): calls of 3, 4 and 6
iterations against option no_persist_planar over several seeds, naming the first tensor that differs and where.
SGPMP_LIB_PATH picks a flavour of the library (make EXTRA=-DSGPMP_PERSIST_KA=7 OUT=ab_libs/ka_7 ...)."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench as B
dev = torch.device("cuda", 0)
for shape in ((21, 64, 64), (256, 64, 128)):
    P, S, T = shape
    bad = 0
    for trial in range(int(os.environ.get('TRIALS', 6))):
        pls = []
        for off in (0, 1):
            pl, obs, _ = B.build_planner(torch, "planar", P, S, T, torch.float32, dev, store_free=True, goals=1 if P == 21 else 4, seed=trial)
            pl._engine.set_option("no_persist_planar", off)
            pls.append(pl)
        for K in (3, 4, 6):
            for pl in pls: pl.optimize(opt_iters=K, **obs)
            for nm in ("particle_means", "_costs", "_weights_buf", "_grad", "_means_prev", "state_samples"):
                x, y = getattr(pls[0], nm), getattr(pls[1], nm)
                if not torch.equal(x, y):
                    d = (x != y).nonzero()
                    bad += 1
                    print(os.environ.get("SGPMP_LIB_PATH", "default")[-40:], shape, "trial", trial, "K", K, nm, "differs at", d.shape[0], "of", x.numel(), "first", d[0].tolist(), "max abs", float((x - y).abs().max()))
                    break
            rc = (pls[0]._engine.row_counts() != pls[1]._engine.row_counts()).sum()
            if rc: print("row counts differ", int(rc))
    print(os.environ.get("SGPMP_LIB_PATH", "default")[-40:], shape, "mismatching calls:", bad)
