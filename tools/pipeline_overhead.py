"""Fixed cost per optimize(opt_iters=K) call of the two-chain mode: time calls of K = 2 .. 200 iterations, two chains
against single-iteration calls (config 3).   usage: pipeline_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pl, obs, _ = bench.build_planner(torch, "panda", 1024, 128, 64, torch.float32, dev)
    bench.time_loop(torch, pl, obs, 200, 20)
    if len(sys.argv) > 1:
        pl._engine.set_option("pipe_split", int(sys.argv[1]))
        print("first chain's share:", sys.argv[1], "/ 16", " ".join(sys.argv[2:]))
    for K in (2, 5, 10, 20, 50, 100, 200):
        row = []
        for one_call in (True, False):
            best = []
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if one_call:
                    pl.optimize(opt_iters=K, **obs)
                else:
                    for _ in range(K):
                        pl.optimize(opt_iters=1, **obs)
                torch.cuda.synchronize()
                best.append(time.perf_counter() - t0)
            best.sort()
            row.append(best[len(best) // 2] * 1e3)
        print(f"K={K:4d}  two chains {row[0]:8.3f} ms ({row[0] / K:.4f}/iter)   single calls {row[1]:8.3f} ms ({row[1] / K:.4f}/iter)",
              flush=True)


main()
