"""Would two particle sub-ranges, each running its own sampler+sweep -> update chain on its own HIP stream, keep
the chip busier than one chain (whose update kernel and launch gaps leave it idle ~8 % of an iteration)?
Emulated with two independent planners of P0 and P1 particles (config 3 otherwise) on two torch streams,
against one planner of P0 + P1.   usage: two_chain_probe.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def build(P, dev):
    pl, obs, _ = bench.build_planner(torch, "panda", P, 128, 64, torch.float32, dev)
    return pl, obs


def run_one(pl, obs, iters):
    for _ in range(20):
        pl.optimize(opt_iters=1, **obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        pl.optimize(opt_iters=1, **obs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def run_two(a, b, iters):
    (pa, oa), (pb, ob) = a, b
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()

    def loop(n):
        for _ in range(n):
            with torch.cuda.stream(sa):
                pa.optimize(opt_iters=1, **oa)
            with torch.cuda.stream(sb):
                pb.optimize(opt_iters=1, **ob)
    loop(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    full = build(1024, dev)
    for rep in range(2):
        print(f"one chain, 1024 particles: {run_one(*full, iters) * 1e3:.4f} ms/iter", flush=True)
        for p0 in (512, 384, 256, 128):
            a, b = build(p0, dev), build(1024 - p0, dev)
            # the planners were built on the default stream: their buffers are ready once it is idle
            torch.cuda.synchronize()
            print(f"two chains, {p0} + {1024 - p0} particles: {run_two(a, b, iters) * 1e3:.4f} ms/iter", flush=True)
            del a, b


main()
