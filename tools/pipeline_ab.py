"""optimize(opt_iters=K) as two particle-half chains (sgpmp_pipeline_begin/_end) against the same K iterations as
single-iteration calls, alternated on one box.   usage: pipeline_ab.py [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

f32 = torch.float32
SPECS = [
    ("config 3 rbf", dict(workload="panda", P_local=1024, S=128, T=64, dtype=f32)),
    ("config 3 sdf", dict(workload="panda", P_local=1024, S=128, T=64, dtype=f32, field="sdf")),
    ("config 5 share", dict(workload="panda", P_local=512, S=256, T=128, dtype=f32, goals=4, shard_of=(3, 8))),
    ("config 2", dict(workload="planar", P_local=256, S=64, T=128, dtype=f32, goals=4)),
]


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    for label, spec in SPECS:
        pl, obs, _ = bench.build_planner(torch, dev=dev, **spec)
        bench.time_loop(torch, pl, obs, 100, 20)
        res = {True: [], False: []}
        for rep in range(3):
            for one_call in (True, False):
                res[one_call].append(bench.time_loop(torch, pl, obs, K, 10, one_call=one_call) / K * 1e3)
        print(label, "two chains:", " ".join(f"{x:.4f}" for x in res[True]), "| single-iteration calls:",
              " ".join(f"{x:.4f}" for x in res[False]), "ms/iter", flush=True)
        del pl
        torch.cuda.empty_cache()


main()
