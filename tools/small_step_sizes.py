"""The launch of small steps (fused_step_small_kernel: one workgroup per item) against the one-wave-per-item launch over problem
sizes, one box, interleaved: it/s inside optimize(opt_iters=K) and the launch's own time (HIP events around single steps).
usage: small_step_sizes.py [S:T:P,P,... ...]"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402

DEFAULT = ["32:64:5,16,64,128,256", "128:64:4,16,32,64,128,256", "64:128:16,64,128"]


def main():
    dev = torch.device("cuda:0")
    for spec in sys.argv[1:] or DEFAULT:
        S, T, Ps = spec.split(":")
        S, T = int(S), int(T)
        for P in [int(p) for p in Ps.split(",")]:
            pls = {}
            for mode in ("one_wave_per_item", "small"):
                pl, obs, _ = bench.build_planner(torch, "panda", P, S, T, torch.float32, dev)
                pl._engine.set_option("no_small_step", 0 if mode == "small" else 1)
                pl._engine.set_option("small_step_items", 1 << 20)
                bench.time_loop(torch, pl, obs, 100, 0)
                pls[mode] = (pl, obs)
            steps = 300
            rates = {m: [] for m in pls}
            for _ in range(3):
                for m, (pl, obs) in pls.items():
                    rates[m].append(steps / bench.time_loop(torch, pl, obs, steps, 10))
            r = {m: round(max(v), 1) for m, v in rates.items()}
            r.update(items=P * ((S + 7) // 8), ratio=round(r["small"] / r["one_wave_per_item"], 3),
                     kernels=[pls[m][0]._engine.last_cost_kernel() for m in pls],
                     same=bool(torch.equal(pls["small"][0].particle_means, pls["one_wave_per_item"][0].particle_means)))
            print(f"panda S={S} T={T} P={P}", json.dumps(r), flush=True)
            del pls
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
