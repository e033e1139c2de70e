"""Where do store-free iterations pay?  Storing against store-free over problem sizes on one box, interleaved: it/s in one
optimize(opt_iters=K) call.  usage: store_free_sizes.py [robot:S:T:goals:P,P,... ...]"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402

# (store_free_min_bytes comes from the environment: SGPMP_STORE_FREE_MIN_BYTES=1 regenerates at every size)
DEFAULT = ["panda:128:64:1:16,64,128,256,384,512,1024", "panda:64:64:1:64,256,512,1024,2048", "panda:128:32:1:256,512,1024,2048",
           "panda:256:64:1:64,128,256,512", "planar:64:64:4:16,64,256,1024,4096", "planar:128:64:4:64,256,1024,4096"]


def main():
    dev = torch.device("cuda:0")
    for spec in sys.argv[1:] or DEFAULT:
        robot, S, T, goals, Ps = spec.split(":")
        S, T, goals = int(S), int(T), int(goals)
        for P in [int(p) for p in Ps.split(",")]:
            pls = {}
            for mode in ("storing", "store_free"):
                pl, obs, _ = bench.build_planner(torch, robot, P, S, T, torch.float32, dev, goals=goals,
                                                 store_free=(mode == "store_free"))
                bench.time_loop(torch, pl, obs, 150, 0)
                pls[mode] = (pl, obs)
            steps = 200
            rates = {m: [] for m in pls}
            for _ in range(4):
                for m, (pl, obs) in pls.items():
                    rates[m].append(steps / bench.time_loop(torch, pl, obs, steps, 10))
            r = {m: round(max(v), 1) for m, v in rates.items()}
            eng = pls["store_free"][0]._engine
            r.update(ratio=round(r["store_free"] / r["storing"], 3), two_chains=eng.pipeline_split_steps() > 0,
                     store_free_steps=eng.store_free_steps(), sample_MB=round(P * S * T * pls["storing"][0].d_state_opt * 4 / 1e6, 1))
            print(f"{robot} S={S} T={T} P={P}", json.dumps(r), flush=True)
            del pls
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
