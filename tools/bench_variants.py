#!/usr/bin/env python3
"""The measurements that used to ride in bench.py's line (rounds 2-5) and do not belong in the one line the driver parses:

    python tools/bench_variants.py [--variants] [--anchors] [--examples] [--cpu] [--membw] [--out gpurun_out/r06/bench_variants.json]

  --variants   config 3 with the sdf field, with 64 spheres (SURVEY 8d stress variant), config 3's shape in fp64 -- each with
               bench.py's own legs (two timed passes, single-iteration calls, event pass, store-free beside it)
  --anchors    the UNSHARDED multi-GPU configurations on one GPU (config 4: 8192 x 128 x 64, 3.8 GB of samples; config 5:
               4096 x 256 x 128, 7.5 GB) -- the N = 1 anchors of a strong-scaling reading
  --examples   the reference's own example problems at their own sizes, one optimize(opt_iters=1) per loop trip as its scripts
               run them (examples/panda_environment.py:29-32,141-147; planar_environment.py:14-20,102-108): us per call
  --cpu        the CPU legs beyond bench.py's bounded cpu_baseline: thread-count sweep and the all-host-cores point of the
               reference-equivalent dense oracle, and the banded fp64 restatement ("fair CPU")
  --membw      this box's streaming figures (tools/membw)
No flag = all of them.  Writes one JSON file (copy it to profiles/rNN/)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def variants(torch, dev):
    specs = [
        ("config 3 with the sdf sphere field", "cfg3_sdf", dict(workload="panda", P_local=1024, S=128, T=64, dtype="f32", field="sdf"), 100),
        ("config 3 with 64 sphere obstacles", "cfg3_64sph", dict(workload="panda", P_local=1024, S=128, T=64, dtype="f32", spheres=64), 40),
        ("config 3's shape in fp64 (Panda 1024 x 128 x 64)", "cfg3_f64", dict(workload="panda", P_local=1024, S=128, T=64, dtype="f64"), 20),
        ("config 3's shape in fp64, link fields in fp32 (option f64_fields_f32)", "cfg3_f64_mixed",
         dict(workload="panda", P_local=1024, S=128, T=64, dtype="f64", options={"f64_fields_f32": 1}), 20),
        ("config 3's shape in fp64 as rounds 1-5 ran it: sampler, sweep, update as three launches (option no_fused_step)", "cfg3_f64_unfused",
         dict(workload="panda", P_local=1024, S=128, T=64, dtype="f64", options={"no_fused_step": 1}), 20),
        ("config 2's shape in fp64 (planar 256 x 64 x 128)", "cfg2_f64", dict(workload="planar", P_local=256, S=64, T=128, dtype="f64", goals=4), 100),
    ]
    return [B.config_row(torch, dev, label, key, spec, steps, passes=2, single_calls=True) for label, key, spec, steps in specs]


def anchors(torch, dev):
    out = []
    for label, key, spec, steps in (
            ("config 4 unsharded: Panda 8192 x 128 x 64 fp32 on one GPU", "cfg4_whole",
             dict(workload="panda", P_local=8192, S=128, T=64, dtype="f32"), 20),
            ("config 5 unsharded: Panda 4 goals, 4096 x 256 x 128 fp32 on one GPU", "cfg5_whole",
             dict(workload="panda", P_local=4096, S=256, T=128, dtype="f32", goals=4), 10)):
        try:
            out.append(B.config_row(torch, dev, label, key, spec, steps, passes=2))
        except Exception as e:                              # (reported, the other rows still count)
            out.append({"config": label, "error": f"{type(e).__name__}: {e}"})
        torch.cuda.empty_cache()
    return out


def reference_examples(torch):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    out = {}
    for which in ("panda", "planar"):
        try:
            ex = __import__(which + "_environment")
            pl, _ = ex.main(opt_iters=20, seed=0, verbose=False)
            obs = {}
            if which == "panda":
                import numpy as np
                sph = np.zeros((1, 5, 4))
                sph[0, :, :3] = [[0.8, 0., 0.8], [0.7, -0.1, 0.7], [0.9, 0.1, 0.9], [0.65, 0.15, 0.95], [0.95, -0.15, 0.65]]
                sph[0, :, 3] = 0.12
                obs = {"obstacle_spheres": torch.from_numpy(sph).to(**pl.tensor_args)}
            calls = 2000
            for _ in range(200):
                pl.optimize(**obs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                pl.optimize(**obs)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            out[which] = {"shape": f"{pl.num_particles} x {pl.num_samples} x {pl.traj_len}", "dtype": str(pl.tensor_args["dtype"]).split(".")[-1],
                          "us_per_call": 1e6 * t_all / calls, "host_enqueue_us_per_call": 1e6 * t_host / calls,
                          "launches_per_call": pl._engine.last_step_launches(), "cost_kernel": pl._engine.last_cost_kernel()}
            # the same problem inside one optimize(opt_iters=K) call
            pl.optimize(opt_iters=200, **obs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pl.optimize(opt_iters=calls, **obs)
            torch.cuda.synchronize()
            out[which]["us_per_iteration_inside_one_call"] = 1e6 * (time.perf_counter() - t0) / calls
            del pl
        except Exception as e:
            out[which] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return out


def cpu_legs(torch):
    """Thread sweep + all-cores point of the dense oracle at P = 4 (config 3's S, T), P = 4, 8, 16 at the best count, and the
    banded fp64 restatement at P = 32."""
    from tests import scenarios as SC
    from stoch_gpmp_amd import workloads as W
    from oracle import banded_equiv as BE
    cores = B.host_cores()
    S, T, P_full, n = 128, 64, 1024, 7
    obs = {"obstacle_spheres": torch.as_tensor(W.panda_spheres(num=5)).to(torch.float32)}

    def timed(o, iters, warm=True):
        if warm:
            o.step(**obs)
        t0 = time.perf_counter()
        for _ in range(iters):
            o.step(**obs)
        return (time.perf_counter() - t0) / iters
    ora = SC.oracle_panda_planner(W.PANDA, T, 4, S, dtype=torch.float32, field_type="rbf", seed=0)
    by_threads = {}
    for threads in sorted({1, min(cores, 8), min(cores, 16), min(cores, 32), min(cores, 64)}):
        torch.set_num_threads(threads)
        by_threads[threads] = 1.0 / timed(ora, 2)
    best = max(by_threads, key=by_threads.get)
    torch.set_num_threads(cores)
    dta = timed(ora, 1, warm=False)
    all_cores = {"threads": cores, "it_per_s_at_P4": 1.0 / dta, "value_extrapolated_per_particle": (1.0 / dta) * 4 / P_full}
    torch.set_num_threads(best)
    points = []
    for P in (4, 8, 16):
        o = SC.oracle_panda_planner(W.PANDA, T, P, S, dtype=torch.float32, field_type="rbf", seed=0)
        points.append({"particles": P, "s_per_it": timed(o, 2)})
        del o
    (p1, t1), (p2, t2) = [(q["particles"], q["s_per_it"]) for q in points[-2:]]
    t_full = t2 + (t2 - t1) / (p2 - p1) * (P_full - p2)
    dense = {"value": 1.0 / t_full, "unit": "iterations/s", "cores": best, "host_cores": cores, "kind": "port",
             "it_per_s_at_P4_by_torch_threads": by_threads, "all_host_cores": all_cores, "measured_points": points,
             "how": f"affine extrapolation through P = {p1}, {p2} to P = {P_full}"}
    # banded restatement
    c = W.PANDA
    Pc, dtype = 32, torch.float64
    goal = torch.tensor([c["goal_q"] + [0.] * n], dtype=dtype)
    start = torch.tensor(c["start_q"] + [0.] * n, dtype=dtype)
    means = torch.stack([start + (goal[0] - start) * t / (T - 1) for t in range(T)]).repeat(Pc, 1, 1)
    band = BE.BandedPlanner(Pc, S, T, c["dt"], n, start, goal, BE.panda_chunk_cost(c, T, S, goal, "rbf"),
                            c["step_size"], c["temperature"], c["sigma_start_sample"], c["sigma_goal_sample"],
                            c["sigma_gp_sample"], means, chunk=8)
    sph = torch.as_tensor(W.panda_spheres(num=5)).to(dtype)
    g = torch.Generator().manual_seed(0)
    fair_by = {}
    for threads in sorted({min(cores, 16), min(cores, 64)}):
        torch.set_num_threads(threads)
        eps = torch.randn(S, Pc, T * 2 * n, generator=g, dtype=dtype)
        band.step(eps, obstacle_spheres=sph)
        t0 = time.perf_counter()
        for _ in range(2):
            eps = torch.randn(S, Pc, T * 2 * n, generator=g, dtype=dtype)   # noise generation is part of an iteration
            band.step(eps, obstacle_spheres=sph)
        fair_by[threads] = (time.perf_counter() - t0) / 2
    bt = min(fair_by, key=fair_by.get)
    fair = {"value": (1.0 / fair_by[bt]) * Pc / P_full, "unit": "iterations/s", "cores": bt, "kind": "port (banded fp64 restatement)",
            "sample_particles": Pc, "s_per_it_by_threads": fair_by, "how": "linear in the particle count"}
    return {"cpu_dense_reference_equivalent": dense, "cpu_fair_banded": fair}


def membw():
    exe = os.path.join(ROOT, "tools", "membw")
    if not os.path.exists(exe):
        return None
    return subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout


def main():
    ap = argparse.ArgumentParser()
    for f in ("variants", "anchors", "examples", "cpu", "membw"):
        ap.add_argument("--" + f, action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_variants.json"))
    a = ap.parse_args()
    every = not (a.variants or a.anchors or a.examples or a.cpu or a.membw)
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {}
    if a.variants or every:
        out["variants"] = variants(torch, dev)
    if a.anchors or every:
        out["single_gpu_anchors"] = anchors(torch, dev)
    if a.examples or every:
        out["reference_examples"] = reference_examples(torch)
    if a.membw or every:
        out["membw"] = membw()
    if a.cpu or every:
        out["cpu"] = cpu_legs(torch)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(B._clean(out), open(a.out, "w"), indent=1)
    print(json.dumps(B._clean(out, 5))[:6000])


if __name__ == "__main__":
    main()
