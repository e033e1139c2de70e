#!/usr/bin/env python3
"""Benchmark of the StochGPMP inner loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is ONE planner iteration = one body of the loop at reference planner.py:289-299 (draw S
samples per particle, evaluate the composite cost, softmax-reweight, update the particle means) on
synthetic data.  Workload at N = 1: BASELINE.json configs[2] -- Panda 7-DoF (14-D state), 1024
particles x 128 samples x 64 waypoints, 5 synthetic sphere obstacles (rbf field) + self-collision,
fp32 compute with the prior factored in fp64.  At N > 1 every rank holds 1024 particles of a
(1024 N)-particle problem (configs[3] at N = 8: weak scaling, particles sharded, no data-path
collective; a [64,4]-double statistics all-reduce over RCCL per iteration, enqueued by sgpmp_step).

With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`) BEFORE anything touches the GPU
and relays rank 0's JSON line and the exit code; under a launcher (WORLD_SIZE set) it is a rank.

Prints one JSON line (rank 0).  `value` / `roofline` are the STORING mode -- every iteration writes its samples,
the mode SURVEY.md 8(d)'s algorithmic bytes are defined on: `roofline.frac` prices the dominant kernel against
the HBM peak using those bytes over the TIMED pass's ms_per_step (a lower bound: the step also holds the update
kernel), `roofline.bound` says what really bounds the launch (the vector ALU: `valu_frac` = its instruction-issue
floor / the launch's duration; `moved_frac` = bytes the counters saw / time / HBM peak); the launch's own
average duration measured with HIP events on the launch stream is in `roofline_detail`.  `store_free` reports the
product's default inside optimize(opt_iters=K) since round 5 -- iterations 1 .. K - 1 do not write their samples,
bit-identical results -- beside it, never instead of it; `parity` is a free-running K = 10 comparison with the fp64 CPU oracle made by this run; `cpu_baseline` times the reference-equivalent PyTorch-CPU
oracle on a bounded sample of the same workload on this box's host cores, `cpu_fair` the banded fp64
restatement (what a careful CPU implementation of the same mathematics costs).  `other_configs` are BASELINE.json's other
single-GPU configurations; `reference_examples` the reference's own two example problems at their own sizes, one
optimize(opt_iters=1) per call as its scripts run them (microseconds per call).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0       # same guide: 6.29 TB/s measured float4 copy (79 % of spec)
PROFILE_ROUNDS = ("r05", "r04", "r03")   # committed rocprofv3 --pmc sets (traffic / VALU figures quoted beside the live timings): newest first
GOALS4_PLANAR = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="panda", choices=["panda", "planar"])
    ap.add_argument("--particles", type=int, default=None, help="particles per GPU")
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--traj-len", type=int, default=None)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--field", default="rbf", choices=["rbf", "sdf", "occupancy"])
    ap.add_argument("--spheres", type=int, default=5, help="number of sphere obstacles (panda; 64 = stress variant)")
    ap.add_argument("--goals", type=int, default=1, help="panda: number of goals (config 5: 4)")
    ap.add_argument("--shard-of", default=None, metavar="R,W",
                    help="N=1 only: build shard R of W of a (particles x W)-particle problem without a process group "
                         "(config 5's per-GPU share: --goals 4 --particles 512 --samples 256 --traj-len 128 --shard-of 3,8)")
    ap.add_argument("--store-free", action="store_true",
                    help="the MAIN planner runs optimize(opt_iters=K) store-free (the product's default) and the event pass "
                         "times store-free launches: for rocprofv3 runs of that mode (tools/profile_config.sh); the line's "
                         "`mode` says so.  Without it value / roofline are the storing mode and `store_free` reports the other.")
    ap.add_argument("--single-iteration-calls", action="store_true",
                    help="time K calls of optimize(opt_iters=1) instead of one optimize(opt_iters=K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-store-free", action="store_true", help="skip the store-free leg (a second planner of the same workload)")
    ap.add_argument("--no-parity", action="store_true", help="skip the free-running K = 10 parity leg (fp64 CPU oracle, ~10 s)")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=2)
    return ap.parse_args()


# --------------------------------------------------------------------------------------- N > 1 launch
def self_launch(args):
    """Parent of a multi-GPU run: start one rank per GPU and relay.  Nothing here may initialise the
    GPU (a process that has done so must never be replaced or forked into ranks), so torch is not
    even imported."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    return proc.returncode if proc.returncode != 0 or line is not None else 1


# --------------------------------------------------------------------------------------- workloads
PANDA_GOALS = [[0.5, 0.2, 0.3, -1.5, 0.1, 2.0, 0.3], [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5],
               [0.9, -0.2, 0.4, -1.1, -0.3, 1.9, 0.8], [-0.8, 0.1, 0.6, -2.4, 0.4, 2.6, -0.2]]


def build_planner(torch, workload, P_local, S, T, dtype, dev, rank=0, world=1, field="rbf", spheres=5,
                  goals=1, shard_of=None, seed=0, **kw):
    """-> (planner, observation dict, workload name).  `shard_of` = (rank, world_size) builds one shard
    of a bigger problem without a process group (config 5's per-GPU share on one GPU)."""
    from stoch_gpmp_amd import workloads as W
    ta = {"device": dev, "dtype": dtype}
    if shard_of is not None:
        rank, world = shard_of
    if workload == "panda":
        assert (P_local * world) % goals == 0
        gl = None if goals == 1 else [g + [0.] * 7 for g in PANDA_GOALS[:goals]]
        pl = W.hip_panda_planner(W.PANDA, T, P_local * world // goals, S, ta, field_type=field, seed=seed,
                                 goals=gl, rank=rank, world_size=world, **kw)
        obs = {"obstacle_spheres": torch.as_tensor(W.panda_spheres(num=spheres)).to(**ta)}
        name = f"Panda 7-DoF {P_local * world}p ({P_local}/GPU) x {S}s x {T}t, {goals} goal, GP+goal+self+{spheres} spheres {field}"
    else:
        from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
        gl = GOALS4_PLANAR[:goals]
        om = synthetic_obstacle_map(seed=0, tensor_args=ta)
        assert (P_local * world) % len(gl) == 0
        pl = W.hip_planar_planner(W.PLANAR, T, gl, P_local * world // len(gl), S, om, ta, seed=seed,
                                  rank=rank, world_size=world, **kw)
        obs = {}
        name = f"2-D point mass {P_local * world}p ({P_local}/GPU) x {S}s x {T}t, {len(gl)} goals, GP+goal+200x200 grid"
    return pl, obs, name


def box_copy_bandwidth():
    """This box's own streaming figures (tools/membw.hip: 470 MB tensors, float4 per lane): GB/s of a device-to-device
    copy (read + write) -- the denominator SURVEY.md 8d asks for beside the 8 TB/s spec.  None if the binary is absent."""
    exe = os.path.join(ROOT, "tools", "membw")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout
        best = {}
        for ln in out.splitlines():
            f = ln.split()
            if len(f) >= 6 and f[0] in ("fill", "read", "copy") and f[-1] == "TB/s":
                best[f[0]] = max(best.get(f[0], 0.0), float(f[-2]) * 1e3)
        return best or None
    except Exception:
        return None


def profiled(config_key, kernel):
    """HBM traffic / VALU figures of `kernel` from the committed rocprofv3 --pmc passes of THIS configuration
    (profiles/<round>/traffic_by_config.json, assembled by tools/collect_traffic.py); None when not profiled."""
    for rnd in PROFILE_ROUNDS:
        tf = os.path.join(ROOT, "profiles", rnd, "traffic_by_config.json")
        if not os.path.exists(tf):
            continue
        prof = json.load(open(tf)).get(config_key)
        if not prof:
            continue
        k = prof["kernels"].get(kernel.split("<")[0].split(" ")[0])
        src = f"profiles/{rnd}/{prof.get('files', '')} (rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE; {prof.get('command', '')})"
        return k, src
    return None, None


def roofline_of(kernel, kernel_ms, N_elems, w, costs_bytes, fused, config_key, copy_gbs, step_ms=None, step_mode="", field=None):
    """`roofline` object of one configuration.  Algorithmic bytes per launch = SURVEY.md 8(d): N w (sampler write) + N w
    (sweep read) + P S 8 for the fused launch, N w + P S 8 for the sweep alone.
    `achieved` / `frac` divide them by the TIMED pass's ms_per_step (the whole iteration: this launch, the update kernel
    and whatever the schedule does not hide) -- a lower bound of the launch's own figure that needs no second timing mode
    (round-3 verdict: the event pass runs one launch chain with events between the kernels, the timed pass two
    particle-half chains).  The launch's own average duration from the event pass, and the fraction it gives, are in
    roofline_detail ("event_pass")."""
    alg = (2 if fused else 1) * N_elems * w + costs_bytes
    t_ms = step_ms if step_ms else kernel_ms
    achieved = alg / (t_ms * 1e-3) / 1e9
    ev = alg / (kernel_ms * 1e-3) / 1e9
    k, src = profiled(config_key, kernel)
    # what bounds the launch (round-4 verdict): the vector ALU when the committed counters of this configuration show it busy
    # > 70 % of the launch while the bytes they saw are < 60 % of the algorithmic ones -- the 8(d) fraction `frac` stays (it is
    # what north_star's 40 % clause is about), `valu_frac` = instruction-issue floor / this run's launch duration and
    # `moved_frac` = counter bytes / this run's launch duration / HBM peak say where the launch really stands
    moved = k.get("bytes") if k else None
    valu_busy = k.get("valu_busy_frac_under_profiler") if k else None
    bound = "valu" if (valu_busy is not None and moved is not None and valu_busy > 0.7 and moved / alg < 0.6) else "hbm"
    r = {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": moved,
         "valu_frac": (k["valu_floor_ms"] / kernel_ms) if (k and "valu_floor_ms" in k) else None,
         "moved_frac": (moved / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if moved else None,
         "valu_busy_under_profiler": valu_busy, "field": field, "launch_ms": kernel_ms,
         "timed_by": ("ms_per_step of the timed pass (" + step_mode + ")") if step_ms else "HIP events around the launch"}
    detail = {"algorithmic_bytes_per_launch": alg, "divided_by_ms": t_ms,
              "event_pass": {"avg_launch_ms": kernel_ms, "achieved_GBs": ev, "frac": ev / HBM_PEAK_GBS,
                             "how": "optimize(opt_iters=1) calls, one launch chain, HIP events around every kernel on the launch stream"},
              "traffic_source": src,
              "frac_of_box_copy_bw": achieved / copy_gbs["copy"] if copy_gbs and "copy" in copy_gbs else None,
              "frac_of_guide_copy_bw": achieved / HBM_COPY_GBS,
              "box_streaming_GBs": copy_gbs}
    if k and "valu_floor_ms" in k:
        detail["compute"] = {"valu_wave_insts_per_launch": k.get("valu_insts"),
                             "valu_busy_cycles_per_simd": k.get("valu_busy_cycles_per_simd"),
                             "valu_floor_ms": k["valu_floor_ms"], "frac_of_valu_floor": k["valu_floor_ms"] / kernel_ms,
                             "valu_busy_frac_under_profiler": k.get("valu_busy_frac_under_profiler"),
                             "profiled_clock_ghz": k.get("clock_ghz")}
    return r, detail


def kernel_profile(torch, pl, obs, steps, unread=False):
    """Per-kernel device time with HIP events on the launch stream (a separate pass: the events sit
    between the kernels, so this pass is never the one whose wall time is reported).  unread: the steps run store-free
    (what iterations 1 .. K - 1 of an optimize(opt_iters=K) call do)."""
    pl._engine.profile_enable(True)
    for _ in range(steps):
        if unread:
            pl.step(_samples_unread=True, **obs)
        else:
            pl.optimize(opt_iters=1, **obs)
    torch.cuda.synchronize()
    kms, launches = pl._engine.profile_read()
    pl._engine.profile_enable(False)
    assert launches == steps
    return {k: v / launches for k, v in kms.items()}


def time_loop(torch, pl, obs, steps, warmup, barrier=None, one_call=True):
    """W warm-up iterations, then exactly `steps` timed iterations between two barriers.  one_call: the loop of
    the reference itself, `optimize(opt_iters=steps)` (planner.py:289-299) -- nobody looks at the buffers between
    its iterations, so the context may run them as two particle-half chains; else `steps` calls of
    optimize(opt_iters=1), each returning its own tensors (one chain by construction)."""
    if one_call:
        if warmup:
            pl.optimize(opt_iters=warmup, **obs)
    else:
        for _ in range(warmup):
            pl.optimize(opt_iters=1, **obs)
    (barrier or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    if one_call:
        pl.optimize(opt_iters=steps, **obs)
    else:
        for _ in range(steps):
            pl.optimize(opt_iters=1, **obs)
    (barrier or torch.cuda.synchronize)()
    return time.perf_counter() - t0


def store_free_leg(torch, spec_kwargs, obs_builder, config_key, steps, N_elems, w, costs_bytes, storing_ms, dev):
    """The product's default inside optimize(opt_iters = K) since round 5, reported BESIDE the storing headline (round-4
    verdict, item 1c): iterations 1 .. K - 1 do not write their samples (SGPMP_STEP_NO_SAMPLES; update_kernel regenerates the
    rows that carry weight from their noise keys -- every returned tensor bit-identical to the storing mode's,
    tests/test_gpu_planner.py::test_store_free_*).  SURVEY 8(d)'s algorithmic bytes are defined on the storing mode (sampler
    write + sweep read), so no HBM fraction is quoted for this launch: it is bound by the vector ALU (`valu_frac`), and what
    it moves is what the committed counters of this mode saw."""
    pl, obs, _ = obs_builder(store_free=True, **spec_kwargs)
    time_loop(torch, pl, obs, 150, 0)
    els = [time_loop(torch, pl, obs, steps, 10), time_loop(torch, pl, obs, steps, 0)]
    el = min(els)
    kms = kernel_profile(torch, pl, obs, min(steps, 60), unread=True)
    kernel = pl._engine.last_cost_kernel()
    ran = pl._engine.store_free_steps()
    k, src = profiled(config_key + "_store_free", kernel)
    launch_ms = kms["cost_sweep"]
    out = {"mode": "optimize(opt_iters=K): iterations 1 .. K-1 without the sample stores (the default), the K-th storing",
           "bound": ("valu" if kernel.startswith("fused_step") else "latency: one workgroup per CU in lockstep, three barriers, the update's dependent chain")
           if ran else None, "kernel": kernel + (" with the update inside the launch" if ran and pl._engine.last_step_launches() == 2 and kernel.startswith("fused_planar_seg") else ""), "iterations_per_s": steps / el, "ms_per_step": 1e3 * el / steps,
           "ms_per_step_of_each_pass": [1e3 * e / steps for e in els], "steps": steps,
           "vs_storing": (storing_ms / (1e3 * el / steps)) if storing_ms else None,
           "launch_ms": launch_ms, "update_ms": kms["update"],
           "valu_frac": (k["valu_floor_ms"] / launch_ms) if (k and "valu_floor_ms" in k) else None,
           "moved_bytes_per_launch": k.get("bytes") if k else None,
           "moved_frac_of_hbm_peak": (k["bytes"] / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (k and k.get("bytes")) else None,
           "valu_wave_insts_per_launch": k.get("valu_insts") if k else None,
           "counters": src, "store_free_steps_run": ran,
           "bytes_not_written_per_launch": N_elems * w if ran else 0,
           "results": "bit-identical to the storing mode (means, returned 6-tuple, state_samples): tests/test_gpu_planner.py::test_store_free_*"}
    del pl
    torch.cuda.empty_cache()
    return out


def other_configs(torch, dev, copy_gbs=None):
    """The other single-GPU configurations of BASELINE.json, each timed for a few hundred iterations
    (they cost milliseconds): configs[0] in fp64, configs[1], configs[2] with the sdf field, and the
    per-GPU share of configs[4] (512 of 4096 particles, 4 goals; shard 3 of 8)."""
    f32, f64 = torch.float32, torch.float64
    specs = [
        ("config 1: planar 4 x 16 x 64, fp64", "cfg1", dict(workload="planar", P_local=4, S=16, T=64, dtype=f64, goals=2), 300),
        ("config 2: planar 256 x 64 x 128, fp32", "cfg2", dict(workload="planar", P_local=256, S=64, T=128, dtype=f32, goals=4), 300),
        ("config 3 with the sdf sphere field", "cfg3_sdf", dict(workload="panda", P_local=1024, S=128, T=64, dtype=f32, field="sdf"), 100),
        ("config 3 with 64 sphere obstacles (SURVEY 8d stress variant)", "cfg3_64sph",
         dict(workload="panda", P_local=1024, S=128, T=64, dtype=f32, spheres=64), 40),
        ("config 5 share: Panda 4 goals, 512 of 4096 particles x 256 x 128, fp32 (shard 3 of 8)", "cfg5",
         dict(workload="panda", P_local=512, S=256, T=128, dtype=f32, goals=4, shard_of=(3, 8)), 60),
        # north_star's fp64 clause (means within 1e-5) at configs[2]'s shape: the fp64 context runs sampler, generic sweep and
        # update as three launches (parity: tests/test_gpu_planner.py::test_config3_shape_fp64_free_running_*)
        ("config 3's shape in fp64 (Panda 1024 x 128 x 64)", "cfg3_f64", dict(workload="panda", P_local=1024, S=128, T=64, dtype=f64), 20),
    ]
    out = []
    for label, key, spec, steps in specs:
        # (value / roofline: the storing mode, as for the headline; the store-free figure of the Panda launches beside it)
        pl, obs, name = build_planner(torch, dev=dev, store_free=False, **spec)
        time_loop(torch, pl, obs, 150, 0)                        # (clock and chain-stream warm-up, see main())
        # (two passes, the faster one reported and both kept: these runs last 7-40 ms, and one host hiccup inside a pass --
        # round 4 saw a 29 ms stall once -- would otherwise be read as a 5 x slower kernel.  The headline is one pass of K.)
        els = [time_loop(torch, pl, obs, steps, 10), time_loop(torch, pl, obs, steps, 0)]
        el = min(els)
        el1 = time_loop(torch, pl, obs, steps, 10, one_call=False)
        kms = kernel_profile(torch, pl, obs, min(steps, 30))
        kernel = pl._engine.last_cost_kernel()
        w = 4 if spec["dtype"] == f32 else 8
        fused = kernel.startswith("fused_")
        roof, roof_detail = roofline_of(kernel + (" (K2+K3 in one launch)" if fused else " (K3)"), kms["cost_sweep"],
                                        spec["P_local"] * spec["S"] * spec["T"] * pl.d_state_opt, w,
                                        spec["P_local"] * spec["S"] * 8, fused, key, copy_gbs,
                                        step_ms=1e3 * el / steps, step_mode="optimize(opt_iters=K)",
                                        field=spec.get("field", "rbf") if spec["workload"] == "panda" else "occupancy grid")
        sf = None
        if kernel == "fused_step_kernel" or (kernel == "fused_planar_seg_kernel" and spec["S"] == 64):
            sf = store_free_leg(torch, spec, lambda **kw: build_planner(torch, dev=dev, **kw), key, steps,
                                spec["P_local"] * spec["S"] * spec["T"] * pl.d_state_opt, w, spec["P_local"] * spec["S"] * 8,
                                1e3 * el / steps, dev)
        out.append({"config": label, "workload": name, "iterations_per_s": steps / el, "store_free": sf,
                    "ms_per_step": 1e3 * el / steps, "steps": steps, "ms_per_step_of_each_pass": [1e3 * e / steps for e in els],
                    "iterations_per_s_single_iteration_calls": steps / el1, "kernel_ms_per_step": kms,
                    "cost_kernel": kernel, "roofline": roof, "roofline_detail": roof_detail,
                    "dtype": "f32" if spec["dtype"] == f32 else "f64"})
        del pl
        torch.cuda.empty_cache()
    return out


def reference_examples(torch):
    """The reference's own example problems at their own sizes, run the way its scripts run them -- one optimize(opt_iters=1)
    per loop trip (examples/panda_environment.py:29-32,107,141-147: 5 particles x 32 samples x 64 waypoints, fp32, five cost
    terms incl. the end-effector goal; planar_environment.py:14-20,82,102-108: 15 x 128 x 64, fp64): microseconds per call,
    free-running (the queue drains behind the host) and with the host's enqueue alone.  Latency-bound launches: the step goes
    out as fused_step_small_kernel / the few-wave sampler."""
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
            calls = 500
            for _ in range(50):
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
            del pl
        except Exception as e:                                    # (reported, never fatal for the bench line)
            out[which] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return out


# --------------------------------------------------------------------------------------- CPU figures
def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(args, torch, S, T, P_full):
    """Reference-equivalent PyTorch-CPU path (oracle/ref_equiv.py: replicated [P,M,M] precision,
    MultivariateNormal rebuilt per iteration, dense sampling, dense IS matmul) on a bounded sample:
    `cpu_particles` particles of the workload at its full S and T."""
    from tests import scenarios as SC
    from stoch_gpmp_amd import workloads as W
    cores = host_cores()
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    Pc = args.cpu_particles
    if args.workload == "panda":
        try:
            ora = SC.oracle_panda_planner(W.PANDA, T, Pc, S, dtype=dtype, field_type=args.field, seed=0)
        except ValueError:
            # torch's MultivariateNormal validation can reject stiff fp32 priors (reference
            # README.md:35); the reference must then be run in fp64, and so is its stand-in
            dtype = torch.float64
            ora = SC.oracle_panda_planner(W.PANDA, T, Pc, S, dtype=dtype, field_type=args.field, seed=0)
        obs = {"obstacle_spheres": torch.as_tensor(W.panda_spheres(num=args.spheres)).to(dtype)}
    else:
        from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
        om = synthetic_obstacle_map(seed=0, tensor_args={"device": torch.device("cpu"), "dtype": torch.float64})
        Pc = max(Pc // 4, 1) * 4
        dtype = torch.float64      # the reference cannot build the planar priors in fp32 (README.md:35)
        ora = SC.oracle_planar_planner(W.PLANAR, T, GOALS4_PLANAR, Pc // 4, S, om.map, om.cell_size,
                                       [om.origin_xi, om.origin_yi], seed=0)
        obs = {}
    # The dense algorithm is dominated by batched small-matrix LAPACK / bmm calls that do not scale
    # with threads (256 threads is ~30x SLOWER than 16 on this workload), so a few thread counts are
    # tried and the best one is reported: the baseline is the reference algorithm at its best -- and the
    # all-host-cores figure BASELINE.md section 3 asks for is measured beside it.
    def timed(o, iters, warm=True):
        if warm:
            o.step(**obs)
        t0 = time.perf_counter()
        for _ in range(iters):
            o.step(**obs)
        return (time.perf_counter() - t0) / iters

    best = None
    by_threads = {}
    leg0 = time.perf_counter()
    for threads in sorted({min(cores, 16), min(cores, 32)}):   # (16 has been the best count on every box so far)
        torch.set_num_threads(threads)
        dt = timed(ora, args.cpu_iters)
        by_threads[threads] = 1.0 / dt
        if best is None or dt < best[0]:
            best = (dt, threads)
        if time.perf_counter() - leg0 > 8.0:            # keep the whole baseline leg bounded
            break
    dt, threads = best
    # measured points at 2 x and 4 x the particles (best thread count) while the leg stays inside ~25 s: the value is
    # extrapolated from the TWO LARGEST measured P (time is affine in P: a fixed part + a per-particle part), not from the
    # smallest sample alone (round-4 verdict: the P = 8 point lay 11 % off the line through P = 4 and the origin)
    torch.set_num_threads(threads)
    points = [{"particles": Pc, "it_per_s": 1.0 / dt, "s_per_it": dt}]
    Pk = Pc
    while args.workload == "panda" and len(points) < 3:
        Pk *= 2
        est = points[-1]["s_per_it"] * 2.2 * (args.cpu_iters + 1)
        if time.perf_counter() - leg0 + est > 25.0 or Pk * 0.45e9 > 48e9:
            break
        try:
            ora_k = SC.oracle_panda_planner(W.PANDA, T, Pk, S, dtype=dtype, field_type=args.field, seed=0)
            dk = timed(ora_k, args.cpu_iters)
            points.append({"particles": Pk, "it_per_s": 1.0 / dk, "s_per_it": dk})
            del ora_k
        except Exception:                                   # (memory) keep what was measured
            break
    # all host cores (one iteration, no warm-up beyond the runs above: it is the slow point)
    all_cores = None
    # (LAST: at 256 threads one iteration of the dense algorithm takes ~15 s, which must not eat the budget of the measured points)
    if cores not in by_threads and time.perf_counter() - leg0 < 30.0:
        torch.set_num_threads(cores)
        dta = timed(ora, 1, warm=False)
        torch.set_num_threads(threads)
        all_cores = {"threads": cores, "it_per_s_at_sample": 1.0 / dta, "value_extrapolated_per_particle": (1.0 / dta) * Pc / P_full}
        by_threads[cores] = 1.0 / dta
    if len(points) >= 2:
        (p1, t1), (p2, t2) = [(q["particles"], q["s_per_it"]) for q in points[-2:]]
        t_full = t2 + (t2 - t1) / (p2 - p1) * (P_full - p2)
        how = (f"affine extrapolation of the time per iteration through the two largest measured points (P = {p1}: {t1:.3f} s, "
               f"P = {p2}: {t2:.3f} s) to P = {P_full}")
    else:
        t_full = dt * P_full / Pc
        how = f"per-particle linear extrapolation of the single measured point (P = {Pc}) to P = {P_full}"
    its = 1.0 / dt
    return {
        "value": 1.0 / t_full, "unit": "iterations/s", "cores": threads, "kind": "port",
        "sample": (f"{', '.join(str(q['particles']) for q in points)} of {P_full} particles at full S={S}, T={T}, "
                   f"{str(dtype).split('.')[-1]}, {args.cpu_iters} iterations after 1 warm-up each, best of several torch thread "
                   f"counts ({threads} threads of {cores} host cores); value = {how} (the dense reference algorithm needs "
                   "~0.4 GB per particle)"),
        "measured_it_per_s_at_sample": its, "sample_particles": Pc, "measured_points": points,
        "it_per_s_at_sample_by_torch_threads": by_threads, "all_host_cores": all_cores,
        "torch_threads": threads, "host_cores": cores, "extrapolation": how,
    }


def cpu_fair(args, torch, S, T, P_full):
    """The banded fp64 restatement (oracle/banded_equiv.py): prior factored once, per-DOF 2T x 2T
    sampling GEMM, IS term as a dot product, same cost functions -- what a careful CPU implementation
    of the same mathematics costs.  Panda workloads only; bounded sample, linear in the particles."""
    if args.workload != "panda":
        return None
    from oracle import banded_equiv as B
    from stoch_gpmp_amd import workloads as W
    c, n = W.PANDA, 7
    cores = host_cores()
    Pc = 32
    dtype = torch.float64
    goal = torch.tensor([c["goal_q"] + [0.] * n], dtype=dtype)
    start = torch.tensor(c["start_q"] + [0.] * n, dtype=dtype)
    means = torch.stack([start + (goal[0] - start) * t / (T - 1) for t in range(T)]).repeat(Pc, 1, 1)
    band = B.BandedPlanner(Pc, S, T, c["dt"], n, start, goal, B.panda_chunk_cost(c, T, S, goal, args.field),
                           c["step_size"], c["temperature"], c["sigma_start_sample"], c["sigma_goal_sample"],
                           c["sigma_gp_sample"], means, chunk=8)
    sph = torch.as_tensor(W.panda_spheres(num=args.spheres)).to(dtype)
    g = torch.Generator().manual_seed(0)
    best = None
    leg0 = time.perf_counter()
    for threads in sorted({min(cores, 16), min(cores, 64)}):
        torch.set_num_threads(threads)
        eps = torch.randn(S, Pc, T * 2 * n, generator=g, dtype=dtype)
        band.step(eps, obstacle_spheres=sph)
        t0 = time.perf_counter()
        iters = 2
        for _ in range(iters):
            eps = torch.randn(S, Pc, T * 2 * n, generator=g, dtype=dtype)   # noise generation is part of an iteration
            band.step(eps, obstacle_spheres=sph)
        dt = (time.perf_counter() - t0) / iters
        if best is None or dt < best[0]:
            best = (dt, threads)
        if time.perf_counter() - leg0 > 8.0:
            break
    dt, threads = best
    return {"value": (1.0 / dt) * Pc / P_full, "unit": "iterations/s", "cores": threads, "kind": "port (banded restatement)",
            "sample": (f"{Pc} of {P_full} particles at full S={S}, T={T}, float64, 2 iterations after 1 warm-up, "
                       f"{threads} torch threads of {cores} host cores; measured {1.0 / dt:.3f} it/s at P={Pc}; "
                       "value = linear extrapolation in the particle count (every step of this algorithm is "
                       "linear in P)"),
            "measured_it_per_s_at_sample": 1.0 / dt, "sample_particles": Pc, "host_cores": cores}


# --------------------------------------------------------------------------------------- parity leg
def parity_leg(torch, args, dev, P_local, S, T, goals, K=10):
    """Free-running fp32 (or fp64) parity, measured by THIS run, outside every timed region ("max rel err of
    particle_means after K = 10 iterations"): a fresh HIP planner of the benchmarked workload at its FULL size steps K
    iterations; four of its particles are followed by the fp64 CPU oracle (oracle/ref_equiv.py: the reference's dense
    algorithm restated and pinned to the reference's own runs, tests/golden) on the restated noise of exactly those global
    particle indices (oracle/native_noise.py) -- the oracle is given the four particles' means ONCE, before iteration 1, and
    nothing afterwards.  The oracle is the checker here, never the thing measured."""
    from oracle import native_noise
    from oracle.native_noise import native_eps
    from tests import scenarios as SC
    from stoch_gpmp_amd import workloads as W
    from stoch_gpmp_amd import _lib
    native_noise.DEFAULT_ROUNDS = int(_lib.load().sgpmp_philox_rounds())
    t0 = time.perf_counter()
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    seed = 97
    if args.workload == "panda":
        if goals != 1 or args.shard_of:
            return None                                  # (the multi-goal share is covered by tests/test_gpu_planner.py)
        pl, obs, _ = build_planner(torch, "panda", P_local, S, T, dtype, dev, field=args.field, spheres=args.spheres, seed=seed)
        sub = sorted({0, 1, P_local // 2 - 1, P_local - 1} & set(range(P_local)))
        n = 7
        ora = SC.oracle_panda_planner(W.PANDA, T, len(sub), S, field_type=args.field, seed=seed,
                                      eps_init=torch.zeros(len(sub), 1, T * 2 * n, dtype=torch.float64))
        ora_obs = {"obstacle_spheres": torch.as_tensor(W.panda_spheres(num=args.spheres)).double()}
    else:
        from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
        pl, obs, _ = build_planner(torch, "planar", P_local, S, T, dtype, dev, goals=4, seed=seed)
        nppg = P_local // 4
        sub = [g * nppg + (5 * g) % nppg for g in range(4)]
        n = 2
        om = synthetic_obstacle_map(seed=0, tensor_args={"device": torch.device("cpu"), "dtype": torch.float64})
        ora = SC.oracle_planar_planner(W.PLANAR, T, GOALS4_PLANAR, 1, S, om.map, om.cell_size, [om.origin_xi, om.origin_yi],
                                       seed=seed, eps_init=torch.zeros(1, 4, T * 2 * n, dtype=torch.float64))
        ora_obs = {}
    torch.set_num_threads(min(host_cores(), 16))
    idx = torch.as_tensor(sub, device=dev)
    k = len(sub)
    ora.particle_means.copy_(pl.particle_means[idx].cpu().double())
    ora.prior.set_mean(ora.particle_means.view(k, -1))
    tracking = [True] * k
    per_iter, departures, unexplained = [], [], []
    worst_cost = worst_means = 0.0
    scale = None
    for it in range(1, K + 1):
        eps = torch.from_numpy(native_eps(seed, pl._draw, [pl.p0 + i for i in sub], S, T, n,
                                          "float32" if dtype == torch.float32 else "float64")).double()
        costs_o, _ = ora.step(eps=eps, **ora_obs)
        costs = pl.optimize(opt_iters=1, **obs)[4]
        scale = scale or float(ora.particle_means.abs().max())
        c_hip = costs[idx].cpu().double()
        d = (pl.particle_means[idx].cpu().double() - ora.particle_means).abs().amax(dim=(1, 2)) / scale
        for j in range(k):
            if not tracking[j]:
                continue
            worst_cost = max(worst_cost, float(((c_hip[j] - costs_o[j]).abs() / costs_o[j].abs()).max()))
            if float(d[j]) < 1e-3:
                worst_means = max(worst_means, float(d[j]))
                continue
            a, b = int(c_hip[j].argmin()), int(costs_o[j].argmin())
            gap = float((costs_o[j, a] - costs_o[j, b]).abs() / costs_o[j, b].abs())
            tracking[j] = False
            rec = {"iteration": it, "particle": int(pl.p0 + sub[j]), "near_tie_gap": gap, "means_rel": float(d[j])}
            (departures if (a != b and gap < 2e-5) else unexplained).append(rec)
        per_iter.append(sum(tracking) / k)
    kernel = pl._engine.last_cost_kernel()
    del pl
    torch.cuda.empty_cache()
    tol = 1e-3 if dtype == torch.float32 else 1e-5
    return {
        "measured_by": "this run (bench.py parity_leg; outside the timed region)",
        "what": f"{k} particles {[int(i) for i in sub]} of a fresh full-size run ({P_local} x {S} x {T}), {K} iterations, "
                "HIP planner and fp64 dense oracle each free-running from the same initial means on the same restated noise",
        "kernel": kernel, "iterations": K, "resynchronised": False,
        "tolerance_on_means": tol,
        "means_rel_err_max_after_K": worst_means if all(tracking) else None,
        "means_rel_err_max_while_tracking": worst_means,
        "particles_within_tolerance_per_iteration": per_iter,
        "cost_rel_err_max_while_tracking": worst_cost,
        "departures_on_near_ties": departures,       # arg-min flips between two samples whose oracle costs differ by < 2e-5
        "unexplained_departures": unexplained,
        "ok": (not unexplained) and worst_cost < 5e-3 and (worst_means < tol),
        "leg_seconds": time.perf_counter() - t0,
    }


# --------------------------------------------------------------------------------------- main (a rank)
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE line, the JSON: whatever a library prints to fd 1 meanwhile (RCCL's
    # version banner, ...) is sent to stderr, and the line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    use_dist = world > 1 or os.environ.get("SGPMP_BENCH_FORCE_DIST") == "1"   # (1-rank RCCL smoke test)
    # TEST HOOK (tests/test_gpu_dist.py): every rank on cuda:0, gloo process group, libsgpmp.so bound to the
    # shared-memory test double (SGPMP_RCCL_LIB) -- runs THIS script's N > 1 code on a 1-GPU box.  The line it prints
    # says so ("shared_gpu_test_double": true) and is not a scaling measurement.
    shared_gpu = os.environ.get("SGPMP_BENCH_SHARED_GPU") == "1"
    if shared_gpu and not (os.environ.get("SGPMP_RCCL_LIB") and os.environ.get("SGPMP_LIB_PATH")):
        raise SystemExit("bench.py: SGPMP_BENCH_SHARED_GPU needs SGPMP_RCCL_LIB and SGPMP_LIB_PATH = the test-hooks build of the "
                         "library (real RCCL refuses ranks that share a GPU; the product library ignores SGPMP_RCCL_LIB)")
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (before the first HIP call: dmabuf IPC only)
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    if args.workload == "panda":
        P_local, S, T = args.particles or 1024, args.samples or 128, args.traj_len or 64
        goals = args.goals
    else:
        P_local, S, T = args.particles or 256, args.samples or 64, args.traj_len or 128
        goals = 4
    pl, obs, name = build_planner(torch, args.workload, P_local, S, T, dtype, dev, rank, world,
                                  field=args.field, spheres=args.spheres, goals=goals,
                                  shard_of=tuple(int(v) for v in args.shard_of.split(",")) if args.shard_of and world == 1 else None,
                                  force_stats_allreduce=use_dist and world == 1, store_free=bool(args.store_free))
    w = 4 if dtype == torch.float32 else 8
    d = pl.d_state_opt

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Pass 1: per-kernel device times (HIP events between the kernels; feeds `roofline`).  It runs first so
    # that the wall-clock pass below starts on a GPU that is already at its working clock: with a cold
    # device the first ~100 iterations run 10-20 % slow, which a 5-step warm-up does not cover.
    # The context's two chain streams (optimize(opt_iters >= 2)) need the same: in a fresh process their first
    # ~150 iterations run up to 25 % slow (tools/first_calls_probe.py), whatever the call sizes.
    for _ in range(2):
        pl.optimize(opt_iters=100, **obs)                 # (device + chain-stream warm-up)
    kms = kernel_profile(torch, pl, obs, 100, unread=bool(args.store_free))
    # Pass 2: W untimed warm-up steps, then EXACTLY K timed steps between barriers (the reported value)
    split0 = pl._engine.pipeline_split_steps()
    elapsed = time_loop(torch, pl, obs, args.steps, args.warmup, barrier, one_call=not args.single_iteration_calls)
    split_steps = pl._engine.pipeline_split_steps() - split0
    # Pass 3 (reported beside it): the same K iterations as K optimize(opt_iters=1) calls
    # (--store-free: skipped -- single-iteration calls always store, and a profile of the store-free launches must not mix them in)
    elapsed_calls = time_loop(torch, pl, obs, args.steps, args.warmup, barrier, one_call=False) if not args.store_free else None
    rank_rates = None
    if use_dist:
        mine = torch.tensor([elapsed, elapsed_calls or 0.0], device="cpu" if shared_gpu else dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_rates = [args.steps / float(e[0]) for e in every]           # each rank's own clock, before the max
        elapsed, elapsed_calls = max(float(e[0]) for e in every), (max(float(e[1]) for e in every) if elapsed_calls else None)
    mean_cost, mean_min_cost = pl.global_stats()
    # what the communicator behind the C ABI itself reports (ncclCommCount / ncclCommUserRank / ncclGetVersion):
    # proof that RCCL saw `world` ranks, not a number this script made up
    comm_world, comm_rank, rccl_version = pl._engine.comm_info()
    comm_lib, comm_hooks = pl._engine.comm_library()     # what the C ABI bound for its collectives ("" = none yet)

    if rank == 0:
        N_elems = P_local * S * T * d
        sweep_kernel = pl._engine.last_cost_kernel()          # what the dispatcher really launched
        fused = sweep_kernel.startswith("fused_")
        copy_gbs = box_copy_bandwidth() if world == 1 else None
        is_headline = args.workload == "panda" and (P_local, S, T, args.dtype, args.field, args.spheres, goals) == \
            (1024, 128, 64, "f32", "rbf", 5, 1)
        cfg_key = "cfg3" if is_headline else ("cfg2" if (args.workload, P_local, S, T) == ("planar", 256, 64, 128) else
                                              "cfg5" if (args.workload, P_local, S, T, goals) == ("panda", 512, 256, 128, 4) else "")
        # K2 and K3 in one launch: samples are written once and never re-read by the sweep; the algorithmic bytes of
        # the pair stay SURVEY.md 8(d)'s N w + N w + P S 8 -- traffic the fusion legitimately avoids raises the fraction
        ms_step = 1e3 * elapsed / args.steps
        roof, roof_detail = roofline_of(sweep_kernel + (" (K2+K3 in one launch)" if fused else " (K3)"), kms["cost_sweep"],
                                        N_elems, w, P_local * S * 8, fused, cfg_key + ("_store_free" if args.store_free and cfg_key else ""),
                                        copy_gbs, step_ms=ms_step,
                                        step_mode="optimize(opt_iters=1) x K" if args.single_iteration_calls else "optimize(opt_iters=K)",
                                        field=args.field if args.workload == "panda" else "occupancy grid")
        # the store-free mode of the same workload (the product's default inside one optimize() call), beside the headline
        sfree = None
        if world == 1 and not args.store_free and not args.no_store_free and fused:
            sfree = store_free_leg(torch, dict(workload=args.workload, P_local=P_local, S=S, T=T, dtype=dtype, field=args.field,
                                               spheres=args.spheres, goals=goals,
                                               shard_of=tuple(int(v) for v in args.shard_of.split(",")) if args.shard_of else None),
                                   lambda **kw: build_planner(torch, dev=dev, **kw), cfg_key, args.steps, N_elems, w,
                                   P_local * S * 8, ms_step, dev)
        # bytes the iteration really moves: K4 reads only the sample rows whose softmax weight is not exactly zero
        nnz_rows = int((pl._weights_buf != 0).sum())
        iter_alg = 3 * N_elems * w + 2 * P_local * T * d * w + 2 * P_local * S * 8
        iter_moved = ((1 if fused else 2) * N_elems * w + nnz_rows * T * d * w + 4 * P_local * T * d * w
                      + 3 * P_local * S * 8)
        # free-running K = 10 parity of the benchmarked workload, measured by this run (rank 0 at N = 1, like cpu_baseline)
        parity = None
        if world == 1 and not args.no_parity:
            parity = parity_leg(torch, args, dev, P_local, S, T, goals)
        cpu = cpu_detail = fair = None
        if world == 1 and not args.no_cpu_baseline:
            t_cpu = time.perf_counter()
            cpu_detail = cpu_baseline(args, torch, S, T, P_local)
            cpu_detail["leg_seconds"] = time.perf_counter() - t_cpu
            cpu = {"value": cpu_detail["value"], "unit": cpu_detail["unit"], "cores": cpu_detail["cores"],
                   "kind": cpu_detail["kind"],
                   "sample": f"{', '.join(str(q['particles']) for q in cpu_detail['measured_points'])} of {P_local} particles, "
                             f"full S x T, {args.cpu_iters} iterations each, extrapolated through the two largest; all "
                             f"{cpu_detail['host_cores']} cores: "
                             + (f"{cpu_detail['all_host_cores']['value_extrapolated_per_particle']:.4g} it/s" if cpu_detail.get('all_host_cores') else "= cores used")}
            t_cpu = time.perf_counter()
            fair = cpu_fair(args, torch, S, T, P_local)
            if fair:
                fair["leg_seconds"] = time.perf_counter() - t_cpu
        value = world * args.steps / elapsed
        # The FIRST 1500 characters carry what the driver's record keeps: value, single_iteration_calls, roofline,
        # cpu_baseline.  Everything verbose follows.
        out = {
            "metric": "planner iterations/sec (and ms/iter) at fixed particles x samples x T",
            # whole-job aggregate: every rank advances its 1024-particle shard by one iteration per step (weak
            # scaling); at N = 1 this is plain planner iterations/s of the BASELINE config
            "value": value,
            "unit": "iterations/s" if world == 1 else f"iterations/s of a {P_local}-particle shard, summed over {world} shards",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if dtype == torch.float32 else "f64", "data": "synthetic",
            "field": args.field if args.workload == "panda" else "occupancy grid",
            "single_call_iterations_per_s": (world * args.steps / elapsed_calls) if elapsed_calls else None,   # K x optimize(opt_iters=1), the reference examples' loop
            "single_iteration_calls": {"iterations_per_s": world * args.steps / elapsed_calls,
                                       "ms_per_step": 1e3 * elapsed_calls / args.steps} if elapsed_calls else None,
            "roofline": roof,
            "cpu_baseline": cpu,
            "mode": "store_free (--store-free: optimize(opt_iters=K) skips the sample stores of iterations 1 .. K-1)" if args.store_free
                    else "storing (every iteration writes its samples: the mode SURVEY 8(d)'s bytes are defined on)",
            "store_free": sfree,
            "config": {"workload": name, "particles_per_gpu": P_local, "particles_total": P_local * world,
                       "samples": S, "traj_len": T, "state_dim": d,
                       "parallelism": f"particle-sharded x{world}, RCCL statistics all-reduce inside sgpmp_step"
                       if world > 1 else "single GPU",
                       "noise": f"philox4x32-{pl._engine.lib.sgpmp_philox_rounds()} + Box-Muller (in-kernel)",
                       "prior_factor_dtype": "f64"},
            "rccl": {"ranks": comm_world, "rank": comm_rank, "version": rccl_version,
                     "library": comm_lib, "test_hooks_build": bool(comm_hooks),
                     "how": "ncclCommCount / ncclCommUserRank / ncclGetVersion of the communicator sgpmp_step all-reduces on "
                            "(0 ranks: no communicator attached, single GPU)"},
            "shared_gpu_test_double": shared_gpu or bool(comm_hooks and "rccl" not in comm_lib),
            "per_rank_iterations_per_s": None if rank_rates is None else
            {"min": min(rank_rates), "max": max(rank_rates), "all": rank_rates},
            "planner_iterations_per_s": args.steps / elapsed,
            "speedup_vs_cpu_baseline": value / cpu["value"] if cpu else None,
            "parity": parity,
            "roofline_detail": roof_detail,
            "cpu_baseline_detail": cpu_detail,
            "cpu_fair": fair,
            "speedup_vs_cpu_fair": value / fair["value"] if fair else None,
            "kernel_ms_per_step": kms,
            "launches_per_iteration": pl._engine.last_step_launches(),
            "passes": "1: 2 x optimize(opt_iters=100) untimed (clock and stream warm-up), then 100 iterations with HIP "
                      "events between the kernels (kernel_ms_per_step, roofline_detail.event_pass); "
                      "2: optimize(opt_iters=W) untimed, then optimize(opt_iters=K) -- the reference's own loop, "
                      "planner.py:289-299 -- between barriers (value, ms_per_step, roofline.frac); "
                      "3: the same as W + K calls of optimize(opt_iters=1) (single_iteration_calls)",
            "loop": {"call": "optimize(opt_iters=1) x K" if args.single_iteration_calls else "optimize(opt_iters=K)",
                     "steps_run_as_two_particle_half_chains": split_steps,
                     "note": "inside one optimize() call nobody reads the buffers between iterations, so the context "
                             "runs the call's iterations as two particle-half launch sequences on two streams of "
                             "its own (sgpmp_pipeline_begin/_end): one half's update kernel runs under the other "
                             "half's sampler + sweep launch; bit-identical results"},
            "iteration_roofline": {"algorithmic_bytes": iter_alg, "moved_bytes": iter_moved,
                                   "k4_rows_read": nnz_rows,
                                   "note": "algorithmic_bytes is SURVEY 8d's three-pass model (sampler write, sweep read, update "
                                           "read); the fused launch and the one-hot update move moved_bytes -- the fraction "
                                           "of the HBM peak is quoted for THAT",
                                   "moved_GBs": iter_moved / (ms_step * 1e-3) / 1e9,
                                   "moved_frac_of_hbm_peak": iter_moved / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "last_iteration": {"mean_cost_sum": mean_cost, "mean_min_cost": mean_min_cost},
        }
        if world == 1 and not args.no_other_configs and is_headline:
            del pl
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(torch, dev, copy_gbs)
            out["reference_examples"] = reference_examples(torch)
        line = json.dumps(out)
        head = line[:1500]
        assert all(k in head for k in ('"value"', '"single_iteration_calls"', '"roofline"', '"cpu_baseline"')), len(head)
        os.write(json_fd, (line + "\n").encode())
    # the context (and its RCCL communicator) goes before torch's process group does
    pl = None
    import gc
    gc.collect()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
