#!/usr/bin/env python3
"""Benchmark of the StochGPMP inner loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is ONE planner iteration = one body of the loop at reference planner.py:289-299 (draw S samples per
particle, evaluate the composite cost, softmax-reweight, update the particle means) on synthetic data.
Workload at N = 1: BASELINE.json configs[2] -- Panda 7-DoF (14-D state), 1024 particles x 128 samples x 64
waypoints, 5 synthetic sphere obstacles (rbf field) + self-collision, fp32 compute, prior factored in fp64.
At N > 1 every rank holds 1024 particles of a (1024 N)-particle problem (configs[3] at N = 8: weak scaling,
particles sharded, no data-path collective; a [64,4]-double statistics all-reduce over RCCL per iteration,
enqueued by sgpmp_step).  With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks
BEFORE anything touches the GPU and relays rank 0's line and the exit code.

Output (rank 0):
  * stdout: ONE compact JSON line, < 4 KB, strict JSON (`compact_line`, tests/test_cpu_host.py holds it to
    that): value / ms_per_step (storing mode: every iteration writes its samples -- the mode SURVEY.md 8(d)'s
    algorithmic bytes are defined on), roofline, cpu_baseline, single_iteration_calls, store_free, parity,
    rccl, one short row per other single-GPU configuration;
  * bench_detail.json beside this script: everything else -- per-kernel event timings, the counters' sources,
    the measured CPU points, per-pass times (stderr only says where it went: nothing JSON-shaped besides the line).

`roofline.frac` = SURVEY 8(d)'s algorithmic bytes of the dominant launch / the TIMED pass's ms_per_step / 8 TB/s
(a lower bound: the step also holds the update kernel); `launch_ms` is the launch's own average duration from
a separate pass with HIP events on the launch stream and `frac_launch` the fraction it gives; `bound` says
what really bounds the launch (`valu_frac` = the vector ALU's instruction-issue floor / launch_ms; `moved_frac`
= bytes the committed rocprofv3 counters saw / launch_ms / 8 TB/s; `traffic` = those bytes per launch).
The variants that used to ride in this line (sdf field, 64 spheres, fp64 shape, the reference's example sizes,
the banded "fair CPU" figure, the all-host-cores point) are tools/bench_variants.py -> profiles/rNN/.
"""
import argparse
import json
import math
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
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03")   # committed rocprofv3 --pmc sets, newest first
LINE_LIMIT = 4096           # the driver parses the whole stdout line or nothing (round-5 verdict): keep it small
DETAIL_FILE = "bench_detail.json"
GOALS4_PLANAR = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
PANDA_GOALS = [[0.5, 0.2, 0.3, -1.5, 0.1, 2.0, 0.3], [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5],
               [0.9, -0.2, 0.4, -1.1, -0.3, 1.9, 0.8], [-0.8, 0.1, 0.6, -2.4, 0.4, 2.6, -0.2]]


def parse(argv=None):
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
                         "times store-free launches: for rocprofv3 runs of that mode (tools/profile_config.sh)")
    ap.add_argument("--single-iteration-calls", action="store_true",
                    help="time K calls of optimize(opt_iters=1) instead of one optimize(opt_iters=K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-store-free", action="store_true", help="skip the store-free leg (a second planner of the same workload)")
    ap.add_argument("--no-parity", action="store_true", help="skip the free-running K = 10 parity leg (fp64 CPU oracle)")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-sweep-alone", action="store_true", help="skip the stand-alone sampler / sweep event pass")
    ap.add_argument("--cpu-particles", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="bound of the cpu_baseline leg")
    ap.add_argument("--detail", default=os.path.join(ROOT, DETAIL_FILE), help="where the detailed record goes")
    return ap.parse_args(argv)


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


# --------------------------------------------------------------------------------------- the line
def _clean(o, digits=None):
    """JSON-safe copy: NaN / Infinity -> None (strict JSON), floats optionally rounded to `digits` significant digits."""
    if isinstance(o, float):
        if not math.isfinite(o):
            return None
        if digits and o != 0.0:
            return float(f"{o:.{digits}g}")
        return o
    if isinstance(o, dict):
        return {str(k): _clean(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v, digits) for v in o]
    if isinstance(o, (str, int, bool)) or o is None:
        return o
    return str(o)


def _pick(d, keys, digits=5):
    return None if not d else {k: _clean(d.get(k), digits) for k in keys if k in d}


def compact_line(full):
    """The ONE stdout line from the detailed record: the contract keys, short, strict JSON, < LINE_LIMIT characters.
    Optional parts are dropped (last first) if a run ever grows past the limit -- the contract keys never are."""
    roof = full.get("roofline") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in ("workload", "particles_total", "particles_per_gpu", "samples", "traj_len",
                                             "parallelism") if k in cfg}
    out["roofline"] = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "valu_frac",
                                   "moved_frac", "launch_ms", "frac_launch"))
    out["cpu_baseline"] = _pick(full.get("cpu_baseline"), ("value", "unit", "cores", "kind", "sample"))
    optional = []                                    # (key, value): dropped from the END if the line is too long

    def opt(key, val):
        if val is not None:
            out[key] = val
            optional.append(key)
    opt("speedup_vs_cpu_baseline", _clean(full.get("speedup_vs_cpu_baseline"), 4))
    opt("single_iteration_calls", _pick(full.get("single_iteration_calls"), ("iterations_per_s", "ms_per_step")))
    opt("store_free", _pick(full.get("store_free"), ("iterations_per_s", "ms_per_step", "valu_frac", "launch_ms")))
    opt("parity", _pick(full.get("parity"), ("ok", "means_rel_err_max_after_K", "cost_rel_err_max_while_tracking",
                                              "iterations", "tolerance_on_means"), 3))
    opt("rccl", _pick(full.get("rccl"), ("ranks", "rank", "version")))
    opt("shared_gpu_test_double", full.get("shared_gpu_test_double"))
    opt("per_rank_iterations_per_s", full.get("per_rank_iterations_per_s"))
    opt("last_iteration", _clean(full.get("last_iteration")))
    opt("sweep_alone_frac", _clean((full.get("sweep_alone") or {}).get("frac"), 4))
    opt("sweep_in_step_frac", _clean((full.get("sweep_alone") or {}).get("in_step_frac"), 4))
    opt("sweep_alone", _pick(full.get("sweep_alone"), ("kernel", "launch_ms", "frac", "in_step_launch_ms", "in_step_frac", "launch_ms_rocprof", "frac_rocprof",
                                                       "sampler_kernel", "sampler_launch_ms", "sampler_frac"), 4))
    opt("mode", full.get("mode"))
    opt("field", full.get("field"))
    opt("launches_per_iteration", full.get("launches_per_iteration"))
    rows = []
    for r in full.get("other_configs") or []:
        rr = r.get("roofline") or {}
        rows.append(_clean({"config": r.get("config"), "it_per_s": r.get("iterations_per_s"),
                            "store_free_it_per_s": (r.get("store_free") or {}).get("iterations_per_s"),
                            "frac": rr.get("frac"), "bound": rr.get("bound"), "kernel": r.get("cost_kernel"),
                            "dtype": r.get("dtype")}, 4))
    opt("other_configs", rows or None)
    opt("detail", full.get("detail_file"))
    out = _clean(out)
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    while len(line) >= LINE_LIMIT and optional:
        out.pop(optional.pop())
        line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    assert len(line) < LINE_LIMIT, len(line)
    return line


def emit(full, json_fd, detail_path):
    """Write the detailed record beside the script, the compact line to the saved stdout."""
    full = dict(full)
    full["detail_file"] = os.path.basename(detail_path) if detail_path else None
    try:
        if detail_path:
            with open(detail_path, "w") as f:
                json.dump(_clean(full), f, indent=1, allow_nan=False)
    except OSError as e:                                  # (a read-only tree must not cost the line)
        full["detail_file"] = None
        print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
    # (nothing JSON-shaped goes to stderr: a driver that scans the combined output for the line must find exactly one)
    print(f"bench.py: detailed record -> {detail_path if full['detail_file'] else '(not written)'}", file=sys.stderr)
    line = compact_line(full)
    os.write(json_fd, (line + "\n").encode())
    return line


# --------------------------------------------------------------------------------------- workloads
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
        if not k:
            continue
        return k, f"profiles/{rnd}/{prof.get('files', '')} ({prof.get('command', '')})"
    return None, None


def roofline_of(kernel, kernel_ms, N_elems, w, costs_bytes, fused, config_key, step_ms=None):
    """(`roofline`, detail) of one configuration.  Algorithmic bytes per launch = SURVEY.md 8(d): N w (sampler write) + N w
    (sweep read) + P S 8 for the fused launch, N w + P S 8 for the sweep alone.  `achieved` / `frac` divide them by the
    TIMED pass's ms_per_step (round-3 verdict: a bound nobody can dispute -- the event pass runs one launch chain with
    events between the kernels, the timed pass two particle-half chains); `launch_ms` / `frac_launch`: the launch's own
    average duration from the event pass.  `bound` = "valu" when the committed counters of this configuration show the
    vector ALU busy > 70 % of the launch while the bytes they saw are < 60 % of the algorithmic ones."""
    alg = (2 if fused else 1) * N_elems * w + costs_bytes
    t_ms = step_ms if step_ms else kernel_ms
    achieved = alg / (t_ms * 1e-3) / 1e9
    k, src = profiled(config_key, kernel)
    moved = k.get("bytes") if k else None
    valu_busy = k.get("valu_busy_frac_under_profiler") if k else None
    bound = "valu" if (valu_busy is not None and moved is not None and valu_busy > 0.7 and moved / alg < 0.6) else "hbm"
    r = {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": moved,
         "valu_frac": (k["valu_floor_ms"] / kernel_ms) if (k and "valu_floor_ms" in k) else None,
         "moved_frac": (moved / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if moved else None,
         "launch_ms": kernel_ms, "frac_launch": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    detail = {"algorithmic_bytes_per_launch": alg, "frac_divided_by_ms": t_ms, "traffic_source": src,
              "frac_of_guide_copy_bw": achieved / HBM_COPY_GBS, "valu_busy_under_profiler": valu_busy}
    if k and "valu_floor_ms" in k:
        detail["compute"] = {f: k.get(f) for f in ("valu_insts", "valu_busy_cycles_per_simd", "valu_floor_ms", "clock_ghz")}
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
    the reference itself, `optimize(opt_iters=steps)` (planner.py:289-299); else `steps` calls of optimize(opt_iters=1),
    each returning its own tensors (the loop the reference's example scripts run)."""
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


def store_free_leg(torch, spec_kwargs, dev, config_key, steps, storing_ms, passes=1):
    """The product's default inside optimize(opt_iters = K), reported BESIDE the storing headline: iterations 1 .. K - 1 do
    not write their samples (SGPMP_STEP_NO_SAMPLES; update_kernel regenerates the rows that carry weight from their noise
    keys -- every returned tensor bit-identical to the storing mode's, tests/test_gpu_planner.py::test_store_free_*).
    SURVEY 8(d)'s bytes are defined on the storing mode, so no HBM fraction is quoted for this launch: `valu_frac`."""
    pl, obs, _ = build_planner(torch, dev=dev, store_free=True, **spec_kwargs)
    time_loop(torch, pl, obs, 150, 0)
    els = [time_loop(torch, pl, obs, steps, 10 if i == 0 else 0) for i in range(passes)]
    el = min(els)
    kms = kernel_profile(torch, pl, obs, min(steps, 60), unread=True)
    kernel = pl._engine.last_cost_kernel()
    ran = pl._engine.store_free_steps()
    # (planar problems with 64 samples per particle: ONE launch runs all K - 1 store-free iterations of a call -- launch_ms below
    # is then the single-iteration launch the step profiler times, iterations_per_launch what the timed call ran)
    multi = pl._engine.multi_iteration_launches()
    k, src = profiled(config_key + "_store_free", kernel + "_tail")   # (tools/summarise_prof.py: the planar launch with its update inside)
    if not k:
        k, src = profiled(config_key + "_store_free", kernel)
    launch_ms = kms["cost_sweep"]
    out = {"kernel": kernel, "iterations_per_s": steps / el, "ms_per_step": 1e3 * el / steps,
           "ms_per_step_of_each_pass": [1e3 * e / steps for e in els], "steps": steps,
           "vs_storing": (storing_ms / (1e3 * el / steps)) if storing_ms else None,
           "launch_ms": launch_ms, "update_ms": kms.get("update"), "launches_per_step": pl._engine.last_step_launches(),
           "valu_frac": (k["valu_floor_ms"] / launch_ms) if (k and "valu_floor_ms" in k) else None,
           "moved_bytes_per_launch": k.get("bytes") if k else None, "counters": src, "store_free_steps_run": ran,
           "multi_iteration_launches": multi, "iterations_per_launch": (steps - 1) if multi else 1}
    del pl
    torch.cuda.empty_cache()
    return out


def sweep_alone_leg(torch, pl, obs, N_elems, w, costs_bytes, reps=60, config_key=""):
    """north_star's ">= 40 % of the HBM roofline on the cost-gradient sweep" is about the STAND-ALONE sweep: sgpmp_cost_eval on the
    planner's own sample tensor (K3 as a launch of its own: what sample_and_eval() and the two-launch steps run), `reps` launches
    back to back between two HIP events on the launch stream -- likewise the stand-alone sampler (sgpmp_sample).  Sweep bytes =
    N w + P S 8; sampler bytes = N w."""
    from stoch_gpmp_amd import _lib as L
    eng, S = pl._engine, pl.num_samples
    sph = pl._spheres(obs)
    isw = eng.is_weights(pl.particle_means, pl.temperature)

    def sweep():
        eng.cost_eval(pl.state_samples, batch_offset=pl.p0 * S, spheres=sph, is_weights=isw, rows_per_particle=S,
                      out=pl._costs, out64=pl._costs64)

    def sampler():
        eng.sample(L.PRIOR_SAMPLE, pl.seed, 1 << 20, pl.particle_means, S, out=pl.state_samples, mode_offset=pl.p0)

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    sweep_ms = timed(sweep)
    kernel = eng.last_cost_kernel()
    samp_ms = timed(sampler)
    pl.optimize(opt_iters=1, **obs)                        # (the planner's buffers hold a step's tensors again)
    # ... and the same kernel where a planner runs it as a launch of its own: inside two-launch STEPS (sampler, sweep, update, one
    # after the other -- option no_fused_step), timed by the context's HIP events around each kernel.  (`reps` launches of nothing
    # but this kernel run into the chip's power limit: rocprofv3's per-dispatch trace shows the first dozen at 120 us and the rest
    # at 151 us -- profiles/r06/sweep_series_trace.txt -- while between a sampler and an update the sweep keeps 120 us.)
    in_step = None
    try:
        eng.set_option("no_fused_step", 1)
        for _ in range(5):
            pl.optimize(opt_iters=1, **obs)
        kms = kernel_profile(torch, pl, obs, 40)
        if eng.last_cost_kernel() == kernel:
            in_step = {"sweep_ms": kms["cost_sweep"], "sampler_ms": kms.get("sample"), "update_ms": kms.get("update"), "steps": 40}
    finally:
        eng.set_option("no_fused_step", 0)
        pl.optimize(opt_iters=1, **obs)
    # beside the live figure (launches back to back: the clock a chip holds under nothing but this kernel), the committed
    # rocprofv3 duration of the same kernel inside a loop of two-launch steps (profiles/rNN: <config>_unfused)
    k, src = profiled(config_key + "_unfused", kernel) if config_key else (None, None)
    prof_ms = k["avg_ns_under_stats"] * 1e-6 if (k and "avg_ns_under_stats" in k) else None
    return {"kernel": kernel, "launch_ms": sweep_ms, "algorithmic_bytes": N_elems * w + costs_bytes,
            "launch_ms_rocprof": prof_ms, "rocprof_source": src,
            "frac_rocprof": ((N_elems * w + costs_bytes) / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if prof_ms else None,
            "frac": (N_elems * w + costs_bytes) / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "in_step_launch_ms": in_step["sweep_ms"] if in_step else None,
            "in_step_frac": ((N_elems * w + costs_bytes) / (in_step["sweep_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if in_step else None,
            "in_step": in_step,
            "in_step_how": "HIP events of the context around each kernel of 40 two-launch steps (sampler, sweep, update; option no_fused_step)",
            "sampler_kernel": "sample_iso_kernel", "sampler_launch_ms": samp_ms,
            "sampler_frac": N_elems * w / (samp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "launches_timed": reps,
            "how": "launches back to back between two HIP events on the launch stream"}


OTHER_SPECS = [
    ("config 1: planar 4x16x64 fp64", "cfg1", dict(workload="planar", P_local=4, S=16, T=64, dtype="f64", goals=2), 300),
    ("config 2: planar 256x64x128 fp32", "cfg2", dict(workload="planar", P_local=256, S=64, T=128, dtype="f32", goals=4), 300),
    ("config 5 share: Panda 4 goals, 512 of 4096 x256x128 fp32", "cfg5",
     dict(workload="panda", P_local=512, S=256, T=128, dtype="f32", goals=4, shard_of=(3, 8)), 60),
]


def config_row(torch, dev, label, key, spec, steps, passes=1, single_calls=False):
    """One other configuration: `passes` timed passes of K iterations (the fastest kept), an event pass, the store-free
    mode beside it where the launch has one.  spec["options"]: development switches of the context (sgpmp_set_option)."""
    spec = dict(spec)
    dtype = {"f32": torch.float32, "f64": torch.float64}[spec.pop("dtype")]
    options = spec.pop("options", {})
    pl, obs, name = build_planner(torch, dev=dev, dtype=dtype, store_free=False, **spec)
    for k, v in options.items():
        pl._engine.set_option(k, v)
    time_loop(torch, pl, obs, 150, 0)                        # (clock and chain-stream warm-up, see main())
    els = [time_loop(torch, pl, obs, steps, 10 if i == 0 else 0) for i in range(passes)]
    el = min(els)
    el1 = time_loop(torch, pl, obs, steps, 10, one_call=False) if single_calls else None
    kms = kernel_profile(torch, pl, obs, min(steps, 30))
    kernel = pl._engine.last_cost_kernel()
    w = 4 if dtype == torch.float32 else 8
    fused = kernel.startswith("fused_")
    N_elems = spec["P_local"] * spec["S"] * spec["T"] * pl.d_state_opt
    roof, roof_detail = roofline_of(kernel, kms["cost_sweep"], N_elems, w, spec["P_local"] * spec["S"] * 8, fused, key,
                                    step_ms=1e3 * el / steps)
    launches = pl._engine.last_step_launches()
    del pl
    torch.cuda.empty_cache()
    sf = None
    if kernel == "fused_step_kernel" or (kernel == "fused_planar_seg_kernel" and spec["S"] == 64):
        sf = store_free_leg(torch, dict(spec, dtype=dtype), dev, key, steps, 1e3 * el / steps, passes)
    return {"config": label, "workload": name, "iterations_per_s": steps / el, "store_free": sf,
            "ms_per_step": 1e3 * el / steps, "steps": steps, "ms_per_step_of_each_pass": [1e3 * e / steps for e in els],
            "iterations_per_s_single_iteration_calls": (steps / el1) if el1 else None, "kernel_ms_per_step": kms,
            "cost_kernel": kernel, "launches_per_iteration": launches, "roofline": roof, "roofline_detail": roof_detail,
            "dtype": "f32" if dtype == torch.float32 else "f64"}


def other_configs(torch, dev):
    """The other single-GPU configurations of BASELINE.json: configs[0] in fp64, configs[1], and the per-GPU share of
    configs[4] (512 of 4096 particles, 4 goals; shard 3 of 8).  One pass each (they cost milliseconds)."""
    return [config_row(torch, dev, *spec) for spec in OTHER_SPECS]


# --------------------------------------------------------------------------------------- CPU figure
def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(args, torch, S, T, P_full):
    """Reference-equivalent PyTorch-CPU path (oracle/ref_equiv.py: replicated [P,M,M] precision, MultivariateNormal rebuilt
    per iteration, dense sampling, dense IS matmul) on a BOUNDED sample: `cpu_particles`, 2 x and 4 x that many particles of the
    workload at its full S and T (while they fit `cpu_seconds`), on 16 torch threads (the dense algorithm is dominated by batched small-matrix LAPACK / bmm
    calls that do not scale with threads: 16 was the best count on every box of rounds 2-5; thread sweep, all-cores point
    and the banded "fair CPU" figure: tools/bench_variants.py --cpu).  value = affine extrapolation of the time per
    iteration through the measured points to the full particle count (time = fixed part + per-particle part)."""
    from tests import scenarios as SC
    from stoch_gpmp_amd import workloads as W
    cores = host_cores()
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    Pc = args.cpu_particles
    leg0 = time.perf_counter()

    def make(P):
        nonlocal dtype
        if args.workload == "panda":
            try:
                return SC.oracle_panda_planner(W.PANDA, T, P, S, dtype=dtype, field_type=args.field, seed=0)
            except ValueError:
                # torch's MultivariateNormal validation can reject stiff fp32 priors (reference
                # README.md:35); the reference must then be run in fp64, and so is its stand-in
                dtype = torch.float64
                return SC.oracle_panda_planner(W.PANDA, T, P, S, dtype=dtype, field_type=args.field, seed=0)
        from stoch_gpmp_amd.envs.obst_map import synthetic_obstacle_map
        om = synthetic_obstacle_map(seed=0, tensor_args={"device": torch.device("cpu"), "dtype": torch.float64})
        dtype = torch.float64      # the reference cannot build the planar priors in fp32 (README.md:35)
        return SC.oracle_planar_planner(W.PLANAR, T, GOALS4_PLANAR, max(P // 4, 1), S, om.map, om.cell_size,
                                        [om.origin_xi, om.origin_yi], seed=0)

    def timed(o, iters, obs):
        o.step(**obs)                                       # warm-up
        t0 = time.perf_counter()
        for _ in range(iters):
            o.step(**obs)
        return (time.perf_counter() - t0) / iters

    if args.workload != "panda":
        Pc = max(Pc // 4, 1) * 4
    ora = make(Pc)
    obs = {"obstacle_spheres": torch.as_tensor(W.panda_spheres(num=args.spheres)).to(dtype)} if args.workload == "panda" else {}
    dt = timed(ora, args.cpu_iters, obs)
    points = [{"particles": Pc, "s_per_it": dt, "iterations": args.cpu_iters}]
    del ora
    # more points at 2 x, 4 x the particles while the leg stays inside its bound (1 warm-up + `cpu_iters` timed iterations, 1 for
    # the last point that fits)
    Pk = Pc
    while args.workload == "panda" and len(points) < 3:
        Pk *= 2
        per_it = 2.2 * points[-1]["s_per_it"]
        left = args.cpu_seconds - (time.perf_counter() - leg0)
        iters = args.cpu_iters if per_it * (args.cpu_iters + 1) < left else 1
        if per_it * (iters + 1) > left or Pk * 0.45e9 > 48e9:
            break
        try:
            ora = make(Pk)
            points.append({"particles": Pk, "s_per_it": timed(ora, iters, obs), "iterations": iters})
            del ora
        except Exception:                                   # (memory) keep what was measured
            break
    if len(points) >= 2:
        (p1, t1), (p2, t2) = [(q["particles"], q["s_per_it"]) for q in points[-2:]]
        t_full = t2 + (t2 - t1) / (p2 - p1) * (P_full - p2)
        how = "affine extrapolation through the measured points"
    else:
        t_full = dt * P_full / Pc
        how = "per-particle linear extrapolation of the one measured point"
    return {"value": 1.0 / t_full, "unit": "iterations/s", "cores": threads, "kind": "port",
            "sample": f"{'+'.join(str(q['particles']) for q in points)} of {P_full} particles at full S={S} T={T}, "
                      f"{str(dtype).split('.')[-1]}, {threads} of {cores} host cores; {how} to P={P_full}",
            "measured_points": points, "host_cores": cores, "torch_threads": threads,
            "leg_seconds": time.perf_counter() - leg0}


# --------------------------------------------------------------------------------------- parity leg
def parity_leg(torch, args, dev, P_local, S, T, goals, K=10, followed=2):
    """Free-running fp32 (or fp64) parity, measured by THIS run, outside every timed region ("max rel err of
    particle_means after K = 10 iterations"): a fresh HIP planner of the benchmarked workload at its FULL size steps K
    iterations; `followed` of its particles are followed by the fp64 CPU oracle (oracle/ref_equiv.py: the reference's dense
    algorithm restated and pinned to the reference's own runs, tests/golden) on the restated noise of exactly those global
    particle indices (oracle/native_noise.py) -- the oracle is given the particles' means ONCE, before iteration 1, and
    nothing afterwards.  The oracle is the checker here, never the thing measured.  (Whole populations:
    tests/test_gpu_planner.py::test_whole_population_*.)"""
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
        sub = sorted({0, P_local - 1, P_local // 2 - 1, 1}.intersection(range(P_local)))
        if followed < len(sub):
            sub = sorted(set([0, P_local - 1] + sub[:followed - 2]))[:followed]
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
        if dtype == torch.float64:
            # (fp64 contexts draw the fp32 normals, widened: the oracle gets the eps the kernels draw, read back through the C ABI
            # -- the numpy restatement follows the hardware's log2 / sin / cos to an ulp of fp32 only, which an fp64 comparison sees)
            eps = torch.cat([pl._engine.noise(seed, pl._draw, 1, S, mode_offset=pl.p0 + i) for i in sub], dim=1).cpu()
        else:
            eps = torch.from_numpy(native_eps(seed, pl._draw, [pl.p0 + i for i in sub], S, T, n, "float32")).double()
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
    return {"particles_followed": [int(i) for i in sub], "full_size": [P_local, S, T], "kernel": kernel, "iterations": K,
            "resynchronised": False, "tolerance_on_means": tol,
            "means_rel_err_max_after_K": worst_means if all(tracking) else None,
            "means_rel_err_max_while_tracking": worst_means,
            "particles_within_tolerance_per_iteration": per_iter,
            "cost_rel_err_max_while_tracking": worst_cost,
            "departures_on_near_ties": departures,       # arg-min flips between two samples whose oracle costs differ by < 2e-5
            "unexplained_departures": unexplained,
            "ok": (not unexplained) and worst_cost < 5e-3 and (worst_means < tol),
            "leg_seconds": time.perf_counter() - t0}


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
    t_start = time.perf_counter()

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
    shard_of = tuple(int(v) for v in args.shard_of.split(",")) if args.shard_of and world == 1 else None
    pl, obs, name = build_planner(torch, args.workload, P_local, S, T, dtype, dev, rank, world,
                                  field=args.field, spheres=args.spheres, goals=goals, shard_of=shard_of,
                                  force_stats_allreduce=use_dist and world == 1, store_free=bool(args.store_free))
    w = 4 if dtype == torch.float32 else 8
    d = pl.d_state_opt

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Pass 1: per-kernel device times (HIP events between the kernels; feeds `roofline.launch_ms`).  It runs first so
    # that the wall-clock pass below starts on a GPU that is already at its working clock: with a cold
    # device the first ~100 iterations run 10-20 % slow, which a 5-step warm-up does not cover.
    # The context's two chain streams (optimize(opt_iters >= 2)) need the same: in a fresh process their first
    # ~150 iterations run up to 25 % slow (tools/first_calls_probe.py), whatever the call sizes.
    for _ in range(2):
        pl.optimize(opt_iters=100, **obs)                 # (device + chain-stream warm-up)
    kms = kernel_profile(torch, pl, obs, 100, unread=bool(args.store_free))
    # Pass 2: W untimed warm-up steps, then EXACTLY K timed steps between barriers (the reported value)
    split0 = pl._engine.pipeline_split_steps()
    t_timed = time.perf_counter()
    elapsed = time_loop(torch, pl, obs, args.steps, args.warmup, barrier, one_call=not args.single_iteration_calls)
    split_steps = pl._engine.pipeline_split_steps() - split0
    # Pass 3 (reported beside it): the same K iterations as K optimize(opt_iters=1) calls -- the loop of the reference's
    # example scripts (--store-free: skipped -- single-iteration calls always store)
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
        is_headline = args.workload == "panda" and (P_local, S, T, args.dtype, args.field, args.spheres, goals) == \
            (1024, 128, 64, "f32", "rbf", 5, 1)
        cfg_key = "cfg3" if is_headline else ("cfg2" if (args.workload, P_local, S, T) == ("planar", 256, 64, 128) else
                                              "cfg5" if (args.workload, P_local, S, T, goals) == ("panda", 512, 256, 128, 4) else
                                              "cfg3_f64" if (args.workload, P_local, S, T, args.dtype, goals) == ("panda", 1024, 128, 64, "f64", 1) else "")
        if args.workload == "panda" and cfg_key == "cfg3" and args.field == "sdf":
            cfg_key = "cfg3_sdf"
        if cfg_key and sweep_kernel == "fused_step_f64_mixed_kernel":
            cfg_key += "_mixed"
        if cfg_key and not fused and args.workload == "panda":
            cfg_key += "_unfused"
        ms_step = 1e3 * elapsed / args.steps
        launches_per_iteration = pl._engine.last_step_launches()
        roof, roof_detail = roofline_of(sweep_kernel, kms["cost_sweep"], N_elems, w, P_local * S * 8, fused,
                                        cfg_key + ("_store_free" if args.store_free and cfg_key else ""), step_ms=ms_step)
        # bytes the iteration really moves: K4 reads only the sample rows whose softmax weight is not exactly zero
        nnz_rows = int((pl._weights_buf != 0).sum())
        iter_alg = 3 * N_elems * w + 2 * P_local * T * d * w + 2 * P_local * S * 8
        iter_moved = ((1 if fused else 2) * N_elems * w + nnz_rows * T * d * w + 4 * P_local * T * d * w
                      + 3 * P_local * S * 8)
        # the stand-alone sampler and sweep of the same planner (north_star's 40 % clause is about THAT launch)
        alone = None
        if world == 1 and fused and not args.no_sweep_alone and not args.store_free:
            alone = sweep_alone_leg(torch, pl, obs, N_elems, w, P_local * S * 8, config_key=cfg_key)
        # the store-free mode of the same workload (the product's default inside one optimize() call), beside the headline
        sfree = None
        spec = dict(workload=args.workload, P_local=P_local, S=S, T=T, dtype=dtype, field=args.field, spheres=args.spheres,
                    goals=goals, shard_of=shard_of)
        if world == 1 and not args.store_free and not args.no_store_free and fused:
            sfree = store_free_leg(torch, spec, dev, cfg_key, args.steps, ms_step)
        # free-running K = 10 parity of the benchmarked workload, measured by this run (rank 0 at N = 1, like cpu_baseline)
        parity = None
        if world == 1 and not args.no_parity:
            parity = parity_leg(torch, args, dev, P_local, S, T, goals)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, torch, S, T, P_local)
        value = world * args.steps / elapsed
        full = {
            "metric": "planner iterations/sec (and ms/iter) at fixed particles x samples x T",
            # whole-job aggregate: every rank advances its 1024-particle shard by one iteration per step (weak
            # scaling); at N = 1 this is plain planner iterations/s of the BASELINE config
            "value": value,
            "unit": "iterations/s" if world == 1 else f"iterations/s of a {P_local}-particle shard, summed over {world} shards",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if dtype == torch.float32 else "f64", "data": "synthetic",
            "field": args.field if args.workload == "panda" else "occupancy grid",
            "config": {"workload": name, "particles_per_gpu": P_local, "particles_total": P_local * world,
                       "samples": S, "traj_len": T, "state_dim": d,
                       "parallelism": f"particle-sharded x{world}, RCCL stats all-reduce in sgpmp_step" if world > 1 else "single GPU",
                       "noise": f"philox4x32-{pl._engine.lib.sgpmp_philox_rounds()} + Box-Muller (in-kernel)",
                       "prior_factor_dtype": "f64"},
            "roofline": roof, "roofline_detail": roof_detail,
            "cpu_baseline": cpu,
            "speedup_vs_cpu_baseline": value / cpu["value"] if cpu else None,
            "single_iteration_calls": {"iterations_per_s": world * args.steps / elapsed_calls,
                                       "ms_per_step": 1e3 * elapsed_calls / args.steps} if elapsed_calls else None,
            "mode": "store_free" if args.store_free else "storing",
            "store_free": sfree,
            "sweep_alone": alone,
            "parity": parity,
            "rccl": {"ranks": comm_world, "rank": comm_rank, "version": rccl_version, "library": comm_lib,
                     "test_hooks_build": bool(comm_hooks)},
            "shared_gpu_test_double": shared_gpu or bool(comm_hooks and "rccl" not in comm_lib),
            "per_rank_iterations_per_s": None if rank_rates is None else
            {"min": min(rank_rates), "max": max(rank_rates), "all": rank_rates},
            "kernel_ms_per_step": kms,
            "launches_per_iteration": launches_per_iteration,
            "loop": {"call": "optimize(opt_iters=1) x K" if args.single_iteration_calls else "optimize(opt_iters=K)",
                     "steps_run_as_two_particle_half_chains": split_steps},
            "iteration_roofline": {"algorithmic_bytes": iter_alg, "moved_bytes": iter_moved, "k4_rows_read": nnz_rows,
                                   "moved_GBs": iter_moved / (ms_step * 1e-3) / 1e9,
                                   "moved_frac_of_hbm_peak": iter_moved / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "last_iteration": {"mean_cost_sum": mean_cost, "mean_min_cost": mean_min_cost},
            "timed_region_started_s": t_timed - t_start,
        }
        if world == 1 and not args.no_other_configs and is_headline:
            del pl
            torch.cuda.empty_cache()
            full["other_configs"] = other_configs(torch, dev)
        full["wall_seconds"] = time.perf_counter() - t_start
        emit(full, json_fd, args.detail if world == 1 or not shared_gpu else None)
    # the context (and its RCCL communicator) goes before torch's process group does
    pl = None
    import gc
    gc.collect()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
