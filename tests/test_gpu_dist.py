"""The multi-GPU path on real devices (`-m gpu`): the RCCL collectives behind the C ABI with one rank
(always runnable on the 1-GPU box) and with two processes when two GPUs are visible; plus bench.py's
own N > 1 launch.  The world-size-2 logic is also covered on CPU with gloo (tests/test_cpu_dist.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests import scenarios as SC
from tests.hip_builders import hip_panda_planner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda:0")
F32 = {"device": DEV, "dtype": torch.float32}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def one_rank_group():
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (a port that was free a moment ago can be taken by the time the store listens on it -- EADDRINUSE once in this build's
    # ~2000 fixture set-ups: try another one)
    for attempt in range(5):
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                    device_id=DEV)
            break
        except Exception as e:                            # torch.distributed.DistNetworkError
            if "EADDRINUSE" not in str(e) and "address already in use" not in str(e) or attempt == 4:
                raise
    yield dist
    dist.destroy_process_group()


def test_one_rank_rccl_statistics_allreduce_inside_the_step(one_rank_group):
    """force_stats_allreduce=True attaches an RCCL communicator (sgpmp_comm_init) on a 1-rank group:
    every sgpmp_step then enqueues ncclAllReduce of its statistics on the side stream.  The all-reduced
    statistics must equal the locally summed ones and the all-gather must be the identity."""
    T, P, S = 32, 48, 32
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    pl = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=2, force_stats_allreduce=True)
    assert pl._comm_attached and pl._engine.comm_info()[:2] == (1, 0)      # what RCCL itself reports
    ref = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=2)                    # no communicator
    assert not ref._comm_attached
    for it in range(5):
        _, _, _, _, costs, _ = pl.optimize(obstacle_spheres=sph)
        ref.optimize(obstacle_spheres=sph)
        mean_sum, mean_min = pl.global_stats()
        c = costs.double()
        assert abs(mean_sum / float(c.sum(1).mean()) - 1) < 1e-6
        assert abs(mean_min / float(c.min(1)[0].mean()) - 1) < 1e-6
        assert pl.global_stats() == ref.global_stats()
    assert torch.equal(pl.particle_means, ref.particle_means)
    assert pl.gather_particle_means() is pl.particle_means                     # world_size == 1: identity
    # the stand-alone entry points
    eng = pl._engine
    stats = torch.arange(256, device=DEV, dtype=torch.float64).reshape(64, 4).contiguous()
    want = stats.clone()
    eng.allreduce_stats(stats)
    eng.stats_wait(stats)
    torch.cuda.synchronize()
    assert torch.equal(stats, want)
    out = eng.allgather_means(pl.particle_means, 1)
    torch.cuda.synchronize()
    assert torch.equal(out, pl.particle_means)
    # reset() keeps the context and its communicator
    pl.reset()
    assert pl._comm_attached
    pl.optimize(obstacle_spheres=sph)
    assert pl.global_stats()[0] > 0


def test_empty_shard_still_joins_the_statistics_allreduce(one_rank_group):
    """A context with zero particles (a rank of a world bigger than the particle count) has no kernels to run, but
    sgpmp_step must still take part in the per-step all-reduce: it contributes a zeroed ring slot.  With one rank the
    all-reduced statistics are then exactly zero, and nothing hangs (stats_wait, reset, destroy)."""
    from stoch_gpmp_amd.engine import Engine
    eng = Engine(7, 16, 0, 8, 1, 4, particle_offset=4, num_particles_global=4, tensor_args=F32)
    eng.comm_init(eng.comm_unique_id(), 1, 0)
    assert eng.comm_info()[:2] == (1, 0) and eng.comm_info()[2] > 0
    stats = torch.full((64, 4), 7.0, device=DEV, dtype=torch.float64)
    empty = torch.empty(0, 16, 14, **F32)
    for draw in range(10):                                # (more steps than ring slots)
        eng.step(0, draw, empty, torch.empty(0, 8, 16, 14, **F32), 1.0, 0.1, stats=stats)
    eng.stats_wait(stats)
    torch.cuda.synchronize()
    assert float(stats.abs().sum()) == 0.0
    eng.close()


def _direct_mode_moments(means, G):
    m = means.double().reshape(G, -1, means.shape[-2], means.shape[-1])
    return m.mean(1), m.var(1, unbiased=False)


def test_mode_statistics_of_eight_shards_sum_to_the_unsharded_run():
    """The all-reduce's north_star content: per-goal sums and sums of squares of the particle means
    (sgpmp_mode_stats).  Two goals x 64 particles: global_mode_stats() of the unsharded planner equals torch's moments
    of its means; the eight `rank = r, world_size = 8` shards' local sums add up to the unsharded sums (what
    ncclAllReduce computes) to rounding -- shard boundaries fall inside a goal and on a goal boundary."""
    T, nppg, S, n = 32, 64, 16, 7
    goals = [SC.PANDA["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n]
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    full = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=12, goals=goals)
    for _ in range(3):
        full.optimize(obstacle_spheres=sph)
    mean, var, cnt = full.global_mode_stats()
    dm, dv = _direct_mode_moments(full.particle_means, 2)
    assert cnt.tolist() == [64., 64.]
    assert float((mean - dm).abs().max()) < 1e-12 * float(dm.abs().max())
    assert float((var - dv).abs().max()) < 1e-9 * float(dv.abs().max()) + 1e-18
    total = torch.zeros(2, T * 14 + 1, 2, device=DEV, dtype=torch.float64)
    for r in range(8):
        sh = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=12, goals=goals, rank=r, world_size=8)
        sh.particle_means.copy_(full.particle_means[sh.p0:sh.p1])
        total += sh._engine.mode_stats(sh.particle_means)
    ref = full._engine.mode_stats(full.particle_means)
    assert float(((total - ref).abs() / ref.abs().clamp(min=1e-30)).max()) < 1e-12
    assert total[:, -1, 0].tolist() == [64., 64.]


def test_mode_statistics_every_step_through_the_rccl_group(one_rank_group):
    """mode_stats=True: every sgpmp_step leaves the per-goal mean statistics -- update kernel snapshot, per-goal
    reduction and ncclAllReduce on the side stream -- and they equal the on-demand ones after every call, single
    iterations and several per call (more calls than snapshot slots)."""
    T, nppg, S, n = 32, 48, 32, 7
    goals = [SC.PANDA["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n]
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=14, goals=goals, force_stats_allreduce=True, mode_stats=True)
    b = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=14, goals=goals, force_stats_allreduce=True)
    c = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=14, goals=goals, mode_stats=True)          # no communicator
    assert a._comm_attached and b._comm_attached and not c._comm_attached
    for k in (1, 1, 4, 1, 3):
        for pl in (a, b, c):
            pl.optimize(opt_iters=k, obstacle_spheres=sph)
        assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a.particle_means, c.particle_means)
        ma, va, ca = a.global_mode_stats()
        mb, vb, cb = b.global_mode_stats()
        mc, vc, cc = c.global_mode_stats()
        dm, dv = _direct_mode_moments(a.particle_means, 2)
        for m_, v_, c_ in ((ma, va, ca), (mb, vb, cb), (mc, vc, cc)):
            assert c_.tolist() == [48., 48.]
            assert float((m_ - dm).abs().max()) < 1e-12 * float(dm.abs().max())
            assert float((v_ - dv).abs().max()) < 1e-9 * float(dv.abs().max()) + 1e-18
        assert a.global_stats() == b.global_stats()          # the cost statistics keep working beside them
    a.reset()
    a.optimize(obstacle_spheres=sph)
    assert a.global_mode_stats()[2].tolist() == [48., 48.]


def test_mode_statistics_describe_the_current_means_not_the_last_step():
    """mode_stats=True keeps the statistics the LAST step left.  Before the first step, after reset() and after the caller
    edited particle_means they describe other means than the current ones (round-3 advisor finding): global_mode_stats()
    must then answer from the current means, as the mode_stats=False path does."""
    T, nppg, S, n = 32, 24, 16, 7
    goals = [SC.PANDA["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n]
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=15, goals=goals, mode_stats=True)

    def check():
        m, v, c = a.global_mode_stats()
        dm, dv = _direct_mode_moments(a.particle_means, 2)
        assert c.tolist() == [float(nppg)] * 2
        assert float((m - dm).abs().max()) < 1e-12 * float(dm.abs().max())
        assert float((v - dv).abs().max()) < 1e-9 * float(dv.abs().max()) + 1e-18
    check()                                                  # before the first step (round 3: count 0, mean 0)
    a.optimize(opt_iters=2, obstacle_spheres=sph)
    check()                                                  # the per-step buffer
    a.particle_means.mul_(1.01)                              # the caller edits the means
    check()
    a.optimize(obstacle_spheres=sph)
    check()
    a.reset()                                                # new initial means, same engine
    check()


def test_two_chain_steps_with_the_rccl_statistics_allreduce(one_rank_group):
    """optimize(opt_iters=K) with a communicator attached: the iterations run as two particle-half chains, each
    accumulating into its own block of the statistics ring slot; the all-reduce waits for both update kernels
    and adds the blocks.  Same buffers as single-iteration calls, same all-reduced statistics."""
    T, P, S = 32, 128, 128
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    a = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=8, force_stats_allreduce=True)
    b = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=8, force_stats_allreduce=True, pipeline_steps=False)
    assert a._comm_attached and b._comm_attached
    for k in (12, 1, 3):                                 # 12 > the ring of 8 slots: a slot comes round again
        a.optimize(opt_iters=k, obstacle_spheres=sph)
        for _ in range(k):
            b.optimize(opt_iters=1, obstacle_spheres=sph)
        sa, sb = a.global_stats(), b.global_stats()
        assert abs(sa[0] / sb[0] - 1) < 1e-12 and abs(sa[1] / sb[1] - 1) < 1e-12
        c = a._costs.double()
        assert abs(sa[0] / float(c.sum(1).mean()) - 1) < 1e-6
        assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a._costs, b._costs)
    assert a._engine.pipeline_split_steps() == 15 and b._engine.pipeline_split_steps() == 0


def test_torch_collective_fallback_matches(one_rank_group):
    """collective='torch' (the path the gloo CPU tests exercise) on HIP tensors over the nccl backend."""
    T, P, S = 16, 8, 8
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    a = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=2, force_stats_allreduce=True, collective='torch')
    b = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=2, force_stats_allreduce=True)
    assert not a._comm_attached and b._comm_attached
    for _ in range(3):
        a.optimize(obstacle_spheres=sph)
        b.optimize(obstacle_spheres=sph)
    assert a.global_stats() == b.global_stats()


def _test_double():
    """(libfakerccl.so, libsgpmp_testhooks.so): the stand-in for librccl and the ONLY build of the library that will bind
    it -- comm.hip compiled with -DSGPMP_TEST_HOOKS (csrc/Makefile `testhooks`); the product library ignores
    SGPMP_RCCL_LIB."""
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
    hooks = os.path.join(ROOT, "tests", "fake_rccl", "libsgpmp_testhooks.so")
    if not os.path.exists(fake):
        subprocess.run(["make", "-C", os.path.dirname(fake)], check=True, capture_output=True)
    if not os.path.exists(hooks):
        subprocess.run(["make", "-C", os.path.join(ROOT, "stoch_gpmp_amd", "csrc"), "testhooks"], check=True, capture_output=True)
    return fake, hooks


def test_product_library_ignores_the_rccl_override():
    """SGPMP_RCCL_LIB is a TEST hook: only libsgpmp_testhooks.so honours it.  The product library, with the variable
    set to the stand-in, still binds librccl and says so (sgpmp_comm_library)."""
    fake, hooks = _test_double()
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from stoch_gpmp_amd import _lib; lib = _lib.load(); "
            "buf = C.create_string_buffer(128); rc = lib.sgpmp_comm_unique_id(buf); h = C.c_int(-1); "
            "print('BOUND', rc, lib.sgpmp_comm_library(C.byref(h)).decode(), h.value)" % ROOT)
    env = dict(os.environ, SGPMP_RCCL_LIB=fake)
    env.pop("SGPMP_LIB_PATH", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    rc, name, hooked = p.stdout.strip().splitlines()[-1].split()[1:]
    assert rc == "0" and "librccl" in name and "fake" not in name and hooked == "0", p.stdout
    p = subprocess.run([sys.executable, "-c", code], env=dict(env, SGPMP_LIB_PATH=hooks), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    rc, name, hooked = p.stdout.strip().splitlines()[-1].split()[1:]
    assert rc == "0" and name == fake and hooked == "1", p.stdout


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_ranks_sharing_one_gpu_through_the_rccl_test_double(world):
    """The library's N > 1 protocol on a 1-GPU box: `world` processes on cuda:0, libsgpmp.so bound to
    tests/fake_rccl/libfakerccl.so (stream-ordered shared-memory all-reduce / all-gather) through SGPMP_RCCL_LIB, which
    only the test-hooks build of the library (SGPMP_LIB_PATH=tests/fake_rccl/libsgpmp_testhooks.so) honours.  The
    worker checks shards against the unsharded run bit for bit, all-reduced cost and per-goal mean statistics, ring-slot
    reuse, two-chain steps, the all-gather and a rank with an empty shard (world = 3: ragged shards too)."""
    fake, hooks = _test_double()
    env = dict(os.environ, SGPMP_RCCL_LIB=fake, SGPMP_LIB_PATH=hooks)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "fake_rccl_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and f"FAKE_RCCL_OK world={world}" in p.stdout, p.stdout[-2000:] + p.stderr[-6000:]
    if world == 2:                         # and the double is not blind: ranks that disagree about a collective fail
        p = subprocess.run(cmd, env=dict(env, FAKE_RCCL_NEGATIVE="1"), capture_output=True, text=True, timeout=900)
        assert p.returncode != 0 and "different collectives in the same position" in p.stderr, p.stderr[-4000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_process_rccl_shards_equal_the_unsharded_run():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "DIST_OK world=2" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: rc 0, one JSON line, n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
                        "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_bench_multi_rank_code_on_one_gpu_through_the_test_double(gpus):
    """bench.py's N > 1 code (own launcher, barriers, max over ranks, per-rank rates, the communicator's own rank
    count, the whole-job value) with `gpus` ranks sharing cuda:0 through tests/fake_rccl: exactly what the driver runs
    as `bench.py --gpus N`, minus RCCL and the other GPUs.  Not a measurement -- the line says so."""
    fake, hooks = _test_double()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SGPMP_BENCH_SHARED_GPU="1", SGPMP_RCCL_LIB=fake, SGPMP_LIB_PATH=hooks)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "20",
                        "--warmup", "5", "--particles", "128"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-6000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == gpus and line["scaling"] == "weak" and line["shared_gpu_test_double"] is True
    assert line["rccl"]["ranks"] == gpus and line["rccl"]["rank"] == 0 and line["rccl"]["version"] == 1
    rates = line["per_rank_iterations_per_s"]["all"]
    assert len(rates) == gpus and min(rates) > 0
    # whole-job value = shard-iterations/s summed over the ranks at the SLOWEST rank's clock (weak scaling)
    assert abs(line["value"] - gpus * min(rates)) <= 1e-6 * line["value"]
    assert line["config"]["particles_total"] == 128 * gpus and line["config"]["particles_per_gpu"] == 128
    assert line["last_iteration"]["mean_cost_sum"] > 0
    assert line["cpu_baseline"] is None               # (rank 0 at N = 1 only)


def test_bench_single_rank_rccl_smoke():
    """bench.py as a child process with the 1-rank RCCL path forced: the line carries roofline and the
    dispatcher's kernel name (not a Python re-derivation)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SGPMP_BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-other-configs", "--particles", "128"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, lines                       # stdout is the JSON line and nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and "fused_step_kernel" in line["roofline"]["kernel"]
    assert line["roofline"]["frac"] > 0 and line["last_iteration"]["mean_cost_sum"] > 0
