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
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=DEV)
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
