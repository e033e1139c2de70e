"""World-size-2 checks of the multi-GPU plumbing on CPU tensors with the gloo backend: the same
functions (stoch_gpmp_amd/dist.py) the planner calls with RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, P, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stoch_gpmp_amd.dist import allgather_means, allreduce_stats_async, shard_range
        T, d = 5, 4
        full = torch.arange(P * T * d, dtype=torch.float64).reshape(P, T, d)
        p0, p1 = shard_range(P, rank, world)
        gathered = allgather_means(full[p0:p1].clone(), P, world)
        ok_gather = torch.equal(gathered, full)
        # per-iteration statistics: each rank contributes its shard's sums
        costs = torch.arange(P * 3, dtype=torch.float64).reshape(P, 3) + 1.0
        local = costs[p0:p1]
        stats = torch.tensor([float(local.sum()), float(local.min(1)[0].sum()), float(p1 - p0), 0.0],
                             dtype=torch.float64)
        work = allreduce_stats_async(stats)
        work.wait()
        expect = torch.tensor([float(costs.sum()), float(costs.min(1)[0].sum()), float(P), 0.0],
                              dtype=torch.float64)
        # per-goal mean statistics (the all-reduce's second payload): local sums by goal -> summed over ranks
        from stoch_gpmp_amd.dist import allreduce_mode_sums, mode_moments
        G, nppg = 2, P // 2
        M = T * d
        means = torch.sin(torch.arange(P * M, dtype=torch.float64)).reshape(P, T, d)
        buf = torch.zeros(G, M + 1, 2, dtype=torch.float64)
        for p in range(p0, p1):
            g = min(p // nppg, G - 1)
            buf[g, :M, 0] += means[p].reshape(-1)
            buf[g, :M, 1] += means[p].reshape(-1) ** 2
            buf[g, M, 0] += 1
        allreduce_mode_sums(buf)
        mean, var, cnt = mode_moments(buf, T, d)
        by_goal = [means[[p for p in range(P) if min(p // nppg, G - 1) == g]] for g in range(G)]
        ok_modes = all(torch.allclose(mean[g], by_goal[g].mean(0), atol=1e-13) and
                       torch.allclose(var[g], by_goal[g].var(0, unbiased=False), atol=1e-13) and
                       float(cnt[g]) == float(len(by_goal[g])) for g in range(G))
        q.put((rank, ok_gather, torch.equal(stats, expect) and ok_modes))
    finally:
        dist.destroy_process_group()


def _run(P):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, P, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_gather, ok_stats in res:
        assert ok_gather, f"rank {rank}: all-gather of particle means wrong"
        assert ok_stats, f"rank {rank}: statistics all-reduce wrong"


def test_world2_even_shards():
    _run(8)


def test_world2_ragged_shards():
    _run(7)
