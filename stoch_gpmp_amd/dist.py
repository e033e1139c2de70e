"""Multi-GPU plumbing: one process per GPU, particles sharded in contiguous ranges.

Particles never interact in StochGPMP (every reduction of planner.py:263-275 is over the samples
of one particle), so the data path needs NO collective.  What crosses GPUs:
  * per iteration, an asynchronous all-reduce (RCCL over xGMI; `nccl` backend) of a 4-double
    statistics vector (sum of costs, sum of per-particle min cost, particle count, spare) -- the
    reference's `print_info` statistic (planner.py:668-672) made global;
  * on request, an all-gather of the particle means.
With the `gloo` backend the same functions run on CPU tensors (tests/test_cpu_dist.py).
"""
import torch
import torch.distributed as dist


def shard_range(num_particles, rank, world_size):
    """Contiguous particle range [p0, p1) of `rank` (balanced, remainder to the low ranks)."""
    base, rem = divmod(num_particles, world_size)
    p0 = rank * base + min(rank, rem)
    return p0, p0 + base + (1 if rank < rem else 0)


def allreduce_stats_async(stats, group=None):
    """Start the per-iteration statistics all-reduce; returns the work handle (wait lazily)."""
    return dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_mode_sums(buf, group=None):
    """Sum the per-goal mean statistics [G, M + 1, 2] (local sums, sums of squares, counts: include/sgpmp.h
    sgpmp_mode_stats) over all ranks, in place -- the torch.distributed twin of sgpmp_allreduce_f64."""
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def mode_moments(buf, T, d):
    """[G, T*d + 1, 2] sums -> (mean [G,T,d], variance [G,T,d], particle count [G]) of the particle means of every
    goal: the first two moments of each mode of the trajectory distribution."""
    M = T * d
    cnt = buf[:, M, 0].clamp(min=1.0)
    mean = buf[:, :M, 0] / cnt[:, None]
    var = (buf[:, :M, 1] / cnt[:, None] - mean * mean).clamp(min=0.0)
    G = buf.shape[0]
    return mean.reshape(G, T, d), var.reshape(G, T, d), buf[:, M, 0].clone()


def allgather_means(local_means, num_particles, world_size, group=None):
    """[P_local,T,d] on every rank -> [P,T,d] on every rank (ragged shards allowed)."""
    sizes = [shard_range(num_particles, r, world_size) for r in range(world_size)]
    outs = [torch.empty((b - a,) + tuple(local_means.shape[1:]), dtype=local_means.dtype,
                        device=local_means.device) for a, b in sizes]
    if all((b - a) == (sizes[0][1] - sizes[0][0]) for a, b in sizes):
        dist.all_gather(outs, local_means.contiguous(), group=group)
    else:                       # ragged: broadcast shard by shard
        rank = dist.get_rank(group)
        for r, (a, b) in enumerate(sizes):
            if r == rank:
                outs[r].copy_(local_means)
            dist.broadcast(outs[r], src=dist.get_global_rank(group, r) if group is not None else r,
                           group=group)
    return torch.cat(outs, dim=0)
