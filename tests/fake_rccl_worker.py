"""One rank of the N-ranks-on-ONE-GPU check (launched by tests/test_gpu_dist.py): every rank uses cuda:0, the process
group is gloo (it only carries the communicator id) and libsgpmp.so binds tests/fake_rccl/libfakerccl.so instead of
RCCL (SGPMP_RCCL_LIB) -- a stream-ordered shared-memory all-reduce / all-gather.  What is under test is the LIBRARY's
side of the N > 1 protocol, which has never met a second rank on the 1-GPU boxes: statistics ring slots and their
events, two-chain steps feeding one all-reduce, empty shards, the per-goal mean statistics, the all-gather."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import scenarios as SC
    from tests.hip_builders import hip_panda_planner
    ta = {"device": dev, "dtype": torch.float32}
    n = 7
    goals = [SC.PANDA["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n]
    P, S, T = 32 * world, 32, 32                                               # nppg = P / 2 per goal
    sph = torch.as_tensor(SC.panda_spheres()).to(**ta)
    full = hip_panda_planner(SC.PANDA, T, P // 2, S, ta, seed=4, goals=goals)                       # unsharded, same GPU
    shard = hip_panda_planner(SC.PANDA, T, P // 2, S, ta, seed=4, goals=goals, rank=rank, world_size=world,
                              mode_stats=True)
    assert shard._comm_attached, "communicator not attached"
    info = shard._engine.comm_info()
    assert info[:2] == (world, rank) and info[2] == 1, info                     # (version 1 = the test double answered)
    assert torch.equal(shard.particle_means, full.particle_means[shard.p0:shard.p1])
    if os.environ.get("FAKE_RCCL_NEGATIVE"):
        # the double must NOTICE ranks that disagree about a collective (else this whole test proves nothing)
        buf = torch.zeros(8 if rank == 0 else 16, device=dev, dtype=torch.float64)
        shard._engine.allreduce_f64(buf)
        torch.cuda.synchronize()
        raise SystemExit("the test double let mismatched collectives through")
    for it in range(12):                                                        # more steps than ring slots
        full.optimize(obstacle_spheres=sph)
        shard.optimize(obstacle_spheres=sph)
        if it % 4 == 3:
            gs, gf = shard.global_stats(), full.global_stats()
            assert abs(gs[0] / gf[0] - 1) < 1e-12 and abs(gs[1] / gf[1] - 1) < 1e-12, (it, gs, gf)
            ms, mf = shard.global_mode_stats(), full.global_mode_stats()
            for a, b in zip(ms, mf):
                assert float((a - b).abs().max()) <= 1e-12 * max(float(b.abs().max()), 1.0), it
    assert torch.equal(shard.particle_means, full.particle_means[shard.p0:shard.p1]), "means differ"
    assert torch.equal(shard._costs, full._costs[shard.p0:shard.p1]), "costs differ"
    allm = shard.gather_particle_means()
    assert torch.equal(allm, full.particle_means), "all-gathered means differ"
    # several iterations in one call: two particle-half chains per rank, one all-reduce per step from both halves
    P2 = 128 * world
    full2 = hip_panda_planner(SC.PANDA, T, P2, 128, ta, seed=6, pipeline_steps=False)
    shard2 = hip_panda_planner(SC.PANDA, T, P2, 128, ta, seed=6, rank=rank, world_size=world)
    for k in (10, 1, 9, 2, 3):
        if k == 2:                                                              # the update kernel's own stop event as the hand-over
            shard2._engine.set_option("comm_packet_event", 1)
        if k == 3:
            shard2._engine.set_option("comm_packet_event", 0)
        full2.optimize(opt_iters=k, obstacle_spheres=sph)
        shard2.optimize(opt_iters=k, obstacle_spheres=sph)
        gs2, gf2 = shard2.global_stats(), full2.global_stats()
        assert abs(gs2[0] / gf2[0] - 1) < 1e-12 and abs(gs2[1] / gf2[1] - 1) < 1e-12, (k, gs2, gf2)
    assert shard2._engine.pipeline_split_steps() >= 19
    assert torch.equal(shard2.particle_means, full2.particle_means[shard2.p0:shard2.p1]), "means differ (two chains)"
    m_on_demand = shard2.global_mode_stats()                                    # computed and all-reduced on demand
    m_full = full2.global_mode_stats()
    for a, b in zip(m_on_demand, m_full):
        assert float((a - b).abs().max()) <= 1e-12 * max(float(b.abs().max()), 1.0)
    # more ranks than particles: the rank with the empty shard still joins every collective
    P3 = world - 1
    full3 = hip_panda_planner(SC.PANDA, T, P3, S, ta, seed=9)
    shard3 = hip_panda_planner(SC.PANDA, T, P3, S, ta, seed=9, rank=rank, world_size=world, mode_stats=True)
    assert shard3._comm_attached and (shard3.num_particles_local == 0) == (rank == world - 1)
    for k in (1, 3):
        full3.optimize(opt_iters=k, obstacle_spheres=sph)
        shard3.optimize(opt_iters=k, obstacle_spheres=sph)
    gs3, gf3 = shard3.global_stats(), full3.global_stats()
    assert abs(gs3[0] / gf3[0] - 1) < 1e-12 and abs(gs3[1] / gf3[1] - 1) < 1e-12, (gs3, gf3)
    c3, cf3 = shard3.global_mode_stats()[2], full3.global_mode_stats()[2]
    assert torch.equal(c3.cpu(), cf3.cpu())
    shard3.reset()
    torch.cuda.synchronize()
    dist.barrier()
    del shard, shard2, shard3
    if rank == 0:
        print(f"FAKE_RCCL_OK world={world}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
