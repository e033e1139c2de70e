"""One rank of the 2-GPU parity check (launched by tests/test_gpu_dist.py through
`python -m torch.distributed.run`): a particle-sharded StochGPMP over RCCL must reproduce, bit for
bit, the slice of the unsharded planner that this rank owns, and its all-reduced statistics must be
the unsharded run's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from tests import scenarios as SC
    from tests.hip_builders import hip_panda_planner
    ta = {"device": dev, "dtype": torch.float32}
    P, S, T, iters = 64 * world, 32, 32, 3
    sph = torch.as_tensor(SC.panda_spheres()).to(**ta)
    full = hip_panda_planner(SC.PANDA, T, P, S, ta, seed=4)                     # unsharded, on this GPU
    shard = hip_panda_planner(SC.PANDA, T, P, S, ta, seed=4, rank=rank, world_size=world)
    assert shard._comm_attached, "RCCL communicator not attached"
    assert shard._engine.comm_info()[:2] == (world, rank), shard._engine.comm_info()   # RCCL's own count / rank
    assert torch.equal(shard.particle_means, full.particle_means[shard.p0:shard.p1])
    for _ in range(iters):
        full.optimize(obstacle_spheres=sph)
        shard.optimize(obstacle_spheres=sph)
    assert torch.equal(shard.particle_means, full.particle_means[shard.p0:shard.p1]), "means differ"
    assert torch.equal(shard._costs, full._costs[shard.p0:shard.p1]), "costs differ"
    gs, gf = shard.global_stats(), full.global_stats()
    assert abs(gs[0] / gf[0] - 1) < 1e-12 and abs(gs[1] / gf[1] - 1) < 1e-12, (gs, gf)
    allm = shard.gather_particle_means()
    assert torch.equal(allm, full.particle_means), "all-gathered means differ"
    # several iterations in one call: two particle-half chains per rank, statistics all-reduced per step from both
    # blocks of the ring slot (big enough shards for the split: 128 particles x 128 samples per rank)
    P2 = 128 * world
    full2 = hip_panda_planner(SC.PANDA, T, P2, 128, ta, seed=6, pipeline_steps=False)
    shard2 = hip_panda_planner(SC.PANDA, T, P2, 128, ta, seed=6, rank=rank, world_size=world)
    full2.optimize(opt_iters=10, obstacle_spheres=sph)
    shard2.optimize(opt_iters=10, obstacle_spheres=sph)
    assert shard2._engine.pipeline_split_steps() == 10
    assert torch.equal(shard2.particle_means, full2.particle_means[shard2.p0:shard2.p1]), "means differ (two chains)"
    gs2, gf2 = shard2.global_stats(), full2.global_stats()
    assert abs(gs2[0] / gf2[0] - 1) < 1e-12 and abs(gs2[1] / gf2[1] - 1) < 1e-12, (gs2, gf2)
    # more ranks than particles: the ranks with an empty shard still join every step's statistics all-reduce
    # (a collective) -- the others would hang in global_stats() / reset() otherwise
    P3 = world - 1
    full3 = hip_panda_planner(SC.PANDA, T, P3, S, ta, seed=9)
    shard3 = hip_panda_planner(SC.PANDA, T, P3, S, ta, seed=9, rank=rank, world_size=world)
    assert shard3._comm_attached and (shard3.num_particles_local == 0) == (rank == world - 1)
    for k in (1, 3):
        full3.optimize(opt_iters=k, obstacle_spheres=sph)
        shard3.optimize(opt_iters=k, obstacle_spheres=sph)
    gs3, gf3 = shard3.global_stats(), full3.global_stats()
    assert abs(gs3[0] / gf3[0] - 1) < 1e-12 and abs(gs3[1] / gf3[1] - 1) < 1e-12, (gs3, gf3)
    assert torch.equal(shard3.particle_means, full3.particle_means[shard3.p0:shard3.p1])
    shard3.reset()
    dist.barrier()
    if rank == 0:
        print(f"DIST_OK world={world} stats={gs}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
