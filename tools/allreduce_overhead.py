"""Where the ~5 % of the per-iteration statistics all-reduce goes (1-rank RCCL group on one GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from stoch_gpmp_amd import workloads as W
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
sph = torch.as_tensor(W.panda_spheres()).to(**ta)
def run(mode):
    pl = W.hip_panda_planner(W.PANDA, 64, 1024, 128, ta, seed=0, rank=0, world_size=1, force_stats_allreduce=(mode != "off"))
    if mode == "nowait":
        orig = pl._reduce_stats
        def rs(slot):
            pl._pending_reduce.append(dist.all_reduce(pl._stats[slot], async_op=True))
            if len(pl._pending_reduce) > 64: pl._pending_reduce = pl._pending_reduce[-2:]
        pl._reduce_stats = rs
    for _ in range(30): pl.optimize(obstacle_spheres=sph)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): pl.optimize(obstacle_spheres=sph)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print(f"{mode:7s}: {dt*1e6:.1f} us per iteration ({1/dt:.0f} it/s)")
for m in ("off", "on", "nowait", "off", "on"):
    run(m)
dist.destroy_process_group()
