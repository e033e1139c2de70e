"""Cost of the per-iteration statistics all-reduce at one rank (1-rank RCCL group on one GPU):
off   -- no communicator
rccl  -- the C-ABI path: sgpmp_step enqueues ncclAllReduce on the context's side stream (chained by an event recorded
         behind the update kernel; rccl/packet-event: by the update kernel's own stop event, round 2's default)
torch -- collective='torch': torch.distributed.all_reduce(async_op=True) from Python (round-1 path)
rccl+modes -- rccl, plus the per-goal mean statistics every step (mode_stats=True: update-kernel snapshot, per-goal
         reduction and a second ncclAllReduce of [G][T d + 1][2] doubles on the side stream)"""
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


def build(mode):
    kw = {} if mode == "off" else dict(force_stats_allreduce=True, collective=mode.split("+")[0],
                                       mode_stats=mode.endswith("+modes"))
    pl = W.hip_panda_planner(W.PANDA, 64, 1024, 128, ta, seed=0, rank=0, world_size=1, **kw)
    assert pl._comm_attached == mode.startswith("rccl")
    for _ in range(30):
        pl.optimize(obstacle_spheres=sph)
    return pl


def timeit(pl, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        pl.optimize(obstacle_spheres=sph)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


pls = {m: build(m) for m in ("off", "rccl", "torch", "rccl+modes")}
pls["rccl/packet-event"] = build("rccl")
pls["rccl/packet-event"]._engine.set_option("comm_packet_event", 1)
res = {m: [] for m in pls}
for rnd in range(5):                       # interleaved rounds in one process
    for m, pl in pls.items():
        res[m].append(timeit(pl))
base = min(res["off"])
for m, v in res.items():
    print(f"{m:16s}: min {min(v)*1e6:7.1f} us  median {sorted(v)[len(v)//2]*1e6:7.1f} us per iteration "
          f"(+{(min(v)-base)*1e6:5.1f} us vs off)")
dist.destroy_process_group()
