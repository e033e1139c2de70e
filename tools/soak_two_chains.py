"""Soak: thousands of two-chain iterations (with and without the 1-rank RCCL statistics all-reduce) must end with
the bits of the single-chain twin -- a race between the chains, the ring slots or the join would show up here."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests import scenarios as SC  # noqa: E402
from tests.hip_builders import hip_panda_planner  # noqa: E402

dev = torch.device("cuda", 0)
F32 = {"device": dev, "dtype": torch.float32}
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for comm in (False, True):
    a = hip_panda_planner(SC.PANDA, 32, 128, 128, F32, seed=9, force_stats_allreduce=comm)
    b = hip_panda_planner(SC.PANDA, 32, 128, 128, F32, seed=9, force_stats_allreduce=comm, pipeline_steps=False)
    done = 0
    for k in (1500, 7, 1, 2, N - 1510):
        a.optimize(opt_iters=k, obstacle_spheres=sph)
        b.optimize(opt_iters=k, obstacle_spheres=sph)
        done += k
        torch.cuda.synchronize()
        ok = torch.equal(a.particle_means, b.particle_means) and torch.equal(a._costs, b._costs)
        sa, sb = a.global_stats(), b.global_stats()
        ok = ok and abs(sa[0] / sb[0] - 1) < 1e-12
        print(f"comm={comm} after {done} iterations: {'identical' if ok else 'DIFFERENT'}; split steps {a._engine.pipeline_split_steps()}",
              flush=True)
        assert ok
dist.destroy_process_group()
print("SOAK_OK")
