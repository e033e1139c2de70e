"""Free-running fp32 parity over MANY iterations (the suite follows 10): config 3 at full size, four particles, K iterations
without resynchronisation against the dense fp64 oracle.  python3 tools/free_run_long.py [K] on the GPU box; prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from tests import scenarios as SC
from tests.hip_builders import hip_panda_planner
from tests.test_gpu_planner import _free_run, F32
from oracle.native_noise import native_eps
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, S, P, seed, n = 64, 128, 1024, 47, 7
sph = torch.as_tensor(SC.panda_spheres(num=5))
pl = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=seed)
sub = [0, 5, 511, 1023]
ora = SC.oracle_panda_planner(SC.PANDA, T, len(sub), S, seed=seed, eps_init=torch.zeros(len(sub), 1, T * 2 * n, dtype=torch.float64))
def set_means(mu):
    ora.particle_means.copy_(mu)
    ora.prior.set_mean(ora.particle_means.view(len(sub), -1))
def step(gidx, draw):
    eps = torch.from_numpy(native_eps(seed, draw, gidx, S, T, n, "float32")).double()
    costs, _ = ora.step(eps=eps, obstacle_spheres=sph)
    return costs, ora.particle_means.clone()
rec = _free_run(f"config 3: Panda 1024 x 128 x 64 fp32 (fused launch), {K} free iterations", pl, sub, set_means, step, K,
                {"obstacle_spheres": sph.to(**F32)}, "fused_step_kernel")
print(json.dumps(rec))
