import sys, time; sys.path.insert(0,'.')
import torch
from stoch_gpmp_amd import workloads as W
ta={"device":torch.device("cuda:0"),"dtype":torch.float32}
pl=W.hip_panda_planner(W.PANDA,64,1024,128,ta,seed=0)
sph=torch.as_tensor(W.panda_spheres()).to(**ta).reshape(-1,4).contiguous()
eng=pl._engine
for stats in (pl._stats[0], None):
    for _ in range(5): eng.update(pl._costs, pl.state_samples, pl.particle_means, 1.0, 0.1, weights=pl._weights_buf, grad=pl._grad, means_prev=pl._means_prev, stats=stats)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): eng.update(pl._costs, pl.state_samples, pl.particle_means, 1.0, 0.1, weights=pl._weights_buf, grad=pl._grad, means_prev=pl._means_prev, stats=stats)
    torch.cuda.synchronize(); print('stats' if stats is not None else 'nostats', (time.perf_counter()-t0)/200*1e6,'us')
