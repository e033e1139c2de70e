"""Fused launch vs sampler + chunked sweep vs sampler + two-trajectory sweep on one trajectory of means: at this small
size the two-launch paths take sample_iso_small_kernel, whose samples differ from the fused launch's in the last bit
(one fused multiply-add ordered differently), hence costs differ by ~1.5e-6 relative -- enough to flip a near-tie."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tests import scenarios as SC
from tests.hip_builders import hip_panda_planner
F32 = {"device": torch.device("cuda:0"), "dtype": torch.float32}
sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
a = hip_panda_planner(SC.PANDA, 32, 48, 32, F32, seed=5)
b = hip_panda_planner(SC.PANDA, 32, 48, 32, F32, seed=5)
b._engine.set_option("no_fused_step", 1)
c = hip_panda_planner(SC.PANDA, 32, 48, 32, F32, seed=5)
c._engine.set_option("no_fused_step", 1); c._engine.set_option("no_chunked_sweep", 1)
for it in range(8):
    for p in (a, b, c): p.optimize(obstacle_spheres=sph)
    print(it, a._engine.last_cost_kernel(), b._engine.last_cost_kernel(), c._engine.last_cost_kernel(),
          "samples a==b", torch.equal(a.state_samples, b.state_samples),
          "costs a==b", torch.equal(a._costs, b._costs), float((a._costs - b._costs).abs().max() / a._costs.abs().max()),
          "a==c", torch.equal(a._costs, c._costs), float((a._costs - c._costs).abs().max() / a._costs.abs().max()),
          "means a==b", torch.equal(a.particle_means, b.particle_means), "argmin same", bool((a._costs.argmin(1) == b._costs.argmin(1)).all()))
    b.particle_means.copy_(a.particle_means); c.particle_means.copy_(a.particle_means)
