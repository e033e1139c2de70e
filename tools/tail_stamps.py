"""Where the in-launch update (csrc/fused_tail.inc) spends its time: SGPMP_TAIL_DEBUG=16 makes the last wave of every
particle write s_memtime deltas at its phase boundaries into weights[p, 100..105] (diagnostic; weights are garbage there)."""
import os, sys
os.environ["SGPMP_TAIL_DEBUG"] = os.environ.get("SGPMP_TAIL_DEBUG", "16")
os.environ["SGPMP_TAIL_UPDATE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
pl = W.hip_panda_planner(W.PANDA, 64, 1024, 128, ta, seed=0)
obs = {"obstacle_spheres": torch.as_tensor(W.panda_spheres(num=5)).to(**ta)}
for _ in range(30):
    pl.optimize(**obs)
torch.cuda.synchronize()
assert pl._engine.last_step_launches() == 1
st = pl._weights_buf[:, 100:106].double().cpu()
names = ["RT1 costs/Qinv", "softmax+compaction", "RT2 rows+gather", "stores+LDS", "isw", "drain (counter)"]
prev = torch.zeros(st.shape[0], dtype=torch.float64)
print(f"tail total: mean {st[:, 5].mean():.0f} ticks, median {st[:, 5].median():.0f}, max {st[:, 5].max():.0f}  (s_memtime ticks = shader cycles, ~2.2 GHz under this load)")
for i, n in enumerate(names):
    d = st[:, i] - prev
    print(f"  {n:22s} mean {d.mean():8.0f}  median {d.median():8.0f}  max {d.max():8.0f}")
    prev = st[:, i]
