"""Update kernel (K4) when the softmax is NOT one-hot: all S rows carry weight, so the weighted sum
streams the whole sample tensor (config 3: 470 MB)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from stoch_gpmp_amd.engine import Engine
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
P, S, T, n = 1024, 128, 64, 7
eng = Engine(n, T, P, S, tensor_args=ta)
samples = torch.randn(P, S, T, 2 * n, **ta)
means = torch.zeros(P, T, 2 * n, **ta)
for name, costs in (("one-hot", torch.arange(S, **ta).repeat(P, 1) * 1e6),
                    ("all rows", torch.rand(P, S, **ta))):
    costs = costs.contiguous()
    for _ in range(3): eng.update(costs, samples, means, 1.0, 0.1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.update(costs, samples, means, 1.0, 0.1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name:9s}: {dt*1e6:8.1f} us" + (f"  ({samples.numel()*4/dt/1e12:.2f} TB/s of sample reads)" if name != "one-hot" else ""))
