"""Where a wave of fused_step_kernel spends its cycles (diagnostic build -DFUSED_STAMPS loaded through SGPMP_LIB_PATH;
`make -C stoch_gpmp_amd/csrc EXTRA=-DFUSED_STAMPS BUILD=... OUT=...`): s_memtime cycles per phase summed over the chunks of
the wave's item, config 3 by default.  usage: fused_stamps.py [P S T [spheres]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stoch_gpmp_amd import workloads as W
ta = {"device": torch.device("cuda:0"), "dtype": torch.float32}
P, S, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 128, 64)))
nsph = int(sys.argv[4]) if len(sys.argv) > 4 else 5
pl = W.hip_panda_planner(W.PANDA, T, P, S, ta, seed=0, pipeline_steps=False)
sph = torch.as_tensor(W.panda_spheres(num=nsph)).to(**ta)
pl._engine.set_option("no_step_pipeline", 1)
UNREAD = os.environ.get("FUSED_STAMPS_STORE_FREE") == "1"      # the last stamped step as a store-free one (no sample stores)
for _ in range(30):
    pl.optimize(opt_iters=1, obstacle_spheres=sph)
if UNREAD:
    pl._engine.set_option("store_free_min_bytes", 1)
    pl.step(_samples_unread=True, obstacle_spheres=sph)
    print("store-free step:", pl._engine.store_free_steps())
torch.cuda.synchronize()
assert pl._engine.last_cost_kernel() in ("fused_step_kernel", "fused_step_small_kernel")
print("kernel:", pl._engine.last_cost_kernel())
c = pl._costs.reshape(-1, 8).double().cpu()          # [item = wave][slot]
PRO = os.environ.get("FUSED_STAMPS_PROLOGUE") == "1"          # library built with -DFUSED_STAMPS=2: slots 0..2 = cycles from kernel entry
names = ["phase A (noise + recurrence)", "wait + phase B (x = mu + y, stores)", "phase C loads + quadratic forms",
         "kinematics + fields", "whole item"]
tot = c[:, 4]
xcc_all = (c[:, 6].long() >> 16).double()
print(f"{c.shape[0]} waves, {T // 16} chunks each; whole item: median {tot.median():.0f}, mean {tot.mean():.0f}, max {tot.max():.0f} cycles")
if PRO:
    for i, n in enumerate(["sphere / state tables staged", "barrier passed", "static sphere sum done"]):
        print(f"  from kernel entry to: {n:32s} median {c[:, i].median():8.0f}  mean {c[:, i].mean():8.0f} cycles")
    print(f"  from kernel entry to: first chunk                      median {c[:, 7].median():8.0f}  mean {c[:, 7].mean():8.0f} cycles")
for i, n in enumerate([] if PRO else names[:4]):
    print(f"  {n:38s} median {c[:, i].median():8.0f}  mean {c[:, i].mean():8.0f}  ({100 * c[:, i].mean() / tot.mean():5.1f} % of the item)  per chunk {c[:, i].mean() / (T // 16):7.0f}")
print(f"  {'prologue (kernel entry -> first chunk)':38s} median {c[:, 7].median():8.0f}  mean {c[:, 7].mean():8.0f}  ({100 * c[:, 7].mean() / tot.mean():5.1f} % of the wave)")
print(f"  {'epilogue / unstamped':38s} mean {(tot - c[:, :4].sum(1) - c[:, 7]).mean():8.0f}")
start = c[:, 5]
rel = torch.zeros_like(start)                          # (per XCD: the cycle counters of different XCDs need not agree)
for x in xcc_all.unique():
    m = xcc_all == x
    # 24-bit stamps may wrap inside the launch: the origin is the start that follows the largest circular gap
    cand = start[m]
    srt, _ = torch.sort(cand)
    gaps_c = torch.cat([srt[1:] - srt[:-1], (srt[:1] + (1 << 24)) - srt[-1:]])
    o = srt[(int(gaps_c.argmax()) + 1) % len(srt)]
    rel[m] = (cand - o) % (1 << 24)
print(f"wave starts after the first: median {rel.median():.0f}, 90 % {rel.quantile(0.9):.0f}, max {rel.max():.0f} cycles")

# ---- occupancy timeline per SIMD: (xcd, se, sh, cu, simd) from HW_ID / XCC_ID
hw = c[:, 6].long() & 0xffff
xcc = c[:, 6].long() >> 16
simd_key = (xcc << 16) | (hw & 0xfff0)                # everything but the wave slot
slot = hw & 0xf
import collections
by = collections.defaultdict(list)
for i in range(c.shape[0]):
    by[int(simd_key[i])].append((float(rel[i]), float(rel[i] + tot[i]), int(slot[i])))
print(f"{len(by)} distinct SIMDs saw waves; waves per SIMD: min {min(len(v) for v in by.values())}, max {max(len(v) for v in by.values())}")
span = float((rel + tot).max())
occ, gaps, first, last = [], [], [], []
for v in by.values():
    v.sort()
    occ.append(sum(e - b for b, e, _ in v) / span)
    first.append(v[0][0]); last.append(max(e for _, e, _ in v))
    per_slot = collections.defaultdict(list)
    for b, e, sl in v:
        per_slot[sl].append((b, e))
    for iv in per_slot.values():
        iv.sort()
        gaps += [iv[j + 1][0] - iv[j][1] for j in range(len(iv) - 1)]
occ = torch.tensor(occ); gaps = torch.tensor(gaps) if gaps else torch.zeros(1); first = torch.tensor(first); last = torch.tensor(last)
print(f"launch span (first wave start .. last wave end) {span:.0f} cycles = {span / 2.3e3:.1f} us at 2.3 GHz")
print(f"waves resident per SIMD, time-averaged over the span: mean {occ.mean():.2f}, min {occ.min():.2f}, max {occ.max():.2f}")
print(f"first wave of a SIMD starts: median {first.median():.0f}, max {first.max():.0f}; last wave of a SIMD ends: median {last.median():.0f}, min {last.min():.0f} (span {span:.0f})")
print(f"gap between a wave's last stamp and the ENTRY of the next wave in the same slot (store drain at s_endpgm + dispatch): median {gaps.median():.0f}, mean {gaps.mean():.0f}, 90 % {gaps.quantile(0.9):.0f}, max {gaps.max():.0f} cycles ({len(gaps)} gaps)")
slots_used = sorted(set(int(x) for x in slot))
print("wave slots used:", slots_used)
