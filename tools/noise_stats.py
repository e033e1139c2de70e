"""Statistical check of the in-kernel noise stream (csrc/rng.h) through its CPU restatement (oracle/native_noise.py,
pinned to the Random123 known-answer vectors): >= 1e8 standard normals in the kernel's own counter layout
(seed, draw, particle, sample, waypoint pair, dof), for a given Philox round count.

    python tools/noise_stats.py --rounds 7 --normals 1e8   ->  one JSON object on stdout

Reported with the standard error a perfect N(0,1) i.i.d. source would show at the same sample size:
  moments (mean, variance, skewness, excess kurtosis); chi-square of the 256-bin probability-integral transform;
  autocorrelation along the waypoint axis for lags 1..T-1 (max |rho| and the number beyond 4 sigma);
  correlations between the two outputs of a Box-Muller pair, between position and velocity noise of a waypoint,
  between neighbouring dofs, samples, particles and draws.
Test infrastructure / evidence only -- nothing in the product imports it."""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.native_noise import native_eps  # noqa: E402


def corr(a, b):
    a = a.ravel().astype(np.float64)
    b = b.ravel().astype(np.float64)
    return float(np.mean(a * b) - a.mean() * b.mean()) / float(a.std() * b.std())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--normals", type=float, default=1e8)
    ap.add_argument("--seed", type=int, default=20261003)
    args = ap.parse_args()
    T, n, S = 64, 7, 128
    per_particle = S * T * 2 * n
    P = int(math.ceil(args.normals / per_particle))
    chunk = 64
    N = 0
    s1 = s2 = s3 = s4 = 0.0
    hist = np.zeros(256, dtype=np.int64)
    lag_sum = np.zeros(T, dtype=np.float64)              # sum over streams of x_t x_{t+lag}
    lag_cnt = np.zeros(T, dtype=np.float64)
    pairs = {k: [0.0, 0] for k in ("box_muller_pair(pos,vel)", "neighbour dof", "neighbour sample", "neighbour particle",
                                   "next draw", "pos_t vs vel_t+1 (same Philox block)")}
    from scipy.special import ndtr
    for p0 in range(0, P, chunk):
        idx = list(range(p0, min(P, p0 + chunk)))
        e = native_eps(args.seed, 2, idx, S, T, n, "float32", rounds=args.rounds).astype(np.float64)
        e = e.reshape(S, len(idx), T, 2 * n)
        x = e.ravel()
        N += x.size
        s1 += x.sum(); s2 += (x ** 2).sum(); s3 += (x ** 3).sum(); s4 += (x ** 4).sum()
        hist += np.bincount(np.minimum((ndtr(x) * 256).astype(np.int64), 255), minlength=256)
        pos = e[..., :n]                                  # [S, p, T, n]
        for lag in range(1, T):
            lag_sum[lag] += float((pos[:, :, :-lag] * pos[:, :, lag:]).sum())
            lag_cnt[lag] += pos[:, :, :-lag].size
        def acc(key, a, b):
            pairs[key][0] += float((a * b).sum()); pairs[key][1] += a.size
        acc("box_muller_pair(pos,vel)", e[..., :n], e[..., n:])
        acc("neighbour dof", e[..., :n - 1], e[..., 1:n])
        acc("neighbour sample", e[:-1], e[1:])
        acc("neighbour particle", e[:, :-1], e[:, 1:])
        acc("pos_t vs vel_t+1 (same Philox block)", e[:, :, 0::2, :n], e[:, :, 1::2, n:])
        if p0 == 0:
            e2 = native_eps(args.seed, 3, idx, S, T, n, "float32", rounds=args.rounds).astype(np.float64).reshape(e.shape)
            acc("next draw", e, e2)
    m = s1 / N
    var = s2 / N - m * m
    m3 = (s3 / N - 3 * m * s2 / N + 2 * m ** 3) / var ** 1.5
    m4 = (s4 / N - 4 * m * s3 / N + 6 * m * m * s2 / N - 3 * m ** 4) / var ** 2 - 3.0
    exp = N / 256.0
    chi2 = float(((hist - exp) ** 2 / exp).sum())
    rho = lag_sum[1:] / lag_cnt[1:]
    z = rho * np.sqrt(lag_cnt[1:])
    out = {
        "rounds": args.rounds, "normals": int(N), "layout": f"{P} particles x {S} samples x {T} waypoints x {2 * n} (fp32 stream)",
        "mean": m, "mean_se": 1 / math.sqrt(N), "variance": var, "variance_se": math.sqrt(2 / N),
        "skewness": m3, "skewness_se": math.sqrt(6 / N), "excess_kurtosis": m4, "excess_kurtosis_se": math.sqrt(24 / N),
        "pit_chi2_255dof": chi2, "pit_chi2_z": (chi2 - 255) / math.sqrt(2 * 255),
        "autocorr_waypoint_axis": {"lags": T - 1, "max_abs_rho": float(np.abs(rho).max()), "max_abs_z": float(np.abs(z).max()),
                                   "beyond_4_sigma": int((np.abs(z) > 4).sum()), "rms_z": float(np.sqrt((z ** 2).mean()))},
        "cross_correlations": {k: {"rho": v[0] / v[1], "z": v[0] / v[1] * math.sqrt(v[1])} for k, v in pairs.items()},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
