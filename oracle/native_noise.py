"""CPU restatement (numpy) of the NATIVE noise source of the HIP sampler -- TEST INFRASTRUCTURE ONLY.

The reference draws eps = randn(S, P, M) from torch's sequential global generator (planner.py:48-49,
torch multivariate_normal.py:250-253); the HIP sampler's default mode replaces that by a counter-based
stream (csrc/rng.h): Philox4x32-R, R = 7 by default, 10 as a build option (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
SC'11 -- the generator behind curand / torch.cuda) + the fp32 Box-Muller (fp64 contexts draw the same normals, widened), keyed on
(seed, draw, global particle, sample, waypoint pair, dof).  The device forms log2 / sin / cos on the hardware's approximating units;
this restatement forms them in fp64 and rounds: equal to an ulp of fp32, not bit for bit -- tests that need the device's exact
eps read it back through sgpmp_noise (Engine.noise).  This file restates that stream so that the
native mode has a deterministic checker too: `native_eps(...)` returns the noise in torch's
`randn(S, P, M)` layout, ready for `oracle.ref_equiv.TrajPrior.sample(eps=...)`.

Pinned by the Random123 known-answer vectors for philox4x32 with 7 and with 10 rounds (tests/test_oracle_golden.py).
"""
import numpy as np

# Round count of the stream being restated: the tests set it to what the library under test reports
# (sgpmp_philox_rounds(); csrc/rng.h SGPMP_PHILOX_ROUNDS).  10 = Random123's default, 7 = its Crush-resistant minimum.
DEFAULT_ROUNDS = 7

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32-R: counters are uint32 arrays (broadcastable), keys Python ints."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(rounds):
        p0 = M0 * c0                                   # < 2^64: exact in uint64
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    return philox4x32(c0, c1, c2, c3, k0, k1, 10)


def box_muller_f32(a, b):
    """csrc/rng.h box_muller_f32: u1 = a*2^-32 + 2^-33 in (0,1], angle = b*2^-32 revolutions."""
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    u1 = a * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)
    u2 = b * np.float32(2.3283064365386963e-10)
    r = np.sqrt(np.float32(-2.0 * 0.6931471805599453) * np.log2(u1.astype(np.float64)).astype(np.float32))
    ang = (2.0 * np.pi) * u2.astype(np.float64)
    return (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)


def native_eps(seed, draw, particles, S, T, n, dtype="float32", rounds=None):
    """Noise of the HIP sampler for global particle indices `particles` -> [S, len(particles), T*2n]
    (torch.randn(S, P, M) layout: element t*d + k is the position noise of dof k at waypoint t,
    t*d + n + k the velocity noise)."""
    if rounds is None:
        rounds = DEFAULT_ROUNDS
    particles = np.asarray(particles, dtype=np.uint64)
    P, d = len(particles), 2 * n
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    s_idx = np.arange(S, dtype=np.uint64).reshape(S, 1, 1, 1)
    m_idx = particles.reshape(1, P, 1, 1)
    k_idx = np.arange(n, dtype=np.uint64).reshape(1, 1, 1, n)
    c3 = np.uint64(draw & 0xFFFFFFFF)
    # (one stream for both precisions since round 6: fp64 contexts draw the fp32 normals, widened -- csrc/rng.h NoiseGen<double>)
    out = np.zeros((S, P, T, d), dtype=np.float64 if dtype == "float64" else np.float32)
    if True:
        nb = (T + 1) // 2
        b_idx = np.arange(nb, dtype=np.uint64).reshape(1, 1, nb, 1)
        x, y, z, w = philox4x32(b_idx | (k_idx << np.uint64(20)), s_idx, m_idx, c3, k0, k1, rounds)
        e0, e1 = box_muller_f32(x, y)                  # waypoint 2b:   (pos, vel)
        e2, e3 = box_muller_f32(z, w)                  # waypoint 2b+1: (pos, vel)
        out[:, :, 0::2, :n], out[:, :, 0::2, n:] = e0[:, :, :(T + 1) // 2], e1[:, :, :(T + 1) // 2]
        out[:, :, 1::2, :n], out[:, :, 1::2, n:] = e2[:, :, :T // 2], e3[:, :, :T // 2]
    return out.reshape(S, P, T * d)
