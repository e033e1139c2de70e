// Counter-based standard-normal noise for the sampler (K2): Philox4x32-R + Box-Muller, R = SGPMP_PHILOX_ROUNDS.
//
// The reference draws eps = randn(S, P, M) from torch's global CPU generator
// (planner.py:48-49 -> torch multivariate_normal.py:250-253); that stream is sequential and cannot
// be reproduced per element.  The native path therefore keys every normal on
// (seed, draw, global particle, sample, waypoint, dof) so that results are independent of how
// particles are sharded over GPUs and nothing has to be stored to revisit a sample.
// Parity with the reference is checked in the separate external-eps mode of the kernels.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

struct Philox4 { uint32_t x, y, z, w; };

// Rounds of the Philox4x32 bijection.  Random123 (Salmon et al., SC'11) ships 10 as the default "with a safety
// margin" and documents 7 as the smallest count that passes BigCrush ("Crush-resistant"); both have known-answer
// vectors (tests/test_oracle_golden.py pins oracle/native_noise.py to them), and sgpmp_philox_rounds() tells the
// host which one this library was built with.  The noise source is ~30 % of the fused launch's vector instructions.
#ifndef SGPMP_PHILOX_ROUNDS
#define SGPMP_PHILOX_ROUNDS 7
#endif
static_assert(SGPMP_PHILOX_ROUNDS >= 7 && SGPMP_PHILOX_ROUNDS <= 10, "Philox4x32: 7 (Crush-resistant minimum) .. 10 rounds");

__device__ __forceinline__ Philox4 philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
    // Rounds 1 and 2 are written with plain xors grouped so that everything WAVE-UNIFORM folds on the scalar
    // unit: the callers key the noise with c2 = particle and c3 = draw, uniform per wave like the key, so in
    // round 1 the product M1 * c2 and the terms (hi1 ^ k0), (c3 ^ k1) are scalar work (s_mul_hi_u32 / s_mul_i32 /
    // s_xor), and in round 2 c1 = lo1 still is.  v_bitop3_b32 takes one scalar operand only: with two uniform
    // inputs the compiler spent a v_mov per round on top of a v_mad_u64_u32 whose result every lane shared.
    // Same integers as the textbook rounds below (Random123 known-answer vectors: tests/test_oracle_golden.py).
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = (hi1 ^ k0) ^ c1;
        const uint32_t n2 = (c3 ^ k1) ^ hi0;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = (c1 ^ k0) ^ hi1;
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int r = 2; r < SGPMP_PHILOX_ROUNDS; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of separate lo / hi multiplies
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // three-input xor in ONE instruction: v_bitop3_b32 (new on gfx950) with truth table 0x96
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// Two independent N(0,1) from two 32-bit words (fp32).  u1 in (0,1], tail reach 6.7 sigma.
__device__ __forceinline__ void box_muller_f32(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float u2 = (float)b * 2.3283064365386963e-10f;            // revolutions
    const float r = __builtin_amdgcn_sqrtf(-2.0f * 0.6931471805599453f * __log2f(u1));   // raw v_sqrt_f32
    z0 = r * __builtin_amdgcn_cosf(u2);                              // cos(2 pi u2)
    z1 = r * __builtin_amdgcn_sinf(u2);
}

// ---- one step of the scan recurrence  y_t = H_t y_{t-1} + G_t eps_t  of an isotropic prior, per (sample, dof):
//     pn = g11 e_pos + h11 p + h12 v,     vn = g21 e_pos + g22 e_vel + h21 p + h22 v
// in ONE evaluation order, every product-sum an explicit fma, so that hipcc has no choice of which multiply to fuse with which
// add (left to -ffp-contract it pairs them differently from kernel to kernel -- and from one unroll factor to another: the
// round-4 `fused_pipe` experiment came out 1 ulp off the default launch for exactly that reason).  Every launch that samples
// with the native stream on an isotropic prior goes through this order -- scalar here, two-wide in scan_step2 below -- which
// makes "the fused launch's samples = sample_iso_kernel's" a property of the source instead of an observation.
// c: row of PriorDev::iso64 / iso32 = [g11 g21 g22 h11 h12 h21 h22 0].
__device__ __forceinline__ float sg_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double sg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// (in two halves: the noise terms do not depend on the state -- the few-wave sampler forms them in parallel and leaves the two
// dependent fmas per waypoint to its serial phase.  h = c[3 .. 6].)
template <typename real, typename CP>
__device__ __forceinline__ void scan_step_noise(CP c, real e_pos, real e_vel, real& tp, real& tv) {
    tp = c[0] * e_pos;
    tv = sg_fma((real)c[2], e_vel, c[1] * e_pos);
}
template <typename real, typename HP>
__device__ __forceinline__ void scan_step_state(HP h, real tp, real tv, real& p, real& v) {
    tp = sg_fma((real)h[0], p, tp);
    tv = sg_fma((real)h[2], p, tv);
    p = sg_fma((real)h[1], v, tp);
    v = sg_fma((real)h[3], v, tv);
}
template <typename real, typename CP>
__device__ __forceinline__ void scan_step(CP c, real e_pos, real e_vel, real& p, real& v) {
    real tp, tv;
    scan_step_noise<real>(c, e_pos, e_vel, tp, tv);
    scan_step_state<real>(c + 3, tp, tv, p, v);
}

// The same step on the packed-fp32 unit: (p, v) as one register pair, four instructions (v_pk_mul, v_fma, 2 x v_pk_fma with the
// broadcast operand selected by op_sel) instead of the seven the compiler made of the scalar form.  c: row of PriorDev::iso32p =
// [g11 g21 | h11 h21 | h12 h22 | g22 0] -- the pairs a packed instruction takes as ONE aligned scalar-register pair.
typedef float sg_f2 __attribute__((ext_vector_type(2)));
// (in two halves: the noise term does not depend on the state -- fused_step_small_kernel forms it for a whole chunk BEFORE the
// state arrives from the wave that has the previous chunk, and only the two dependent fmas per waypoint wait for it)
template <typename CP>
__device__ __forceinline__ sg_f2 scan_step2_noise(CP c, float e_pos, float e_vel) {
    const sg_f2 g = {c[0], c[1]};
    sg_f2 t = g * (sg_f2){e_pos, e_pos};
    t.y = __builtin_fmaf(c[6], e_vel, t.y);
    return t;
}
__device__ __forceinline__ void scan_step2_state(sg_f2 h1, sg_f2 h2, sg_f2 t, sg_f2& pv) {
    t = __builtin_elementwise_fma(h1, (sg_f2){pv.x, pv.x}, t);
    pv = __builtin_elementwise_fma(h2, (sg_f2){pv.y, pv.y}, t);
}
template <typename CP>
__device__ __forceinline__ void scan_step2(CP c, float e_pos, float e_vel, sg_f2& pv) {
    const sg_f2 h1 = {c[2], c[3]}, h2 = {c[4], c[5]};
    scan_step2_state(h1, h2, scan_step2_noise(c, e_pos, e_vel), pv);
}

// ... with the two coefficient rows of a Philox block (waypoints t, t + 1) already in 16 scalar registers (one s_load_dwordx16)
typedef float sg_f16 __attribute__((ext_vector_type(16)));
template <int R>
__device__ __forceinline__ void scan_step2v(const sg_f16& c, float e_pos, float e_vel, sg_f2& pv) {
    const sg_f2 g = {c[R + 0], c[R + 1]}, h1 = {c[R + 2], c[R + 3]}, h2 = {c[R + 4], c[R + 5]};
    sg_f2 t = g * (sg_f2){e_pos, e_pos};
    t.y = __builtin_fmaf(c[R + 6], e_vel, t.y);
    t = __builtin_elementwise_fma(h1, (sg_f2){pv.x, pv.x}, t);
    pv = __builtin_elementwise_fma(h2, (sg_f2){pv.y, pv.y}, t);
}

// ---- the segment form of the same scan (fused_planar_seg.inc: a wave owns L waypoints; update.hip regenerates a row in that form):
// the in-segment recurrence is scan_step from a zero state; a segment's true start state is the chain  y <- z_j + A_j y  over
// the earlier segments (A_j: the segment's 2 x 2 propagator, z_j: its zero-start end state), and the fix-up adds Pre_t y_start.
// Explicit fmas here too: the update kernel's regenerated row must equal the launch's row bit for bit.
__device__ __forceinline__ void seg_chain(float a00, float a01, float a10, float a11, float zx, float zy, float& yp, float& yv) {
    const float np_ = __builtin_fmaf(a01, yv, __builtin_fmaf(a00, yp, zx));
    const float nv_ = __builtin_fmaf(a11, yv, __builtin_fmaf(a10, yp, zy));
    yp = np_; yv = nv_;
}
__device__ __forceinline__ void seg_fixup(float p00, float p01, float p10, float p11, float ysp, float ysv, float& y_pos, float& y_vel) {
    y_pos = y_pos + __builtin_fmaf(p01, ysv, p00 * ysp);
    y_vel = y_vel + __builtin_fmaf(p11, ysv, p10 * ysp);
}

// Box-Muller for both pairs of one Philox block, the plain multiplies / fmas two-wide: same operations on the same values as
// two box_muller_f32 calls (z02 = (z0, z2), z13 = (z1, z3)).
__device__ __forceinline__ void box_muller2_f32(const Philox4& r, sg_f2& z02, sg_f2& z13) {
    const sg_f2 k = {2.3283064365386963e-10f, 2.3283064365386963e-10f}, h = {1.1641532182693481e-10f, 1.1641532182693481e-10f};
    const sg_f2 u1 = __builtin_elementwise_fma((sg_f2){(float)r.x, (float)r.z}, k, h);
    const sg_f2 u2 = (sg_f2){(float)r.y, (float)r.w} * k;          // revolutions
    const sg_f2 m = (sg_f2){__log2f(u1.x), __log2f(u1.y)} * (sg_f2){-2.0f * 0.6931471805599453f, -2.0f * 0.6931471805599453f};
    const sg_f2 rr = {__builtin_amdgcn_sqrtf(m.x), __builtin_amdgcn_sqrtf(m.y)};
    z02 = rr * (sg_f2){__builtin_amdgcn_cosf(u2.x), __builtin_amdgcn_cosf(u2.y)};
    z13 = rr * (sg_f2){__builtin_amdgcn_sinf(u2.x), __builtin_amdgcn_sinf(u2.y)};
}

// Noise for waypoint t, dof k of (mode, sample): returns (eps[t, k], eps[t, n + k]).
// fp32: one Philox call serves waypoints (t, t^1); the caller may cache it via `cache`.
template <typename real> struct NoiseGen;

template <> struct NoiseGen<float> {
    uint32_t k0, k1, c1, c2, c3, kk;
    float z[4];
    int have = -1;
    __device__ __forceinline__ void init(uint64_t seed, uint64_t draw, uint32_t mode, uint32_t s,
                                         uint32_t k) {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        c1 = s; c2 = mode; c3 = (uint32_t)draw; kk = k << 20;
    }
    __device__ __forceinline__ void get(int t, float& e_pos, float& e_vel) {
        const int blk = t >> 1;
        if (blk != have) {
            const Philox4 r = philox4x32_r((uint32_t)blk | kk, c1, c2, c3, k0, k1);
            box_muller_f32(r.x, r.y, z[0], z[1]);
            box_muller_f32(r.z, r.w, z[2], z[3]);
            have = blk;
        }
        e_pos = (t & 1) ? z[2] : z[0];
        e_vel = (t & 1) ? z[3] : z[1];
    }
    // both waypoints of an even-aligned pair: e = (pos_t, vel_t, pos_{t+1}, vel_{t+1}); same stream as get()
    __device__ __forceinline__ void get4(int t_even, float (&e)[4]) {
        const Philox4 r = philox4x32_r((uint32_t)(t_even >> 1) | kk, c1, c2, c3, k0, k1);
        box_muller_f32(r.x, r.y, e[0], e[1]);
        box_muller_f32(r.z, r.w, e[2], e[3]);
    }
    // the same four normals as register pairs: z02 = (pos_t, pos_{t+1}), z13 = (vel_t, vel_{t+1})
    __device__ __forceinline__ void get4p(int t_even, sg_f2& z02, sg_f2& z13) {
        const Philox4 r = philox4x32_r((uint32_t)(t_even >> 1) | kk, c1, c2, c3, k0, k1);
        box_muller2_f32(r, z02, z13);
    }
};

// fp64 contexts draw the SAME normals as fp32 contexts (round 6): the fp32 Box-Muller of the block of waypoints (t, t ^ 1), widened.
// Rounds 1-5 gave them a stream of their own -- 53-bit uniforms, log / sqrt / sincospi in double: ~120 software fp64 operations
// per pair of normals, a third of an fp64 step's vector instructions -- for digits of the NOISE that no result depends on (the
// reference draws torch.randn from a generator that cannot be matched natively anyway; parity with it is the external-eps mode).
// One stream for both precisions also lets an fp32 planner be compared with its fp64 twin on identical noise at any size.
// The exact values a context's kernels use can be read back with sgpmp_noise (the hardware's log2 / sin / cos are approximations
// that a CPU restatement reproduces to an ulp of fp32, not bit for bit).
template <> struct NoiseGen<double> {
    NoiseGen<float> g;
    __device__ __forceinline__ void init(uint64_t seed, uint64_t draw, uint32_t mode, uint32_t s, uint32_t k) { g.init(seed, draw, mode, s, k); }
    __device__ __forceinline__ void get(int t, double& e_pos, double& e_vel) {
        float p, v;
        g.get(t, p, v);
        e_pos = (double)p; e_vel = (double)v;
    }
    __device__ __forceinline__ void get4(int t_even, double (&e)[4]) {
        float f[4];
        g.get4(t_even, f);
        e[0] = (double)f[0]; e[1] = (double)f[1]; e[2] = (double)f[2]; e[3] = (double)f[3];
    }
};
