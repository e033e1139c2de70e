// Device-side definitions of the cost sweep family (K3 and the fused launches): arithmetic traits, kernel-argument
// structs, wave helpers, the grid lookup, and the three forward-kinematics / link-field code paths.  Included by
// cost_sweep.hip (the build-time kernels) AND by the run-time translation unit that chain_rtc.hip hands to hiprtc for a
// chain without build-time generated code -- so: device code and plain structs only, no host library headers.
#pragma once
#include "sgpmp_internal.h"

template <typename real> struct RealOps;
template <> struct RealOps<float> {
    static __device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
    static __device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
    static __device__ __forceinline__ float exp_(float a) { return expf(a); }
    static __device__ __forceinline__ float sqrt_(float a) { return sqrtf(a); }
    static __device__ __forceinline__ float floor_(float a) { return floorf(a); }
    static __device__ __forceinline__ void sincos_(float a, float* s, float* c) { sincosf(a, s, c); }
    // fast forms used by the register path
    static __device__ __forceinline__ float exp2_(float a) { return __builtin_amdgcn_exp2f(a); }
    static __device__ __forceinline__ void fsincos_(float a, float* s, float* c) {
        const float rev = a * 0.15915494309189535f;            // radians -> revolutions
        *s = __builtin_amdgcn_sinf(rev);
        *c = __builtin_amdgcn_cosf(rev);
    }
    static __device__ __forceinline__ float rcp_(float a) { return __builtin_amdgcn_rcpf(a); }
};
template <> struct RealOps<double> {
    static __device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ double exp_(double a) { return exp(a); }
    static __device__ __forceinline__ double sqrt_(double a) { return sqrt(a); }
    static __device__ __forceinline__ double floor_(double a) { return floor(a); }
    static __device__ __forceinline__ void sincos_(double a, double* s, double* c) { sincos(a, s, c); }
    // The two transcendentals of the fp64 link fields, 47 exponentials and 7 sine / cosine pairs per waypoint of the Panda program:
    // the device library's general forms are ~24 and ~65 fp64 instructions each (overflow / underflow selects; a double-double
    // argument reduction in front of a Payne-Hanek branch) -- a third of an fp64 step's vector instructions (round 6).  These
    // are the same algorithms cut to the arguments the field code forms, full double accuracy (checked against the reference's
    // fixtures at 1e-11: tests/test_gpu_kernels.py g3 / g4):
    //  * 2^a for a = k d^2, k < 0: split a = n + r, |r| <= 1/2, the Taylor polynomial of 2^r to degree 12 (truncation 1.7e-16),
    //    v_ldexp_f64; a clamped to [-1100, 1000] instead of selects on the result (2^-1100 is 0 either way)
    static __device__ __forceinline__ double exp2_(double a) {
        a = fmin(fmax(a, -1100.0), 1000.0);
        const double n = __builtin_rint(a);
        const double r = a - n;
        double p = 0x1.c3bd650fc2986p-36;
        p = __builtin_fma(p, r, 0x1.e8cac7351bb25p-32);
        p = __builtin_fma(p, r, 0x1.e4cf5158b8ecap-28);
        p = __builtin_fma(p, r, 0x1.b5253d395e7c4p-24);
        p = __builtin_fma(p, r, 0x1.62c0223a5c824p-20);
        p = __builtin_fma(p, r, 0x1.ffcbfc588b0c7p-17);
        p = __builtin_fma(p, r, 0x1.430912f86c787p-13);
        p = __builtin_fma(p, r, 0x1.5d87fe78a6731p-10);
        p = __builtin_fma(p, r, 0x1.3b2ab6fba4e77p-7);
        p = __builtin_fma(p, r, 0x1.c6b08d704a0c0p-5);
        p = __builtin_fma(p, r, 0x1.ebfbdff82c58fp-3);
        p = __builtin_fma(p, r, 0x1.62e42fefa39efp-1);
        p = __builtin_fma(p, r, 1.0);
        return ldexp(p, (int)n);
    }
    //  * sin / cos of a joint angle: Cody-Waite reduction by three 33-bit pieces of pi/2 (n pi/2 exact for |n| < 2^20), the
    //    classic degree-13 / degree-14 kernels on |r| <= pi/4, quadrant by select and sign bit; |a| >= 1e6 (never a joint angle)
    //    takes the library's path
    static __device__ __forceinline__ void fsincos_(double a, double* s, double* c) {
        if (__builtin_expect(!(fabs(a) < 1.0e6), 0)) { sincos(a, s, c); return; }
        const double n = __builtin_rint(a * 6.36619772367581382433e-01);
        double r = __builtin_fma(-n, 1.57079632673412561417e+00, a);
        r = __builtin_fma(-n, 6.07710050630396597660e-11, r);
        r = __builtin_fma(-n, 2.02226624871116645580e-21, r);
        const int q = (int)n;
        const double z = r * r;
        double ps = 1.58969099521155010221e-10;
        ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
        ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
        ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
        ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
        ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
        const double sr = __builtin_fma(r * z, ps, r);
        double pc = -1.13596475577881948265e-11;
        pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
        pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
        pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
        pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
        pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
        const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
        const bool swap = (q & 1) != 0;
        const double ss = swap ? cr : sr, cc = swap ? sr : cr;
        *s = __hiloint2double(__double2hiint(ss) ^ ((q & 2) << 30), __double2loint(ss));
        *c = __hiloint2double(__double2hiint(cc) ^ (((q + 1) & 2) << 30), __double2loint(cc));
    }
    static __device__ __forceinline__ double rcp_(double a) { return 1.0 / a; }
};

#define SGPMP_LOG2E 1.4426950408889634

// ---------------------------------------------------------------------------------- kernarg structs
template <typename real>
struct TermK {                    // CostTerm with every constant pre-converted to the compute type
    int kind, flags;
    real K, K2, dt, c11, c12, c22, selfc, inv_cell, off_x, off_y;
    const void* dev_data;
    int dim0, dim1;
    long long rows_per_goal;
    int n_points, n_interp, interp_lo, interp_hi;
    real alpha[SGPMP_MAX_INTERP];
};

template <typename real>
struct ProgK {
    int n_terms, needs_fk;
    TermK<real> t[SGPMP_MAX_TERMS];
};

// Chain constants and FkPlan weights are read through the CONSTANT address space (scalar loads into
// SGPRs, usable directly as VALU operands).  `opaque` hides the pointer from loop-invariant code
// motion: hoisted out of the trajectory loop the ~230 scalars would be spilled lane-by-lane into VGPRs.
#define SGPMP_CONST __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const SGPMP_CONST T* as_const(const T* p) {
    return (const SGPMP_CONST T*)p;
}
template <typename T>
__device__ __forceinline__ const SGPMP_CONST T* opaque(const SGPMP_CONST T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
typedef const SGPMP_CONST ChainDev* ChainC;

template <typename real> struct JointK;
template <> struct JointK<float> {
    static __device__ __forceinline__ float R(ChainC ch, int j, int i) { return ch->Rf[j][i]; }
    static __device__ __forceinline__ float t(ChainC ch, int j, int i) { return ch->tf[j][i]; }
};
template <> struct JointK<double> {
    static __device__ __forceinline__ double R(ChainC ch, int j, int i) { return ch->j[j].R[i]; }
    static __device__ __forceinline__ double t(ChainC ch, int j, int i) { return ch->j[j].t[i]; }
};

// ---------------------------------------------------------------------------------- wave helpers
template <typename real>
__device__ __forceinline__ real shfl_up1(real v) { return __shfl_up(v, 1, 64); }
template <typename real>
__device__ __forceinline__ real shfl_idx(real v, int src) { return __shfl(v, src, 64); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------- grid lookup
// ObstacleMap.get_collisions (obst_map.py:164-182): idx = floor(X * (1/cell) + c_offset) with the
// multiply and the add rounded separately (no FMA) so that cell boundaries fall where the
// reference's do; x clamped by shape[0]-1, y by shape[1]-1, value = map[y, x].
template <typename real>
__device__ __forceinline__ real grid_value(const TermK<real>& tm, real x, real y) {
    using O = RealOps<real>;
    const real fx = O::floor_(O::add_rn(O::mul_rn(x, tm.inv_cell), tm.off_x));
    const real fy = O::floor_(O::add_rn(O::mul_rn(y, tm.inv_cell), tm.off_y));
    const real hx = (real)(tm.dim0 - 1), hy = (real)(tm.dim1 - 1);
    const int ix = (int)fmin(fmax(fx, (real)0), hx);       // clamp in float first: no int overflow
    const int iy = (int)fmin(fmax(fy, (real)0), hy);
    const real* grid = (const real*)tm.dev_data;
    return grid[(size_t)iy * tm.dim1 + ix];
}

// ---------------------------------------------------------------------------------- generic FK (LDS)
// Positions of all link frames for joint vector q, written to LDS column `col` (SoA, `stride`
// reals between consecutive scalars).  H_child = H_parent * Trans(xyz) * RPY * Rz(q).
template <typename real, int N>
__device__ __forceinline__ void fk_points(const ChainDev* __restrict__ ch, const real (&q)[N], real* col,
                                          int stride) {
    using O = RealOps<real>;
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    real p[3] = {0, 0, 0};
    col[0] = 0; col[stride] = 0; col[2 * stride] = 0;
    const int nj = ch->n_joints;
    for (int j = 0; j < nj; ++j) {
        const JointDev& J = ch->j[j];
        real F[9], tt[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = (real)J.R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) tt[i] = (real)J.t[i];
#pragma unroll
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3 + 0] * tt[0] + R[r * 3 + 1] * tt[1] + R[r * 3 + 2] * tt[2];
        real Rn[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Rn[r * 3 + c] = R[r * 3 + 0] * F[c] + R[r * 3 + 1] * F[3 + c] + R[r * 3 + 2] * F[6 + c];
        if (J.revolute) {
            real qv = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) qv = (J.qidx == i) ? q[i] : qv;
            real s, c;
            O::sincos_(qv, &s, &c);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const real a = Rn[r * 3 + 0], b = Rn[r * 3 + 1];
                Rn[r * 3 + 0] = a * c + b * s;
                Rn[r * 3 + 1] = b * c - a * s;
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        real* o = col + (size_t)(j + 1) * 3 * stride;
        o[0] = p[0]; o[stride] = p[1]; o[2 * stride] = p[2];
    }
}

// Append the interpolated points of a field term (fields.py:68-74) after the link points.
template <typename real>
__device__ __forceinline__ void add_interp_points(const TermK<real>& tm, int n_links, real* col, int stride) {
    int o = n_links;
    for (int i = tm.interp_lo; i < tm.interp_hi; ++i) {
        const real ax = col[(i * 3 + 0) * stride], ay = col[(i * 3 + 1) * stride], az = col[(i * 3 + 2) * stride];
        const real bx = col[((i + 1) * 3 + 0) * stride], by = col[((i + 1) * 3 + 1) * stride],
                   bz = col[((i + 1) * 3 + 2) * stride];
        for (int a = 0; a < tm.n_interp; ++a, ++o) {
            const real al = tm.alpha[a];
            col[(o * 3 + 0) * stride] = ax + (bx - ax) * al;
            col[(o * 3 + 1) * stride] = ay + (by - ay) * al;
            col[(o * 3 + 2) * stride] = az + (bz - az) * al;
        }
    }
}

// LinkDistanceField.compute_cost on the np points of this lane (fields.py:75-86).
template <typename real>
__device__ __forceinline__ real spheres_field(const TermK<real>& tm, int np, const real* col, int stride,
                                              const real* __restrict__ sph, int n_sph) {
    using O = RealOps<real>;
    const int type = tm.flags & 15;
    real acc = (type == SGPMP_FIELD_SDF) ? (real)-1e30 : (real)0;
    for (int o = 0; o < n_sph; ++o) {
        const real cx = sph[o * 4 + 0], cy = sph[o * 4 + 1], cz = sph[o * 4 + 2], r = sph[o * 4 + 3];
        const real r2 = r * r;
        for (int i = 0; i < np; ++i) {
            const real dx = col[(i * 3 + 0) * stride] - cx, dy = col[(i * 3 + 1) * stride] - cy,
                       dz = col[(i * 3 + 2) * stride] - cz;
            const real d2 = dx * dx + dy * dy + dz * dz;
            if (type == SGPMP_FIELD_RBF) {
                acc += O::exp_((real)-0.5 * d2 / r2);
            } else if (type == SGPMP_FIELD_SDF) {
                real sdf = r - O::sqrt_(d2);
                if (tm.flags & SGPMP_FLAG_SDF_CLAMP) sdf = fmin(sdf, (real)0);
                acc = fmax(acc, sdf);
            } else {
                acc += (O::sqrt_(d2) < r) ? (real)1 : (real)0;
            }
        }
    }
    return acc;
}

// LinkSelfDistanceField.compute_cost: full np x np sum incl. the diagonal (fields.py:124).
template <typename real>
__device__ __forceinline__ real self_field(const TermK<real>& tm, int np, const real* col, int stride) {
    using O = RealOps<real>;
    const real k = tm.K2;                             // -1 / (2 margin^2)
    real acc = (real)np;                              // diagonal: exp(0)
    for (int i = 1; i < np; ++i) {
        const real ax = col[(i * 3 + 0) * stride], ay = col[(i * 3 + 1) * stride], az = col[(i * 3 + 2) * stride];
        for (int j = 0; j < i; ++j) {
            const real dx = ax - col[(j * 3 + 0) * stride], dy = ay - col[(j * 3 + 1) * stride],
                       dz = az - col[(j * 3 + 2) * stride];
            acc += (real)2 * O::exp_((dx * dx + dy * dy + dz * dz) * k);
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------- generated-chain path
// FKMODE >= 1000: the chain has build-time generated code (chain_code_generated.h): FK is
// straight-line code with the joint constants folded in, only the DISTINCT link positions exist,
// the loops over links / q-dependent pairs are unrolled with immediate weights, and the sphere terms
// of links that never move are evaluated once per wave (`stat`) instead of once per waypoint.
template <int FKMODE> struct ChainOf { using type = ChainCode_panda; };     // 1000 -> panda

template <typename real, class CC>
__device__ __forceinline__ void fk_cg(const real (&q)[CC::N], real (&Pq)[CC::NREP][3]) {
    using O = RealOps<real>;
    if constexpr (sizeof(real) == 4) CC::template fk_snapped<real, O>(q, Pq);
    else CC::template fk_exact<real, O>(q, Pq);
}

// sphere field restricted to the links with is_static == WANT_STATIC; `acc` carries the other part
template <typename real, class CC, bool WANT_STATIC>
__device__ __forceinline__ real spheres_field_cg(const TermK<real>& tm, const real (&Pq)[CC::NREP][3],
                                                 const real* __restrict__ sph, int n_sph, real acc) {
    using O = RealOps<real>;
    const int type = tm.flags & 15;
    for (int o = 0; o < n_sph; ++o) {
        const real cx = sph[o * 4 + 0], cy = sph[o * 4 + 1], cz = sph[o * 4 + 2], r = sph[o * 4 + 3];
        if (type == SGPMP_FIELD_RBF) {
            const real k = (real)(-0.5 * SGPMP_LOG2E) * O::rcp_(r * r);
#pragma unroll
            for (int l = 0; l < CC::NREP; ++l) {
                if (CC::is_static(l) != WANT_STATIC) continue;
                const real dx = Pq[l][0] - cx, dy = Pq[l][1] - cy, dz = Pq[l][2] - cz;
                acc += (real)CC::mult(l) * O::exp2_((dx * dx + dy * dy + dz * dz) * k);
            }
        } else if (type == SGPMP_FIELD_SDF) {
#pragma unroll
            for (int l = 0; l < CC::NREP; ++l) {
                if (CC::is_static(l) != WANT_STATIC) continue;
                const real dx = Pq[l][0] - cx, dy = Pq[l][1] - cy, dz = Pq[l][2] - cz;
                real sdf = r - O::sqrt_(dx * dx + dy * dy + dz * dz);
                if (tm.flags & SGPMP_FLAG_SDF_CLAMP) sdf = fmin(sdf, (real)0);
                acc = fmax(acc, sdf);
            }
        } else {
#pragma unroll
            for (int l = 0; l < CC::NREP; ++l) {
                if (CC::is_static(l) != WANT_STATIC) continue;
                const real dx = Pq[l][0] - cx, dy = Pq[l][1] - cy, dz = Pq[l][2] - cz;
                acc += (O::sqrt_(dx * dx + dy * dy + dz * dz) < r) ? (real)CC::mult(l) : (real)0;
            }
        }
    }
    return acc;
}

template <typename real, class CC>
__device__ __forceinline__ real self_field_cg(const TermK<real>& tm, const real (&Pq)[CC::NREP][3]) {
    using O = RealOps<real>;
    const real k = tm.K2 * (real)SGPMP_LOG2E;
    real acc = tm.selfc;                                 // diagonal, coincident and rigid pairs (host)
#pragma unroll
    for (int p = 0; p < CC::NPAIR; ++p) {
        const int i = CC::pair_i(p), j = CC::pair_j(p);
        const real dx = Pq[i][0] - Pq[j][0], dy = Pq[i][1] - Pq[j][1], dz = Pq[i][2] - Pq[j][2];
        acc += (real)CC::pair_w(p) * O::exp2_((dx * dx + dy * dy + dz * dz) * k);
    }
    return acc;
}

// ---------------------------------------------------------------------------------- register FK path
template <typename real, int N, int NJ>
__device__ __forceinline__ void fk_points_reg(ChainC chain, const real (&q)[N],
                                              real (&PX)[NJ + 1], real (&PY)[NJ + 1], real (&PZ)[NJ + 1]) {
    using O = RealOps<real>;
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    real p[3] = {0, 0, 0};
    PX[0] = 0; PY[0] = 0; PZ[0] = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        ChainC ch = opaque(chain);                         // this joint's 12 constants: loaded here
        real Fm[9], tt[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Fm[i] = JointK<real>::R(ch, j, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) tt[i] = JointK<real>::t(ch, j, i);
#pragma unroll
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3 + 0] * tt[0] + R[r * 3 + 1] * tt[1] + R[r * 3 + 2] * tt[2];
        real Rn[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Rn[r * 3 + c] = R[r * 3 + 0] * Fm[c] + R[r * 3 + 1] * Fm[3 + c] + R[r * 3 + 2] * Fm[6 + c];
        if (j < N) {                                       // revolute-first chain: joint j turns by q[j]
            real s, c;
            O::fsincos_(q[j < N ? j : 0], &s, &c);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const real a = Rn[r * 3 + 0], b = Rn[r * 3 + 1];
                Rn[r * 3 + 0] = a * c + b * s;
                Rn[r * 3 + 1] = b * c - a * s;
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        PX[j + 1] = p[0]; PY[j + 1] = p[1]; PZ[j + 1] = p[2];
    }
}

template <typename real, int NJ>
__device__ __forceinline__ real spheres_field_reg(const TermK<real>& tm, ChainC chain,
                                                  const real (&PX)[NJ + 1], const real (&PY)[NJ + 1],
                                                  const real (&PZ)[NJ + 1], const real* __restrict__ sph,
                                                  int n_sph) {
    using O = RealOps<real>;
    constexpr int NL = NJ + 1;
    const int type = tm.flags & 15;
    real acc = (type == SGPMP_FIELD_SDF) ? (real)-1e30 : (real)0;
    ChainC ch = opaque(chain);
    real mult[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) mult[l] = (real)ch->plan.mult[l];
    for (int o = 0; o < n_sph; ++o) {
        const real cx = sph[o * 4 + 0], cy = sph[o * 4 + 1], cz = sph[o * 4 + 2], r = sph[o * 4 + 3];
        if (type == SGPMP_FIELD_RBF) {
            const real k = (real)(-0.5 * SGPMP_LOG2E) * O::rcp_(r * r);
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const real m = mult[l];
                if (m != 0) {
                    const real dx = PX[l] - cx, dy = PY[l] - cy, dz = PZ[l] - cz;
                    acc += m * O::exp2_((dx * dx + dy * dy + dz * dz) * k);
                }
            }
        } else if (type == SGPMP_FIELD_SDF) {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (mult[l] != 0) {
                    const real dx = PX[l] - cx, dy = PY[l] - cy, dz = PZ[l] - cz;
                    real sdf = r - O::sqrt_(dx * dx + dy * dy + dz * dz);
                    if (tm.flags & SGPMP_FLAG_SDF_CLAMP) sdf = fmin(sdf, (real)0);
                    acc = fmax(acc, sdf);
                }
            }
        } else {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const real m = mult[l];
                if (m != 0) {
                    const real dx = PX[l] - cx, dy = PY[l] - cy, dz = PZ[l] - cz;
                    acc += (O::sqrt_(dx * dx + dy * dy + dz * dz) < r) ? m : (real)0;
                }
            }
        }
    }
    return acc;
}

template <typename real, int NJ>
__device__ __forceinline__ real self_field_reg(const TermK<real>& tm, ChainC chain,
                                               const real (&PX)[NJ + 1], const real (&PY)[NJ + 1],
                                               const real (&PZ)[NJ + 1]) {
    using O = RealOps<real>;
    constexpr int NL = NJ + 1;
    const real k = tm.K2 * (real)SGPMP_LOG2E;            // exp(K2 d^2) = exp2(K2 log2(e) d^2)
    real acc = tm.selfc;                                 // diagonal, coincident and rigid pairs
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        ChainC ch = opaque(chain);                         // one row of pair weights at a time
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const real w = (real)ch->plan.wpair[i * SGPMP_MAX_LINKS + j];
            if (w != 0) {
                const real dx = PX[i] - PX[j], dy = PY[i] - PY[j], dz = PZ[i] - PZ[j];
                acc += w * O::exp2_((dx * dx + dy * dy + dz * dz) * k);
            }
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------- the sweep
template <typename real>
struct CostArgs {
    int T;
    const ChainDev* chain;      // generic path only
    int n_links;
    const real* trajs;
    long long batch, batch_offset;
    const real* spheres;
    int n_spheres;
    const real* isw;            // [particles][T+1][d] or null
    int rows_per_particle;
    real is_dt;                 // time step of the sampling prior (Phi of the IS term)
    real* costs;
    double* costs64;
    int rpp_shift, rpg_shift;   // log2 of rows_per_particle / rows_per_goal when a power of two, else -1
};

// A cost program with at most one term of each kind, as named fields (see cost_sweep_kernel.inc).
template <typename real>
struct FlatProg {
    int has_gp, has_goal, has_grid, has_self, has_sph, sph_index;
    TermK<real> gp, goal, grid, self, sph;
};

