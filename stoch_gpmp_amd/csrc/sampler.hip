// K2 -- trajectory sampler:  out[m, s] = means[m] + scale_tril @ eps[s, m]   as an O(T d) scan.
//
// Replaces MultiMPPrior.sample (costs/factors/mp_priors_multi.py:204-207) ->
// torch MultivariateNormal.rsample (multivariate_normal.py:250-253), whose dense
// [P,M,M] @ [S,P,M,1] product is 81 % of the reference's iteration time.
//
// scale_tril @ eps solves L_inv y = eps with L_inv block-bidiagonal (prior_factor.hip), i.e.
//     y_t = G_t eps_t + H_t y_{t-1}.
// Isotropic path (every prior the public API can build: K_s, K_g, Q_c are multiples of I, so
// G_t = g_t (x) I_n and H_t = h_t (x) I_n with 2x2 g_t, h_t): one thread per (sample, dof) runs a
// 7-FMA recurrence on (position_k, velocity_k); coefficients are wave-uniform scalar loads.
// Dense path (user-supplied Q_c_inv): one thread per sample carries the full d-vector.
#include <cstdlib>
#include "sgpmp_internal.h"
#include "rng.h"

// The kernel is VALU-bound (Philox + Box-Muller dominate), so the inner loop is kept lean:
//   * two waypoints per trip = one Philox4x32 call = four normals (fp32);
//   * the perturbation y is written to an LDS tile [samples][TC*d (+pad)]; the mean is added while the
//     tile is flushed, once per VW-wide vector instead of per scalar;
//   * tiles are flushed as whole row segments with VW-wide stores (16 B per lane when the row pitch
//     allows), so a wave writes contiguous spans instead of 4-byte fragments 3.5 KB apart.
// A step whose importance-sampling weights were prepared by the previous update kernel has no K5 launch to zero
// its statistics: the sampler's first workgroup does it (K4 accumulates into them two launches later).
#define SGPMP_ZERO_STATS(ptr)                                                                     \
    do {                                                                                          \
        if ((ptr) && blockIdx.x == 0 && blockIdx.y == 0)                                          \
            for (int i_ = threadIdx.x; i_ < SGPMP_STAT_SHARDS * 4; i_ += blockDim.x) (ptr)[i_] = 0.; \
    } while (0)
#define SGPMP_SAMPLE_TC 16
#define SGPMP_SAMPLE_PAD(n) ((n) > 4 ? 8 : 4)

#ifndef SGPMP_SAMPLE_STORE_NT      // sample_iso_kernel's stores of x = mu + y (whole rows, written once, read by another kernel): non-temporal --
#define SGPMP_SAMPLE_STORE_NT 1    // 121.4 -> 111.7 us at config 3 (same box, three passes each: profiles/r06/store_policy_ab.txt)
#endif
#if SGPMP_SAMPLE_STORE_NT
#define SGPMP_SAMPLE_STORE(p_, v_) __builtin_nontemporal_store((v_), (p_))
#else
#define SGPMP_SAMPLE_STORE(p_, v_) (*(p_) = (v_))
#endif

template <typename real, int VW>
__global__ void __launch_bounds__(256)
sample_iso_kernel(int n, int T, int S, int spw, const real* __restrict__ coef /*[T][8]*/,
                  const real* __restrict__ means, const real* __restrict__ eps, int eps_modes,
                  int eps_mode_offset, int mode_offset, uint64_t seed, uint64_t draw, int lpr_shift,
                  real* __restrict__ out, double* __restrict__ zero_stats) {
    SGPMP_ZERO_STATS(zero_stats);
    // Each WAVE owns spw = 64 / n samples (one lane per (sample, dof)) and its own rows of the LDS
    // tile, so producing a tile and flushing it need no workgroup barrier: waves run out of step and
    // one wave's stores overlap the others' arithmetic.
    typedef real vec __attribute__((ext_vector_type(VW)));
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int TC = SGPMP_SAMPLE_TC;
    const int m = blockIdx.y;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int slw = ln / n, k = ln - slw * n;               // sample within the wave, dof
    const int s0 = (blockIdx.x * (blockDim.x >> 6) + wv) * spw;   // first sample of this wave
    const int s = s0 + slw;
    const bool active = slw < spw && s < S;
    const int d = 2 * n;
    // reals per tile row: 16 d = 32 n is a multiple of the 32 banks, so the pad alone places consecutive sample
    // rows in bank space -- n lanes wide each: 4 banks apart for n <= 4, 8 apart beyond (with 4, the rows of a
    // 7-dof wave overlapped on 3 of their 7 banks: SQ_LDS_BANK_CONFLICT 3.8 M of 4.5 M LDS cycles in round 1)
    const int pitch = TC * d + SGPMP_SAMPLE_PAD(n);
    real* tile = reinterpret_cast<real*>(lds_raw) + (size_t)wv * spw * pitch;     // this wave's rows
    const size_t M = (size_t)T * d;
    const real* mu = means + (size_t)m * M;
    const real* erow = (eps && active) ? eps + ((size_t)s * eps_modes + eps_mode_offset + m) * M : nullptr;
    NoiseGen<real> gen;
    gen.init(seed, draw, (uint32_t)(mode_offset + m), (uint32_t)s, (uint32_t)k);
    // Warm the scalar cache: touch every 64-B line of the coefficient table once with all loads in
    // flight together.  The recurrence reads 32 B of coefficients per waypoint through s_load; a cold
    // line costs ~1 us, which a launch with few waves (configs 1 and 2) cannot hide behind other waves
    // (config 2: 41.6 -> 33.7 us).
    {
        const int nlines = (int)(((size_t)T * 8 * sizeof(real) + 63) >> 6);
        for (int i = 0; i < nlines; ++i) {
            uint32_t sink;
            asm volatile("s_load_dword %0, %1, %2" : "=s"(sink) : "s"(coef), "s"(i << 6) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    real p = 0, v = 0;
    real* trow = tile + (size_t)slw * pitch + k;
    const int rows = min(spw, S - s0);                   // rows of this wave that exist (may be <= 0)
    const int lpr = 1 << lpr_shift;
    const int groups = 64 >> lpr_shift;
    const int j0 = ln & (lpr - 1);
    for (int t0 = 0; t0 < T; t0 += TC) {
        const int tc = min(TC, T - t0);
        // the means of this chunk, fetched BEFORE the arithmetic so that the flush below issues stores
        // only: on gfx9 stores count in vmcnt, and a load + s_waitcnt vmcnt(0) inside the flush loop would
        // make every row wait for the previous row's HBM write to be acknowledged
        const int seg = tc * d / VW;                     // vectors per row segment
        const bool one_j = seg <= lpr;                   // each lane owns at most one vector of a row
        vec mu0 = {};
        if (one_j && j0 < seg) mu0 = *reinterpret_cast<const vec*>(mu + (size_t)t0 * d + j0 * VW);
        if (active) {
            const real* c = coef + (size_t)t0 * 8;
            real* o = trow;
            if (erow) {
                const real* e = erow + (size_t)t0 * d + k;
                for (int tt = 0; tt < tc; ++tt, c += 8, o += d, e += d) {
                    scan_step(c, e[0], e[n], p, v);
                    o[0] = p; o[n] = v;
                }
            } else {
                int tt = 0;
                for (; tt + 2 <= tc; tt += 2, c += 16, o += 2 * d) {     // t0 is even: one RNG block
                    real e[4];
                    gen.get4(t0 + tt, e);
                    scan_step(c, e[0], e[1], p, v);
                    o[0] = p; o[n] = v;
                    scan_step(c + 8, e[2], e[3], p, v);
                    o[d] = p; o[d + n] = v;
                }
                if (tt < tc) {                                       // odd tail (last waypoint)
                    real e1, e2;
                    gen.get(t0 + tt, e1, e2);
                    scan_step(c, e1, e2, p, v);
                    o[0] = p; o[n] = v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // tile rows are wave-private
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // flush: each group of 2^lpr_shift lanes owns one row at a time (no integer division)
        if (one_j) {
            if (j0 < seg)
                for (int r = ln >> lpr_shift; r < rows; r += groups) {
                    vec val = *reinterpret_cast<const vec*>(tile + (size_t)r * pitch + j0 * VW);
                    val += mu0;                                                          // x = mu + y
                    SGPMP_SAMPLE_STORE(reinterpret_cast<vec*>(out + ((size_t)m * S + s0 + r) * M + (size_t)t0 * d + j0 * VW), val);
                }
        } else {
            for (int r = ln >> lpr_shift; r < rows; r += groups) {
                const real* trow_r = tile + (size_t)r * pitch;
                real* orow = out + ((size_t)m * S + s0 + r) * M + (size_t)t0 * d;
                for (int j = j0; j < seg; j += lpr) {
                    vec val = *reinterpret_cast<const vec*>(trow_r + j * VW);
                    val += *reinterpret_cast<const vec*>(mu + (size_t)t0 * d + j * VW);
                    SGPMP_SAMPLE_STORE(reinterpret_cast<vec*>(orow + j * VW), val);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Few-wave launches (BASELINE config 1: 4 waves on 1024 SIMDs).  In the kernel above a
// wave walks its chains through all T waypoints, one Philox block after the other: with nothing else
// resident to hide behind, that is T/2 dependent ~0.5 us steps (fp64: T steps of ~0.7 us).  Here the
// noise of a 32-waypoint chunk is produced by ALL threads of the workgroup in parallel (one Philox
// block each), parked in the LDS tile, and only the cheap 7-FMA recurrence runs serially, in place;
// the tile is then flushed as row segments.  Same counter layout and the same recurrence expressions
// as above, so the two kernels return identical samples.
// (SPB samples x TC waypoints per chunk and workgroup: the launch's choice.)
// NC: the dof count as a compile-time constant (0: whatever n says) -- the serial phase is ONE wave issuing one instruction every
// four cycles with nothing to overlap, and with run-time strides a third of its instructions were address arithmetic.
template <typename real, int NC>
__global__ void __launch_bounds__(256)
sample_iso_small_kernel(int n_arg, int T, int S, int SPB, int TC, const real* __restrict__ coef, const real* __restrict__ means,
                        int mode_offset, uint64_t seed, uint64_t draw, real* __restrict__ out,
                        double* __restrict__ zero_stats) {
    const int n = NC ? NC : n_arg;
    SGPMP_ZERO_STATS(zero_stats);
    typedef real vec __attribute__((ext_vector_type(2)));
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int m = blockIdx.y, d = 2 * n, pitch = TC * d + 4;
    real* tile = reinterpret_cast<real*>(lds_raw);        // [SPB][TC*d + 4]
    real* cf = tile + (size_t)SPB * pitch;                // [TC][8] recurrence coefficients of the chunk
    const int s0 = blockIdx.x * SPB;
    const int rows = min(SPB, S - s0);
    const int chains = rows * n;
    const int tid = threadIdx.x;
    const size_t M = (size_t)T * d;
    const real* mu = means + (size_t)m * M;
    const int sl_scan = tid / n, k_scan = tid - sl_scan * n;
    real p = 0, v = 0;
#ifdef SGPMP_SMALL_STAMPS      // diagnostic build: cycles of workgroup (0, 0)'s phases over the first waypoints of its first row
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t;
#define SST_NOW() ([]() { unsigned long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); return t_; }())
#define SST(i) do { const unsigned long long n_ = SST_NOW(); st_[i] += n_ - st_t; st_t = n_; } while (0)
    st_t = SST_NOW();
#else
#define SST(i) do {} while (0)
#endif
    for (int t0 = 0; t0 < T; t0 += TC) {
        const int tc = min(TC, T - t0);
        // ---- phase 1: noise of this chunk, one RNG block per thread and trip -- left in the tile as the recurrence's NOISE TERMS
        // (rng.h scan_step_noise: the half of a step that does not depend on the state, three products per waypoint), so that the
        // serial phase is the two dependent fmas per waypoint and nothing else -- + the coefficients it needs (it must not wait for
        // global memory at every waypoint)
        for (int w = tid; w < tc * 8; w += blockDim.x) cf[w] = coef[(size_t)t0 * 8 + w];
#if defined(SGPMP_SMALL_SKIP) && SGPMP_SMALL_SKIP == 1
        if (false) {} else if (false)
#endif
        {   // (both precisions: a Philox block serves the waypoints (2b, 2b + 1) -- fp64 contexts draw the fp32 stream, widened: rng.h)
            const int pairs = (tc + 1) >> 1;              // t0 is even: blocks cover waypoints (2b, 2b+1)
            for (int w = tid; w < chains * pairs; w += blockDim.x) {
                const int c = w / pairs, b = w - c * pairs;
                const int sl = c / n, k = c - sl * n;
                const real* cw = coef + (size_t)(t0 + 2 * b) * 8;        // (rows of waypoints t0 + 2b, + 1: the table is padded to even T)
                NoiseGen<real> gen;
                gen.init(seed, draw, (uint32_t)(mode_offset + m), (uint32_t)(s0 + sl), (uint32_t)k);
                real e[4];
                gen.get4(t0 + 2 * b, e);
                real* o = tile + (size_t)sl * pitch + (2 * b) * d + k;
                scan_step_noise<real>(cw, e[0], e[1], o[0], o[n]);
                if (2 * b + 1 < tc) scan_step_noise<real>(cw + 8, e[2], e[3], o[d], o[d + n]);
            }
        }
        SST(0);
        __syncthreads();
        SST(1);
        // ---- phase 2: the recurrence, in place, one thread per chain: blocks of four waypoints, the next block's noise terms and
        // coefficients on their way out of LDS while this block's fmas run (as one loop the chain waited for an LDS round trip
        // per waypoint: 32 x ~250 cycles per chunk, half of this kernel's time)
#if defined(SGPMP_SMALL_SKIP) && SGPMP_SMALL_SKIP == 2
        if (false)
#endif
        if (tid < chains) {
            constexpr int SB = 8;
            real* o = tile + (size_t)sl_scan * pitch + k_scan;
            struct Blk { real tp[SB], tv[SB], h[SB][4]; } A, B;
            auto fetch = [&](int tb, Blk& x) {
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int tt = tb + i;                                 // (up to 2 SB waypoints past the chunk: inside the allocation -- the launch pads it -- and never used)
                    x.tp[i] = o[tt * d]; x.tv[i] = o[tt * d + n];
                    x.h[i][0] = cf[tt * 8 + 3]; x.h[i][1] = cf[tt * 8 + 4]; x.h[i][2] = cf[tt * 8 + 5]; x.h[i][3] = cf[tt * 8 + 6];
                }
            };
            // (no guard per waypoint: a short chunk is the LAST one, its steps past tc stay inside the tile -- TC is a multiple of the
            // 16 waypoints of a loop trip -- and nothing reads them or the state after them; with guards every step was a basic
            // block of its own, two taken branches per waypoint: 170 cycles per step instead of 40)
            auto run = [&](int tb, const Blk& x) {
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    scan_step_state<real>(x.h[i], x.tp[i], x.tv[i], p, v);
                    o[(tb + i) * d] = p; o[(tb + i) * d + n] = v;
                }
            };
            // two register sets, alternating: the loads of one block are issued before the other block's fmas (in source order, so
            // nothing has to be proved about the in-place stores)
            fetch(0, A);
            for (int tb = 0; tb < tc; tb += 2 * SB) {
                fetch(tb + SB, B);
                run(tb, A);
                fetch(tb + 2 * SB, A);
                run(tb + SB, B);
            }
        }
        SST(2);
        __syncthreads();
        SST(3);
        // ---- phase 3: flush row segments, x = mu + y
        const int seg = tc * d / 2;
        for (int w = tid; w < rows * seg; w += blockDim.x) {
            const int r = w / seg, j = w - r * seg;
            vec val = *reinterpret_cast<const vec*>(tile + (size_t)r * pitch + j * 2);
            val += *reinterpret_cast<const vec*>(mu + (size_t)t0 * d + j * 2);
            *reinterpret_cast<vec*>(out + ((size_t)m * S + s0 + r) * M + (size_t)t0 * d + j * 2) = val;
        }
        SST(4);
        __syncthreads();
        SST(5);
    }
#ifdef SGPMP_SMALL_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < 6) out[tid] = (real)st_[tid];
#endif
}

// ---- dense d x d factors (user-supplied Q_c^-1: gp_factor.py:25-27 accepts any matrix; per-mode precisions of
// MultiMPPrior.set_Sigma_invs): y_t = G_t eps_t + H_t y_{t-1} with full blocks, on the matrix cores.
// One wave = 16 samples of one mode = the 16 columns of a [state x sample] tile; per waypoint the two d x d products
// are eight MFMA 16x16x4 (fp32: v_mfma_f32_16x16x4_f32, fp64: v_mfma_f64_16x16x4_f64), the state block padded to
// 16.  The K index a lane serves in MFMA call kb is chosen as the ROW its accumulator register kb holds, so the
// tile y_{t-1} goes back in as the B operand of the next waypoint's H_t y_{t-1} without a single lane exchange,
// and the noise is drawn directly in that layout.
template <typename real> struct DenseTile;
template <> struct DenseTile<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    // accumulator register r of lane (q = lane >> 4, j = lane & 15) holds element [4 q + r][j]
    static __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
template <> struct DenseTile<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    // accumulator register r of lane (q, j) holds element [q + 4 r][j]
    static __device__ __forceinline__ int row(int q, int r) { return q + 4 * r; }
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
};

template <typename real>
__global__ void __launch_bounds__(256)
sample_dense_kernel(int n, int T, int S, const real* __restrict__ G, const real* __restrict__ H, size_t mode_stride,
                    const real* __restrict__ means, const real* __restrict__ eps, int eps_modes,
                    int eps_mode_offset, int mode_offset, uint64_t seed, uint64_t draw,
                    real* __restrict__ out, double* __restrict__ zero_stats) {
    SGPMP_ZERO_STATS(zero_stats);
    using DT = DenseTile<real>;
    const int D = 2 * n;
    const int m = blockIdx.y;
    G += (size_t)m * mode_stride;                        // per-mode factors (set_Sigma_invs) or 0: shared
    H += (size_t)m * mode_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;              // A operand: row i, K slot q;  B / accumulator: column i, slot q
    const int s = (blockIdx.x * 4 + wave) * 16 + i;      // this lane's sample (tile column)
    const bool live = s < S;
    const size_t M = (size_t)T * D;
    const real* mu = means + (size_t)m * M;
    real* row_out = out + ((size_t)m * S + (live ? s : 0)) * M;
    const real* erow = (eps && live) ? eps + ((size_t)s * eps_modes + eps_mode_offset + m) * M : nullptr;
    int rows[4];                                         // state rows of the four accumulator registers (= K indices served)
    NoiseGen<real> gen[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rows[r] = DT::row(q, r);
        const int k = rows[r] < n ? rows[r] : rows[r] - n;              // dof of that row (position or velocity part)
        gen[r].init(seed, draw, (uint32_t)(mode_offset + m), (uint32_t)(live ? s : 0), (uint32_t)k);
    }
    typename DT::acc_t y = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
        real e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            real v = 0;
            if (rows[r] < D && live) {
                if (erow) v = erow[(size_t)t * D + rows[r]];
                else {
                    real ep, ev;
                    gen[r].get(t, ep, ev);
                    v = rows[r] < n ? ep : ev;
                }
            }
            e[r] = v;
        }
        const real* Gt = G + (size_t)t * D * D;
        const real* Ht = H + (size_t)t * D * D;
        typename DT::acc_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int k = rows[kb];                      // K index of slot q in call kb: the row register kb holds
            const bool in = i < D && k < D;
            const real g = (in && k <= i) ? Gt[i * D + k] : (real)0;       // G_t = B_t^-1 is lower triangular
            const real h = in ? Ht[i * D + k] : (real)0;
            acc = DT::mma(g, e[kb], acc);
            acc = DT::mma(h, y[kb], acc);
        }
        y = acc;
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rows[r] < D) row_out[(size_t)t * D + rows[r]] = mu[(size_t)t * D + rows[r]] + y[r];
        }
    }
}

template <typename real>
static hipError_t sample_dispatch(int n, int T, const PriorDev& prior, uint64_t seed, uint64_t draw,
                                  const real* means, int n_modes, int mode_offset, int S,
                                  const real* eps, int eps_modes, int eps_mode_offset, real* out,
                                  hipStream_t stream, const SgpmpToggles& tg, double* zero_stats) {
    constexpr bool f64 = sizeof(real) == 8;
    if (prior.isotropic) {
        const real* coef = f64 ? (const real*)prior.iso64 : (const real*)prior.iso32;
        const int d = 2 * n;
        const int spw = 64 / n;                          // samples per wave (one lane per (sample, dof))
        int waves = (S + spw - 1) / spw;                 // waves needed per mode
        if (!eps && (long long)waves * n_modes < 256 && !tg.no_small_sampler) {
            // less than one wave per CU: noise in parallel, recurrence from LDS (config 1: 48 -> 16 us;
            // at config 2's 512 waves the standard kernel is still the faster one, 33 vs 45 us)
            // (a workgroup's share: 8 or 4 samples x up to 64 waypoints per chunk.  Measured on the reference's planar example, 15 x 128
            // x 64 fp64 -- tools/small_sampler_shapes.sh, tools/small_sampler_stamps.py: the serial recurrence of a workgroup, 64
            // steps of ~90 cycles on one wave, is what the launch waits for; more, smaller workgroups only add to it)
            // fp64: 4 samples per workgroup while that keeps the grid within two workgroups per CU -- the noise phase (a double-precision
            // Box-Muller per block) halves, 10.6 -> 9.9 us on the example; 2 samples: 13.9 us
            int tc = T > 32 ? 64 : 32, spb = (f64 && (long long)((S + 3) / 4) * n_modes <= 512) ? 4 : 8;
#ifdef SGPMP_SMALL_SHAPE_ENV      // diagnostic build (tools/small_sampler_shapes.sh): SGPMP_SMALL_SHAPE="spb,tc"
            if (const char* e = getenv("SGPMP_SMALL_SHAPE")) sscanf(e, "%d,%d", &spb, &tc);
#endif
            dim3 sgrid((S + spb - 1) / spb, n_modes);
            const size_t slds = ((size_t)spb * (tc * d + 4) + (size_t)tc * 8 + 16 * (size_t)(d + 8)) * sizeof(real);   // (+ the serial phase's read-ahead)
#define SMALL_LAUNCH(NC_) hipLaunchKernelGGL((sample_iso_small_kernel<real, NC_>), sgrid, dim3(256), slds, stream, n, T, S, spb, tc, coef, means, \
                                              mode_offset, seed, draw, out, zero_stats)
            if (n == 2) SMALL_LAUNCH(2); else if (n == 3) SMALL_LAUNCH(3); else if (n == 7) SMALL_LAUNCH(7); else SMALL_LAUNCH(0);
#undef SMALL_LAUNCH
            return hipGetLastError();
        }
        const int wpb = waves < 4 ? waves : 4;
        dim3 grid((waves + wpb - 1) / wpb, n_modes), block(64 * wpb);
        const size_t lds = (size_t)wpb * spw * (SGPMP_SAMPLE_TC * d + SGPMP_SAMPLE_PAD(n)) * sizeof(real);
        // widest store that every row segment start is aligned to: rows are T*d reals apart and
        // tiles start every 16*d reals; d is even
        const bool v16 = ((size_t)T * d * sizeof(real)) % 16 == 0;
        const int vw = (f64 || !v16) ? 2 : 4;
        int lpr_shift = 0;                               // lanes that share one row while flushing
        while ((1 << lpr_shift) < SGPMP_SAMPLE_TC * d / vw && lpr_shift < 6) ++lpr_shift;
        if (f64 || !v16)
            hipLaunchKernelGGL((sample_iso_kernel<real, 2>), grid, block, lds, stream, n, T, S, spw, coef,
                               means, eps, eps_modes, eps_mode_offset, mode_offset, seed, draw, lpr_shift, out, zero_stats);
        else
            hipLaunchKernelGGL((sample_iso_kernel<real, 4>), grid, block, lds, stream, n, T, S, spw, coef,
                               means, eps, eps_modes, eps_mode_offset, mode_offset, seed, draw, lpr_shift, out, zero_stats);
        return hipGetLastError();
    }
    const real* G = f64 ? (const real*)prior.G : (const real*)prior.G32;
    const real* H = f64 ? (const real*)prior.H : (const real*)prior.H32;
    if (prior.n_factor_modes > 0 && n_modes > prior.n_factor_modes) return hipErrorInvalidValue;
    const size_t mode_stride = prior.n_factor_modes > 0 ? (size_t)T * 4 * n * n : 0;
    if (n < 1 || n > 8) return hipErrorInvalidValue;     // state block = one 16 x 16 tile
    dim3 grid((S + 63) / 64, n_modes), block(256);       // 4 waves x 16 samples
    hipLaunchKernelGGL((sample_dense_kernel<real>), grid, block, 0, stream, n, T, S, G, H, mode_stride, means, eps,
                       eps_modes, eps_mode_offset, mode_offset, seed, draw, out, zero_stats);
    return hipGetLastError();
}

hipError_t launch_sample(int dtype, int n, int T, const PriorDev& prior, uint64_t seed, uint64_t draw,
                         const void* means, int n_modes, int mode_offset, int n_samples,
                         const void* eps, int eps_modes, int eps_mode_offset, void* out,
                         hipStream_t stream, const SgpmpToggles& tg, double* zero_stats) {
    if (dtype == SGPMP_F64)
        return sample_dispatch<double>(n, T, prior, seed, draw, (const double*)means, n_modes,
                                       mode_offset, n_samples, (const double*)eps, eps_modes,
                                       eps_mode_offset, (double*)out, stream, tg, zero_stats);
    return sample_dispatch<float>(n, T, prior, seed, draw, (const float*)means, n_modes, mode_offset,
                                  n_samples, (const float*)eps, eps_modes, eps_mode_offset,
                                  (float*)out, stream, tg, zero_stats);
}

// ---------------------------------------------------------------------------------- the noise itself
// eps[s][m][t * d + k] (position noise of dof k at waypoint t), eps[s][m][t * d + n + k] (velocity noise): torch.randn(S, P, M)'s
// layout (multivariate_normal.py:250-253), the values every sampling kernel of this library draws for (seed, draw, global
// particle mode_offset + m, sample s) -- the same calls, NoiseGen<real>::get4 on the block of waypoints (2b, 2b + 1).
template <typename real>
__global__ void __launch_bounds__(256)
noise_dump_kernel(int n, int T, int n_modes, int mode_offset, int S, uint64_t seed, uint64_t draw, real* __restrict__ out) {
    const long long pairs = (T + 1) / 2, total = (long long)S * n_modes * pairs * n;
    const size_t d = 2 * (size_t)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % n);
        long long r = i / n;
        const int b = (int)(r % pairs);
        r /= pairs;
        const int m = (int)(r % n_modes), s = (int)(r / n_modes);
        NoiseGen<real> gen;
        gen.init(seed, draw, (uint32_t)(mode_offset + m), (uint32_t)s, (uint32_t)k);
        real e[4];
        gen.get4(2 * b, e);
        real* o = out + ((size_t)s * n_modes + m) * (size_t)T * d + (size_t)(2 * b) * d;
        o[k] = e[0]; o[n + k] = e[1];
        if (2 * b + 1 < T) { o[d + k] = e[2]; o[d + n + k] = e[3]; }
    }
}

hipError_t launch_noise(int dtype, int n, int T, int n_modes, int mode_offset, int S, uint64_t seed, uint64_t draw, void* out,
                        hipStream_t stream) {
    const long long total = (long long)S * n_modes * ((T + 1) / 2) * n;
    if (total <= 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL(noise_dump_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, stream, n, T, n_modes, mode_offset, S, seed, draw, (double*)out);
    else
        hipLaunchKernelGGL(noise_dump_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, n, T, n_modes, mode_offset, S, seed, draw, (float*)out);
    return hipGetLastError();
}
