// K4 -- exponentiated-cost reweight and mean update;  K5 -- importance-sampling weights.
//
// K4 replaces StochGPMP._update_distribution (planner.py:263-275):
//     w = softmax(-costs / temperature, dim=samples);  grad = sum_s w_s (x_s - mu);
//     mu += step_size * grad
// (the reference then rebuilds a MultivariateNormal -> a [P,M,M] Cholesky -- per iteration; the
// factor does not depend on the mean, so nothing is left to do here).
// One workgroup per particle: the S costs are reduced in LDS, then threads stride over the M = T*d
// trajectory elements, each streaming its column of the [S, M] sample block (coalesced rows).
//
// K5 replaces the Sigma_inv @ mu part of StochGPMP._get_costs (planner.py:233-236) in factored form:
//     x^T Sigma^-1 mu = (A x)^T a,   a = blkdiag(K_s, Q^-1 x (T-1), K_g) (A mu)
// with A x = (x_0, e_0(x), .., e_{T-2}(x), x_{T-1}) (mp_priors_multi.py:185-198); `a` is computed in
// fp64 once per particle per iteration and consumed by the cost sweep (K3).  The last block
// (x_{T-1} . K_g mu_{T-1}) is re-expressed on (x_0, e_0, ..) and folded into the first T blocks, so
// K3 needs exactly one weight vector per waypoint.
#include <hip/hip_ext.h>
#include <cstring>

#include "sgpmp_internal.h"
#include "update_common.h"

// LDS of update_kernel without the regeneration: weights + indices (+ the new means for the tail when they fit)
static size_t update_base_lds(int dtype, int n, int T, int S) {
    const size_t M = (size_t)T * 2 * n, lds = (size_t)S * (sizeof(double) + sizeof(int));
    const size_t with_tail = ((lds + 15) & ~(size_t)15) + M * (dtype == SGPMP_F64 ? 8 : 4);
    return with_tail + 256 <= 65536 ? with_tail : lds;
}

// Rows update_kernel can regenerate per round for this shape (4, 2, 1), or 0: a store-free step is not possible (the caller
// then stores the samples as always).  fp32, M a multiple of 4 (16-byte elements), T even (a Philox block is two waypoints).
int update_regen_rows(int dtype, int n, int T, int S, int recipe) {
    if (dtype != SGPMP_F32 || (T * 2 * n) % 4 != 0 || T % 2 != 0 || recipe < 1 || recipe > 2) return 0;
    const size_t base = (update_base_lds(dtype, n, T, S) + 15) & ~(size_t)15;
    for (int R = 4; R >= 1; R >>= 1)
        if (base + regen_lds_bytes(recipe, T, n, R) + 256 <= 65536) return R;
    return 0;
}

// does update_kernel's scratch leave room for the folded end-effector term (S more doubles)?
bool update_ee_fold_fits(int dtype, int n, int T, int S) {
    return ((update_base_lds(dtype, n, T, S) + 15) & ~(size_t)15) + (size_t)S * sizeof(double) + 256 <= 65536;
}

hipError_t launch_update(int dtype, int n, int T, int P, int S, const void* costs, int costs_dtype,
                         const void* samples, void* means, double temperature, double step_size,
                         void* weights, void* grad, void* means_prev, double* stats,
                         hipStream_t stream, hipEvent_t done, const PriorDev* isw_prior, void* isw_next,
                         bool* isw_written, void* means_copy, const float* part, unsigned* nnz, unsigned nnz_threshold,
                         const RegenHost* regen, const EeFoldHost* ee) {
    const int M = T * 2 * n;
    size_t lds = (size_t)S * (sizeof(double) + sizeof(int));
    // (the softmax coefficients of the partials reuse the index array as doubles: 8 bytes per group of 8 rows fit its 4 S)
    if (isw_prior) {
        // + the new means for the tail.  Shapes whose weights fit the 64 KB of a workgroup but not together with
        // the means (e.g. S = 4096, T = 512, n = 7 in fp32: 77 KB) run WITHOUT the tail: the next step then
        // computes its importance-sampling weights with K5, as before the tail existed.
        const size_t with_tail = ((lds + 15) & ~(size_t)15) + (size_t)M * (dtype == SGPMP_F64 ? 8 : 4);
        if (with_tail + 256 <= 65536) lds = with_tail;
        else isw_prior = nullptr;
    }
    if (isw_written) *isw_written = isw_prior != nullptr && P > 0;
    if (P <= 0) return hipSuccess;
    // store-free step: the rows with weight are regenerated behind the kernel's usual scratch
    RegenArgs rg;
    std::memset(&rg, 0, sizeof(rg));
    unsigned regen_off = 0;
    if (regen && regen->recipe != 0) {
        if (dtype != SGPMP_F32 || !nnz) return hipErrorInvalidValue;
        const int R = update_regen_rows(dtype, n, T, S, regen->recipe);
        if (R < 1) return hipErrorInvalidValue;                  // (the caller asked update_regen_rows before the launch that skipped the stores)
        rg.recipe = regen->recipe; rg.N = n; rg.L = regen->L; rg.R = R; rg.seed = regen->seed; rg.draw = regen->draw;
        rg.mode_offset = regen->mode_offset; rg.coef = regen->coef; rg.pre = regen->pre; rg.store_threshold = regen->store_threshold;
        regen_off = (unsigned)((update_base_lds(dtype, n, T, S) + 15) & ~(size_t)15);
        // (the scratch sized WITH the tail's means: without them the offset is merely generous)
        if (lds < regen_off) lds = regen_off;
        lds = regen_off + regen_lds_bytes(rg.recipe, T, n, R);
    }
    // the step's end-effector goal term inside this kernel: S doubles behind everything else
    unsigned ee_off = 0;
    if (ee && ee->term) {
        ee_off = (unsigned)((lds + 15) & ~(size_t)15);
        lds = ee_off + (size_t)S * sizeof(double);
        if (lds + 256 > 65536) return hipErrorInvalidValue;      // (the caller checks update_ee_fold_fits first)
    }
    dim3 grid(P), block(256);
    // `done` (multi-GPU statistics): the event is signalled by this kernel's own dispatch packet
    // (hipExtLaunchKernelGGL stop event) instead of a separate barrier packet behind it
#define UPD(REAL, COST, VW)                                                                          \
    hipExtLaunchKernelGGL((update_kernel<REAL, COST, VW>), grid, block, (unsigned)lds, stream, (hipEvent_t) nullptr, \
                          done, 0u, M, S, (const COST*)costs, (const REAL*)samples, (REAL*)means, temperature, \
                          step_size, (REAL*)weights, (REAL*)grad, (REAL*)means_prev, stats,          \
                          IswNext<REAL>{isw_prior ? (REAL*)isw_next : nullptr, isw_prior ? isw_prior->Qinv : nullptr, \
                                        isw_prior ? isw_prior->ks : 0., isw_prior ? isw_prior->kg : -1., \
                                        isw_prior ? isw_prior->dt : 0., n, isw_prior ? isw_prior->isotropic : 1}, \
                          (REAL*)means_copy, (dtype == SGPMP_F32 && M % 4 == 0) ? part : (const float*)nullptr, (S + 7) / 8, nnz, nnz_threshold, rg, regen_off, \
                          (ee && ee->term) ? EeFold<REAL>{ee->d_chain, make_ee_target<REAL>(*ee->term), n, T, (REAL*)ee->costs} : EeFold<REAL>{nullptr, EeTarget<REAL>{}, n, T, nullptr}, ee_off)
    if (dtype == SGPMP_F64) {
        if (M % 4 == 0) UPD(double, double, 4); else UPD(double, double, 2);
    } else if (costs_dtype == SGPMP_F64) {
        if (M % 4 == 0) UPD(float, double, 4); else UPD(float, double, 2);
    } else {
        if (M % 4 == 0) UPD(float, float, 4); else UPD(float, float, 2);
    }
#undef UPD
    return hipGetLastError();
}

// Per-goal statistics of the particle means -- what the multi-GPU all-reduce carries besides the cost sums (SURVEY.md
// 8e: "per-goal sum_p mu_p and sum_p mu_p mu_p^T-diagonal"; north_star: "weighted-mean/covariance statistics"): the
// modes of the trajectory distribution are the goals (p = g * nppg + k, planner.py:215), and the first two moments of
// a mode's particles are what a covariance adaptation (MultiMPPrior.set_Sigma_invs, mp_priors_multi.py:125-128) or a
// mode summary would consume.  out [G][M + 1][2] doubles is ACCUMULATED into (zero it first): [g][m] = (sum of
// mu_p[m], sum of mu_p[m]^2) over this shard's particles of goal g, [g][M] = (their number, 0).
// grid = (ceil((M + 1) / 64), G), 1024 threads: lane = element, the 16 waves take the goal's particles of this shard
// round-robin; the waves' partial sums meet in LDS and are added in wave order -- NO atomics, so the sums are bitwise
// reproducible run to run for a given sharding (round 3's version added up to 32 slices with fp64 atomics in whatever
// order they arrived).  Across different shardings the order of the additions differs: equal to rounding only.
#define SGPMP_MS_WAVES 16
template <typename real>
__global__ void __launch_bounds__(64 * SGPMP_MS_WAVES)
mode_stats_kernel(int M, int P, long long p_offset, int nppg, int G, const real* __restrict__ means,
                  double* __restrict__ out) {
    __shared__ double red[SGPMP_MS_WAVES][64][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + lane;
    const int g = blockIdx.y;
    // local particles of goal g: global [g nppg, (g + 1) nppg) -- the last goal also takes what lies beyond (as the
    // clamp min(G - 1, p / nppg) of the cost kernels does) -- cut to this shard [p_offset, p_offset + P)
    const long long lo = (long long)g * nppg, hi = g == G - 1 ? (long long)1 << 62 : lo + nppg;
    const long long a = lo > p_offset ? lo - p_offset : 0;
    const long long b = hi - p_offset < (long long)P ? hi - p_offset : (long long)P;
    double s1 = 0., s2 = 0.;
    if (m < M) {
        long long p = a + wave;
        for (; p + 3 * SGPMP_MS_WAVES < b; p += 4 * SGPMP_MS_WAVES) {            // four independent loads in flight
            const double v0 = (double)means[(size_t)p * M + m], v1 = (double)means[(size_t)(p + SGPMP_MS_WAVES) * M + m];
            const double v2 = (double)means[(size_t)(p + 2 * SGPMP_MS_WAVES) * M + m];
            const double v3 = (double)means[(size_t)(p + 3 * SGPMP_MS_WAVES) * M + m];
            s1 += v0; s2 += v0 * v0; s1 += v1; s2 += v1 * v1; s1 += v2; s2 += v2 * v2; s1 += v3; s2 += v3 * v3;
        }
        for (; p < b; p += SGPMP_MS_WAVES) {
            const double v = (double)means[(size_t)p * M + m];
            s1 += v; s2 += v * v;
        }
    }
    red[wave][lane][0] = s1; red[wave][lane][1] = s2;
    __syncthreads();
    if (wave == 0) {
        double t1 = red[0][lane][0], t2 = red[0][lane][1];
#pragma unroll
        for (int w = 1; w < SGPMP_MS_WAVES; ++w) { t1 += red[w][lane][0]; t2 += red[w][lane][1]; }
        double* o = out + ((size_t)g * (M + 1) + m) * 2;
        if (m < M) { o[0] += t1; o[1] += t2; }                                   // (ACCUMULATES, as documented: one writer per element)
        else if (m == M) o[0] += (double)(b > a ? b - a : 0);
    }
}

hipError_t launch_mode_stats(int dtype, int n, int T, int P, long long p_offset, int nppg, int G, const void* means,
                             double* out, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    const int M = T * 2 * n;
    dim3 grid((unsigned)((M + 1 + 63) / 64), (unsigned)G), block(64 * SGPMP_MS_WAVES);
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((mode_stats_kernel<double>), grid, block, 0, stream, M, P, p_offset, nppg, G, (const double*)means, out);
    else
        hipLaunchKernelGGL((mode_stats_kernel<float>), grid, block, 0, stream, M, P, p_offset, nppg, G, (const float*)means, out);
    return hipGetLastError();
}

// statistics of the second particle half of a pipelined step (api.hip, StepPipe) added to the first half's
__global__ void stats_add_kernel(double* __restrict__ dst, const double* __restrict__ src) {
    if (threadIdx.x < SGPMP_STAT_SHARDS * 4) dst[threadIdx.x] += src[threadIdx.x];
}

hipError_t launch_stats_add(double* dst, const double* src, hipStream_t stream) {
    hipLaunchKernelGGL(stats_add_kernel, dim3(1), dim3(SGPMP_STAT_SHARDS * 4), 0, stream, dst, src);
    return hipGetLastError();
}

// K5: one thread per (particle, block row t in [0,T], component i).
template <typename real>
__global__ void is_weights_kernel(int n, int T, int P, const real* __restrict__ means,
                                  const double* __restrict__ Qinv, double ks, double kg, double dt,
                                  double temperature, int isotropic, real* __restrict__ out,
                                  double* __restrict__ zero_stats) {
    const int d = 2 * n;
    // grid = (ceil((T+1) d / 256), P): 32-bit index arithmetic (64-bit div / mod cost more than the math)
    const int p = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_stats && p == 0 && e < SGPMP_STAT_SHARDS * 4) zero_stats[e] = 0.;   // this step's statistics start from zero
    if (e >= (T + 1) * d) return;
    out[(size_t)p * (T + 1) * d + e] = is_weight_elem<real>(n, T, means + (size_t)p * T * d, Qinv, ks, kg, dt,
                                                             temperature, isotropic, e);
}

hipError_t launch_is_weights(int dtype, int n, int T, const PriorDev& prior, const void* means,
                             int n_particles, double temperature, void* out, double* zero_stats,
                             hipStream_t stream) {
    const long long total = (long long)n_particles * (T + 1) * 2 * n;
    if (total <= 0) return hipSuccess;
    const int block = 256;
    const dim3 grid((unsigned)(((T + 1) * 2 * n + block - 1) / block), (unsigned)n_particles);
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((is_weights_kernel<double>), grid, dim3(block), 0, stream, n, T,
                           n_particles, (const double*)means, prior.Qinv, prior.ks, prior.kg, prior.dt,
                           temperature, prior.isotropic, (double*)out, zero_stats);
    else
        hipLaunchKernelGGL((is_weights_kernel<float>), grid, dim3(block), 0, stream, n, T,
                           n_particles, (const float*)means, prior.Qinv, prior.ks, prior.kg, prior.dt,
                           temperature, prior.isotropic, (float*)out, zero_stats);
    return hipGetLastError();
}
