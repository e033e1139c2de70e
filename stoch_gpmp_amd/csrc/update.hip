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

#include "sgpmp_internal.h"
#include "update_common.h"

template <typename T> __device__ __forceinline__ T block_reduce(T v, T* scratch, bool is_min) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const T o = __shfl_xor(v, off, 64);
        v = is_min ? (o < v ? o : v) : (v + o);
    }
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < nw; ++i) r = is_min ? (scratch[i] < r ? scratch[i] : r) : (r + scratch[i]);
    __syncthreads();
    return r;
}

// VW = elements per thread and load (4 when M % 4 == 0, else 2; M = T * 2n is always even).
template <typename real, typename cost_t, int VW>
__global__ void __launch_bounds__(256)
update_kernel(int M, int S, const cost_t* __restrict__ costs, const real* __restrict__ samples,
              real* __restrict__ means, double temperature, double step_size,
              real* __restrict__ weights, real* __restrict__ grad, real* __restrict__ means_prev,
              double* __restrict__ stats, IswNext<real> nx, real* __restrict__ means_copy) {
    typedef real vec __attribute__((ext_vector_type(VW)));
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* w = reinterpret_cast<double*>(lds_raw);                  // [S] weights
    int* idx = reinterpret_cast<int*>(lds_raw + (size_t)S * 8);      // [S] samples with weight != 0
    real* mu_lds = reinterpret_cast<real*>(lds_raw + (((size_t)S * 12 + 15) & ~(size_t)15));   // [M] new means (tail)
    __shared__ double scratch[8];
    __shared__ int nnz_s;
    const int p = blockIdx.x;
    const cost_t* c = costs + (size_t)p * S;
    // (issued first, consumed last: the four distinct entries of an isotropic prior's one-step precision -- the tail
    // below would otherwise pay two dependent global loads per element)
    const bool iso_tail = nx.out && nx.isotropic;
    const int nd = 2 * nx.n;
    const double q00 = iso_tail ? nx.Qinv[0] : 0., q01 = iso_tail ? nx.Qinv[nx.n] : 0.;
    const double q10 = iso_tail ? nx.Qinv[nx.n * nd] : 0., q11 = iso_tail ? nx.Qinv[nx.n * nd + nx.n] : 0.;

    // softmax(-c / temperature) exactly as torch.softmax: exp(z - max z) / sum
    double zmax = -1.7976931348623157e308, csum = 0., cmin = 1.7976931348623157e308;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double cv = (double)c[s];
        const double z = -cv / temperature;
        w[s] = z;
        zmax = z > zmax ? z : zmax;
        cmin = cv < cmin ? cv : cmin;
        csum += cv;
    }
    // one combined reduction for (max z, sum c, min c): three block reductions cost nine barriers
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double oz = __shfl_xor(zmax, off, 64), os = __shfl_xor(csum, off, 64), om = __shfl_xor(cmin, off, 64);
            zmax = oz > zmax ? oz : zmax;
            csum += os;
            cmin = om < cmin ? om : cmin;
        }
        __shared__ double red3[3 * 4];
        if (lane == 0) { red3[wave] = zmax; red3[4 + wave] = csum; red3[8 + wave] = cmin; }
        __syncthreads();
        zmax = red3[0]; csum = red3[4]; cmin = red3[8];
        for (int i = 1; i < nw; ++i) {
            zmax = red3[i] > zmax ? red3[i] : zmax;
            csum += red3[4 + i];
            cmin = red3[8 + i] < cmin ? red3[8 + i] : cmin;
        }
    }
    double part = 0.;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double e = exp(w[s] - zmax);
        w[s] = e;
        part += e;
    }
    double Z;
    {   // (one barrier: `scratch` is written once in this kernel)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) scratch[wave] = part;
        __syncthreads();
        Z = scratch[0];
        for (int i = 1; i < nw; ++i) Z += scratch[i];
    }
    if (stats) {
        const double tot = csum, mn = cmin;
        if (threadIdx.x == 0) {
            // 64 shards of 4 doubles: a thousand workgroups adding to one address serialise
            // at the memory side (~30 us); the consumer sums the shards
            double* sh = stats + (blockIdx.x & (SGPMP_STAT_SHARDS - 1)) * 4;
            atomicAdd(&sh[0], tot);
            atomicAdd(&sh[1], mn);
            atomicAdd(&sh[2], 1.0);
        }
    }
    const double invZ = 1. / Z;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double ws = w[s] * invZ;
        w[s] = ws;
        if (weights) weights[(size_t)p * S + s] = (real)ws;
    }
    __syncthreads();
    // Samples whose weight underflowed to exactly 0 contribute exactly 0 to the sum below, so their
    // rows need not be read at all (with the reference's hyper-parameters the softmax is one-hot
    // and this turns a pass over [S, M] into a pass over one row).  Order-preserving compaction.
    if (threadIdx.x < 64) {
        int base = 0;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + (int)threadIdx.x;
            const bool nz = s < S && w[s] != 0.;
            const unsigned long long mask = __ballot(nz);
            const int pos = base + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
            if (nz) idx[pos] = s;
            base += __popcll(mask);
        }
        if (threadIdx.x == 0) nnz_s = base;
    }
    __syncthreads();
    const int nnz = nnz_s;

    const real* X = samples + (size_t)p * S * M;
    real* mu = means + (size_t)p * M;
    for (int m = threadIdx.x * VW; m < M; m += blockDim.x * VW) {
        const vec mu_m = *reinterpret_cast<const vec*>(mu + m);
        double acc[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = 0.;
        int k = 0;
        for (; k + 4 <= nnz; k += 4) {                   // four rows in flight per thread
            vec v[4];
            double ws[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = idx[k + u];
                ws[u] = w[s];
                v[u] = *reinterpret_cast<const vec*>(X + (size_t)s * M + m);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < VW; ++i) acc[i] = fma(ws[u], (double)(v[u][i] - mu_m[i]), acc[i]);
        }
        for (; k < nnz; ++k) {
            const int s = idx[k];
            const double ws = w[s];
            const vec v = *reinterpret_cast<const vec*>(X + (size_t)s * M + m);
#pragma unroll
            for (int i = 0; i < VW; ++i) acc[i] = fma(ws, (double)(v[i] - mu_m[i]), acc[i]);
        }
        vec g, mn;
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            g[i] = (real)acc[i];
            mn[i] = (real)fma(step_size, acc[i], (double)mu_m[i]);
        }
        if (grad) *reinterpret_cast<vec*>(grad + (size_t)p * M + m) = g;
        if (means_prev) *reinterpret_cast<vec*>(means_prev + (size_t)p * M + m) = mu_m;
        *reinterpret_cast<vec*>(mu + m) = mn;
        if (means_copy) *reinterpret_cast<vec*>(means_copy + (size_t)p * M + m) = mn;   // (snapshot for the side stream's statistics)
        if (nx.out) *reinterpret_cast<vec*>(mu_lds + m) = mn;
    }
    // The NEXT iteration's importance-sampling weights, from the means just written (K5's arithmetic, same
    // function): sgpmp_step then starts with the sampler + sweep launch instead of a K5 launch, provided the
    // caller vouches that nobody touched the means in between (SGPMP_STEP_MEANS_KEPT).
    if (nx.out) {
        __syncthreads();                                  // the new means of this particle are in LDS
        const int d = 2 * nx.n, Tn = M / d;
        const real* mu_new = mu_lds;
        for (int e = threadIdx.x; e < (Tn + 1) * d; e += blockDim.x)
            nx.out[(size_t)p * (Tn + 1) * d + e] =
                iso_tail ? is_weight_elem_iso<real>(nx.n, Tn, mu_new, q00, q01, q10, q11, nx.ks, nx.kg, nx.dt, temperature, e)
                         : is_weight_elem<real>(nx.n, Tn, mu_new, nx.Qinv, nx.ks, nx.kg, nx.dt, temperature, nx.isotropic, e);
    }
}

hipError_t launch_update(int dtype, int n, int T, int P, int S, const void* costs, int costs_dtype,
                         const void* samples, void* means, double temperature, double step_size,
                         void* weights, void* grad, void* means_prev, double* stats,
                         hipStream_t stream, hipEvent_t done, const PriorDev* isw_prior, void* isw_next,
                         bool* isw_written, void* means_copy) {
    const int M = T * 2 * n;
    size_t lds = (size_t)S * (sizeof(double) + sizeof(int));
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
                          (REAL*)means_copy)
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
// grid = (ceil(M / 256), particle slices); a slice adds its partial sums with one atomic per element and goal run.
template <typename real>
__global__ void __launch_bounds__(256)
mode_stats_kernel(int M, int P, long long p_offset, int nppg, int G, const real* __restrict__ means,
                  double* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(P, p0 + per);
    if (p0 >= p1) return;
    int g = (int)min((long long)(G - 1), (p_offset + p0) / nppg);
    double s1 = 0., s2 = 0.;
    int cnt = 0;
    auto flush = [&]() {
        if (m < M) {
            atomicAdd(&out[((size_t)g * (M + 1) + m) * 2], s1);
            atomicAdd(&out[((size_t)g * (M + 1) + m) * 2 + 1], s2);
        } else if (m == M) {
            atomicAdd(&out[((size_t)g * (M + 1) + M) * 2], (double)cnt);
        }
        s1 = s2 = 0.; cnt = 0;
    };
    for (int p = p0; p < p1; ++p) {
        const int gp = (int)min((long long)(G - 1), (p_offset + p) / nppg);
        if (gp != g) { flush(); g = gp; }
        if (m < M) {
            const double v = (double)means[(size_t)p * M + m];
            s1 += v; s2 += v * v;
        }
        ++cnt;
    }
    flush();
}

hipError_t launch_mode_stats(int dtype, int n, int T, int P, long long p_offset, int nppg, int G, const void* means,
                             double* out, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    const int M = T * 2 * n;
    const int slices = P >= 32 ? 32 : P;
    dim3 grid((unsigned)((M + 1 + 255) / 256), (unsigned)slices), block(256);
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
