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
// fp64 once per particle per iteration and consumed by the cost sweep (K3).
#include "sgpmp_internal.h"

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

template <typename real, typename cost_t>
__global__ void __launch_bounds__(256)
update_kernel(int M, int S, const cost_t* __restrict__ costs, const real* __restrict__ samples,
              real* __restrict__ means, double temperature, double step_size,
              real* __restrict__ weights, real* __restrict__ grad, real* __restrict__ means_prev,
              double* __restrict__ stats) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* w = reinterpret_cast<double*>(lds_raw);      // [S]
    __shared__ double scratch[8];
    const int p = blockIdx.x;
    const cost_t* c = costs + (size_t)p * S;

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
    zmax = -block_reduce<double>(-zmax, scratch, true);
    double part = 0.;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double e = exp(w[s] - zmax);
        w[s] = e;
        part += e;
    }
    const double Z = block_reduce<double>(part, scratch, false);
    if (stats) {
        const double tot = block_reduce<double>(csum, scratch, false);
        const double mn = block_reduce<double>(cmin, scratch, true);
        if (threadIdx.x == 0) {
            atomicAdd(&stats[0], tot);
            atomicAdd(&stats[1], mn);
            atomicAdd(&stats[2], 1.0);
        }
    }
    const double invZ = 1. / Z;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double ws = w[s] * invZ;
        w[s] = ws;
        if (weights) weights[(size_t)p * S + s] = (real)ws;
    }
    __syncthreads();

    const real* X = samples + (size_t)p * S * M;
    real* mu = means + (size_t)p * M;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const real mu_m = mu[m];
        double acc = 0.;
        for (int s = 0; s < S; ++s) acc += w[s] * (double)(X[(size_t)s * M + m] - mu_m);
        if (grad) grad[(size_t)p * M + m] = (real)acc;
        if (means_prev) means_prev[(size_t)p * M + m] = mu_m;
        mu[m] = (real)((double)mu_m + step_size * acc);
    }
}

hipError_t launch_update(int dtype, int n, int T, int P, int S, const void* costs, int costs_dtype,
                         const void* samples, void* means, double temperature, double step_size,
                         void* weights, void* grad, void* means_prev, double* stats,
                         hipStream_t stream) {
    const int M = T * 2 * n;
    const size_t lds = (size_t)S * sizeof(double);
    if (P <= 0) return hipSuccess;
    dim3 grid(P), block(256);
    if (dtype == SGPMP_F64) {
        hipLaunchKernelGGL((update_kernel<double, double>), grid, block, lds, stream, M, S,
                           (const double*)costs, (const double*)samples, (double*)means, temperature,
                           step_size, (double*)weights, (double*)grad, (double*)means_prev, stats);
    } else if (costs_dtype == SGPMP_F64) {
        hipLaunchKernelGGL((update_kernel<float, double>), grid, block, lds, stream, M, S,
                           (const double*)costs, (const float*)samples, (float*)means, temperature,
                           step_size, (float*)weights, (float*)grad, (float*)means_prev, stats);
    } else {
        hipLaunchKernelGGL((update_kernel<float, float>), grid, block, lds, stream, M, S,
                           (const float*)costs, (const float*)samples, (float*)means, temperature,
                           step_size, (float*)weights, (float*)grad, (float*)means_prev, stats);
    }
    return hipGetLastError();
}

// K5: one thread per (particle, block row t in [0,T], component i).
template <typename real>
__global__ void is_weights_kernel(int n, int T, int P, const real* __restrict__ means,
                                  const double* __restrict__ Qinv, double ks, double kg, double dt,
                                  double temperature, real* __restrict__ out,
                                  double* __restrict__ zero_stats) {
    const int d = 2 * n;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_stats && idx < 4) zero_stats[idx] = 0.;      // this step's statistics start from zero
    const long long total = (long long)P * (T + 1) * d;
    if (idx >= total) return;
    const int i = (int)(idx % d);
    const int t = (int)((idx / d) % (T + 1));
    const int p = (int)(idx / ((long long)d * (T + 1)));
    const real* mu = means + (size_t)p * T * d;
    double v;
    if (t == 0) {
        v = ks * (double)mu[i];
    } else if (t == T) {
        v = kg >= 0. ? kg * (double)mu[(size_t)(T - 1) * d + i] : 0.;
    } else {
        const real* a = mu + (size_t)(t - 1) * d;      // mu_{t-1}
        const real* b = mu + (size_t)t * d;            // mu_t
        v = 0.;
        for (int j = 0; j < d; ++j) {
            double e;                                   // e_{t-1}(mu)_j = (mu_t - Phi mu_{t-1})_j
            if (j < n) e = (double)b[j] - ((double)a[j] + dt * (double)a[n + j]);
            else e = (double)b[j] - (double)a[j];
            v += Qinv[i * d + j] * e;
        }
    }
    out[idx] = (real)(temperature * v);
}

hipError_t launch_is_weights(int dtype, int n, int T, const PriorDev& prior, const void* means,
                             int n_particles, double temperature, void* out, double* zero_stats,
                             hipStream_t stream) {
    const long long total = (long long)n_particles * (T + 1) * 2 * n;
    if (total <= 0) return hipSuccess;
    const int block = 256;
    const unsigned grid = (unsigned)((total + block - 1) / block);
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((is_weights_kernel<double>), dim3(grid), dim3(block), 0, stream, n, T,
                           n_particles, (const double*)means, prior.Qinv, prior.ks, prior.kg, prior.dt,
                           temperature, (double*)out, zero_stats);
    else
        hipLaunchKernelGGL((is_weights_kernel<float>), dim3(grid), dim3(block), 0, stream, n, T,
                           n_particles, (const float*)means, prior.Qinv, prior.ks, prior.kg, prior.dt,
                           temperature, (float*)out, zero_stats);
    return hipGetLastError();
}
