// Arithmetic of the update kernel (update.hip): the importance-sampling weight element (K5, also its own kernel), the
// descriptor of "the next step's weights", the update of one particle, and the regeneration of sample rows from their
// noise keys for store-free steps.
#pragma once
#include "sgpmp_internal.h"
#include "rng.h"

// K5 element: component e = t * d + i of one particle's importance-sampling weight vector [T+1][d], from that
// particle's means mu [T][d] (fp64 arithmetic on the context-dtype means).
template <typename real>
__device__ __forceinline__ real is_weight_elem(int n, int T, const real* __restrict__ mu, const double* __restrict__ Qinv,
                                               double ks, double kg, double dt, double temperature, int isotropic, int e) {
    const int d = 2 * n;
    const int t = e / d, i = e - t * d;
    if (t == T) return (real)0;         // spare block (kept for layout stability): unused by K3
    double v;
    if (t == 0) {
        v = ks * (double)mu[i];
    } else {
        const real* a = mu + (size_t)(t - 1) * d;      // mu_{t-1}
        const real* b = mu + (size_t)t * d;            // mu_t
        v = 0.;
        if (isotropic) {                                // Q^-1 = q (x) I_n: two non-zeros per row
            const int k = i < n ? i : i - n;
            const double ep = (double)b[k] - ((double)a[k] + dt * (double)a[n + k]);
            const double ev = (double)b[n + k] - (double)a[n + k];
            v = Qinv[i * d + k] * ep + Qinv[i * d + n + k] * ev;
        } else {
            for (int j = 0; j < d; ++j) {
                double ej;                              // e_{t-1}(mu)_j = (mu_t - Phi mu_{t-1})_j
                if (j < n) ej = (double)b[j] - ((double)a[j] + dt * (double)a[n + j]);
                else ej = (double)b[j] - (double)a[j];
                v += Qinv[i * d + j] * ej;
            }
        }
    }
    // goal block b = K_g mu_{T-1} of A x = (x_0, e_0.., x_{T-1}) folded into the per-waypoint weights:
    // x_{T-1} = sum_j Phi^{T-2-j} e_j + Phi^{T-1} x_0 and (Phi^T)^k (b_p, b_v) = (b_p, k dt b_p + b_v)
    if (kg >= 0.) {
        const real* last = mu + (size_t)(T - 1) * d;
        const double k = (double)(T - 1 - t);
        v += (i < n) ? kg * (double)last[i] : kg * (k * dt * (double)last[i - n] + (double)last[i]);
    }
    return (real)(temperature * v);
}

// Importance-sampling weight element for an ISOTROPIC prior with the four distinct entries of its one-step precision
// Q^-1 = [[q00, q01], [q10, q11]] (x) I_n held in registers -- is_weight_elem's arithmetic (update_common.h) on the
// same numbers, without its two global loads per element: under the launch's store traffic every dependent
// vector-memory round trip of the in-launch update costs ~3 us, and the update kernel saves a dependent trip per element.
template <typename real>
__device__ __forceinline__ real is_weight_elem_iso(int N, int T, const real* __restrict__ mu, double q00, double q01, double q10,
                                                   double q11, double ks, double kg, double dt, double temperature, int e) {
    const int d = 2 * N;
    const int t = e / d, i = e - t * d;
    if (t == T) return (real)0;
    double v;
    if (t == 0) {
        v = ks * (double)mu[i];
    } else {
        const real* a = mu + (size_t)(t - 1) * d;
        const real* b = mu + (size_t)t * d;
        const int k = i < N ? i : i - N;
        const double ep = (double)b[k] - ((double)a[k] + dt * (double)a[N + k]);
        const double ev = (double)b[N + k] - (double)a[N + k];
        v = (i < N ? q00 : q10) * ep + (i < N ? q01 : q11) * ev;
    }
    if (kg >= 0.) {
        const real* last = mu + (size_t)(T - 1) * d;
        const double k = (double)(T - 1 - t);
        v += (i < N) ? kg * (double)last[i] : kg * (k * dt * (double)last[i - N] + (double)last[i]);
    }
    return (real)(temperature * v);
}

template <typename real> struct IswNext {            // K4's optional tail (next step's IS weights)
    real* out;               // [P][T+1][d] or null
    const double* Qinv;
    double ks, kg, dt;
    int n, isotropic;
};

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

// ---- end-effector goal term ------------------------------------------------------------------------------------------------
// CostGoal + EESE3DistanceField (cost_functions.py:308-321, fields.py:146-150): K * dist^2 on the LAST waypoint, dist =
// SE3_distance(H_ee, H_target).  SE3_distance is third-party (torch_robotics) and un-vendored; this build defines it as
// w_pos |p - p*| + w_rot angle(R*^T R)  (DESIGN.md).  One trajectory per thread: the chain's frames down to the end effector.
// Evaluated by ee_goal_kernel (cost_sweep.hip) behind a sweep -- or, inside sgpmp_step, by update_kernel itself before its
// softmax (EeFold below: the same function, the same additions in the same order: bit-identical, one launch less).
template <typename real> struct EeTarget { real R[9], p[3], w_pos, w_rot, K; int square; };
__device__ __forceinline__ void sg_sincos(float a, float* s, float* c) { sincosf(a, s, c); }
__device__ __forceinline__ void sg_sincos(double a, double* s, double* c) { sincos(a, s, c); }
__device__ __forceinline__ float sg_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ double sg_sqrt(double a) { return sqrt(a); }

template <typename real>
__device__ __forceinline__ real se3_distance(const real (&R)[9], const real (&p)[3], const EeTarget<real>& tg) {
    const real dx = p[0] - tg.p[0], dy = p[1] - tg.p[1], dz = p[2] - tg.p[2];
    const real dpos = sg_sqrt(dx * dx + dy * dy + dz * dz);
    real tr = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) tr += R[i] * tg.R[i];                  // trace(R*^T R)
    const real c = fmin(fmax((tr - (real)1) * (real)0.5, (real)-1), (real)1);
    return tg.w_pos * dpos + tg.w_rot * acos(c);
}

// The chain's joint constants as the term reads them, staged ONCE per workgroup (LDS): the frames are a dependent chain joint
// after joint, and with the constants behind a global (scalar) load in every trip each joint waited for memory -- 5 of the 12 us
// of update_kernel with the term inside.  jr [joint][12] = R (row-major 3 x 3), t; ji [joint][2] = revolute, index into q.
#define SGPMP_EE_JR 12
template <typename real>
__device__ __forceinline__ void ee_stage_chain(const ChainDev* __restrict__ ch, real* jr, int* ji, int tid, int nthr) {
    const int nj = ch->n_joints;
    for (int i = tid; i < nj * SGPMP_EE_JR; i += nthr) {
        const int j = i / SGPMP_EE_JR, e = i - j * SGPMP_EE_JR;
        jr[i] = e < 9 ? (real)ch->j[j].R[e] : (real)ch->j[j].t[e - 9];
    }
    for (int j = tid; j < nj; j += nthr) { ji[2 * j] = ch->j[j].revolute; ji[2 * j + 1] = ch->j[j].qidx; }
}

// the term's cost of the trajectory whose last waypoint's state is q [2n] (positions first)
template <typename real>
__device__ __forceinline__ double ee_goal_add(const real* jr, const int* ji, int nj, const real* __restrict__ q, const EeTarget<real>& tg) {
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    for (int j = 0; j < nj; ++j) {
        const real* F = jr + j * SGPMP_EE_JR;
        const real* tt = F + 9;
        real Rn[9];
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3 + 0] * tt[0] + R[r * 3 + 1] * tt[1] + R[r * 3 + 2] * tt[2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rn[r * 3 + c] = R[r * 3 + 0] * F[c] + R[r * 3 + 1] * F[3 + c] + R[r * 3 + 2] * F[6 + c];
        if (ji[2 * j]) {
            real s, c;
            sg_sincos(q[ji[2 * j + 1]], &s, &c);
            for (int r = 0; r < 3; ++r) {
                const real aa = Rn[r * 3 + 0], bb = Rn[r * 3 + 1];
                Rn[r * 3 + 0] = aa * c + bb * s;
                Rn[r * 3 + 1] = bb * c - aa * s;
            }
        }
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    real dist = se3_distance<real>(R, p, tg);
    if (tg.square) dist = dist * dist;
    return (double)(tg.K * dist);
}

// update_kernel evaluates the step's end-effector goal term itself (sgpmp_step, one such term): what ee_goal_kernel would have
// been launched with.  ch == null: no term.
template <typename real> struct EeFold {
    const ChainDev* ch;
    EeTarget<real> tg;
    int n, T;
    real* costs;                // [P][S] the step's cost output in the context dtype, or null (the fp64 scratch update_kernel reads
                                // its costs from stays WITHOUT the term: it is the step's own, nothing reads it after this kernel)
};

#ifndef __HIPCC_RTC__
template <typename real>
static inline EeTarget<real> make_ee_target(const CostTerm& t) {
    EeTarget<real> g;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) g.R[r * 3 + c] = (real)t.target[r * 4 + c];
        g.p[r] = (real)t.target[r * 4 + 3];
    }
    g.w_pos = (real)t.w_pos; g.w_rot = (real)t.w_rot; g.K = (real)t.K;
    g.square = (t.flags & SGPMP_FLAG_EE_SQUARE) ? 1 : 0;
    return g;
}
#endif

// ---- store-free steps: the rows a particle's update needs, regenerated from their noise keys -----------------------------
// A step whose samples nobody reads (SGPMP_STEP_NO_SAMPLES: iterations 1 .. K - 1 of optimize(opt_iters = K); the reference
// returns the last iteration's tensors only, planner.py:289-317) does not write them: 470 MB per launch at config 3.  With the
// reference's hyper-parameters the softmax of planner.py:263-275 is one-hot, so the update wants ONE row per particle -- and
// every row is a pure function of (seed, draw, global particle, sample) and the particle's means: Philox + Box-Muller, the
// scan recurrence, x = mu + y.  Evaluated here with the calls the launch itself makes (rng.h: get4p / get4, scan_step2 /
// scan_step, seg_chain, seg_fixup: every product-sum an explicit fma), the row comes out bit for bit as the launch would have
// stored it, so a store-free iteration leaves the same means, weights and gradient as a storing one.
struct RegenArgs {
    int recipe;                 // 0: every row is in memory; 1: the scan order of sample_iso_kernel = fused_step_kernel; 2: fused_planar_seg_kernel's segments of L waypoints
    int N, L, R;                // dofs; waypoints per segment (recipe 2); rows regenerated per round (sized by the launch's LDS)
    unsigned long long seed, draw;
    int mode_offset;            // global index of local particle 0 (noise key)
    const float* coef;          // recipe 1: PriorDev::iso32p [T][8] (pair order); recipe 2: PriorDev::iso32 [T][8]
    const float* pre;           // recipe 2: [T][4] prefix products of the propagators inside each segment (PriorDev::slabpre)
    unsigned store_threshold;   // the launch stored the rows of particle p iff nnz_prev[p] > store_threshold
};

// Rows idx[0 .. nb) of particle p (global index gp) -> ybuf [nb][M] as PERTURBATIONS y (the caller forms x = mu + y in fp32,
// the launch's phase B).  tab: workgroup scratch of T * 12 floats, zl: nb * (T / L) * N * 2 floats (recipe 2).  All 256 threads
// of the workgroup call; barriers inside; ybuf is complete on return.
__device__ __forceinline__ void regen_rows(const RegenArgs& rg, int T, unsigned gp, const int* idx, int nb, float* ybuf, float* tab,
                                           float* zl, int tid, int nthr, bool stage_tables) {
    const int N = rg.N, D = 2 * N, M = T * D, half = T >> 1;
    // the coefficient table(s) through LDS: the recurrence below is a dependent chain, a global load per step would dominate it
    if (stage_tables) {
        for (int i = tid; i < T * 8; i += nthr) tab[i] = rg.coef[i];
        if (rg.recipe == 2) for (int i = tid; i < T * 4; i += nthr) tab[T * 8 + i] = rg.pre[i];
    }
    // noise: one Philox block = (waypoints 2 tp, 2 tp + 1) x (position, velocity) of dof k -- one block per thread and trip
    for (int it = tid; it < nb * half * N; it += nthr) {
        const int k = it % N, q = it / N, tp = q % half, r = q / half;
        NoiseGen<float> gen;
        gen.init((uint64_t)rg.seed, (uint64_t)rg.draw, (uint32_t)gp, (uint32_t)idx[r], (uint32_t)k);
        float* o = ybuf + (size_t)r * M + (size_t)(2 * tp) * D;
        if (rg.recipe == 1) {
            sg_f2 z02, z13;
            gen.get4p(2 * tp, z02, z13);
            o[k] = z02.x; o[N + k] = z13.x; o[D + k] = z02.y; o[D + N + k] = z13.y;
        } else {
            float e[4];
            gen.get4(2 * tp, e);
            o[k] = e[0]; o[N + k] = e[1]; o[D + k] = e[2]; o[D + N + k] = e[3];
        }
    }
    __syncthreads();
    if (rg.recipe == 1) {
        // the launch's phase A: lane = (sample, dof), two steps per Philox block, state carried over the whole trajectory
        if (tid < nb * N) {
            const int r = tid / N, k = tid - r * N;
            float* o = ybuf + (size_t)r * M + k;
            const float* c = tab;
            sg_f2 pv = {0.f, 0.f};
#pragma unroll 4
            for (int t = 0; t < T; t += 2, c += 16, o += 2 * D) {
                const float e0 = o[0], e1 = o[N], e2 = o[D], e3 = o[D + N];
                scan_step2(c, e0, e1, pv);
                o[0] = pv.x; o[N] = pv.y;
                scan_step2(c + 8, e2, e3, pv);
                o[D] = pv.x; o[D + N] = pv.y;
            }
        }
        __syncthreads();
        return;
    }
    // recipe 2 -- fused_planar_seg_kernel: thread = (row, segment, dof); zero-start recurrence over the segment's L waypoints,
    // the chain over the earlier segments' end states, the fix-up
    const int L = rg.L, G = T / L;
    const float* const pre = tab + T * 8;
    if (tid < nb * G * N) {
        const int k = tid % N, q = tid / N, g = q % G, r = q / G;
        float* o = ybuf + (size_t)r * M + (size_t)(g * L) * D + k;
        const float* c = tab + (size_t)(g * L) * 8;
        float zp = 0.f, zv = 0.f;
        for (int i = 0; i < L; ++i, c += 8, o += D) {
            scan_step<float>(c, o[0], o[N], zp, zv);
            o[0] = zp; o[N] = zv;
        }
        zl[((size_t)(r * G + g) * N + k) * 2] = zp;
        zl[((size_t)(r * G + g) * N + k) * 2 + 1] = zv;
    }
    __syncthreads();
    if (tid < nb * G * N) {
        const int k = tid % N, q = tid / N, g = q % G, r = q / G;
        float ysp = 0.f, ysv = 0.f;
        for (int j = 0; j < g; ++j) {
            const float* A = pre + (size_t)(j * L + L - 1) * 4;           // segment j's propagator: the prefix product at its last waypoint
            const float* z = zl + ((size_t)(r * G + j) * N + k) * 2;
            seg_chain(A[0], A[1], A[2], A[3], z[0], z[1], ysp, ysv);
        }
        float* o = ybuf + (size_t)r * M + (size_t)(g * L) * D + k;
        const float* P = pre + (size_t)(g * L) * 4;
        for (int i = 0; i < L; ++i, P += 4, o += D) seg_fixup(P[0], P[1], P[2], P[3], ysp, ysv, o[0], o[N]);
    }
    __syncthreads();
}

// LDS of the regeneration: rows + tables + segment end states + (more rows with weight than one round holds) the carried sums
static inline size_t regen_lds_bytes(int recipe, int T, int n, int R) {
    const size_t M = (size_t)T * 2 * n;
    return (size_t)R * M * 4 + (size_t)T * 12 * 4 + (recipe == 2 ? (size_t)R * 16 * n * 2 * 4 : 0) + M * 8;
}

// The update of ONE particle by one 256-thread workgroup -- update_kernel's body.  `c`: the particle's S costs, `X`: its S
// sample rows, `x_pitch` elements apart (global [S][M]), `lds_raw`: S * 12 (+ 16-byte round-up + M elements when nx.out)
// bytes of workgroup scratch (+ regen_lds_bytes behind them when `regen`).  `stats` are ACCUMULATED into by atomics (shard p & 63).
// VW = elements per thread and load (4 when M % 4 == 0, else 2; M = T * 2n is always even).
// regen: the particle's rows are NOT in memory (store-free step) -- the rows that carry weight are regenerated (rg says how).
#define SGPMP_UPD_THREADS 256
template <typename real, typename cost_t, int VW>
__device__ __forceinline__ void update_particle(int p, int M, int S, const cost_t* c, const real* X, size_t x_pitch,
                                                real* __restrict__ means, double temperature, double step_size,
                                                real* __restrict__ weights, real* __restrict__ grad, real* __restrict__ means_prev,
                                                double* __restrict__ stats, const IswNext<real>& nx, real* __restrict__ means_copy,
                                                unsigned char* lds_raw, const real* mu_rd = nullptr,
                                                const float* __restrict__ partials = nullptr, int gpp = 0, unsigned* __restrict__ nnz_out = nullptr,
                                                const RegenArgs& rg = RegenArgs{}, bool regen = false, unsigned char* regen_lds = nullptr,
                                                const double* c_add = nullptr) {
    // (rg by reference and a separate flag: a POINTER to the kernel-argument struct made every thread copy the struct to scratch
    // -- 72 bytes of private segment, 16 KB of scratch stores per workgroup: +2.6 us on update_kernel until it was found)
    typedef real vec __attribute__((ext_vector_type(VW)));
    double* w = reinterpret_cast<double*>(lds_raw);                  // [S] weights
    int* idx = reinterpret_cast<int*>(lds_raw + (size_t)S * 8);      // [S] samples with weight != 0
    real* mu_lds = reinterpret_cast<real*>(lds_raw + (((size_t)S * 12 + 15) & ~(size_t)15));   // [M] new means (tail)
    __shared__ double scratch[8];
    __shared__ int nnz_s;
    const int tid = threadIdx.x < SGPMP_UPD_THREADS ? (int)threadIdx.x : 1 << 30;   // (extra threads: every loop is empty)
    const int nthr = SGPMP_UPD_THREADS;
    // (issued first, consumed last: the four distinct entries of an isotropic prior's one-step precision -- the tail
    // below would otherwise pay two dependent global loads per element)
    const bool iso_tail = nx.out && nx.isotropic;
    const int nd = 2 * nx.n;
    const double q00 = iso_tail ? nx.Qinv[0] : 0., q01 = iso_tail ? nx.Qinv[nx.n] : 0.;
    const double q10 = iso_tail ? nx.Qinv[nx.n * nd] : 0., q11 = iso_tail ? nx.Qinv[nx.n * nd + nx.n] : 0.;

    // softmax(-c / temperature) exactly as torch.softmax: exp(z - max z) / sum
    double zmax = -1.7976931348623157e308, csum = 0., cmin = 1.7976931348623157e308;
    for (int s = tid; s < S; s += nthr) {
        double cv = (double)c[s];
        if (c_add) cv += c_add[s];                      // (LDS: a cost term the caller evaluated itself -- update_kernel's EeFold)
        const double z = -cv / temperature;
        w[s] = z;
        zmax = z > zmax ? z : zmax;
        cmin = cv < cmin ? cv : cmin;
        csum += cv;
    }
    // one combined reduction for (max z, sum c, min c): three block reductions cost nine barriers
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ((blockDim.x < nthr ? blockDim.x : nthr) + 63) >> 6;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double oz = __shfl_xor(zmax, off, 64), os = __shfl_xor(csum, off, 64), om = __shfl_xor(cmin, off, 64);
            zmax = oz > zmax ? oz : zmax;
            csum += os;
            cmin = om < cmin ? om : cmin;
        }
        __shared__ double red3[3 * 4];
        if (lane == 0 && wave < nw) { red3[wave] = zmax; red3[4 + wave] = csum; red3[8 + wave] = cmin; }
        __syncthreads();
        zmax = red3[0]; csum = red3[4]; cmin = red3[8];
        for (int i = 1; i < nw; ++i) {
            zmax = red3[i] > zmax ? red3[i] : zmax;
            csum += red3[4 + i];
            cmin = red3[8 + i] < cmin ? red3[8 + i] : cmin;
        }
    }
    double part = 0.;
    for (int s = tid; s < S; s += nthr) {
        const double e = exp(w[s] - zmax);
        w[s] = e;
        part += e;
    }
    double Z;
    {   // (one barrier: `scratch` is written once in this kernel)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ((blockDim.x < nthr ? blockDim.x : nthr) + 63) >> 6;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0 && wave < nw) scratch[wave] = part;
        __syncthreads();
        Z = scratch[0];
        for (int i = 1; i < nw; ++i) Z += scratch[i];
    }
    if (stats) {
        const double tot = csum, mn = cmin;
        if (threadIdx.x == 0) {
            // 64 shards of 4 doubles: a thousand workgroups adding to one address serialise
            // at the memory side (~30 us); the consumer sums the shards
            double* sh = stats + (p & (SGPMP_STAT_SHARDS - 1)) * 4;
            atomicAdd(&sh[0], tot);
            atomicAdd(&sh[1], mn);
            atomicAdd(&sh[2], 1.0);
        }
    }
    const double invZ = 1. / Z;
    for (int s = tid; s < S; s += nthr) {
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
    if (nnz_out && threadIdx.x == 0) *nnz_out = (unsigned)nnz;        // (the next step's fused launch decides on it: FusedArgs::nnz_prev)
    // Dense-weight regime: the fused launch left softmax partials of the particle's gpp = S / 8 row groups (FusedArgs::part):
    // sum_s w_s (x_s - mu) = sum_j [exp(-cmin_j / temperature - zmax) / Z] acc_j -- gpp rows instead of nnz
    const bool use_part = sizeof(real) == 4 && VW == 4 && partials != nullptr && nnz > gpp;
    double* coef = reinterpret_cast<double*>(idx);                  // [gpp] (the compacted indices are not needed on this path)
    if (use_part) {
        for (int j = tid; j < gpp; j += nthr)
            coef[j] = exp(-(double)partials[(size_t)j * (M + 4)] / temperature - zmax) * invZ;
        __syncthreads();
    }

    real* mu = means + (size_t)p * M;
    // what every path does with an element's finished sum: gradient, pre-update means, new means (+ copies)
    auto finish = [&](int m, const vec& mu_m, const double (&acc)[VW]) {
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
    };
    bool regenerated = false;
    if constexpr (sizeof(real) == 4) {
        if (regen && !use_part) {
            regenerated = true;
            // Store-free step: rounds of R rows with weight -- regenerated into LDS as perturbations y, x = mu + y formed as the
            // launch forms it (one fp32 add), then the row-gather's arithmetic in the row-gather's order (ascending sample index,
            // one fma per row): the same sums, bit for bit.  One round is the rule (one-hot weights: one row).
            const int Tn = M / (2 * rg.N), R = rg.R;
            float* ybuf = reinterpret_cast<float*>(regen_lds);
            float* tab = ybuf + (size_t)R * M;
            float* zl = tab + (size_t)Tn * 12;
            double* accl = reinterpret_cast<double*>(zl + (rg.recipe == 2 ? (size_t)R * 16 * rg.N * 2 : 0));
            for (int b0 = 0; b0 < nnz; b0 += R) {
                const int nb = nnz - b0 < R ? nnz - b0 : R;
                regen_rows(rg, Tn, (unsigned)(rg.mode_offset + p), idx + b0, nb, ybuf, tab, zl, tid, nthr, b0 == 0);
                for (int m = tid < nthr ? tid * VW : M; m < M; m += nthr * VW) {
                    const vec mu_m = *reinterpret_cast<const vec*>((mu_rd ? mu_rd : mu) + m);
                    double acc[VW];
#pragma unroll
                    for (int i = 0; i < VW; ++i) acc[i] = b0 > 0 ? accl[m + i] : 0.;
                    for (int k = 0; k < nb; ++k) {
                        const double ws = w[idx[b0 + k]];
                        const vec y = *reinterpret_cast<const vec*>(ybuf + (size_t)k * M + m);
                        const vec x = y + mu_m;
#pragma unroll
                        for (int i = 0; i < VW; ++i) acc[i] = fma(ws, (double)(x[i] - mu_m[i]), acc[i]);
                    }
                    if (b0 + R < nnz) {
#pragma unroll
                        for (int i = 0; i < VW; ++i) accl[m + i] = acc[i];
                    } else {
                        finish(m, mu_m, acc);
                    }
                }
                __syncthreads();                         // (the next round overwrites the rows; the tail below reads mu_lds)
            }
        }
    }
    for (int m = (tid < nthr && !regenerated) ? tid * VW : M; m < M; m += nthr * VW) {
        const vec mu_m = *reinterpret_cast<const vec*>((mu_rd ? mu_rd : mu) + m);   // (mu_rd: an LDS copy of the same means)
        double acc[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = 0.;
        int k = 0;
        if (use_part) {
            if constexpr (sizeof(real) == 4 && VW == 4) {
                for (int j = 0; j < gpp; j += 8) {           // eight partial rows in flight per thread (S / 8 of them: 16 at config 3)
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    f4 v[8];
                    double cj[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int jj = j + u < gpp ? j + u : gpp - 1;
                        cj[u] = j + u < gpp ? coef[jj] : 0.;
                        v[u] = *reinterpret_cast<const f4*>(partials + (size_t)jj * (M + 4) + 4 + m);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int i = 0; i < VW; ++i) acc[i] = fma(cj[u], (double)v[u][i], acc[i]);
                }
            }
            k = nnz;                                     // (skip the row gather)
        }
        for (; k + 4 <= nnz; k += 4) {                   // four rows in flight per thread
            vec v[4];
            double ws[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = idx[k + u];
                ws[u] = w[s];
                v[u] = *reinterpret_cast<const vec*>(X + (size_t)s * x_pitch + m);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < VW; ++i) acc[i] = fma(ws[u], (double)(v[u][i] - mu_m[i]), acc[i]);
        }
        for (; k < nnz; ++k) {
            const int s = idx[k];
            const double ws = w[s];
            const vec v = *reinterpret_cast<const vec*>(X + (size_t)s * x_pitch + m);
#pragma unroll
            for (int i = 0; i < VW; ++i) acc[i] = fma(ws, (double)(v[i] - mu_m[i]), acc[i]);
        }
        finish(m, mu_m, acc);
    }
    // The NEXT iteration's importance-sampling weights, from the means just written (K5's arithmetic, same
    // function): sgpmp_step then starts with the sampler + sweep launch instead of a K5 launch, provided the
    // caller vouches that nobody touched the means in between (SGPMP_STEP_MEANS_KEPT).
    if (nx.out) {
        __syncthreads();                                  // the new means of this particle are in LDS
        const int d = 2 * nx.n, Tn = M / d;
        const real* mu_new = mu_lds;
        for (int e = tid; e < (Tn + 1) * d; e += nthr)
            nx.out[(size_t)p * (Tn + 1) * d + e] =
                iso_tail ? is_weight_elem_iso<real>(nx.n, Tn, mu_new, q00, q01, q10, q11, nx.ks, nx.kg, nx.dt, temperature, e)
                         : is_weight_elem<real>(nx.n, Tn, mu_new, nx.Qinv, nx.ks, nx.kg, nx.dt, temperature, nx.isotropic, e);
    }
}

template <typename real, typename cost_t, int VW>
__global__ void __launch_bounds__(256)
update_kernel(int M, int S, const cost_t* __restrict__ costs, const real* __restrict__ samples,
              real* __restrict__ means, double temperature, double step_size,
              real* __restrict__ weights, real* __restrict__ grad, real* __restrict__ means_prev,
              double* __restrict__ stats, IswNext<real> nx, real* __restrict__ means_copy,
              const float* __restrict__ part, int gpp, unsigned* __restrict__ nnz, unsigned nnz_threshold,
              RegenArgs rg, unsigned regen_lds_offset, EeFold<real> ee, unsigned ee_lds_offset) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int p = blockIdx.x;
    // the step's end-effector goal term, evaluated here instead of by a launch of its own in front of this one: thread s has
    // row s of the particle; the sums are ee_goal_kernel's (cost64 + add in update_particle; costs = (real)((double)costs + add))
    double* c_add = nullptr;
    if (ee.ch != nullptr) {
        __shared__ real ee_jr[SGPMP_MAX_JOINTS * SGPMP_EE_JR];
        __shared__ int ee_ji[SGPMP_MAX_JOINTS * 2];
        ee_stage_chain<real>(ee.ch, ee_jr, ee_ji, threadIdx.x, blockDim.x);
        const int nj = ee.ch->n_joints;
        __syncthreads();
        c_add = reinterpret_cast<double*>(lds_raw + ee_lds_offset);
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            const size_t b = (size_t)p * S + s;
            const double add = ee_goal_add<real>(ee_jr, ee_ji, nj, samples + (b * ee.T + (ee.T - 1)) * 2 * ee.n, ee.tg);
            c_add[s] = add;
            if (ee.costs) ee.costs[b] = (real)((double)ee.costs[b] + add);
        }
        __syncthreads();
    }
    // partials of this particle exist iff the fused launch of THIS step saw nnz[p] above the threshold: the same word,
    // read here before this particle's new count replaces it -- and, after a store-free step, its rows exist iff that word
    // was above the store threshold
    const unsigned prev = nnz ? nnz[p] : 0u;
    const float* part_p = nullptr;
    if (part && nnz && prev > nnz_threshold) part_p = part + (size_t)p * gpp * (M + 4);
    const bool regen = rg.recipe != 0 && !(prev > rg.store_threshold);
    update_particle<real, cost_t, VW>(p, M, S, costs + (size_t)p * S, samples + (size_t)p * S * M, (size_t)M, means, temperature,
                                      step_size, weights, grad, means_prev, stats, nx, means_copy, lds_raw, nullptr, part_p, gpp,
                                      nnz ? nnz + p : nullptr, rg, regen, lds_raw + regen_lds_offset, c_add);
}

