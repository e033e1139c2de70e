// K3 -- composite cost sweep over the trajectory batch: trajs [B,T,d] -> costs [B].
//
// Replaces CostComposite.eval (costs/cost_functions.py:47-58) and everything it calls:
//   CostGP / CostGPTrajectory.eval   cost_functions.py:128-146, 202-215  (+ gp_factor.py:54-67,
//                                    unary_factor.py:21-22)
//   CostGoalPrior.eval               cost_functions.py:376-388
//   CostCollision.eval               cost_functions.py:247-261 + field_factor.py:18-40 with
//       ObstacleMap.get_collisions   envs/obst_map.py:164-182
//       LinkDistanceField            costs/fields.py:63-86   (rbf / sdf / occupancy)
//       LinkSelfDistanceField        costs/fields.py:114-124
//   the FK callable                  cost_functions.py:51-52 (URDF chain)
//   the importance-sampling term     planner.py:233-236, in the factored form
//       x^T Sigma^-1 mu = (A x)^T [blkdiag(K_s, Q^-1.., K_g) A mu],  A x = (x_0, e_0.., x_{T-1})
//
// Mapping: ONE WAVE PER TRAJECTORY, ONE LANE PER WAYPOINT (64 waypoints per pass).  A lane loads its
// own d-vector (the wave reads one contiguous 64*d*w-byte span), gets x_{t-1} from its neighbour
// lane, evaluates every per-waypoint term, and the wave reduces the 64 partial costs in fp64.
// Link positions of the lane's waypoint live in LDS (structure-of-arrays, one column per lane, so
// accesses are bank-conflict free) because the pair loops index them dynamically.
#include "sgpmp_internal.h"

template <typename real> struct RealOps;
template <> struct RealOps<float> {
    static __device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
    static __device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
    static __device__ __forceinline__ float exp_(float a) { return expf(a); }
    static __device__ __forceinline__ float sqrt_(float a) { return sqrtf(a); }
    static __device__ __forceinline__ float floor_(float a) { return floorf(a); }
    static __device__ __forceinline__ void sincos_(float a, float* s, float* c) { sincosf(a, s, c); }
};
template <> struct RealOps<double> {
    static __device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ double exp_(double a) { return exp(a); }
    static __device__ __forceinline__ double sqrt_(double a) { return sqrt(a); }
    static __device__ __forceinline__ double floor_(double a) { return floor(a); }
    static __device__ __forceinline__ void sincos_(double a, double* s, double* c) { sincos(a, s, c); }
};

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
__device__ __forceinline__ real grid_value(const CostTerm& tm, real x, real y) {
    using O = RealOps<real>;
    const real fx = O::floor_(O::add_rn(O::mul_rn(x, (real)tm.inv_cell), (real)tm.off_x));
    const real fy = O::floor_(O::add_rn(O::mul_rn(y, (real)tm.inv_cell), (real)tm.off_y));
    const real hx = (real)(tm.dim0 - 1), hy = (real)(tm.dim1 - 1);
    const int ix = (int)fmin(fmax(fx, (real)0), hx);       // clamp in float first: no int overflow
    const int iy = (int)fmin(fmax(fy, (real)0), hy);
    const real* grid = (const real*)tm.dev_data;
    return grid[(size_t)iy * tm.dim1 + ix];
}

// ---------------------------------------------------------------------------------- FK
// Positions of all link frames for joint vector q, written to LDS column `col` (SoA, `stride`
// reals between consecutive scalars).  H_child = H_parent * Trans(xyz) * RPY * Rz(q).
template <typename real> struct JointConst;
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
__device__ __forceinline__ void add_interp_points(const CostTerm& tm, int n_links, real* col, int stride) {
    int o = n_links;
    for (int i = tm.interp_lo; i < tm.interp_hi; ++i) {
        const real ax = col[(i * 3 + 0) * stride], ay = col[(i * 3 + 1) * stride], az = col[(i * 3 + 2) * stride];
        const real bx = col[((i + 1) * 3 + 0) * stride], by = col[((i + 1) * 3 + 1) * stride],
                   bz = col[((i + 1) * 3 + 2) * stride];
        for (int a = 0; a < tm.n_interp; ++a, ++o) {
            const real al = (real)tm.alpha[a];
            col[(o * 3 + 0) * stride] = ax + (bx - ax) * al;
            col[(o * 3 + 1) * stride] = ay + (by - ay) * al;
            col[(o * 3 + 2) * stride] = az + (bz - az) * al;
        }
    }
}

// LinkDistanceField.compute_cost on the np points of this lane (fields.py:75-86).
template <typename real>
__device__ __forceinline__ real spheres_field(const CostTerm& tm, int np, const real* col, int stride,
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
__device__ __forceinline__ real self_field(const CostTerm& tm, int np, const real* col, int stride) {
    using O = RealOps<real>;
    const real k = (real)tm.K2;                       // -1 / (2 margin^2)
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

// ---------------------------------------------------------------------------------- the sweep
template <typename real>
struct CostArgs {
    int T;
    const CostProgram* prog;
    const ChainDev* chain;
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
};

template <typename real, int N, bool HAS_FK>
__global__ void __launch_bounds__(256)
cost_sweep_kernel(CostArgs<real> a) {
    constexpr int D = 2 * N;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    real* lds = reinterpret_cast<real*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int stride = blockDim.x;
    real* col = lds + threadIdx.x;
    const CostProgram& P = *a.prog;
    const int T = a.T;
    const int nchunks = (T + 63) >> 6;

    for (long long b = (long long)blockIdx.x * waves_per_block + wave; b < a.batch;
         b += (long long)gridDim.x * waves_per_block) {
        const real* row = a.trajs + (size_t)b * T * D;
        const real* isw = a.isw ? a.isw + (size_t)(b / a.rows_per_particle) * (T + 1) * D : nullptr;
        double acc = 0.;
        real carry[D];
#pragma unroll
        for (int i = 0; i < D; ++i) carry[i] = 0;

        for (int c = 0; c < nchunks; ++c) {
            const int t = (c << 6) + lane;
            const bool valid = t < T;
            real x[D], xp[D];
            if (valid) {
#pragma unroll
                for (int i = 0; i < D; ++i) x[i] = row[(size_t)t * D + i];
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) x[i] = 0;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const real up = shfl_up1(x[i]);
                xp[i] = (lane == 0) ? carry[i] : up;
                carry[i] = shfl_idx(x[i], 63);
            }
            real part = 0;                                   // this waypoint's cost, in `real`
            double part64 = 0.;

            for (int ti = 0; ti < P.n_terms; ++ti) {
                const CostTerm& tm = P.terms[ti];
                if (tm.kind == SGPMP_COST_GP) {
                    if (valid && t == 0 && (tm.flags & SGPMP_FLAG_GP_START)) {
                        const real* st = (const real*)tm.dev_data;
                        real sq = 0;
#pragma unroll
                        for (int i = 0; i < D; ++i) { const real dd = st[i] - x[i]; sq += dd * dd; }
                        part64 += (double)(sq * (real)tm.K2);
                    }
                    if (valid && t >= 1) {           // e_{t-1} = x_t - Phi x_{t-1} (gp_factor.py:54-58)
                        const real dt = (real)tm.dt;
                        real pp = 0, pv = 0, vv = 0;
#pragma unroll
                        for (int k = 0; k < N; ++k) {
                            const real ep = x[k] - (xp[k] + dt * xp[N + k]);
                            const real ev = x[N + k] - xp[N + k];
                            pp += ep * ep; pv += ep * ev; vv += ev * ev;
                        }
                        part64 += (double)((real)tm.K * ((real)tm.c11 * pp + (real)2 * (real)tm.c12 * pv +
                                                         (real)tm.c22 * vv));
                    }
                } else if (tm.kind == SGPMP_COST_GOAL_PRIOR) {
                    if (valid && t == T - 1) {
                        const long long g = (a.batch_offset + b) / tm.rows_per_goal;
                        const real* gl = (const real*)tm.dev_data + (size_t)g * D;
                        real sq = 0;
#pragma unroll
                        for (int i = 0; i < D; ++i) { const real dd = gl[i] - x[i]; sq += dd * dd; }
                        part64 += (double)(sq * (real)tm.K);
                    }
                } else if (tm.kind == SGPMP_COST_GRID) {
                    if (valid && t >= 1) part += (real)tm.K * grid_value<real>(tm, x[0], x[N > 1 ? 1 : 0]);
                }
            }
            if constexpr (HAS_FK) {
                if (P.needs_fk) {
                    // every lane runs FK (uniform control flow); invalid lanes work on zeros
                    real q[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) q[k] = x[k];
                    fk_points<real, N>(a.chain, q, col, stride);
                    for (int ti = 0; ti < P.n_terms; ++ti) {
                        const CostTerm& tm = P.terms[ti];
                        if (tm.kind != SGPMP_COST_SPHERES && tm.kind != SGPMP_COST_SELF) continue;
                        if (tm.n_interp > 0) add_interp_points<real>(tm, a.n_links, col, stride);
                        real f;
                        if (tm.kind == SGPMP_COST_SPHERES)
                            f = spheres_field<real>(tm, tm.n_points, col, stride, a.spheres, a.n_spheres);
                        else
                            f = self_field<real>(tm, tm.n_points, col, stride);
                        if (valid && t >= 1) part += (real)tm.K * f;
                    }
                }
            }
            // importance-sampling term with the sampling prior's (A x)_t
            if (isw && valid) {
                const real* w = isw + (size_t)t * D;
                real dot = 0;
                if (t == 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) dot += x[i] * w[i];
                } else {
                    const real dt = a.is_dt;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const real ep = x[k] - (xp[k] + dt * xp[N + k]);
                        const real ev = x[N + k] - xp[N + k];
                        dot += ep * w[k] + ev * w[N + k];
                    }
                }
                if (t == T - 1) {
                    const real* wg = isw + (size_t)T * D;
#pragma unroll
                    for (int i = 0; i < D; ++i) dot += x[i] * wg[i];
                }
                part64 += (double)dot;
            }
            acc += part64 + (double)part;
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (a.costs) a.costs[b] = (real)acc;
            if (a.costs64) a.costs64[b] = acc;
        }
    }
}

template <typename real>
static hipError_t cost_dispatch(int n, int T, const CostProgram* d_prog, const CostProgram& h_prog,
                                const ChainDev* d_chain, int n_links, const real* trajs, long long batch,
                                long long batch_offset, const real* spheres, int n_spheres,
                                const real* isw, int rows_per_particle, double is_dt, real* costs,
                                double* costs64, hipStream_t stream) {
    CostArgs<real> a;
    a.T = T; a.prog = d_prog; a.chain = d_chain; a.n_links = n_links; a.trajs = trajs;
    a.batch = batch; a.batch_offset = batch_offset; a.spheres = spheres; a.n_spheres = n_spheres;
    a.isw = isw; a.rows_per_particle = rows_per_particle > 0 ? rows_per_particle : 1;
    a.is_dt = (real)is_dt; a.costs = costs; a.costs64 = costs64;
    int block = 256;
    size_t lds = 0;
    const bool fk = h_prog.needs_fk != 0;
    if (fk) {
        int max_pts = n_links;
        for (int i = 0; i < h_prog.n_terms; ++i)
            if (h_prog.terms[i].n_points > max_pts) max_pts = h_prog.terms[i].n_points;
        lds = (size_t)max_pts * 3 * block * sizeof(real);
        while (lds > 64 * 1024 && block > 64) { block >>= 1; lds >>= 1; }
    }
    const int wpb = block / 64;
    long long blocks = (batch + wpb - 1) / wpb;
    const long long cap = 256LL * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
#define COST_CASE(NN)                                                                              \
    case NN:                                                                                       \
        if (fk)                                                                                    \
            hipLaunchKernelGGL((cost_sweep_kernel<real, NN, true>), dim3((unsigned)blocks),        \
                               dim3(block), lds, stream, a);                                       \
        else                                                                                       \
            hipLaunchKernelGGL((cost_sweep_kernel<real, NN, false>), dim3((unsigned)blocks),       \
                               dim3(block), 0, stream, a);                                         \
        break;
    switch (n) {
        COST_CASE(1) COST_CASE(2) COST_CASE(3) COST_CASE(4) COST_CASE(5) COST_CASE(6) COST_CASE(7) COST_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef COST_CASE
    return hipGetLastError();
}

hipError_t launch_cost(int dtype, int n, int T, const CostProgram* d_prog, const CostProgram& h_prog,
                       const ChainDev* d_chain, int n_links, const void* trajs, long long batch,
                       long long batch_offset, const void* spheres, int n_spheres,
                       const void* is_weights, int rows_per_particle, double is_dt, void* costs,
                       double* costs64, hipStream_t stream) {
    if (dtype == SGPMP_F64)
        return cost_dispatch<double>(n, T, d_prog, h_prog, d_chain, n_links, (const double*)trajs, batch,
                                     batch_offset, (const double*)spheres, n_spheres,
                                     (const double*)is_weights, rows_per_particle, is_dt, (double*)costs,
                                     costs64, stream);
    return cost_dispatch<float>(n, T, d_prog, h_prog, d_chain, n_links, (const float*)trajs, batch,
                                batch_offset, (const float*)spheres, n_spheres, (const float*)is_weights,
                                rows_per_particle, is_dt, (float*)costs, costs64, stream);
}

// ---------------------------------------------------------------------------------- standalone ops
// FK callable: q [B,n] -> frames [B,L,4,4]  (cost_functions.py:51-52).
template <typename real>
__global__ void fk_frames_kernel(int n, const ChainDev* __restrict__ ch, const real* __restrict__ q,
                                 long long batch, real* __restrict__ frames) {
    using O = RealOps<real>;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int L = ch->n_joints + 1;
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    real* out = frames + (size_t)b * L * 16;
    auto emit = [&](int l) {
        real* o = out + l * 16;
        for (int r = 0; r < 3; ++r) {
            o[r * 4 + 0] = R[r * 3 + 0]; o[r * 4 + 1] = R[r * 3 + 1]; o[r * 4 + 2] = R[r * 3 + 2];
            o[r * 4 + 3] = p[r];
        }
        o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
    };
    emit(0);
    for (int j = 0; j < ch->n_joints; ++j) {
        const JointDev& J = ch->j[j];
        real F[9], tt[3], Rn[9];
        for (int i = 0; i < 9; ++i) F[i] = (real)J.R[i];
        for (int i = 0; i < 3; ++i) tt[i] = (real)J.t[i];
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3 + 0] * tt[0] + R[r * 3 + 1] * tt[1] + R[r * 3 + 2] * tt[2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rn[r * 3 + c] = R[r * 3 + 0] * F[c] + R[r * 3 + 1] * F[3 + c] + R[r * 3 + 2] * F[6 + c];
        if (J.revolute) {
            real s, c;
            O::sincos_(q[(size_t)b * n + J.qidx], &s, &c);
            for (int r = 0; r < 3; ++r) {
                const real aa = Rn[r * 3 + 0], bb = Rn[r * 3 + 1];
                Rn[r * 3 + 0] = aa * c + bb * s;
                Rn[r * 3 + 1] = bb * c - aa * s;
            }
        }
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        emit(j + 1);
    }
}

hipError_t launch_fk(int dtype, int n, const ChainDev* d_chain, int n_links, const void* q,
                     long long batch, void* frames, hipStream_t stream) {
    const int block = 128;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((fk_frames_kernel<double>), dim3(grid), dim3(block), 0, stream, n, d_chain,
                           (const double*)q, batch, (double*)frames);
    else
        hipLaunchKernelGGL((fk_frames_kernel<float>), dim3(grid), dim3(block), 0, stream, n, d_chain,
                           (const float*)q, batch, (float*)frames);
    return hipGetLastError();
}

template <typename real>
__global__ void grid_lookup_kernel(CostTerm tm, const real* __restrict__ xy, long long batch,
                                   real* __restrict__ out) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    out[b] = grid_value<real>(tm, xy[2 * b], xy[2 * b + 1]);
}

hipError_t launch_grid_lookup(int dtype, const CostTerm& term, const void* xy, long long batch,
                              void* out, hipStream_t stream) {
    const int block = 256;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((grid_lookup_kernel<double>), dim3(grid), dim3(block), 0, stream, term,
                           (const double*)xy, batch, (double*)out);
    else
        hipLaunchKernelGGL((grid_lookup_kernel<float>), dim3(grid), dim3(block), 0, stream, term,
                           (const float*)xy, batch, (float*)out);
    return hipGetLastError();
}

// LinkDistanceField / LinkSelfDistanceField.compute_cost on explicit frames [B,L,4,4] -> [B].
template <typename real>
__global__ void field_eval_kernel(CostTerm tm, const real* __restrict__ frames, long long batch,
                                  int n_links, const real* __restrict__ sph, int n_sph,
                                  real* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    real* col = reinterpret_cast<real*>(lds_raw) + threadIdx.x;
    const int stride = blockDim.x;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long bb = b < batch ? b : batch - 1;
    const real* f = frames + (size_t)bb * n_links * 16;
    for (int l = 0; l < n_links; ++l)
        for (int c = 0; c < 3; ++c) col[(l * 3 + c) * stride] = f[l * 16 + c * 4 + 3];
    if (tm.n_interp > 0) add_interp_points<real>(tm, n_links, col, stride);
    real v;
    if (tm.kind == SGPMP_COST_SPHERES) v = spheres_field<real>(tm, tm.n_points, col, stride, sph, n_sph);
    else v = self_field<real>(tm, tm.n_points, col, stride);
    if (b < batch) out[b] = v;
}

hipError_t launch_field_eval(int dtype, const CostTerm& term, const void* frames, long long batch,
                             int n_links, const void* spheres, int n_spheres, void* out,
                             hipStream_t stream) {
    const int block = 64;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    const size_t esz = dtype == SGPMP_F64 ? 8 : 4;
    const size_t lds = (size_t)term.n_points * 3 * block * esz;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((field_eval_kernel<double>), dim3(grid), dim3(block), lds, stream, term,
                           (const double*)frames, batch, n_links, (const double*)spheres, n_spheres,
                           (double*)out);
    else
        hipLaunchKernelGGL((field_eval_kernel<float>), dim3(grid), dim3(block), lds, stream, term,
                           (const float*)frames, batch, n_links, (const float*)spheres, n_spheres,
                           (float*)out);
    return hipGetLastError();
}
