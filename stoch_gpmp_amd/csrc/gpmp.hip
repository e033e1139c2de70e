// Gauss-Newton step of the reference's `GPMP` planner (planner.py:580-640), fused.
//
// The reference stacks every factor of the cost list into dense A [P, rows, N], b, K (N = T d), forms
// J^T J = A^T K A + damping [P, N, N] and solves it densely (planner.py:607-640).  The normal matrix is
// block-TRIDIAGONAL in d x d waypoint blocks:
//   * start prior (cost_functions.py:154-158): K_s on block (0,0), rhs K_s (s - x_0);
//   * GP factor i (cost_functions.py:160-166; H1 = Phi, H2 = -I): blocks (i,i) += Phi^T Q^-1 Phi,
//     (i+1,i+1) += Q^-1, (i+1,i) = -Q^-1 Phi; rhs_i += Phi^T Q^-1 e_i, rhs_{i+1} -= Q^-1 e_i,
//     e_i = x_{i+1} - Phi x_i;
//   * goal prior (cost_functions.py:390-405): K_g on the last block, rhs K_g (goal - x_{T-1});
//   * link field k at waypoint t >= 1 (cost_functions.py:263-279; A row = H = -grad f): the position
//     block gets the rank-1 term K_k h h^T, rhs K_k h f;
//   * damping (planner.py:613-622): delta I, or delta * diag(mean over ALL particles of A^T K A).
// One wave per particle runs a forward block Cholesky over the waypoints (d x d tiles padded to 16 x 16
// in LDS; products on the fp64 matrix cores, v_mfma_f64_16x16x4_f64) and a backward substitution;
// L_t^-1 and W_t = E L_{t-1}^-T are parked in a context scratch buffer between the two sweeps.
#include "sgpmp_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
#define TS SGPMP_TILE

// C = alpha * op(A) * op(B) + beta * Cin on 16x16 row-major LDS tiles, one wave (operand maps as in
// prior_factor.hip: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[row = (l>>4) + 4r][col = l&15]).
__device__ __forceinline__ void gp_mm16(double* C, const double* A, const double* B, bool tA, bool tB,
                                        double alpha, const double* Cin, double beta) {
    const int l = threadIdx.x;
    const int i = l & 15, kq = l >> 4;
    d4 acc = {0., 0., 0., 0.};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int k = 4 * kb + kq;
        const double a = tA ? A[k * TS + i] : A[i * TS + k];
        const double b = tB ? B[i * TS + k] : B[k * TS + i];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    double cin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cin[r] = Cin ? Cin[(kq + 4 * r) * TS + i] : 0.;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(kq + 4 * r) * TS + i] = alpha * acc[r] + beta * cin[r];
    __syncthreads();
}

template <typename real>
__device__ __forceinline__ double ld(const void* p, size_t i) { return (double)((const real*)p)[i]; }

// constant part of block (t,t) of A^T K A, element (r,c)
__device__ __forceinline__ double gp_diag_const(const GpmpArgs& a, int t, int r, int c) {
    const int n = a.n;
    if ((r % n) != (c % n)) return 0.;
    const bool rp = r < n, cp = c < n;
    const double q = a.Kgp * (rp ? (cp ? a.c11 : a.c12) : (cp ? a.c12 : a.c22));
    const double m = a.c11 * a.dt + a.c12;
    const double pqp = a.Kgp * (rp ? (cp ? a.c11 : m) : (cp ? m : a.c11 * a.dt * a.dt + 2. * a.c12 * a.dt + a.c22));
    double v = 0.;
    if (t >= 1) v += q;                                  // Q^-1 of factor t-1
    if (t <= a.T - 2) v += pqp;                          // Phi^T Q^-1 Phi of factor t
    if (r == c) {
        if (t == 0) v += a.Ks;
        if (t == a.T - 1) v += a.Kg;
    }
    return v;
}

// sum over this rank's particles of the field part of diag(A^T K A) (entries (t >= 1, joint j); the
// rest of diag_sum stays zero): workgroup = 8 particles, threads over the contiguous (t, j) elements of
// a particle's gradient rows, one fp64 atomic per element and workgroup into the zeroed output
#define SGPMP_DIAG_PCHUNK 8
template <typename real>
__global__ void __launch_bounds__(256)
gpmp_diag_kernel(GpmpArgs a, double* __restrict__ diag_sum) {
    const int d = 2 * a.n, per = (a.T - 1) * a.n;
    const int p0 = blockIdx.x * SGPMP_DIAG_PCHUNK, p1 = min(p0 + SGPMP_DIAG_PCHUNK, a.P);
    for (int e = threadIdx.x; e < per; e += blockDim.x) {
        double s = 0.;
        for (int k = 0; k < a.n_fields; ++k)
            for (int p = p0; p < p1; ++p) {
                const double h = ld<real>(a.f[k].grad, (size_t)p * per + e);
                s += a.f[k].K * h * h;
            }
        const int t1 = e / a.n, j = e - t1 * a.n;
        atomicAdd(&diag_sum[(t1 + 1) * d + j], s);
    }
}

template <typename real>
__global__ void __launch_bounds__(64)
gpmp_solve_kernel(GpmpArgs a, real* __restrict__ means, real* __restrict__ d_theta, real* __restrict__ costs) {
    __shared__ double S[TS * TS], L[TS * TS], Li[TS * TS], Lp[TS * TS], W[TS * TS], E[TS * TS];
    __shared__ double g[TS], r[TS], tmp[TS], rinv[TS];
    __shared__ double csum[64];
    // sized by the launch: means and solution [T][16], field values [F][T], field gradients [F][T][8]
    extern __shared__ __align__(16) unsigned char gp_lds_raw[];
    const int l = threadIdx.x, p = blockIdx.x;
    const int n = a.n, d = 2 * n, T = a.T;
    double* mu = reinterpret_cast<double*>(gp_lds_raw);
    double* y = mu + (size_t)T * TS;
    double* fv = y + (size_t)T * TS;                      // [F][T]    (index t-1)
    double* fg = fv + (size_t)a.n_fields * T;             // [F][T][8]
    real* mp = means + (size_t)p * T * d;
    double* scr = a.scratch + (size_t)p * T * 2 * TS * TS;
    for (int e = l; e < T * TS; e += 64) {
        const int t = e / TS, i = e % TS;
        mu[e] = i < d ? (double)mp[t * d + i] : 0.;
    }
    // the particle's field values and Jacobians, staged once (the waypoint loop must not wait for HBM)
    for (int f = 0; f < a.n_fields; ++f) {
        for (int e = l; e < T - 1; e += 64) fv[f * T + e] = ld<real>(a.f[f].val, (size_t)p * (T - 1) + e);
        for (int e = l; e < (T - 1) * n; e += 64) {
            const int t1 = e / n, j = e - t1 * n;
            fg[((size_t)f * T + t1) * 8 + j] = ld<real>(a.f[f].grad, ((size_t)p * (T - 1) + t1) * n + e - t1 * n);
        }
    }
    for (int e = l; e < TS * TS; e += 64) {              // E = block (t, t-1) = -Q^-1 Phi, constant
        const int rr = e / TS, c = e % TS;
        double v = 0.;
        if (rr < d && c < d && (rr % n) == (c % n)) {
            const bool rp = rr < n, cp = c < n;
            const double q1 = rp ? a.c11 : a.c12, q2 = rp ? a.c12 : a.c22;     // row of Q^-1: (pos col, vel col)
            v = -a.Kgp * (cp ? q1 : q1 * a.dt + q2);                          // (Q^-1 Phi)[r][c]
        }
        E[e] = v;
        Lp[e] = 0.; L[e] = 0.; Li[e] = 0.; W[e] = 0.;
    }
    __syncthreads();
    const long long gi = a.Kg > 0. ? (a.p_offset + p) / a.rows_per_goal : 0;
    double cost = 0.;                                    // b^T K b, accumulated by lane (i = l < d)

    for (int t = 0; t < T; ++t) {
        // ---- right-hand side g_t and the cost terms living at waypoint t
        if (l < TS) {
            double v = 0.;
            if (l < d) {
                const int k = l % n;
                const bool pos = l < n;
                if (t == 0 && a.Ks > 0.) {
                    const double e0 = ld<real>(a.start, l) - mu[l];
                    v += a.Ks * e0;
                    cost += a.Ks * e0 * e0;
                }
                if (t == T - 1 && a.Kg > 0.) {
                    const double eg = ld<real>(a.goals, (size_t)gi * d + l) - mu[t * TS + l];
                    v += a.Kg * eg;
                    cost += a.Kg * eg * eg;
                }
                if (t <= T - 2) {                         // factor t: e = x_{t+1} - Phi x_t ; rhs_t += Phi^T Q^-1 e
                    const double ep = mu[(t + 1) * TS + k] - (mu[t * TS + k] + a.dt * mu[t * TS + n + k]);
                    const double ev = mu[(t + 1) * TS + n + k] - mu[t * TS + n + k];
                    const double qp = a.Kgp * (a.c11 * ep + a.c12 * ev), qv = a.Kgp * (a.c12 * ep + a.c22 * ev);
                    v += pos ? qp : a.dt * qp + qv;
                    cost += pos ? ep * qp : ev * qv;      // e^T Q^-1 e, split over the lanes of dof k
                }
                if (t >= 1) {                             // factor t-1: rhs_t -= Q^-1 e
                    const double ep = mu[t * TS + k] - (mu[(t - 1) * TS + k] + a.dt * mu[(t - 1) * TS + n + k]);
                    const double ev = mu[t * TS + n + k] - mu[(t - 1) * TS + n + k];
                    v -= a.Kgp * (pos ? a.c11 * ep + a.c12 * ev : a.c12 * ep + a.c22 * ev);
                    if (pos)
                        for (int f = 0; f < a.n_fields; ++f) {
                            const double fval = fv[f * T + t - 1];
                            v += a.f[f].K * (-fg[((size_t)f * T + t - 1) * 8 + l]) * fval;   // A row = -grad f
                            if (l == 0) cost += a.f[f].K * fval * fval;
                        }
                }
            }
            g[l] = v;
        }
        // ---- S = D_t + damping - W W^T
        for (int e = l; e < TS * TS; e += 64) {
            const int rr = e / TS, c = e % TS;
            double v = 0.;
            if (rr < d && c < d) {
                v = gp_diag_const(a, t, rr, c);
                double fpart = 0.;
                if (t >= 1 && rr < n && c < n)
                    for (int f = 0; f < a.n_fields; ++f) {
                        const double* h = fg + ((size_t)f * T + t - 1) * 8;
                        fpart += a.f[f].K * h[rr] * h[c];
                    }
                v += fpart;
                if (rr == c)
                    v += a.diag_sum ? a.delta * (gp_diag_const(a, t, rr, rr) + a.diag_sum[t * d + rr] * a.inv_particles)
                                    : a.delta;
            } else if (rr == c) {
                v = 1.;                                   // padding keeps the tile positive definite
            }
            S[e] = v;
        }
        __syncthreads();
        if (t >= 1) {
            gp_mm16(W, E, Lp, false, true, 1., nullptr, 0.);         // W = E L_{t-1}^-T
            gp_mm16(S, W, W, false, true, -1., S, 1.);               // S -= W W^T
            if (l < TS) {                                            // r = g - W y_{t-1}
                double v = g[l];
                for (int c = 0; c < d; ++c) v -= W[l * TS + c] * y[(t - 1) * TS + c];
                r[l] = v;
            }
        } else if (l < TS) {
            r[l] = g[l];
        }
        __syncthreads();
        // ---- L L^T = S (lower), one column per step, lanes over rows
        for (int j = 0; j < TS; ++j) {
            if (l >= j && l < TS) {
                double v = S[l * TS + j];
                for (int k = 0; k < j; ++k) v -= L[l * TS + k] * L[j * TS + k];
                tmp[l] = v;
            }
            __syncthreads();
            const double piv = tmp[j];
            if (!(piv > 0.) || !(piv < 1e300)) { if (l == 0) *a.status = 1; }
            const double rt = sqrt(piv > 0. ? piv : 1.);
            const double ri = 1. / rt;                               // one division per column, reused below
            if (l < TS) L[l * TS + j] = l > j ? tmp[l] * ri : (l == j ? rt : 0.);
            if (l == 0) rinv[j] = ri;
            __syncthreads();
        }
        // ---- Li = L^-1 (forward substitution, lanes over columns)
        for (int i = 0; i < TS; ++i) {
            if (l < TS) {
                double v = (i == l) ? 1. : 0.;
                for (int k = 0; k < i; ++k) v -= L[i * TS + k] * Li[k * TS + l];
                Li[i * TS + l] = v * rinv[i];
            }
            __syncthreads();
        }
        if (l < TS) {                                               // y_t = L^-1 r
            double v = 0.;
            for (int c = 0; c <= l; ++c) v += Li[l * TS + c] * r[c];
            y[t * TS + l] = v;
        }
        for (int e = l; e < TS * TS; e += 64) {                     // park L^-1 and W for the back sweep
            scr[(size_t)(2 * t) * TS * TS + e] = Li[e];
            scr[(size_t)(2 * t + 1) * TS * TS + e] = W[e];
            Lp[e] = Li[e];
        }
        __syncthreads();
    }
    // ---- cost of the linearisation point (GPMP._get_costs, planner.py:642-644)
    csum[l] = l < d ? cost : 0.;
    __syncthreads();
    if (l == 0) {
        double c = 0.;
        for (int i = 0; i < d; ++i) c += csum[i];
        if (costs) costs[p] = (real)c;
    }
    // ---- backward: x_t = L_t^-T (y_t - W_{t+1}^T x_{t+1}), written into y in place; the tiles of step
    // t-1 are fetched while step t computes (Li still holds L_{T-1}^-1 from the forward sweep)
    for (int t = T - 1; t >= 0; --t) {
        double nl[4], nw[4];
        if (t >= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                nl[q] = scr[(size_t)(2 * (t - 1)) * TS * TS + l + 64 * q];
                nw[q] = scr[(size_t)(2 * t + 1) * TS * TS + l + 64 * q];   // W_t, needed by step t-1
            }
        }
        if (l < TS) {
            double v = y[t * TS + l];
            if (t < T - 1)
                for (int c = 0; c < d; ++c) v -= W[c * TS + l] * y[(t + 1) * TS + c];   // W currently = W_{t+1}
            r[l] = v;
        }
        __syncthreads();
        if (l < TS) {
            double v = 0.;
            for (int c = l; c < TS; ++c) v += Li[c * TS + l] * r[c];
            y[t * TS + l] = v;
        }
        __syncthreads();
        if (t >= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { Li[l + 64 * q] = nl[q]; W[l + 64 * q] = nw[q]; }
        }
        __syncthreads();
    }
    for (int e = l; e < T * d; e += 64) {
        const int t = e / d, i = e % d;
        const double x = y[t * TS + i];
        if (d_theta) d_theta[(size_t)p * T * d + e] = (real)x;
        mp[e] = (real)(mu[t * TS + i] + a.step_size * x);
    }
}

// Block-Thomas form of the Gauss-Newton solve, the matrices in REGISTERS (round 4).
//
// The normal matrix is block-tridiagonal with diagonal blocks D_t (constant GP / unary part (x) I_n, + one rank-1 term per
// link field on the position block, + damping) and the CONSTANT sub-diagonal block E = -Q^-1 Phi = E2 (x) I_n.  Forward:
//     S_0 = D_0,  S_t = D_t - E M_{t-1} E^T,  r_t = g_t - E M_{t-1} r_{t-1},   M_t = S_t^-1
// backward:  x_{T-1} = M_{T-1} r_{T-1},  x_t = M_t (r_t - E^T x_{t+1}).
// One wave per particle, lane r < d holds ROW r of the current d x d matrix in registers:
//   * E M E^T and E M r need no matrix product: E2 (x) I_n mixes row (pos, i) with row (vel, i) -- one exchange with the partner
//     lane r +- n -- and column (pos, j) with column (vel, j) -- in-lane;
//   * M_t = S_t^-1 by Gauss-Jordan elimination without pivoting (S_t is symmetric positive definite), the pivot row broadcast by
//     v_readlane with literal lane numbers: no LDS, no barrier inside the elimination (round 3's column-by-column Cholesky
//     through LDS paid two barriers and an fp64 sqrt + division per column, 14 us per waypoint; this is ~1.5 us);
//   * M_t rows are parked for the backward sweep (d x d doubles per waypoint instead of two 16 x 16 tiles).
// PPW particles per wave, one per row of 16 lanes (a single wave is ISSUE-bound on this recursion -- ~2500 instructions per
// waypoint at one per 4 cycles -- and a lane-row layout leaves 50 of 64 lanes idle: four particles share every instruction).
// The pivot-row broadcast is then per 16-lane row: DPP row_newbcast instead of v_readlane.
template <int PPW, int K>
__device__ __forceinline__ double row_bcast(double v) {
    if constexpr (PPW == 1) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), K), __builtin_amdgcn_readlane(__double2loint(v), K));
    } else {
        // (every lane of a row has a source lane: the `old` operand is never kept -- handing the source itself spares a move)
        const int hi = __double2hiint(v), lo = __double2loint(v);
        return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x150 + K, 0xf, 0xf, true),
                                __builtin_amdgcn_update_dpp(lo, lo, 0x150 + K, 0xf, 0xf, true));
    }
}
// dst[c] = lane c's value of v, for every c (lane numbers must be literals: recursion over the index)
template <int PPW, int D, int C = 0>
__device__ __forceinline__ void bcast_each(double v, double (&dst)[D]) {
    if constexpr (C < D) {
        dst[C] = row_bcast<PPW, C>(v);
        bcast_each<PPW, D, C + 1>(v, dst);
    }
}
// dst[c] = lane kk's src[c] for every c; kk is a loop constant after unrolling -- a switch keeps it a literal for the DPP control
template <int PPW, int D, int KK = 0>
__device__ __forceinline__ void bcast_row(const double (&src)[D], double (&dst)[D], int kk) {
    if constexpr (KK < D) {
        if (kk == KK) {
#pragma unroll
            for (int c = 0; c < D; ++c) dst[c] = row_bcast<PPW, KK>(src[c]);
        } else {
            bcast_row<PPW, D, KK + 1>(src, dst, kk);
        }
    }
}

template <typename real, int N, int PPW>
__global__ void __launch_bounds__(64)
gpmp_thomas_kernel(GpmpArgs a, real* __restrict__ means, real* __restrict__ d_theta, real* __restrict__ costs) {
    constexpr int D = 2 * N;
    extern __shared__ __align__(16) unsigned char gp_lds_raw[];
    __shared__ double csum[64];
    const int l = threadIdx.x;
    const int sub = PPW == 1 ? 0 : l >> 4, l16 = PPW == 1 ? l : l & 15;       // particle of the wave, lane within its row
    const int p_raw = blockIdx.x * PPW + sub;
    const bool valid = p_raw < a.P;
    const int p = valid ? p_raw : a.P - 1;                // (lanes of a missing particle compute on the last one, write nothing)
    const int T = a.T;
    const size_t per = (size_t)2 * T * TS + (size_t)a.n_fields * T * 9;       // doubles of LDS per particle
    double* mu = reinterpret_cast<double*>(gp_lds_raw) + (size_t)sub * per;   // [T][TS] means
    double* y = mu + (size_t)T * TS;                      // [T][TS] r_t, then the solution
    double* fv = y + (size_t)T * TS;                      // [F][T]    (index t-1)
    double* fg = fv + (size_t)a.n_fields * T;             // [F][T][8]
    double* scr = a.scratch + (size_t)p * T * D * D;      // [T][D][D] M_t, element (row, column) at [t][column][row]
    for (int q = 0; q < PPW; ++q) {                       // all 64 lanes stage each particle of the wave
        const int pq = min(blockIdx.x * PPW + q, a.P - 1);
        double* muq = reinterpret_cast<double*>(gp_lds_raw) + (size_t)q * per;
        double* fvq = muq + (size_t)2 * T * TS;
        double* fgq = fvq + (size_t)a.n_fields * T;
        const real* mq = means + (size_t)pq * T * D;
        for (int e = l; e < T * TS; e += 64) {
            const int t = e / TS, i = e % TS;
            muq[e] = i < D ? (double)mq[t * D + i] : 0.;
        }
        for (int f = 0; f < a.n_fields; ++f) {
            for (int e = l; e < T - 1; e += 64) fvq[f * T + e] = ld<real>(a.f[f].val, (size_t)pq * (T - 1) + e);
            for (int e = l; e < (T - 1) * N; e += 64) {
                const int t1 = e / N, j = e - t1 * N;
                fgq[((size_t)f * T + t1) * 8 + j] = ld<real>(a.f[f].grad, ((size_t)pq * (T - 1) + t1) * N + j);
            }
        }
    }
    __syncthreads();
    const long long gi = a.Kg > 0. ? (a.p_offset + p) / a.rows_per_goal : 0;
    const int r = l16 < D ? l16 : 0;                      // this lane's row (lanes >= D of a row idle along)
    const int k = r % N;
    const bool pos = r < N;
    const int partner = (PPW == 1 ? 0 : (l & 48)) + (pos ? r + N : r - N);
    // E2 = -K [[c11, c11 dt + c12], [c12, c12 dt + c22]]  (rows: pos, vel of the LATER waypoint; columns: of the earlier one)
    const double e00 = -a.Kgp * a.c11, e01 = -a.Kgp * (a.c11 * a.dt + a.c12), e10 = -a.Kgp * a.c12, e11 = -a.Kgp * (a.c12 * a.dt + a.c22);
    const double er0 = pos ? e00 : e10, er1 = pos ? e01 : e11;      // this lane's row of E2
    // constant part of D_t, element (r, c): nonzero for c = k (pos column) and c = N + k (vel column) only
    const double q_p = a.Kgp * (pos ? a.c11 : a.c12), q_v = a.Kgp * (pos ? a.c12 : a.c22);         // Q^-1 row
    const double mm = a.c11 * a.dt + a.c12;
    const double pq_p = a.Kgp * (pos ? a.c11 : mm), pq_v = a.Kgp * (pos ? mm : a.c11 * a.dt * a.dt + 2. * a.c12 * a.dt + a.c22);   // Phi^T Q^-1 Phi row
    double M[D], EM[D];
#pragma unroll
    for (int c = 0; c < D; ++c) { M[c] = 0.; EM[c] = 0.; }
    double rprev = 0., cost = 0.;
    bool bad = false;

    for (int t = 0; t < T; ++t) {
        // ---- right-hand side g_t and the cost terms at waypoint t (as gpmp_solve_kernel)
        double g = 0.;
        {
            if (t == 0 && a.Ks > 0.) {
                const double e0 = ld<real>(a.start, r) - mu[r];
                g += a.Ks * e0;
                cost += a.Ks * e0 * e0;
            }
            if (t == T - 1 && a.Kg > 0.) {
                const double eg = ld<real>(a.goals, (size_t)gi * D + r) - mu[t * TS + r];
                g += a.Kg * eg;
                cost += a.Kg * eg * eg;
            }
            if (t <= T - 2) {
                const double ep = mu[(t + 1) * TS + k] - (mu[t * TS + k] + a.dt * mu[t * TS + N + k]);
                const double ev = mu[(t + 1) * TS + N + k] - mu[t * TS + N + k];
                const double qp = a.Kgp * (a.c11 * ep + a.c12 * ev), qv = a.Kgp * (a.c12 * ep + a.c22 * ev);
                g += pos ? qp : a.dt * qp + qv;
                cost += pos ? ep * qp : ev * qv;
            }
            if (t >= 1) {
                const double ep = mu[t * TS + k] - (mu[(t - 1) * TS + k] + a.dt * mu[(t - 1) * TS + N + k]);
                const double ev = mu[t * TS + N + k] - mu[(t - 1) * TS + N + k];
                g -= a.Kgp * (pos ? a.c11 * ep + a.c12 * ev : a.c12 * ep + a.c22 * ev);
                if (pos)
                    for (int f = 0; f < a.n_fields; ++f) {
                        const double fval = fv[f * T + t - 1];
                        g += a.f[f].K * (-fg[((size_t)f * T + t - 1) * 8 + r]) * fval;
                        if (r == 0) cost += a.f[f].K * fval * fval;
                    }
            }
        }
        // ---- EM = (E M_{t-1}) row r: e2[row][0] * M[(pos, k)][:] + e2[row][1] * M[(vel, k)][:]
        double S[D];
        if (t >= 1) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const double other = __shfl(M[c], partner, 64);
                const double mp_ = pos ? M[c] : other, mv_ = pos ? other : M[c];
                EM[c] = er0 * mp_ + er1 * mv_;
            }
            // r_t = g_t - (E M_{t-1}) r_{t-1}
            double acc = g;
            {
                double rb[D];
                bcast_each<PPW, D>(rprev, rb);            // r_{t-1} of every row to every lane of the particle
#pragma unroll
                for (int c = 0; c < D; ++c) acc -= EM[c] * rb[c];
            }
            g = acc;
            // S = - (E M E^T) row r:  column (beta, j) = e2[beta][0] EM[(pos, j)] + e2[beta][1] EM[(vel, j)]
#pragma unroll
            for (int j = 0; j < N; ++j) {
                S[j] = -(e00 * EM[j] + e01 * EM[N + j]);
                S[N + j] = -(e10 * EM[j] + e11 * EM[N + j]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) S[c] = 0.;
        }
        // ---- + D_t: constant part, link-field rank-1 terms on the position block, damping
        {
            double dk = 0., dnk = 0.;                     // elements (r, k) and (r, N + k) of the constant part
            if (t >= 1) { dk += q_p; dnk += q_v; }
            if (t <= T - 2) { dk += pq_p; dnk += pq_v; }
            double diag_c = pos ? dk : dnk;               // element (r, r) of the constant part without the unary factors
            if (t == 0) diag_c += a.Ks;
            if (t == T - 1) diag_c += a.Kg;
            const double damp = a.diag_sum ? a.delta * (diag_c + a.diag_sum[t * D + r] * a.inv_particles) : a.delta;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double v = 0.;
                if (c == k) v += dk;
                if (c == N + k) v += dnk;
                if (c == r) v += damp + (t == 0 ? a.Ks : 0.) + (t == T - 1 ? a.Kg : 0.);
                S[c] += v;
            }
            if (t >= 1)
                for (int f = 0; f < a.n_fields; ++f) {
                    const double* h = fg + ((size_t)f * T + t - 1) * 8;
                    const double hr = pos ? a.f[f].K * h[r] : 0.;
#pragma unroll
                    for (int c = 0; c < N; ++c) S[c] += hr * h[c];
                }
        }
        // ---- M_t = S^-1: Gauss-Jordan without pivoting, pivot row by readlane (literal lanes).  Step kk, with row kk of
        // the current matrix in SGPRs (rowk) and ip = 1 / pivot:   pivot lane: new[c] = rowk[c] ip (c != kk), new[kk] = ip;
        // other lanes, f = their element of column kk:  new[c] = old[c] - (f ip) rowk[c] (c != kk), new[kk] = -f ip.
        // One form for both: new[c] = A old[c] - fe rowk'[c] with rowk'[kk] = 1 and (A, fe) = (0, -ip) in the pivot lane,
        // (1, f ip) elsewhere (old[kk] is dropped by A' = 0 in that column): a multiply and an FMA per element, no selects.
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            double rowk[D];
            bcast_row<PPW, D>(S, rowk, kk);               // row kk of the particle's matrix to all its lanes
            const double piv = rowk[kk];
            bad = bad || !(piv > 0.) || !(piv < 1e300);
            double ip = __builtin_amdgcn_rcp(piv);        // v_rcp_f64 + two Newton steps: full double precision for normal pivots
            ip = fma(fma(-piv, ip, 1.), ip, ip);
            ip = fma(fma(-piv, ip, 1.), ip, ip);
            const bool pl = l16 == kk;
            const double A = pl ? 0. : 1.;
            const double fe = pl ? -ip : S[kk] * ip;
#pragma unroll
            for (int c = 0; c < D; ++c) S[c] = c == kk ? -fe : fma(-fe, rowk[c], A * S[c]);
        }
#pragma unroll
        for (int c = 0; c < D; ++c) M[c] = S[c];
        rprev = g;
        if (l16 < D) {
            y[t * TS + l16] = g;
            if (valid) {
#pragma unroll
                for (int c = 0; c < D; ++c) scr[((size_t)t * D + c) * D + l16] = M[c];   // [t][column][row]: one contiguous 8 d-byte run per store
            }
        }
    }
    if (bad && valid && l16 == 0) *a.status = 1;
    csum[l] = l16 < D ? cost : 0.;
    __syncthreads();
    if (l16 == 0 && valid) {
        double c = 0.;
        for (int i = 0; i < D; ++i) c += csum[l + i];
        if (costs) costs[p] = (real)c;
    }
    // ---- backward: x_t = M_t (r_t - E^T x_{t+1});  (E^T x)[(gamma, i)] = e2[0][gamma] x[(pos, i)] + e2[1][gamma] x[(vel, i)]
    double xn = 0.;                                       // x_{t+1}[r]
    double Mt[D], Mn[D];                                  // M_t, and M_{t-1} on its way in while step t computes
#pragma unroll
    for (int c = 0; c < D; ++c) { Mt[c] = M[c]; Mn[c] = 0.; }
    for (int t = T - 1; t >= 0; --t) {
        if (t >= 1) {
#pragma unroll
            for (int c = 0; c < D; ++c) Mn[c] = scr[((size_t)(t - 1) * D + c) * D + r];
        }
        double v = y[t * TS + r];
        if (t < T - 1) {
            const double xo = __shfl(xn, partner, 64);
            const double xp = pos ? xn : xo, xv = pos ? xo : xn;
            v -= pos ? e00 * xp + e10 * xv : e01 * xp + e11 * xv;
        }
        double acc = 0.;
        {
            double vb[D];
            bcast_each<PPW, D>(v, vb);
#pragma unroll
            for (int c = 0; c < D; ++c) acc += Mt[c] * vb[c];
        }
        xn = acc;
        if (l16 < D) y[t * TS + l16] = acc;
#pragma unroll
        for (int c = 0; c < D; ++c) Mt[c] = Mn[c];
    }
    __syncthreads();
    for (int q = 0; q < PPW; ++q) {
        const int pq = blockIdx.x * PPW + q;
        if (pq >= a.P) break;
        const double* muq = reinterpret_cast<const double*>(gp_lds_raw) + (size_t)q * per;
        const double* yq = muq + (size_t)T * TS;
        real* mq = means + (size_t)pq * T * D;
        for (int e = l; e < T * D; e += 64) {
            const int t = e / D, i = e % D;
            const double x = yq[t * TS + i];
            if (d_theta) d_theta[(size_t)pq * T * D + e] = (real)x;
            mq[e] = (real)(muq[t * TS + i] + a.step_size * x);
        }
    }
}

hipError_t launch_gpmp_diag(int dtype, const GpmpArgs& a, double* diag_sum, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(diag_sum, 0, (size_t)a.T * 2 * a.n * sizeof(double), stream);
    if (e != hipSuccess || a.P <= 0 || a.n_fields == 0) return e;
    const unsigned grid = (unsigned)((a.P + SGPMP_DIAG_PCHUNK - 1) / SGPMP_DIAG_PCHUNK);
    if (dtype == SGPMP_F64) hipLaunchKernelGGL((gpmp_diag_kernel<double>), dim3(grid), dim3(256), 0, stream, a, diag_sum);
    else hipLaunchKernelGGL((gpmp_diag_kernel<float>), dim3(grid), dim3(256), 0, stream, a, diag_sum);
    return hipGetLastError();
}

hipError_t launch_gpmp_solve(int dtype, const GpmpArgs& a, void* means, void* d_theta, void* costs,
                             hipStream_t stream, bool cholesky) {
    if (a.P <= 0) return hipSuccess;
    const size_t lds = ((size_t)2 * a.T * TS + (size_t)a.n_fields * a.T * 9) * sizeof(double);
    // register-resident block-Thomas solve (round 4) for the instantiated joint counts; `cholesky`: round 3's kernel
    // particles per wave: four (one per row of 16 lanes) while their staging fits the LDS, else two, else one
    const size_t per_particle = lds;
    const int ppw = 4 * per_particle <= 150 * 1024 ? 4 : 2 * per_particle <= 150 * 1024 ? 2 : 1;
#define THOMAS_L(NN, PP)                                                                                            \
    {                                                                                                               \
        const unsigned grid = (unsigned)((a.P + PP - 1) / PP);                                                      \
        if (dtype == SGPMP_F64)                                                                                     \
            hipLaunchKernelGGL((gpmp_thomas_kernel<double, NN, PP>), dim3(grid), dim3(64), PP * per_particle, stream, a,        \
                               (double*)means, (double*)d_theta, (double*)costs);                                   \
        else                                                                                                        \
            hipLaunchKernelGGL((gpmp_thomas_kernel<float, NN, PP>), dim3(grid), dim3(64), PP * per_particle, stream, a,         \
                               (float*)means, (float*)d_theta, (float*)costs);                                      \
        return hipGetLastError();                                                                                   \
    }
#define THOMAS(NN)                                                                                                  \
    if (!cholesky && a.n == NN) {                                                                                   \
        if (ppw == 4) THOMAS_L(NN, 4) else if (ppw == 2) THOMAS_L(NN, 2) else THOMAS_L(NN, 1)                       \
    }
    THOMAS(2) THOMAS(3) THOMAS(6) THOMAS(7)
#undef THOMAS_L
#undef THOMAS
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((gpmp_solve_kernel<double>), dim3(a.P), dim3(64), lds, stream, a, (double*)means,
                           (double*)d_theta, (double*)costs);
    else
        hipLaunchKernelGGL((gpmp_solve_kernel<float>), dim3(a.P), dim3(64), lds, stream, a, (float*)means,
                           (float*)d_theta, (float*)costs);
    return hipGetLastError();
}
