// K1 -- GP-prior precision blocks and their reverse block-Cholesky, fp64, one wave, one launch.
//
// Replaces (reference file:line, relative to the reference tree, stoch_gpmp/):
//   costs/factors/gp_factor.py:36-52      Phi, Q^-1
//   costs/factors/unary_factor.py:19      K = I / sigma^2
//   costs/factors/mp_priors_multi.py:170-202   Sigma^-1 = A^T blkdiag(K_s, Q^-1.., K_g) A
//   costs/factors/mp_priors_multi.py:100-110 -> torch multivariate_normal.py:80-86
//                                          scale_tril = (flip-Cholesky of Sigma^-1)^-1
//
// Sigma^-1 is block-tridiagonal with d x d blocks
//   D_0 = K_s + Phi^T Q^-1 Phi,  D_i = Q^-1 + Phi^T Q^-1 Phi,  D_{T-1} = Q^-1 + K_g,
//   E = Sigma^-1[i+1, i] = -Q^-1 Phi,
// and torch's L_inv (lower, Sigma^-1 = L_inv^T L_inv) is block-BIdiagonal with diagonal blocks B_t
// (lower triangular) and sub-diagonal blocks C_t.  Matching blocks gives the reverse recursion
//   B_{T-1}^T B_{T-1} = D_{T-1};   C_t = B_t^-T E;   B_{t-1}^T B_{t-1} = D_{t-1} - C_t^T C_t.
// A sample  x = mu + scale_tril @ eps  solves  L_inv (x - mu) = eps, i.e. the scan
//   y_t = G_t eps_t + H_t y_{t-1},   G_t = B_t^-1,  H_t = -B_t^-1 C_t,
// so K1 emits G_t, H_t (T blocks each) instead of the dense M x M scale_tril.
//
// The d x d products run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) with the state block
// padded to one 16x16 tile; the triangular factorisations are d sequential steps across lanes.
#include "sgpmp_internal.h"
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

#define TS SGPMP_TILE

// slot of iso32 element l = (g11 g21 g22 h11 h12 h21 h22 0) in a row of iso32p = (g11 g21 | h11 h21 | h12 h22 | g22 0)
__device__ __forceinline__ int iso_pair_slot(int l) {
    return l == 0 ? 0 : l == 1 ? 1 : l == 2 ? 6 : l == 3 ? 2 : l == 4 ? 4 : l == 5 ? 3 : l == 6 ? 5 : 7;
}

// C = alpha * op(A) * op(B) + beta * Cin, all 16x16 row-major tiles in LDS, executed by one wave.
// v_mfma_f64_16x16x4_f64 operand maps: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D[row = (l>>4) + 4*r][col = l&15], r = 0..3.
__device__ __forceinline__ void mm16(double* C, const double* A, const double* B, bool tA, bool tB,
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
    __syncthreads();                       // C may alias an operand
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(kq + 4 * r) * TS + i] = alpha * acc[r] + beta * cin[r];
    __syncthreads();
}

// Dblk / Eblk non-null: PER-MODE precisions given by their blocks (MultiMPPrior.set_Sigma_invs,
// mp_priors_multi.py:125-128): workgroup m factors the block-tridiagonal matrix with diagonal blocks
// Dblk[m][t] and sub-diagonal blocks Eblk[m][t] = Sigma_m^-1[t+1, t]; outputs are indexed [m][t].
__global__ void __launch_bounds__(64)
prior_factor_kernel(int n, int T, double c11, double c12, double c22, double dt, double ks, double kg,
                    const double* __restrict__ qc_inv, int isotropic, const double* __restrict__ Dblk,
                    const double* __restrict__ Eblk, PriorDev out) {
    __shared__ double Phi[TS * TS], Qi[TS * TS], W[TS * TS], PQP[TS * TS];
    __shared__ double Dm[TS * TS], S[TS * TS], B[TS * TS], C[TS * TS], G[TS * TS], Hh[TS * TS];
    __shared__ double tmp[TS];
    __shared__ int bad;
    const int l = threadIdx.x;
    const int d = 2 * n;
    if (l == 0) bad = 0;
    const bool given = Dblk != nullptr;
    const size_t mode = blockIdx.x;
    if (given) {
        Dblk += mode * (size_t)T * d * d;
        Eblk += mode * (size_t)(T > 1 ? T - 1 : 0) * d * d;
        out.G += mode * (size_t)T * d * d; out.H += mode * (size_t)T * d * d;
        out.G32 += mode * (size_t)T * d * d; out.H32 += mode * (size_t)T * d * d;
    }
    auto load_block = [&](double* dst, const double* src) {      // [d][d] global -> zero-padded 16x16 LDS tile
        for (int e = l; e < TS * TS; e += 64) {
            const int r = e / TS, c = e % TS;
            dst[e] = (r < d && c < d) ? src[r * d + c] : 0.;
        }
    };

    // ---- assemble Phi and Q^-1 (gp_factor.py:36-52), zero padded to 16x16
    for (int e = l; e < TS * TS; e += 64) {
        const int r = e / TS, c = e % TS;
        double phi = 0., q = 0.;
        if (!given && r < d && c < d) {
            phi = (r == c) ? 1. : 0.;
            if (r < n && c == r + n) phi = dt;
            const double qc = qc_inv[(r % n) * n + (c % n)];
            const double coef = (r < n) ? ((c < n) ? c11 : c12) : ((c < n) ? c12 : c22);
            q = coef * qc;
        }
        Phi[e] = phi;
        Qi[e] = q;
        B[e] = 0.; C[e] = 0.; G[e] = 0.; Hh[e] = 0.;
    }
    __syncthreads();
    mm16(W, Qi, Phi, false, false, 1., nullptr, 0.);        // W = Q^-1 Phi
    mm16(PQP, Phi, W, true, false, 1., nullptr, 0.);        // Phi^T Q^-1 Phi

    // ---- the four distinct blocks of Sigma^-1 (mp_priors_multi.py:170-202, closed form)
    for (int e = l; e < (given ? 0 : d * d); e += 64) {
        const int r = e / d, c = e % d;
        const double eye = (r == c) ? 1. : 0.;
        const double pqp = PQP[r * TS + c], q = Qi[r * TS + c];
        out.blocks[0 * d * d + e] = ks * eye + pqp;
        out.blocks[1 * d * d + e] = q + pqp;
        out.blocks[2 * d * d + e] = q + (kg >= 0. ? kg * eye : 0.);
        out.blocks[3 * d * d + e] = -W[r * TS + c];
        out.Qinv[e] = q;
    }
    // E = -W kept in W (sign folded into the solves below): E[r][c] = -W[r][c]
    for (int e = l; e < TS * TS; e += 64) W[e] = -W[e];
    // S = D_{T-1}
    for (int e = l; e < TS * TS; e += 64) {
        const int r = e / TS, c = e % TS;
        double v = 0.;
        if (r < d && c < d) {
            const double eye = (r == c) ? 1. : 0.;
            if (T == 1) v = ks * eye + (kg >= 0. ? kg * eye : 0.);
            else v = Qi[e] + (kg >= 0. ? kg * eye : 0.);
        }
        S[e] = v;
    }
    __syncthreads();
    if (given) { load_block(S, Dblk + (size_t)(T - 1) * d * d); __syncthreads(); }

    for (int t = T - 1; t >= 0; --t) {
        // ---- reverse Cholesky: B lower triangular with B^T B = S
        for (int j = d - 1; j >= 0; --j) {
            if (l <= j) {
                double v = S[j * TS + l];
                for (int k = j + 1; k < d; ++k) v -= B[k * TS + j] * B[k * TS + l];
                tmp[l] = v;
            }
            __syncthreads();
            const double piv = tmp[j];
            if (!(piv > 0.) || !(piv < 1e300)) { if (l == 0) bad = 1; }
            const double r = sqrt(piv > 0. ? piv : 1.);
            if (l < j) B[j * TS + l] = tmp[l] / r;
            else if (l == j) B[j * TS + l] = r;
            else if (l < TS) B[j * TS + l] = 0.;
            __syncthreads();
        }
        // ---- G = B^-1 (forward substitution, lanes over columns)
        for (int i = 0; i < d; ++i) {
            if (l < d) {
                double v = (i == l) ? 1. : 0.;
                for (int k = 0; k < i; ++k) v -= B[i * TS + k] * G[k * TS + l];
                G[i * TS + l] = v / B[i * TS + i];
            }
            __syncthreads();
        }
        if (t >= 1) {
            if (given) { load_block(W, Eblk + (size_t)(t - 1) * d * d); __syncthreads(); }
            // ---- C = B^-T E (back substitution on the upper-triangular B^T)
            for (int i = d - 1; i >= 0; --i) {
                if (l < d) {
                    double v = W[i * TS + l];
                    for (int k = i + 1; k < d; ++k) v -= B[k * TS + i] * C[k * TS + l];
                    C[i * TS + l] = v / B[i * TS + i];
                }
                __syncthreads();
            }
            mm16(Hh, G, C, false, false, -1., nullptr, 0.);            // H_t = -B^-1 C
        } else {
            for (int e = l; e < TS * TS; e += 64) Hh[e] = 0.;
            __syncthreads();
        }
        // ---- emit G_t, H_t
        for (int e = l; e < d * d; e += 64) {
            const int r = e / d, c = e % d;
            const double g = G[r * TS + c], h = Hh[r * TS + c];
            out.G[(size_t)t * d * d + e] = g;
            out.H[(size_t)t * d * d + e] = h;
            out.G32[(size_t)t * d * d + e] = (float)g;
            out.H32[(size_t)t * d * d + e] = (float)h;
        }
        if (l < 8 && !given) {
            double v = 0.;
            switch (l) {
                case 0: v = G[0]; break;                 // g11
                case 1: v = G[n * TS + 0]; break;        // g21
                case 2: v = G[n * TS + n]; break;        // g22
                case 3: v = Hh[0]; break;                // h11
                case 4: v = Hh[n]; break;                // h12
                case 5: v = Hh[n * TS + 0]; break;       // h21
                case 6: v = Hh[n * TS + n]; break;       // h22
                default: v = 0.;
            }
            out.iso64[t * 8 + l] = v;
            out.iso32[t * 8 + l] = (float)v;
            out.iso32p[t * 8 + iso_pair_slot(l)] = (float)v;
        }
        if (t >= 1) {
            // ---- Schur complement: S = D_{t-1} - C^T C
            for (int e = l; e < TS * TS; e += 64) {
                const int r = e / TS, c = e % TS;
                double v = 0.;
                if (r < d && c < d) {
                    const double eye = (r == c) ? 1. : 0.;
                    v = PQP[e] + ((t - 1 == 0) ? ks * eye : Qi[e]);
                }
                Dm[e] = v;
            }
            __syncthreads();
            if (given) { load_block(Dm, Dblk + (size_t)(t - 1) * d * d); __syncthreads(); }
            mm16(S, C, C, true, false, -1., Dm, 1.);
        }
    }
    if (l == 0 && (bad || !given)) *out.status = bad;        // (per-mode launch: the host zeroes status first)
}

// K1 for an ISOTROPIC GP prior (Q_c^-1 = q I_n: every BASELINE configuration, gp_factor.py:25-26 with a scalar sigma_gp).
// Every d x d block of Sigma^-1 and of its factor is then a 2 x 2 matrix (x) I_n in the (positions, velocities) ordering:
// the reverse block-Cholesky of prior_factor_kernel collapses to a recursion on 2 x 2 SCALARS -- the same formulas, the
// zero products left out -- which one lane walks in ~1 us per waypoint instead of the 19 us a general 14 x 14 step takes
// (d sequential column steps across lanes, five 16 x 16 MFMA products, a dozen barriers): 1.23 ms -> ~0.07 ms at T = 64.
// The other lanes then write the dense blocks the general consumers read (G, H and their fp32 copies, the four
// precision blocks, Q^-1) from the scalars.  Same outputs to rounding (tests: blocks 1e-13, factor 1e-9 of the oracle).
__global__ void __launch_bounds__(64)
prior_factor_iso_kernel(int n, int T, double c11, double c12, double c22, double dt, double ks, double kg,
                        const double* __restrict__ qc_inv, PriorDev out) {
    extern __shared__ double iso_l[];                    // [T][8]  g11 g21 g22 h11 h12 h21 h22 -
    __shared__ double blk[4][4];                         // D0, D, Dlast, E as (11, 12, 21, 22)
    __shared__ double qs[4];
    const int l = threadIdx.x, d = 2 * n;
    if (l == 0) {
        const double q = qc_inv[0];
        const double Q11 = c11 * q, Q12 = c12 * q, Q22 = c22 * q;                  // Q^-1 = q [[c11, c12], [c12, c22]]
        const double W11 = Q11, W12 = Q11 * dt + Q12, W21 = Q12, W22 = Q12 * dt + Q22;   // W = Q^-1 Phi, Phi = [[1, dt], [0, 1]]
        const double P11 = W11, P12 = W12, P21 = dt * W11 + W21, P22 = dt * W12 + W22;   // Phi^T Q^-1 Phi
        const double kgg = kg >= 0. ? kg : 0.;
        blk[0][0] = ks + P11; blk[0][1] = P12; blk[0][2] = P21; blk[0][3] = ks + P22;
        blk[1][0] = Q11 + P11; blk[1][1] = Q12 + P12; blk[1][2] = Q12 + P21; blk[1][3] = Q22 + P22;
        blk[2][0] = Q11 + kgg; blk[2][1] = Q12; blk[2][2] = Q12; blk[2][3] = Q22 + kgg;
        blk[3][0] = -W11; blk[3][1] = -W12; blk[3][2] = -W21; blk[3][3] = -W22;
        qs[0] = Q11; qs[1] = Q12; qs[2] = Q12; qs[3] = Q22;
        const double E11 = -W11, E12 = -W12, E21 = -W21, E22 = -W22;
        // S = D_{T-1}
        double S11, S12, S22;
        if (T == 1) { S11 = ks + kgg; S12 = 0.; S22 = ks + kgg; }
        else { S11 = Q11 + kgg; S12 = Q12; S22 = Q22 + kgg; }
        int bad = 0;
        for (int t = T - 1; t >= 0; --t) {
            // reverse Cholesky  B^T B = S,  B = [[b11, 0], [b21, b22]]
            if (!(S22 > 0.) || !(S22 < 1e300)) bad = 1;
            const double b22 = sqrt(S22 > 0. ? S22 : 1.);
            const double b21 = S12 / b22;
            const double p11 = S11 - b21 * b21;
            if (!(p11 > 0.) || !(p11 < 1e300)) bad = 1;
            const double b11 = sqrt(p11 > 0. ? p11 : 1.);
            // G = B^-1
            const double g11 = 1. / b11, g22 = 1. / b22, g21 = (-b21 * g11) / b22;
            double h11 = 0., h12 = 0., h21 = 0., h22 = 0.;
            double C11 = 0., C12 = 0., C21 = 0., C22 = 0.;
            if (t >= 1) {
                // C = B^-T E (back substitution on B^T = [[b11, b21], [0, b22]]),  H = -G C
                C21 = E21 / b22; C22 = E22 / b22;
                C11 = (E11 - b21 * C21) / b11; C12 = (E12 - b21 * C22) / b11;
                h11 = -(g11 * C11); h12 = -(g11 * C12);
                h21 = -(g21 * C11 + g22 * C21); h22 = -(g21 * C12 + g22 * C22);
            }
            double* o = iso_l + (size_t)t * 8;
            o[0] = g11; o[1] = g21; o[2] = g22; o[3] = h11; o[4] = h12; o[5] = h21; o[6] = h22; o[7] = 0.;
            if (t >= 1) {
                // Schur complement  S = D_{t-1} - C^T C
                const double* D = (t - 1 == 0) ? blk[0] : blk[1];
                S11 = D[0] - (C11 * C11 + C21 * C21);
                S12 = D[1] - (C11 * C12 + C21 * C22);
                S22 = D[3] - (C12 * C12 + C22 * C22);
            }
        }
        *out.status = bad;
    }
    __syncthreads();
    // ---- the dense forms the general consumers read
    for (int e = l; e < d * d; e += 64) {
        const int r = e / d, c = e % d;
        const bool on = (r % n) == (c % n);
        const int q4 = (r < n ? 0 : 2) + (c < n ? 0 : 1);
        for (int b = 0; b < 4; ++b) out.blocks[(size_t)b * d * d + e] = on ? blk[b][q4] : 0.;
        out.Qinv[e] = on ? qs[q4] : 0.;
    }
    for (size_t e = l; e < (size_t)T * d * d; e += 64) {
        const int t = (int)(e / (d * d)), rc = (int)(e % (d * d)), r = rc / d, c = rc % d;
        const double* o = iso_l + (size_t)t * 8;
        double g = 0., h = 0.;
        if ((r % n) == (c % n)) {
            if (r < n) { g = c < n ? o[0] : 0.; h = c < n ? o[3] : o[4]; }
            else { g = c < n ? o[1] : o[2]; h = c < n ? o[5] : o[6]; }
        }
        out.G[e] = g; out.H[e] = h;
        out.G32[e] = (float)g; out.H32[e] = (float)h;
    }
    for (int e = l; e < T * 8; e += 64) {
        out.iso64[e] = iso_l[e]; out.iso32[e] = (float)iso_l[e];
        out.iso32p[(e & ~7) + iso_pair_slot(e & 7)] = (float)iso_l[e];
    }
}

hipError_t launch_prior_factor(int n, int T, double dt, double ks, double kg, const double* d_qc_inv,
                               int isotropic, PriorDev out, hipStream_t stream) {
    // coefficients exactly as gp_factor.py:45-47 evaluates them in Python floats
    const double c11 = 12. * pow(dt, -3.), c12 = -6. * pow(dt, -2.), c22 = 4. * pow(dt, -1.);
    if (isotropic && !getenv("SGPMP_K1_GENERAL") && (size_t)T * 8 * sizeof(double) <= 48 * 1024) {
        hipLaunchKernelGGL(prior_factor_iso_kernel, dim3(1), dim3(64), (unsigned)((size_t)T * 8 * sizeof(double)), stream, n, T,
                           c11, c12, c22, dt, ks, kg, d_qc_inv, out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(prior_factor_kernel, dim3(1), dim3(64), 0, stream, n, T, c11, c12, c22, dt,
                       ks, kg, d_qc_inv, isotropic, (const double*)nullptr, (const double*)nullptr, out);
    return hipGetLastError();
}

hipError_t launch_prior_factor_blocks(int n, int T, int n_modes, const double* d_D, const double* d_E,
                                      PriorDev out, hipStream_t stream) {
    hipLaunchKernelGGL(prior_factor_kernel, dim3(n_modes), dim3(64), 0, stream, n, T, 0., 0., 0., 0., 0., -1.,
                       (const double*)nullptr, 0, d_D, d_E, out);
    return hipGetLastError();
}

// MultiMPPrior.log_prob (mp_priors_multi.py:209-210 -> torch MultivariateNormal.log_prob): the quadratic
// form (x - mu)^T Sigma^-1 (x - mu) on the block-tridiagonal precision, fp64, one wave per row
// (row r belongs to mode r % n_modes, the layout of a [.., modes, M] batch).
template <typename real>
__global__ void __launch_bounds__(64)
prior_quadform_kernel(int d, int T, long long rows, int n_modes, const real* __restrict__ x,
                      const real* __restrict__ means, const double* __restrict__ blocks4,
                      const double* __restrict__ Dblk, const double* __restrict__ Eblk, double* __restrict__ out) {
    const long long r = blockIdx.x;
    if (r >= rows) return;
    const int m = (int)(r % n_modes);
    const size_t M = (size_t)T * d, dd = (size_t)d * d;
    const real* xr = x + (size_t)r * M;
    const real* mu = means + (size_t)m * M;
    double acc = 0.;
    for (int t = threadIdx.x; t < T; t += 64) {
        const double* Dt = Dblk ? Dblk + ((size_t)m * T + t) * dd
                                : blocks4 + (t == 0 ? 0 : (t == T - 1 ? 2 : 1)) * dd;
        double y[SGPMP_MAX_D], yn[SGPMP_MAX_D];
        for (int i = 0; i < d; ++i) y[i] = (double)xr[(size_t)t * d + i] - (double)mu[(size_t)t * d + i];
        for (int i = 0; i < d; ++i) {
            double v = 0.;
            for (int j = 0; j < d; ++j) v += Dt[i * d + j] * y[j];
            acc += y[i] * v;
        }
        if (t + 1 < T) {                                  // 2 y_{t+1}^T E_t y_t,  E_t = Sigma^-1[t+1, t]
            const double* Et = Dblk ? Eblk + ((size_t)m * (T - 1) + t) * dd : blocks4 + 3 * dd;
            for (int i = 0; i < d; ++i) yn[i] = (double)xr[(size_t)(t + 1) * d + i] - (double)mu[(size_t)(t + 1) * d + i];
            for (int i = 0; i < d; ++i) {
                double v = 0.;
                for (int j = 0; j < d; ++j) v += Et[i * d + j] * y[j];
                acc += 2. * yn[i] * v;
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (threadIdx.x == 0) out[r] = acc;
}

hipError_t launch_prior_quadform(int dtype, int n, int T, long long rows, int n_modes, const void* x,
                                 const void* means, const PriorDev& prior, double* out, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    // per-mode blocks only while they are the prior's current precision (sgpmp_set_prior after
    // sgpmp_set_prior_blocks leaves Dm / Em allocated but stale)
    const double* Dm = prior.n_factor_modes > 0 ? prior.Dm : nullptr;
    const double* Em = prior.n_factor_modes > 0 ? prior.Em : nullptr;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((prior_quadform_kernel<double>), dim3((unsigned)rows), dim3(64), 0, stream, 2 * n, T, rows,
                           n_modes, (const double*)x, (const double*)means, prior.blocks, Dm, Em, out);
    else
        hipLaunchKernelGGL((prior_quadform_kernel<float>), dim3((unsigned)rows), dim3(64), 0, stream, 2 * n, T, rows,
                           n_modes, (const float*)x, (const float*)means, prior.blocks, Dm, Em, out);
    return hipGetLastError();
}
