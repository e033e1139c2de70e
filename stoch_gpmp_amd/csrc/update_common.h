// Arithmetic shared by the update kernel (update.hip) and the in-launch update of the fused step launches
// (fused_tail.inc): the importance-sampling weight element (K5) and the descriptor of "the next step's weights".
#pragma once
#include "sgpmp_internal.h"

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

