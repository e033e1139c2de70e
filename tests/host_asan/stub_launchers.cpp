// Stand-ins for the kernel launchers of libsgpmp.so (the launch_* functions defined beside the kernels in the .hip files) for
// the host-side sanitizer run: each TOUCHES exactly the byte ranges the real kernel reads and writes -- on malloc-backed
// "device" memory (stub_hip.cpp) -- so AddressSanitizer checks every pointer and size the host layer hands to a launch.
// TEST INFRASTRUCTURE (tests/test_cpu_host.py); no arithmetic of the product lives here.
#include <cstdlib>
#include <cstring>

#include "sgpmp_internal.h"

static size_t esz(int dtype) { return dtype == SGPMP_F64 ? 8 : 4; }
static void rd(const void* p, size_t n) {                 // read every byte (a checksum the optimiser cannot drop)
    if (!p || !n) return;
    volatile unsigned char acc = 0;
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i += 61) acc = acc ^ b[i];
    acc = acc ^ b[n - 1];
}
static void wr(void* p, size_t n, int v = 0) { if (p && n) std::memset(p, v, n); }
static bool env1(const char* k) { const char* e = getenv(k); return e && *e == '1'; }

hipError_t launch_prior_factor(int n, int T, double, double, double, const double* qc, int, PriorDev out, hipStream_t) {
    const size_t d = 2 * n;
    rd(qc, sizeof(double) * n * n);
    wr(out.blocks, sizeof(double) * 4 * d * d); wr(out.G, sizeof(double) * T * d * d); wr(out.H, sizeof(double) * T * d * d);
    wr(out.iso64, sizeof(double) * T * 8); wr(out.iso32, sizeof(float) * T * 8); wr(out.Qinv, sizeof(double) * d * d);
    wr(out.G32, sizeof(float) * T * d * d); wr(out.H32, sizeof(float) * T * d * d);
    *out.status = env1("STUB_NOT_PD") ? 1 : 0;
    return hipSuccess;
}
hipError_t launch_prior_factor_blocks(int n, int T, int m, const double* D, const double* E, PriorDev out, hipStream_t) {
    const size_t d = 2 * n, dd = d * d;
    rd(D, sizeof(double) * m * T * dd); rd(E, sizeof(double) * m * (T - 1) * dd);
    wr(out.G, sizeof(double) * m * T * dd); wr(out.H, sizeof(double) * m * T * dd);
    wr(out.G32, sizeof(float) * m * T * dd); wr(out.H32, sizeof(float) * m * T * dd);
    *out.status = 0;
    return hipSuccess;
}
hipError_t launch_prior_quadform(int dtype, int n, int T, long long rows, int m, const void* x, const void* means, const PriorDev&, double* out, hipStream_t) {
    const size_t M = (size_t)T * 2 * n;
    rd(x, rows * M * esz(dtype)); rd(means, (size_t)m * M * esz(dtype)); wr(out, rows * sizeof(double));
    return hipSuccess;
}
hipError_t launch_sample(int dtype, int n, int T, const PriorDev& p, uint64_t, uint64_t, const void* means, int n_modes, int, int S,
                         const void* eps, int eps_modes, int, void* out, hipStream_t, const SgpmpToggles&, double* zero_stats) {
    const size_t M = (size_t)T * 2 * n;
    rd(p.iso32, sizeof(float) * T * 8); rd(means, (size_t)n_modes * M * esz(dtype));
    if (eps) rd(eps, (size_t)S * eps_modes * M * esz(dtype));
    wr(out, (size_t)n_modes * S * M * esz(dtype)); wr(zero_stats, sizeof(double) * SGPMP_STAT_SHARDS * 4);
    return hipSuccess;
}
hipError_t launch_noise(int dtype, int n, int T, int n_modes, int, int S, uint64_t, uint64_t, void* out, hipStream_t) {
    wr(out, (size_t)S * n_modes * T * 2 * n * esz(dtype));
    return hipSuccess;
}
hipError_t launch_cost(int dtype, int n, int T, const CostProgram&, const ChainDev* dch, const ChainDev&, const void* trajs, long long batch, long long,
                       const void* spheres, int ns, const void* isw, int rpp, double, void* costs, double* c64, hipStream_t, const SgpmpToggles&,
                       const char** picked) {
    const size_t M = (size_t)T * 2 * n;
    rd(dch, sizeof(ChainDev)); rd(trajs, batch * M * esz(dtype)); rd(spheres, (size_t)ns * 4 * esz(dtype));
    if (isw && rpp > 0) rd(isw, (size_t)(batch / rpp) * (T + 1) * 2 * n * esz(dtype));
    wr(costs, batch * esz(dtype)); wr(c64, batch * 8);
    *picked = "stub_cost";
    return hipSuccess;
}
bool update_ee_fold_fits(int, int, int, int S) { return S <= 4096; }
int update_regen_rows(int dtype, int, int T, int, int recipe) { return dtype == SGPMP_F32 && T % 2 == 0 && recipe == 1 ? 4 : 0; }
int fused_step_regen_recipe(int dtype, int n, int T, const PriorDev& pr, const CostProgram& prog, const ChainDev& ch, int P, int off, int S, int ns,
                            const SgpmpToggles& tg, int* seg_len) {
    if (seg_len) *seg_len = 0;
    return fused_step_eligible(dtype, n, T, pr, prog, ch, P, off, S, ns, tg) && prog.n_ee == 0 ? 1 : 0;
}
bool planar_seg_step(int, int, int, const PriorDev&, const CostProgram&, const ChainDev&, int, int, int, int, const SgpmpToggles&) { return false; }
// STUB_PERSIST=1 (with STUB_FUSED, STUB_TAIL): as if the launch could run several iterations (fused_planar_seg.inc: PERSIST) --
// sgpmp_optimize's chunking of a call is then exercised on the host; every fused launch is logged (draw, iterations) for the driver
bool planar_persist_step(int dtype, int n, int T, const PriorDev& pr, const CostProgram& prog, const ChainDev& ch, int P, int off, int S, int ns, const SgpmpToggles& tg) {
    return env1("STUB_PERSIST") && env1("STUB_TAIL") && !tg.no_persist_planar && fused_step_eligible(dtype, n, T, pr, prog, ch, P, off, S, ns, tg) && prog.n_ee == 0 &&
           update_regen_rows(dtype, n, T, S, 1) > 0;
}
static unsigned long long g_log_draw[4096];
static int g_log_iters[4096], g_log_n = 0;
extern "C" int stub_launch_log(int i, unsigned long long* draw, int* iters) {     // entry i of the log; returns the number of entries
    if (i >= 0 && i < g_log_n) { *draw = g_log_draw[i]; *iters = g_log_iters[i]; }
    return g_log_n;
}
extern "C" void stub_launch_log_clear() { g_log_n = 0; }
bool planar_tail_step(int, int, int, const PriorDev&, const CostProgram&, const ChainDev&, int, int, int, int, const SgpmpToggles&) { return false; }
bool fused_step_eligible(int dtype, int, int T, const PriorDev&, const CostProgram&, const ChainDev&, int P, int, int S, int, const SgpmpToggles& tg) {
    return env1("STUB_FUSED") && dtype == SGPMP_F32 && !tg.no_fused_step && T % 16 == 0 && S % 8 == 0 && P > 0;
}
hipError_t launch_fused_step(int dtype, int n, int T, const PriorDev& pr, const CostProgram& prog, const ChainDev& ch, uint64_t, uint64_t draw, const void* means,
                             int P, int off, int S, void* samples, const void* spheres, int ns, const void* isw, double* zero_stats, void* costs,
                             double* c64, hipStream_t, const SgpmpToggles& tg, const char** picked, bool* launched, const FusedDenseHost* dense,
                             bool* armed, RegenHost* regen, bool* tail_ran) {
    *launched = fused_step_eligible(dtype, n, T, pr, prog, ch, P, off, S, ns, tg) && samples && isw;
    if (armed) *armed = false;
    if (tail_ran) *tail_ran = false;
    if (regen) std::memset(regen, 0, sizeof(*regen));
    if (!*launched) return hipSuccess;
    const size_t M = (size_t)T * 2 * n;
    rd(means, (size_t)P * M * 4); rd(isw, (size_t)P * (T + 1) * 2 * n * 4); rd(spheres, (size_t)ns * 16);
    const bool nostore = dense && dense->nostore && dense->nnz && regen && prog.n_ee == 0 && update_regen_rows(dtype, n, T, S, 1) > 0;
    if (nostore) { regen->recipe = 1; regen->coef = pr.iso32p; regen->store_threshold = dense->store_threshold; regen->mode_offset = off; }
    else wr(samples, (size_t)P * S * M * 4);
    wr(costs, (size_t)P * S * 4); wr(c64, (size_t)P * S * 8); wr(zero_stats, sizeof(double) * SGPMP_STAT_SHARDS * 4);
    if (dense && dense->nnz) rd(dense->nnz, (size_t)P * 4);
    if (dense && dense->part && dense->nnz) {
        wr(dense->part, (size_t)P * ((S + 7) / 8) * (M + 4) * 4);
        if (armed) *armed = true;
    }
    // STUB_TAIL=1: as if the launch updated its particles itself (fused_planar_seg.inc: seg_update) -- touches what update_kernel would
    if (env1("STUB_TAIL") && nostore && dense->tail_done && tail_ran) {
        regen->recipe = 0;
        rd(dense->tail_done, 4); wr(dense->tail_done, 4); rd(dense->tail_acc, sizeof(double) * SGPMP_STAT_SHARDS * 4);
        wr(dense->stats_out, sizeof(double) * SGPMP_STAT_SHARDS * 4);
        wr(const_cast<void*>(means), (size_t)P * M * 4); wr(dense->weights, (size_t)P * S * 4); wr(dense->grad, (size_t)P * M * 4);
        wr(dense->means_prev, (size_t)P * M * 4); wr(const_cast<void*>(isw), (size_t)P * (T + 1) * 2 * n * 4); wr(dense->nnz, (size_t)P * 4, 1);
        *tail_ran = true;
    }
    if (dense && dense->tail_iters > 1 && !(tail_ran && *tail_ran)) return hipErrorInvalidValue;   // (as the real launcher: the caller asks planar_persist_step first)
    if (g_log_n < 4096) { g_log_draw[g_log_n] = draw; g_log_iters[g_log_n] = (dense && dense->tail_iters > 1) ? dense->tail_iters : 1; ++g_log_n; }
    if (picked) *picked = "stub_fused";
    return hipSuccess;
}
hipError_t launch_is_weights(int dtype, int n, int T, const PriorDev& p, const void* means, int P, double, void* out, double* zero_stats, hipStream_t) {
    rd(p.Qinv, sizeof(double) * 4 * n * n); rd(means, (size_t)P * T * 2 * n * esz(dtype));
    wr(out, (size_t)P * (T + 1) * 2 * n * esz(dtype)); wr(zero_stats, sizeof(double) * SGPMP_STAT_SHARDS * 4);
    return hipSuccess;
}
hipError_t launch_update(int dtype, int n, int T, int P, int S, const void* costs, int cdt, const void* samples, void* means, double, double, void* weights,
                         void* grad, void* means_prev, double* stats, hipStream_t, hipEvent_t done, const PriorDev* ip, void* isw_next, bool* isw_written,
                         void* means_copy, const float* part, unsigned* nnz, unsigned, const RegenHost* regen, const EeFoldHost* ee) {
    const size_t M = (size_t)T * 2 * n, w = esz(dtype);
    rd(costs, (size_t)P * S * esz(cdt));
    if (ee && ee->term) {                                      // the end-effector goal term inside the update: chain, rows, the cost output
        rd(ee->d_chain, sizeof(ChainDev)); rd(samples, (size_t)P * S * M * w);
        if (ee->costs) { rd(ee->costs, (size_t)P * S * w); wr(ee->costs, (size_t)P * S * w); }
    }
    if (regen && regen->recipe) {                              // store-free step: tables instead of rows
        rd(regen->coef, sizeof(float) * T * 8);
        if (regen->recipe == 2) rd(regen->pre, sizeof(float) * T * 4);
    } else {
        rd(samples, (size_t)P * S * M * w);
    }
    wr(means, (size_t)P * M * w); wr(weights, (size_t)P * S * w); wr(grad, (size_t)P * M * w); wr(means_prev, (size_t)P * M * w);
    wr(means_copy, (size_t)P * M * w);
    if (stats) for (int i = 0; i < SGPMP_STAT_SHARDS * 4; ++i) stats[i] += 1.;        // accumulates, as the kernel's atomics do
    if (ip) { rd(ip->Qinv, sizeof(double) * 4 * n * n); wr(isw_next, (size_t)P * (T + 1) * 2 * n * w); }
    if (isw_written) *isw_written = ip != nullptr && P > 0;
    if (part) rd(part, (size_t)P * (S / 8) * (M + 4) * 4);
    if (nnz) { rd(nnz, (size_t)P * 4); wr(nnz, (size_t)P * 4, 1); }
    if (done) return hipEventRecord(done, nullptr);
    return hipSuccess;
}
hipError_t launch_stats_add(double* dst, const double* src, hipStream_t) {
    for (int i = 0; i < SGPMP_STAT_SHARDS * 4; ++i) dst[i] += src[i];
    return hipSuccess;
}
hipError_t launch_mode_stats(int dtype, int n, int T, int P, long long, int, int G, const void* means, double* out, hipStream_t) {
    const size_t M = (size_t)T * 2 * n;
    rd(means, (size_t)P * M * esz(dtype)); rd(out, (size_t)G * (M + 1) * 2 * 8); wr(out, (size_t)G * (M + 1) * 2 * 8);
    return hipSuccess;
}
hipError_t launch_ee_goal(int dtype, int n, int T, const CostTerm&, const ChainDev* ch, const void* trajs, long long batch, void* costs, double* c64, hipStream_t) {
    rd(ch, sizeof(ChainDev)); rd(trajs, (size_t)batch * T * 2 * n * esz(dtype)); wr(costs, batch * esz(dtype)); wr(c64, batch * 8);
    return hipSuccess;
}
hipError_t launch_ee_grad(int dtype, int n, const CostTerm&, const ChainDev*, const void*, long long, int, long long, long long, void*, void*, hipStream_t) { (void)dtype; (void)n; return hipSuccess; }
hipError_t launch_field_grad(int dtype, int n, const CostTerm&, const ChainDev*, int, const void* q, long long batch, int, const void* sph, int ns, void* value, void* grad, hipStream_t) {
    (void)q; rd(sph, (size_t)ns * 4 * esz(dtype)); wr(value, batch * esz(dtype)); wr(grad, batch * n * esz(dtype));
    return hipSuccess;
}
hipError_t launch_gpmp_diag(int, const GpmpArgs& a, double* diag, hipStream_t) { wr(diag, (size_t)a.T * 2 * a.n * 8); return hipSuccess; }
hipError_t launch_gpmp_solve(int dtype, const GpmpArgs& a, void* means, void* d_theta, void* costs, hipStream_t, bool) {
    const size_t M = (size_t)a.T * 2 * a.n;
    wr(a.scratch, (size_t)a.P * a.T * 2 * 256 * 8); wr(means, a.P * M * esz(dtype)); wr(d_theta, a.P * M * esz(dtype)); wr(costs, a.P * esz(dtype));
    *a.status = 0;
    return hipSuccess;
}
hipError_t launch_link_dist(int, const void*, long long, int, const void*, int, int, double, void*, hipStream_t) { return hipSuccess; }
hipError_t launch_fk(int dtype, int n, const ChainDev* ch, int n_links, const void* q, long long batch, void* frames, hipStream_t) {
    rd(ch, sizeof(ChainDev)); rd(q, batch * n * esz(dtype)); wr(frames, (size_t)batch * n_links * 16 * esz(dtype));
    return hipSuccess;
}
hipError_t launch_grid_lookup(int dtype, const CostTerm&, const void* xy, long long batch, void* out, hipStream_t) {
    rd(xy, batch * 2 * esz(dtype)); wr(out, batch * esz(dtype));
    return hipSuccess;
}
hipError_t launch_field_eval(int dtype, const CostTerm&, const void* frames, long long batch, int nl, const void*, int, void* out, hipStream_t) {
    rd(frames, (size_t)batch * nl * 16 * esz(dtype)); wr(out, batch * esz(dtype));
    return hipSuccess;
}
// run-time chain kernels: none in this harness (chain_rtc.hip needs hipModule*; its compile path has its own CPU test)
const char* rtc_chain_get(const char*, int, RtcChain**) { return "no run-time compiler in the sanitizer harness"; }
hipFunction_t rtc_kernel(RtcChain*, int, bool) { return nullptr; }
hipError_t rtc_launch(hipFunction_t, unsigned, hipStream_t, void**, hipEvent_t) { return hipErrorNotSupported; }
const char* rtc_verify(RtcChain*, const ChainDev&, int, int* mismatch) { if (mismatch) *mismatch = 0; return "no run-time compiler in the sanitizer harness"; }
const char* rtc_error(const RtcChain*) { return ""; }
void rtc_stats(const RtcChain*, double* s, int* c, int* f) { if (s) *s = 0.; if (c) *c = 0; if (f) *f = 0; }
long long rtc_compile_check_c(const char*, int, char* err, size_t n) { if (err && n) err[0] = 0; return -1; }
