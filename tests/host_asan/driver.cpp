// Host-side bookkeeping of libsgpmp.so under AddressSanitizer + UBSan, without a GPU: the C ABI (api.hip, comm.hip compiled
// host-only with the sanitizers) over a stub HIP runtime (stub_hip.cpp), stub launchers that touch the kernels' byte ranges
// (stub_launchers.cpp) and the shared-memory stand-in for librccl (tests/fake_rccl).  What runs here is the code the round-3
// advisor found four lifetime bugs in by reading: context create / destroy, prior and cost-program set-up, the step in all its
// sequencings (single chain, two particle-half chains, profiling events, per-step mode statistics, an empty shard), the
// statistics ring (more steps than slots), the "reduced" event table (more buffers than entries), communicator re-attach.
// Exit code 0 and no sanitizer report = pass.  TEST INFRASTRUCTURE (tests/test_cpu_host.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sgpmp.h"

extern "C" int stub_hip_live_objects(void);
extern "C" int stub_launch_log(int i, unsigned long long* draw, int* iters);   // stub_launchers.cpp: every fused launch's (draw, iterations)
extern "C" void stub_launch_log_clear();
static int g_multi = 0;                                // launches of several iterations seen by the checks below
extern "C" int hipMalloc(void**, size_t);
extern "C" int hipFree(void*);

#define CHECK(expr)                                                                                   \
    do {                                                                                              \
        const int rc_ = (expr);                                                                       \
        if (rc_ != SGPMP_OK) { std::fprintf(stderr, "FAILED %s -> %d: %s\n", #expr, rc_, sgpmp_last_error()); std::exit(2); } \
    } while (0)
#define EXPECT(expr, code)                                                                            \
    do {                                                                                              \
        const int rc_ = (expr);                                                                       \
        if (rc_ != (code)) { std::fprintf(stderr, "EXPECTED %d from %s, got %d: %s\n", (code), #expr, rc_, sgpmp_last_error()); std::exit(3); } \
    } while (0)

struct Dev {                                            // a "device" buffer (malloc-backed by the stub runtime)
    void* p = nullptr;
    explicit Dev(size_t bytes) { hipMalloc(&p, bytes); std::memset(p, 0, bytes); }
    ~Dev() { hipFree(p); }
    Dev(const Dev&) = delete;
};

static sgpmp_joint joint(double z, bool rev) { sgpmp_joint j; std::memset(&j, 0, sizeof(j)); j.xyz[2] = z; j.rpy[0] = rev ? 1.5707963 : 0.; j.revolute = rev ? 1 : 0; return j; }

static void run_planner(int n, int T, int P, int P_global, int offset, int S, int G, int dtype, bool with_comm, bool mode_stats, int steps) {
    const int d = 2 * n;
    const size_t M = (size_t)T * d, w = dtype == SGPMP_F64 ? 8 : 4;
    sgpmp_dims dims = {n, T, P, offset, P_global, S, G, P_global / G > 0 ? P_global / G : 1, dtype, 0};
    sgpmp_ctx* c = nullptr;
    CHECK(sgpmp_create(&dims, &c));
    const double ss[2] = {1e-3, 1e-3}, sg[2] = {0.8, 0.1}, sgoal[2] = {0.1, 0.07};
    CHECK(sgpmp_set_priors(c, 0.05, ss, sg, sgoal, nullptr));
    CHECK(sgpmp_set_prior(c, SGPMP_PRIOR_SAMPLE, 0.05, 1e-3, 0.1, -1., nullptr, nullptr));      // re-factor one, not goal-directed
    std::vector<double> start(d, 0.1), goals((size_t)G * d, 0.3), qc((size_t)n * n, 0.);
    for (int i = 0; i < n; ++i) qc[(size_t)i * n + i] = 4.;
    CHECK(sgpmp_set_prior(c, SGPMP_PRIOR_INIT, 0.05, 1e-3, 0.1, 0.1, qc.data(), nullptr));      // explicit Q_c^-1
    // cost program: GP + goal prior (+ link fields with a chain when n >= 3)
    std::vector<sgpmp_cost_desc> descs(2);
    std::memset(descs.data(), 0, sizeof(sgpmp_cost_desc) * descs.size());
    descs[0].kind = SGPMP_COST_GP; descs[0].flags = SGPMP_FLAG_GP_START; descs[0].sigma = 7e-4; descs[0].sigma2 = 1e-4; descs[0].dt = 0.05; descs[0].data = start.data();
    descs[1].kind = SGPMP_COST_GOAL_PRIOR; descs[1].sigma = 20.; descs[1].data = goals.data(); descs[1].dim0 = G; descs[1].dim1 = (P_global / G > 0 ? P_global / G : 1) * S;
    if (n >= 3) {
        sgpmp_cost_desc f; std::memset(&f, 0, sizeof(f));
        f.kind = SGPMP_COST_SELF; f.sigma = 0.01; f.sigma2 = 0.03; descs.push_back(f);
        f.kind = SGPMP_COST_SPHERES; f.flags = SGPMP_FIELD_SDF | SGPMP_FLAG_SDF_CLAMP; f.sigma = 0.01; f.num_interpolate = 2; f.interp_lo = 1; f.interp_hi = 3;
        f.alpha[0] = 1. / 3; f.alpha[1] = 2. / 3; descs.push_back(f);
        // + an end-effector goal term (CostGoal): inside sgpmp_step the update kernel evaluates it (EeFoldHost) -- unless SGPMP_NO_EE_FOLD
        static const double target[16] = {1, 0, 0, 0.3, 0, 1, 0, 0.1, 0, 0, 1, 0.5, 0, 0, 0, 1};
        sgpmp_cost_desc e; std::memset(&e, 0, sizeof(e));
        e.kind = SGPMP_COST_EE_GOAL; e.sigma = 1e-2; e.data = target; e.p0 = 1.0; e.p1 = 0.5; descs.push_back(e);
    }
    CHECK(sgpmp_set_costs(c, descs.data(), (int)descs.size()));
    if (n >= 3) {
        std::vector<sgpmp_joint> chain;
        for (int i = 0; i < n; ++i) chain.push_back(joint(0.1 + 0.05 * i, true));
        chain.push_back(joint(0.1, false));
        chain.push_back(joint(0., false));                       // coincident frame: merged by the chain analysis
        CHECK(sgpmp_set_fk(c, chain.data(), (int)chain.size()));
        int cid = -1;
        CHECK(sgpmp_fk_codegen_info(c, &cid, nullptr, nullptr, nullptr));
        EXPECT(sgpmp_set_fk_codegen(c, "struct ChainCode_rt { static constexpr int N = 3; };     "), SGPMP_EINVAL);   // fp64 / no compiler here
    }
    const size_t Pn = P > 0 ? (size_t)P : 1;
    // HOST_ASAN_INJECT=1: the costs buffer one element short -- the harness must NOT pass (the test of the test)
    const size_t shave = getenv("HOST_ASAN_INJECT") ? w : 0;
    Dev means(Pn * M * w), samples(Pn * S * M * w), costs(Pn * S * w - shave), weights(Pn * S * w), grad(Pn * M * w), prev(Pn * M * w), sph(5 * 4 * w);
    Dev stats_a(sizeof(double) * SGPMP_STAT_SHARDS * 4), stats_b(sizeof(double) * SGPMP_STAT_SHARDS * 4);
    Dev mode(sizeof(double) * (size_t)G * (M + 1) * 2);
    if (with_comm) {
        unsigned char id[128];
        CHECK(sgpmp_comm_unique_id(id));
        CHECK(sgpmp_comm_init(c, id, 1, 0));
        int wld = -1, rk = -1, ver = -1, hooks = -1;
        CHECK(sgpmp_comm_info(c, &wld, &rk, &ver));
        if (wld != 1 || rk != 0) { std::fprintf(stderr, "comm_info: world %d rank %d\n", wld, rk); std::exit(4); }
        const char* lib = sgpmp_comm_library(&hooks);
        if (!lib || !*lib || hooks != 1) { std::fprintf(stderr, "comm_library: '%s' hooks %d\n", lib ? lib : "(null)", hooks); std::exit(4); }
    }
    if (mode_stats) CHECK(sgpmp_set_step_mode_stats(c, (double*)mode.p));
    const int n_sph = n >= 3 ? 5 : 0;
    auto step = [&](int i, int flags) {
        CHECK(sgpmp_step(c, 7, (uint64_t)i, nullptr, 0, 0, means.p, samples.p, costs.p, weights.p, grad.p, prev.p, n_sph ? sph.p : nullptr, n_sph, 1.0, 0.1,
                         (double*)((i & 1) ? stats_b.p : stats_a.p), flags, nullptr));
    };
    for (int i = 0; i < steps; ++i) step(i, i ? SGPMP_STEP_MEANS_KEPT : 0);                       // more steps than ring slots
    CHECK(sgpmp_stats_wait(c, nullptr, nullptr));
    CHECK(sgpmp_profile_enable(c, 1));
    for (int i = 0; i < 5; ++i) step(100 + i, 0);
    double ms[4]; int64_t launches = 0;
    CHECK(sgpmp_profile_read(c, ms, &launches));
    if (launches != (P > 0 ? 5 : 0)) { std::fprintf(stderr, "profile_read: %lld launches\n", (long long)launches); std::exit(5); }
    CHECK(sgpmp_profile_enable(c, 0));
    // two particle-half chains
    CHECK(sgpmp_pipeline_begin(c, nullptr));
    EXPECT(sgpmp_pipeline_begin(c, nullptr), SGPMP_ESTATE);
    for (int i = 0; i < 11; ++i) step(200 + i, SGPMP_STEP_MEANS_KEPT | (i < 10 ? SGPMP_STEP_NO_SAMPLES : 0));   // store-free but the last
    CHECK(sgpmp_pipeline_end(c, nullptr));
    step(220, SGPMP_STEP_NO_SAMPLES);                           // ... and outside a bracket
    {
        std::vector<uint32_t> rc(Pn, 7u);
        CHECK(sgpmp_row_counts_get(c, rc.data()));
        CHECK(sgpmp_row_counts_set(c, rc.data()));
        CHECK(sgpmp_row_counts_set(c, nullptr));
        (void)sgpmp_store_free_steps(c);
        (void)sgpmp_multi_iteration_launches(c);
    }
    CHECK(sgpmp_pipeline_end(c, nullptr));                      // idempotent
    // round 6: the K-loop of optimize() behind the ABI (draw counters, alternating statistics slots, flags, the bracket), the
    // stream-ordered clear of the row counts, the noise read-back
    {
        Dev stats2(sizeof(double) * 2 * SGPMP_STAT_SHARDS * 4), prev_last(Pn * M * w);
        CHECK(sgpmp_optimize(c, 1, 7, 300, means.p, samples.p, costs.p, weights.p, grad.p, prev.p, prev_last.p, n_sph ? sph.p : nullptr, n_sph,
                             1.0, 0.1, (double*)stats2.p, 0, 0, nullptr));
        CHECK(sgpmp_optimize(c, 5, 7, 301, means.p, samples.p, costs.p, weights.p, grad.p, prev.p, prev_last.p, n_sph ? sph.p : nullptr, n_sph,
                             1.0, 0.1, (double*)stats2.p, 1, SGPMP_STEP_MEANS_KEPT | SGPMP_OPT_PIPELINE | SGPMP_OPT_STORE_FREE, nullptr));
        CHECK(sgpmp_optimize(c, 2, 7, 306, means.p, samples.p, costs.p, nullptr, nullptr, nullptr, nullptr, n_sph ? sph.p : nullptr, n_sph,
                             1.0, 0.1, nullptr, 0, SGPMP_OPT_STORE_FREE, nullptr));
        // the call's iterations, however sgpmp_optimize cuts them into launches (several iterations per launch where the launch
        // has that form -- STUB_PERSIST -- and at most `persist_max_iters` of them): every draw counter exactly once, in order
        if (P > 0 && !mode_stats && !with_comm) {
            for (long long cap : {0LL, 7LL, 2LL}) {
                CHECK(sgpmp_set_option(c, "persist_max_iters", cap));
                for (int K : {1, 2, 3, 8, 9, 16, 40}) {
                    stub_launch_log_clear();
                    CHECK(sgpmp_optimize(c, K, 7, 1000, means.p, samples.p, costs.p, weights.p, grad.p, prev.p, prev_last.p, n_sph ? sph.p : nullptr, n_sph,
                                         1.0, 0.1, (double*)stats2.p, 0, SGPMP_STEP_MEANS_KEPT | SGPMP_OPT_STORE_FREE, nullptr));
                    unsigned long long d = 0, next = 1000; int it = 0, total = 0;
                    const int nl = stub_launch_log(-1, &d, &it);
                    for (int i = 0; i < nl; ++i) {
                        stub_launch_log(i, &d, &it);
                        if (d != next || it < 1 || (cap >= 2 && it > cap)) { std::fprintf(stderr, "optimize(K=%d, cap %lld): launch %d has draw %llu x %d, expected draw %llu\n", K, cap, i, d, it, next); std::exit(6); }
                        next += (unsigned long long)it; total += it;
                        g_multi += it > 1 ? 1 : 0;
                    }
                    if (nl > 0 && total != K) { std::fprintf(stderr, "optimize(K=%d, cap %lld): %d iterations launched\n", K, cap, total); std::exit(6); }
                }
            }
            CHECK(sgpmp_set_option(c, "persist_max_iters", 0));
        }
        EXPECT(sgpmp_optimize(c, 0, 7, 0, means.p, samples.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1.0, 0.1, nullptr, 0, 0, nullptr), SGPMP_EINVAL);
        EXPECT(sgpmp_optimize(nullptr, 1, 7, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1.0, 0.1, nullptr, 0, 0, nullptr), SGPMP_EINVAL);
        CHECK(sgpmp_stats_wait(c, nullptr, nullptr));
        CHECK(sgpmp_row_counts_clear(c, nullptr));
        EXPECT(sgpmp_row_counts_clear(nullptr, nullptr), SGPMP_EINVAL);
        if (Pn > 0) {
            Dev eps((size_t)3 * Pn * M * w);
            CHECK(sgpmp_noise(c, 7, 2, Pn, 0, 3, eps.p, nullptr));
        }
        EXPECT(sgpmp_noise(c, 7, 2, 1, 0, 1, nullptr, nullptr), SGPMP_EINVAL);
    }
    if (mode_stats) CHECK(sgpmp_mode_stats_wait(c, nullptr));
    CHECK(sgpmp_mode_stats(c, means.p, (double*)mode.p, nullptr));
    if (with_comm) {
        // more statistics buffers than the "reduced" table keeps
        std::vector<Dev*> bufs;
        for (int i = 0; i < 12; ++i) {
            bufs.push_back(new Dev(sizeof(double) * SGPMP_STAT_SHARDS * 4));
            CHECK(sgpmp_allreduce_stats(c, (double*)bufs.back()->p, nullptr));
            CHECK(sgpmp_stats_wait(c, (double*)bufs.back()->p, nullptr));
        }
        CHECK(sgpmp_allreduce_f64(c, (double*)mode.p, (int64_t)G * (M + 1) * 2, nullptr));
        if ((long long)P == (long long)P_global && P > 0) {
            Dev all(Pn * M * w);
            CHECK(sgpmp_allgather_means(c, means.p, all.p, nullptr));
        }
        CHECK(sgpmp_stats_wait(c, nullptr, nullptr));
        for (Dev* b : bufs) delete b;
        unsigned char id[128];
        CHECK(sgpmp_comm_unique_id(id));
        CHECK(sgpmp_comm_init(c, id, 1, 0));                     // re-attach: the old communicator goes first
        step(300, 0);
        CHECK(sgpmp_comm_destroy(c));
        step(301, 0);
    }
    // the other entry points with sizes of their own
    if (P > 0) {
        Dev out(Pn * 3 * M * w), isw(Pn * (T + 1) * d * w), c64(Pn * S * 8);
        CHECK(sgpmp_sample(c, SGPMP_PRIOR_SAMPLE, 1, 2, means.p, P, offset, 3, nullptr, 0, 0, out.p, nullptr));
        CHECK(sgpmp_is_weights(c, means.p, P, 1.0, isw.p, nullptr));
        CHECK(sgpmp_cost_eval(c, samples.p, (int64_t)P * S, (int64_t)offset * S, n_sph ? sph.p : nullptr, n_sph, isw.p, S, costs.p, (double*)c64.p, nullptr));
        CHECK(sgpmp_update(c, c64.p, SGPMP_F64, samples.p, means.p, 1.0, 0.1, weights.p, grad.p, prev.p, (double*)stats_a.p, nullptr));
        EXPECT(sgpmp_update(c, c64.p, SGPMP_F64, samples.p, means.p, -1.0, 0.1, nullptr, nullptr, nullptr, nullptr, nullptr), SGPMP_EINVAL);
        int64_t dense = 0, armed = 0;
        CHECK(sgpmp_dense_particles(c, &dense, &armed));
        // per-mode precisions and their quadratic forms
        std::vector<double> D((size_t)2 * T * d * d, 0.), E((size_t)2 * (T - 1) * d * d, 0.);
        for (int m = 0; m < 2; ++m) for (int t = 0; t < T; ++t) for (int i = 0; i < d; ++i) D[(((size_t)m * T + t) * d + i) * d + i] = 2.;
        CHECK(sgpmp_set_prior_blocks(c, SGPMP_PRIOR_INIT, 2, D.data(), E.data(), nullptr));
        Dev q(sizeof(double) * 4);
        CHECK(sgpmp_prior_quadform(c, SGPMP_PRIOR_INIT, out.p, 4, means.p, 2, (double*)q.p, nullptr));
        std::vector<double> G_((size_t)2 * T * d * d), H_((size_t)2 * T * d * d);
        CHECK(sgpmp_get_prior(c, SGPMP_PRIOR_INIT, nullptr, G_.data(), H_.data()));
    }
    EXPECT(sgpmp_set_option(c, "no_such_switch", 1), SGPMP_EINVAL);
    for (const char* name : {"no_fused_step", "no_step_pipeline", "no_dense_partials", "gpmp_cholesky", "comm_packet_event", "no_ee_fold", "no_small_step",
                             "planar_store_free", "no_planar_tail"}) {
        CHECK(sgpmp_set_option(c, name, 1));
        step(400, 0);
        CHECK(sgpmp_set_option(c, name, 0));
    }
    CHECK(sgpmp_set_option(c, "k3_blocks", 64));
    CHECK(sgpmp_set_option(c, "pipe_split", 5));
    CHECK(sgpmp_set_option(c, "small_step_items", 64));
    CHECK(sgpmp_set_option(c, "store_free_min_bytes", 1));
    step(401, SGPMP_STEP_NO_SAMPLES);
    sgpmp_destroy(c);
}

int main() {
    EXPECT(sgpmp_create(nullptr, nullptr), SGPMP_EINVAL);
    sgpmp_dims bad = {9, 8, 1, 0, 1, 1, 1, 1, SGPMP_F32, 0};
    sgpmp_ctx* c = nullptr;
    EXPECT(sgpmp_create(&bad, &c), SGPMP_EINVAL);
    EXPECT(sgpmp_step(nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1., 1., nullptr, 0, nullptr), SGPMP_EINVAL);
    if (sgpmp_abi_version() != SGPMP_ABI_VERSION) return 6;
    const bool comm = getenv("SGPMP_RCCL_LIB") != nullptr;
    //           n   T   P   Pglobal off  S   G  dtype      comm   mode   steps
    run_planner(2, 16, 6, 6, 0, 8, 2, SGPMP_F64, false, false, 3);
    run_planner(2, 32, 64, 64, 0, 64, 4, SGPMP_F32, comm, true, 20);
    run_planner(2, 32, 16, 16, 0, 64, 2, SGPMP_F32, false, false, 5);         // planar-like, no communicator: optimize() may put several iterations into one launch
    run_planner(3, 16, 5, 40, 35, 8, 1, SGPMP_F32, comm, false, 11);        // the last shard of a ragged split
    run_planner(7, 16, 1024, 1024, 0, 8, 1, SGPMP_F32, comm, false, 10);      // big enough for two particle-half chains
    run_planner(7, 16, 0, 3, 3, 8, 1, SGPMP_F32, comm, true, 10);             // an empty shard still joins the collectives
    const int live = stub_hip_live_objects();
    if (live != 0) { std::fprintf(stderr, "%d streams / events / device buffers outlived their contexts\n", live); return 7; }
    std::printf("MULTI_ITERATION_LAUNCHES %d\n", g_multi);
    std::printf("HOST_ASAN_OK\n");
    return 0;
}
