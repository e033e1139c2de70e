// Host side of libsgpmp.so: the C ABI declared in include/sgpmp.h.
// Context bookkeeping, descriptor compilation and kernel sequencing only -- no arithmetic on
// trajectory data happens on the host.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "chain_code_generated.h"
#include "rng.h"
#include "sgpmp_internal.h"

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(SGPMP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct StepEvents { hipEvent_t ev[5]; bool has[4]; };    // has[k]: a kernel runs between ev[k] and ev[k+1]

// Consecutive steps of one optimize() call as TWO chains (sgpmp_pipeline_begin .. sgpmp_pipeline_end): the particles
// are independent (planner.py:263-275 reduces over the samples of one particle only), so the two halves of the
// particle range each run their own launch sequence -- sampler + sweep, update, sampler + sweep, ... -- on a stream
// of their own.  One chain leaves the chip idle while its update kernel runs (13.9 us of latency-bound work plus
// two launch gaps in a 210 us iteration at config 3); with two chains the other half's sampler + sweep launch
// fills that time (tools/two_chain_probe.py: 0.2161 -> 0.1975 ms per iteration).  Results are those of the
// unsplit step bit for bit: the kernels see the same particles with the same global indices.
struct StepPipe {
    bool active = false;          // between _begin and _end
    bool forked = false;          // the chains are ahead of the caller's stream
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[2] = {nullptr, nullptr};
    double* stats2 = nullptr;     // statistics slot of the second half [SGPMP_STAT_SHARDS][4]
    double* last_stats = nullptr; // the caller's buffer the last split step accumulated half 0 into
    long long split_steps = 0;    // steps run as two chains so far (tests, bench)
};

struct sgpmp_ctx {
    sgpmp_dims dims;
    int d, M;
    size_t esz;
    PriorDev prior[2];
    double* d_qc;                 // scratch [n*n] for K1's Q_c^-1
    CostProgram h_prog;
    CostProgram* d_prog;
    bool have_costs, prog_dirty;
    ChainDev h_chain;
    ChainDev* d_chain;
    bool have_chain;
    std::vector<void*> owned;     // device buffers holding term data (start / goal states)
    void* d_isw;                  // [P, T+1, d] importance-sampling weights (K5 output)
    // GPMP (Gauss-Newton planner) scratch, allocated on first use
    void* d_fval[4];              // per link-field term: [P, T-1] values
    void* d_fgrad[4];             // [P, T-1, n] gradients
    double* d_gscratch;           // [P][T][2][256] block factors
    double* d_diag;               // [T*d] field part of sum_p diag(A^T K A)
    int* d_gstatus;
    double* d_costs64;            // [P, S]
    bool profiling;
    std::vector<StepEvents> events;
    SgpmpToggles tg;              // development switches, read from the environment at creation
    SgpmpComm* comm;              // RCCL communicator (multi-GPU runs), or null
    // importance-sampling weights prepared by the previous step's update kernel for the means it wrote
    bool isw_ready;
    const void* isw_means;        // the means buffer they belong to
    double isw_temperature;
    const char* last_cost_kernel; // name of the cost-sweep kernel the dispatcher picked last
    StepPipe pipe;                // two-chain execution of consecutive steps (sgpmp_pipeline_begin / _end)
    // the update inside fused_planar_seg_kernel (store-free steps, S = 64): per-launch finished-particle counters and
    // statistics accumulators ([0]: whole range / first half, [1]: second half of a two-chain step); zero between
    // launches by construction (the launch's last particle resets them)
    unsigned* d_done = nullptr;      // [2]
    double* d_tail_acc = nullptr;    // [2][SGPMP_STAT_SHARDS][4]
    // per-goal mean statistics once per iteration (sgpmp_set_step_mode_stats): the update kernel leaves a snapshot of
    // the new means, a side stream reduces it per goal and all-reduces the sums -- nothing on the steps' own stream
    double* ms_buf = nullptr;        // caller's [G][M+1][2] buffer, or null: off
    void* ms_snap[2] = {nullptr, nullptr};   // snapshots of the new means [P][M] (ring of two)
    hipEvent_t ms_read[2] = {nullptr, nullptr};   // side stream: snapshot i has been reduced
    hipEvent_t ms_ready = nullptr;   // steps' stream: snapshot written
    hipStream_t ms_side = nullptr;   // the side stream when no communicator is attached
    bool ms_used[2] = {false, false};
    unsigned long long ms_step = 0;
    // per particle: rows that carried weight in its last in-step update (written by update_kernel, read by the NEXT step's
    // launch: which particles get softmax partials -- dense-weight regime, FusedArgs::part -- and, in a store-free step,
    // which particles' rows are written at all); + the partials themselves
    float* d_part = nullptr;         // [P][ceil(S / 8)][4 + M], allocated by the first step that can use it
    unsigned* d_nnz = nullptr;       // [P], allocated (zero) by the first fp32 step
    long long dense_armed_steps = 0, store_free_steps = 0;
    long long multi_iteration_launches = 0;
    int tail_iters_next = 0;         // sgpmp_optimize -> its next sgpmp_step: iterations the step's launch runs itself (fused_planar_seg.inc: PERSIST)
    int last_step_launches = 0;      // kernels the last sgpmp_step enqueued for its particle range (1: everything in one launch)
    hipStream_t k1_side = nullptr;   // sgpmp_set_priors: the second factorisation's stream
    hipEvent_t k1_fork = nullptr;
};

// name -> field of SgpmpToggles (environment variable = "SGPMP_" + upper-case name)
static const struct { const char* name; int SgpmpToggles::*flag; } kToggleNames[] = {
    {"force_generic_fk", &SgpmpToggles::force_generic_fk}, {"no_flat_program", &SgpmpToggles::no_flat_program},
    {"no_chain_codegen", &SgpmpToggles::no_chain_codegen}, {"no_dual_sweep", &SgpmpToggles::no_dual_sweep},
    {"k3_no_one", &SgpmpToggles::k3_no_one}, {"k3_no_lds_prefetch", &SgpmpToggles::k3_no_lds_prefetch},
    {"no_small_sampler", &SgpmpToggles::no_small_sampler}, {"no_fused_step", &SgpmpToggles::no_fused_step},
    {"no_chunked_sweep", &SgpmpToggles::no_chunked_sweep}, {"no_step_pipeline", &SgpmpToggles::no_step_pipeline},
    {"comm_packet_event", &SgpmpToggles::comm_packet_event}, {"no_planar_seg", &SgpmpToggles::no_planar_seg}, {"planar_store_free", &SgpmpToggles::planar_store_free}, {"no_planar_tail", &SgpmpToggles::no_planar_tail}, {"no_persist_planar", &SgpmpToggles::no_persist_planar},
    {"no_small_step", &SgpmpToggles::no_small_step}, {"no_ee_fold", &SgpmpToggles::no_ee_fold}, {"no_dense_partials", &SgpmpToggles::no_dense_partials}, {"gpmp_cholesky", &SgpmpToggles::gpmp_cholesky},
    {"f64_fields_f32", &SgpmpToggles::f64_fields_f32},
};

static void toggles_from_env(SgpmpToggles& tg) {
    std::memset(&tg, 0, sizeof(tg));
    for (const auto& t : kToggleNames) {
        std::string env = "SGPMP_";
        for (const char* p = t.name; *p; ++p) env += (char)toupper(*p);
        const char* v = getenv(env.c_str());
        tg.*(t.flag) = (v && *v && std::strcmp(v, "0") != 0) ? 1 : 0;
    }
    if (const char* e = getenv("SGPMP_K3_BLOCKS")) tg.k3_blocks = atoll(e);
    if (const char* e = getenv("SGPMP_PIPE_SPLIT")) tg.pipe_split = atoll(e);
    if (const char* e = getenv("SGPMP_STORE_FREE_MIN_BYTES")) tg.store_free_min_bytes = atoll(e);
    if (const char* e = getenv("SGPMP_SMALL_STEP_ITEMS")) tg.small_step_items = atoll(e);
    if (const char* e = getenv("SGPMP_PERSIST_MAX_ITERS")) tg.persist_max_iters = atoll(e);
}

#ifdef SGPMP_HOST_TIMING      // diagnostic build (tools/host_step_cost.py): host nanoseconds of sgpmp_step's segments, printed by sgpmp_destroy
#include <ctime>
static double g_ht[8]; static long long g_htn;
static inline double ht_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e9 * (double)t.tv_sec + (double)t.tv_nsec; }
#define HT_START() double ht_t = ht_now()
#define HT(i) do { const double n_ = ht_now(); g_ht[i] += n_ - ht_t; ht_t = n_; } while (0)
#else
#define HT_START() do {} while (0)
#define HT(i) do {} while (0)
#endif
extern "C" int sgpmp_abi_version(void) { return SGPMP_ABI_VERSION; }
extern "C" int sgpmp_philox_rounds(void) { return SGPMP_PHILOX_ROUNDS; }
extern "C" const char* sgpmp_last_error(void) { return g_err.c_str(); }

static int alloc_prior(sgpmp_ctx* c, PriorDev& p) {
    const int d = c->d, T = c->dims.traj_len;
    std::memset(&p, 0, sizeof(p));
    HIPCHK(hipMalloc(&p.blocks, sizeof(double) * 4 * d * d));
    HIPCHK(hipMalloc(&p.G, sizeof(double) * T * d * d));
    HIPCHK(hipMalloc(&p.H, sizeof(double) * T * d * d));
    // (the fp32 tables hold whole chunks of 16 rows, zero past T: the fused launch's last chunk reads them with one scalar load per
    // two rows whatever T is)
    const size_t Tpad = ((size_t)T + 15) / 16 * 16;
    HIPCHK(hipMalloc(&p.iso64, sizeof(double) * T * 8));
    HIPCHK(hipMalloc(&p.iso32, sizeof(float) * Tpad * 8));
    HIPCHK(hipMalloc(&p.iso32p, sizeof(float) * Tpad * 8));
    HIPCHK(hipMemset(p.iso32, 0, sizeof(float) * Tpad * 8));
    HIPCHK(hipMemset(p.iso32p, 0, sizeof(float) * Tpad * 8));
    HIPCHK(hipMalloc(&p.slabpre, sizeof(float) * 5 * T * 4));
    HIPCHK(hipMalloc(&p.scan64, sizeof(double) * T * 28));
    HIPCHK(hipMalloc(&p.Qinv, sizeof(double) * d * d));
    HIPCHK(hipMalloc(&p.G32, sizeof(float) * T * d * d));
    HIPCHK(hipMalloc(&p.H32, sizeof(float) * T * d * d));
    HIPCHK(hipMalloc(&p.status, sizeof(int)));
    return SGPMP_OK;
}

static void free_prior(PriorDev& p) {
    hipFree(p.blocks); hipFree(p.G); hipFree(p.H); hipFree(p.iso64); hipFree(p.iso32); hipFree(p.iso32p); hipFree(p.slabpre);
    hipFree(p.scan64); hipFree(p.Qinv); hipFree(p.G32); hipFree(p.H32); hipFree(p.status); hipFree(p.Dm); hipFree(p.Em);
    std::memset(&p, 0, sizeof(p));
}

extern "C" int sgpmp_create(const sgpmp_dims* dims, sgpmp_ctx** out) {
    if (!dims || !out) return fail(SGPMP_EINVAL, "sgpmp_create: null argument");
    if (dims->n_dof < 1 || dims->n_dof > SGPMP_MAX_DOF)
        return fail(SGPMP_EINVAL, "sgpmp_create: n_dof must be in [1, 8] (state block = one 16x16 tile)");
    if (dims->traj_len < 2) return fail(SGPMP_EINVAL, "sgpmp_create: traj_len must be >= 2");
    if (dims->num_particles < 0 || dims->num_samples < 1 || dims->num_goals < 1 ||
        dims->num_particles_per_goal < 1)
        return fail(SGPMP_EINVAL, "sgpmp_create: bad particle/sample/goal counts");
    if (dims->dtype != SGPMP_F32 && dims->dtype != SGPMP_F64)
        return fail(SGPMP_EINVAL, "sgpmp_create: dtype must be SGPMP_F32 or SGPMP_F64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SGPMP_EHIP, "sgpmp_create: no HIP device visible (this library has no CPU path)");
    sgpmp_ctx* c = new sgpmp_ctx();
    c->dims = *dims;
    c->d = 2 * dims->n_dof;
    c->M = c->d * dims->traj_len;
    c->esz = dims->dtype == SGPMP_F64 ? 8 : 4;
    c->d_qc = nullptr; c->d_prog = nullptr; c->d_chain = nullptr; c->d_isw = nullptr;
    c->d_costs64 = nullptr;
    for (int i = 0; i < 4; ++i) { c->d_fval[i] = nullptr; c->d_fgrad[i] = nullptr; }
    c->d_gscratch = nullptr; c->d_diag = nullptr; c->d_gstatus = nullptr;
    c->have_costs = false; c->prog_dirty = false; c->have_chain = false; c->profiling = false;
    std::memset(&c->h_prog, 0, sizeof(c->h_prog));
    std::memset(&c->h_chain, 0, sizeof(c->h_chain));
    toggles_from_env(c->tg);
    c->last_cost_kernel = "";
    c->comm = nullptr;
    c->isw_ready = false; c->isw_means = nullptr; c->isw_temperature = 0.;
    int rc;
    if ((rc = alloc_prior(c, c->prior[0])) != SGPMP_OK) return rc;
    if ((rc = alloc_prior(c, c->prior[1])) != SGPMP_OK) return rc;
    HIPCHK(hipMalloc(&c->d_qc, sizeof(double) * 2 * dims->n_dof * dims->n_dof));   // (two: sgpmp_set_priors)
    HIPCHK(hipMalloc(&c->d_prog, sizeof(CostProgram)));
    HIPCHK(hipMalloc(&c->d_chain, sizeof(ChainDev)));
    const size_t P = (size_t)(dims->num_particles > 0 ? dims->num_particles : 1);
    HIPCHK(hipMalloc(&c->d_isw, P * (dims->traj_len + 1) * c->d * c->esz));
    HIPCHK(hipMalloc(&c->d_costs64, P * dims->num_samples * sizeof(double)));
    HIPCHK(hipMalloc(&c->d_done, 2 * sizeof(unsigned)));
    HIPCHK(hipMalloc(&c->d_tail_acc, 2 * SGPMP_STAT_SHARDS * 4 * sizeof(double)));
    HIPCHK(hipMemset(c->d_done, 0, 2 * sizeof(unsigned)));
    HIPCHK(hipMemset(c->d_tail_acc, 0, 2 * SGPMP_STAT_SHARDS * 4 * sizeof(double)));
    *out = c;
    return SGPMP_OK;
}

extern "C" int sgpmp_set_option(sgpmp_ctx* c, const char* name, long long value) {
    if (!c || !name) return fail(SGPMP_EINVAL, "sgpmp_set_option: null argument");
    if (std::strcmp(name, "k3_blocks") == 0) { c->tg.k3_blocks = value; return SGPMP_OK; }
    if (std::strcmp(name, "pipe_split") == 0) { c->tg.pipe_split = value; return SGPMP_OK; }
    if (std::strcmp(name, "store_free_min_bytes") == 0) { c->tg.store_free_min_bytes = value; return SGPMP_OK; }
    if (std::strcmp(name, "small_step_items") == 0) { c->tg.small_step_items = value; return SGPMP_OK; }
    if (std::strcmp(name, "persist_max_iters") == 0) { c->tg.persist_max_iters = value; return SGPMP_OK; }
    for (const auto& t : kToggleNames)
        if (std::strcmp(name, t.name) == 0) { c->tg.*(t.flag) = value != 0; return SGPMP_OK; }
    return fail(SGPMP_EINVAL, std::string("sgpmp_set_option: unknown option ") + name);
}

extern "C" const char* sgpmp_last_cost_kernel(sgpmp_ctx* c) { return c ? c->last_cost_kernel : ""; }

// ---------------------------------------------------------------------------------- collectives
#define COMMCHK(expr)                                                                        \
    do {                                                                                     \
        const char* e_ = (expr);                                                             \
        if (e_) return fail(SGPMP_EHIP, std::string(#expr) + ": " + e_);                     \
    } while (0)

extern "C" int sgpmp_comm_unique_id(unsigned char* out128) {
    if (!out128) return fail(SGPMP_EINVAL, "sgpmp_comm_unique_id: null argument");
    COMMCHK(comm_unique_id(out128));
    return SGPMP_OK;
}

extern "C" int sgpmp_comm_init(sgpmp_ctx* c, const unsigned char* id128, int world_size, int rank) {
    if (!c || !id128 || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(SGPMP_EINVAL, "sgpmp_comm_init: bad argument");
    if (c->comm) { comm_destroy(c->comm); c->comm = nullptr; }
    COMMCHK(comm_create(id128, world_size, rank, &c->comm));
    return SGPMP_OK;
}

extern "C" int sgpmp_comm_destroy(sgpmp_ctx* c) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_comm_destroy: null argument");
    comm_destroy(c->comm);
    c->comm = nullptr;
    return SGPMP_OK;
}

extern "C" int sgpmp_comm_info(sgpmp_ctx* c, int* world, int* rank, int* rccl_version) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_comm_info: null context");
    if (world) *world = 0;
    if (rank) *rank = 0;
    if (rccl_version) *rccl_version = 0;
    if (!c->comm) return SGPMP_OK;
    COMMCHK(comm_info(c->comm, world, rank, rccl_version));
    return SGPMP_OK;
}

extern "C" const char* sgpmp_comm_library(int* test_hooks) {
    if (test_hooks) *test_hooks = comm_test_hooks();
    return comm_library_name();
}

// ---- per-goal statistics of the particle means (the all-reduce's north_star content) -----------------------------
static size_t mode_stats_count(const sgpmp_ctx* c) { return (size_t)c->dims.num_goals * (c->M + 1) * 2; }

extern "C" int sgpmp_mode_stats(sgpmp_ctx* c, const void* means, double* out, void* stream) {
    if (!c || !out || (c->dims.num_particles > 0 && !means)) return fail(SGPMP_EINVAL, "sgpmp_mode_stats: null argument");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(out, 0, mode_stats_count(c) * sizeof(double), st));
    HIPCHK(launch_mode_stats(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, c->dims.num_particles, c->dims.particle_offset,
                             c->dims.num_particles_per_goal, c->dims.num_goals, means, out, st));
    return SGPMP_OK;
}

extern "C" int sgpmp_allreduce_f64(sgpmp_ctx* c, double* buf, int64_t count, void* stream) {
    if (!c || !buf || count < 1) return fail(SGPMP_EINVAL, "sgpmp_allreduce_f64: bad argument");
    if (!c->comm) return fail(SGPMP_ESTATE, "sgpmp_allreduce_f64: no communicator (sgpmp_comm_init)");
    COMMCHK(comm_allreduce_f64(c->comm, buf, (size_t)count, (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_set_step_mode_stats(sgpmp_ctx* c, double* buf) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_set_step_mode_stats: null context");
    if (buf && !c->ms_ready) {
        const size_t bytes = (size_t)(c->dims.num_particles > 0 ? c->dims.num_particles : 1) * c->M * c->esz;
        HIPCHK(hipEventCreateWithFlags(&c->ms_ready, hipEventDisableTiming));
        HIPCHK(hipStreamCreateWithFlags(&c->ms_side, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipMalloc(&c->ms_snap[i], bytes));
            HIPCHK(hipEventCreateWithFlags(&c->ms_read[i], hipEventDisableTiming));
        }
    }
    c->ms_buf = buf;
    return SGPMP_OK;
}

// After the update kernel of a step left snapshot `slot` of the new means (steps' stream `st`): reduce it per goal on
// the side stream into the caller's buffer and sum over the ranks.  The side stream is the communicator's when there
// is one (the collectives of a context stay on one stream, in one order on every rank).
static int step_mode_stats(sgpmp_ctx* c, int slot, hipStream_t st) {
    hipStream_t side = c->comm ? comm_side_stream(c->comm) : c->ms_side;
    HIPCHK(hipEventRecord(c->ms_ready, st));
    HIPCHK(hipStreamWaitEvent(side, c->ms_ready, 0));
    HIPCHK(hipMemsetAsync(c->ms_buf, 0, mode_stats_count(c) * sizeof(double), side));
    HIPCHK(launch_mode_stats(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, c->dims.num_particles, c->dims.particle_offset,
                             c->dims.num_particles_per_goal, c->dims.num_goals, c->ms_snap[slot], c->ms_buf, side));
    HIPCHK(hipEventRecord(c->ms_read[slot], side));
    c->ms_used[slot] = true;
    if (c->comm) COMMCHK(comm_allreduce_f64(c->comm, c->ms_buf, mode_stats_count(c), side));
    return SGPMP_OK;
}

extern "C" int sgpmp_mode_stats_wait(sgpmp_ctx* c, void* stream) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_mode_stats_wait: null context");
    hipStream_t st = (hipStream_t)stream;
    if (c->comm && c->ms_buf) COMMCHK(comm_stats_wait(c->comm, c->ms_buf, st));
    for (int i = 0; i < 2; ++i)
        if (c->ms_used[i]) HIPCHK(hipStreamWaitEvent(st, c->ms_read[i], 0));
    return SGPMP_OK;
}

extern "C" int sgpmp_allreduce_stats(sgpmp_ctx* c, double* stats, void* stream) {
    if (!c || !stats) return fail(SGPMP_EINVAL, "sgpmp_allreduce_stats: null argument");
    if (!c->comm) return fail(SGPMP_ESTATE, "sgpmp_allreduce_stats: no communicator (sgpmp_comm_init)");
    COMMCHK(comm_allreduce_stats(c->comm, stats, (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_stats_wait(sgpmp_ctx* c, double* stats, void* stream) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_stats_wait: null argument");
    if (c->comm) COMMCHK(comm_stats_wait(c->comm, stats, (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_allgather_means(sgpmp_ctx* c, const void* local_means, void* all_means, void* stream) {
    if (!c || !local_means || !all_means) return fail(SGPMP_EINVAL, "sgpmp_allgather_means: null argument");
    if (!c->comm) return fail(SGPMP_ESTATE, "sgpmp_allgather_means: no communicator (sgpmp_comm_init)");
    if ((long long)c->dims.num_particles * comm_world(c->comm) != c->dims.num_particles_global)
        return fail(SGPMP_EINVAL, "sgpmp_allgather_means: shards must be equal (ragged shards: gather on the host side)");
    COMMCHK(comm_allgather(c->comm, local_means, all_means, (size_t)c->dims.num_particles * c->M * c->esz,
                           (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" void sgpmp_destroy(sgpmp_ctx* c) {
#ifdef SGPMP_HOST_TIMING
    if (g_htn) {
        std::fprintf(stderr, "[host timing] %lld steps: checks+split %.2f us, eligibility %.2f, fused launch %.2f, update launch %.2f\n", g_htn,
                     1e-3 * g_ht[0] / g_htn, 1e-3 * g_ht[1] / g_htn, 1e-3 * g_ht[2] / g_htn, 1e-3 * g_ht[3] / g_htn);
        g_htn = 0; for (double& v : g_ht) v = 0.;
    }
#endif
    if (!c) return;
    comm_destroy(c->comm);
    free_prior(c->prior[0]);
    free_prior(c->prior[1]);
    hipFree(c->d_qc); hipFree(c->d_prog); hipFree(c->d_chain); hipFree(c->d_isw);
    hipFree(c->d_costs64); hipFree(c->d_done); hipFree(c->d_tail_acc);
    hipFree(c->d_part); hipFree(c->d_nnz);
    if (c->ms_side) { hipStreamSynchronize(c->ms_side); hipStreamDestroy(c->ms_side); }
    for (int i = 0; i < 2; ++i) { hipFree(c->ms_snap[i]); if (c->ms_read[i]) hipEventDestroy(c->ms_read[i]); }
    if (c->ms_ready) hipEventDestroy(c->ms_ready);
    for (int i = 0; i < 4; ++i) { hipFree(c->d_fval[i]); hipFree(c->d_fgrad[i]); }
    hipFree(c->d_gscratch); hipFree(c->d_diag); hipFree(c->d_gstatus);
    for (void* p : c->owned) hipFree(p);
    for (auto& se : c->events)
        for (auto& e : se.ev) hipEventDestroy(e);
    for (int h = 0; h < 2; ++h) {
        if (c->pipe.side[h]) { hipStreamSynchronize(c->pipe.side[h]); hipStreamDestroy(c->pipe.side[h]); }
        if (c->pipe.join_ev[h]) hipEventDestroy(c->pipe.join_ev[h]);
    }
    if (c->pipe.fork_ev) hipEventDestroy(c->pipe.fork_ev);
    hipFree(c->pipe.stats2);
    if (c->k1_side) hipStreamDestroy(c->k1_side);
    if (c->k1_fork) hipEventDestroy(c->k1_fork);
    delete c;
}

// Prefix products of the scan's 2 x 2 propagators H_t = [[h11, h12], [h21, h22]] from the start of a time segment: what the
// planar launches need to carry a segment's true start state through the segment (fused_planar.inc: the in-chunk segments,
// fused_planar_seg.inc: a wave's 8 or 16 waypoints).  Once per factorisation, in fp64 on the host from K1's coefficients.
static int upload_slab_prefix(sgpmp_ctx* c, PriorDev& p, hipStream_t st) {
    // (only the point-mass launches use the tables: programs without forward kinematics run for n = 2, 3)
    if (!p.isotropic || (c->dims.n_dof != 2 && c->dims.n_dof != 3)) return SGPMP_OK;
    const int T = c->dims.traj_len;
    std::vector<double> iso((size_t)T * 8);
    HIPCHK(hipMemcpyAsync(iso.data(), p.iso64, sizeof(double) * T * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<float> tab((size_t)5 * T * 4, 0.f);
    for (int wi = 2; wi < 5; ++wi) {       // tables: (0, 1: unused since round 5,) the scan's own segments, segments of 8, of 16
        // (segments of the in-chunk scan: 16-waypoint chunks cut into 4 segments for n = 2, 2 for n = 3: fused_planar.inc;
        // segments of 8 / 16 waypoints: one wave each in fused_planar_seg.inc)
        const int L = wi == 2 ? (c->dims.n_dof == 2 ? 4 : 8) : wi == 3 ? 8 : 16;
        double p00 = 1., p01 = 0., p10 = 0., p11 = 1.;
        for (int t = 0; t < T; ++t) {
            if (t % L == 0) { p00 = 1.; p01 = 0.; p10 = 0.; p11 = 1.; }
            const double* h = &iso[(size_t)t * 8];                   // g11 g21 g22 h11 h12 h21 h22 -
            const double n00 = h[3] * p00 + h[4] * p10, n01 = h[3] * p01 + h[4] * p11;
            const double n10 = h[5] * p00 + h[6] * p10, n11 = h[5] * p01 + h[6] * p11;
            p00 = n00; p01 = n01; p10 = n10; p11 = n11;
            float* o = &tab[((size_t)wi * T + t) * 4];
            o[0] = (float)p00; o[1] = (float)p01; o[2] = (float)p10; o[3] = (float)p11;
        }
    }
    HIPCHK(hipMemcpyAsync(p.slabpre, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return SGPMP_OK;
}

// The Kogge-Stone tables of the fp64 one-launch step (cost_sweep_kernel.inc: GenArgs64): the recurrence y_t = H_t y_{t-1} + b_t of
// the 64 waypoints of a pass, all lanes at once, needs the propagator products A_t^(r) = H_t .. H_{t - 2^r + 1} of every round
// and C_t = H_t .. H_{t0} for the state carried into a later pass.  Once per factorisation, fp64, on the host from K1's
// coefficients (the propagators of an isotropic prior are the same for every dof, particle and sample).
static int upload_scan64(sgpmp_ctx* c, PriorDev& p, hipStream_t st) {
    if (!p.isotropic || c->dims.dtype != SGPMP_F64) return SGPMP_OK;
    const int T = c->dims.traj_len;
    std::vector<double> iso((size_t)T * 8);
    HIPCHK(hipMemcpyAsync(iso.data(), p.iso64, sizeof(double) * T * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    struct M2 { double a, b, c, d; };
    auto mul = [](const M2& x, const M2& y) { return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; };
    std::vector<double> tab((size_t)T * 28, 0.);
    std::vector<M2> cur((size_t)T), nxt((size_t)T);
    for (int t = 0; t < T; ++t) cur[t] = M2{iso[(size_t)t * 8 + 3], iso[(size_t)t * 8 + 4], iso[(size_t)t * 8 + 5], iso[(size_t)t * 8 + 6]};
    for (int r = 0; r < 6; ++r) {
        const int d = 1 << r;
        for (int t = 0; t < T; ++t) {
            const int t0 = t & ~63;
            double* o = &tab[(size_t)t * 28 + 4 * r];
            if (t - d >= t0) {
                o[0] = cur[t].a; o[1] = cur[t].b; o[2] = cur[t].c; o[3] = cur[t].d;
                nxt[t] = mul(cur[t], cur[t - d]);                // H_t .. H_{t - 2d + 1} (used only where t - 2d >= t0)
            } else nxt[t] = cur[t];
        }
        cur.swap(nxt);
    }
    M2 pre{1., 0., 0., 1.};
    for (int t = 0; t < T; ++t) {
        if ((t & 63) == 0) pre = M2{1., 0., 0., 1.};
        pre = mul(M2{iso[(size_t)t * 8 + 3], iso[(size_t)t * 8 + 4], iso[(size_t)t * 8 + 5], iso[(size_t)t * 8 + 6]}, pre);
        double* o = &tab[(size_t)t * 28 + 24];
        o[0] = pre.a; o[1] = pre.b; o[2] = pre.c; o[3] = pre.d;
    }
    HIPCHK(hipMemcpyAsync(p.scan64, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return SGPMP_OK;
}

extern "C" int sgpmp_set_prior(sgpmp_ctx* c, int which, double dt, double sigma_start, double sigma_gp,
                               double sigma_goal, const double* qc_inv, void* stream) {
    if (!c || (which != 0 && which != 1)) return fail(SGPMP_EINVAL, "sgpmp_set_prior: bad argument");
    if (!(dt > 0.) || !(sigma_start > 0.) || (!qc_inv && !(sigma_gp > 0.)))
        return fail(SGPMP_EINVAL, "sgpmp_set_prior: dt and sigmas must be positive");
    const int n = c->dims.n_dof;
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> qc((size_t)n * n, 0.);
    if (qc_inv) std::memcpy(qc.data(), qc_inv, sizeof(double) * n * n);
    else for (int i = 0; i < n; ++i) qc[(size_t)i * n + i] = 1. / (sigma_gp * sigma_gp);   // gp_factor.py:25-26
    HIPCHK(hipMemcpyAsync(c->d_qc, qc.data(), sizeof(double) * n * n, hipMemcpyHostToDevice, st));
    PriorDev& p = c->prior[which];
    c->isw_ready = false;                                        // prepared IS weights belong to the old prior
    p.ks = 1. / (sigma_start * sigma_start);                     // unary_factor.py:19
    p.kg = sigma_goal > 0. ? 1. / (sigma_goal * sigma_goal) : -1.;
    p.dt = dt;
    p.isotropic = qc_inv ? 0 : 1;
    p.valid = 0;
    p.n_factor_modes = 0;                                        // back to one shared closed-form factor
    HIPCHK(launch_prior_factor(n, c->dims.traj_len, dt, p.ks, p.kg, c->d_qc, p.isotropic, p, st));
    int status = 0;
    HIPCHK(hipMemcpyAsync(&status, p.status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (status != 0)
        return fail(SGPMP_ENOTPD, "sgpmp_set_prior: prior precision matrix is not positive definite");
    int rcs;
    if ((rcs = upload_slab_prefix(c, p, st)) != SGPMP_OK) return rcs;
    if ((rcs = upload_scan64(c, p, st)) != SGPMP_OK) return rcs;
    p.valid = 1;
    return SGPMP_OK;
}

// Both priors of StochGPMP.reset (planner.py:204-226) at once: K1 is one wave for ~1.2 ms whatever the problem
// size, so the two factorisations run side by side -- the sampling prior's on `stream`, the initialisation prior's
// on a stream of the context -- and the call synchronises once.  Isotropic Q_c only (what the planner builds).
extern "C" int sgpmp_set_priors(sgpmp_ctx* c, double dt, const double* sigma_start, const double* sigma_gp,
                                const double* sigma_goal, void* stream) {
    if (!c || !sigma_start || !sigma_gp || !sigma_goal) return fail(SGPMP_EINVAL, "sgpmp_set_priors: null argument");
    for (int w = 0; w < 2; ++w)
        if (!(dt > 0.) || !(sigma_start[w] > 0.) || !(sigma_gp[w] > 0.))
            return fail(SGPMP_EINVAL, "sgpmp_set_priors: dt and sigmas must be positive");
    const int n = c->dims.n_dof;
    hipStream_t st = (hipStream_t)stream;
    if (!c->k1_side) {
        HIPCHK(hipStreamCreateWithFlags(&c->k1_side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->k1_fork, hipEventDisableTiming));
    }
    c->isw_ready = false;
    std::vector<double> qc((size_t)2 * n * n, 0.);
    for (int w = 0; w < 2; ++w)
        for (int i = 0; i < n; ++i) qc[(size_t)w * n * n + (size_t)i * n + i] = 1. / (sigma_gp[w] * sigma_gp[w]);
    HIPCHK(hipMemcpyAsync(c->d_qc, qc.data(), sizeof(double) * 2 * n * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(c->k1_fork, st));                       // the Q_c copy, and whatever used the old factors,
    HIPCHK(hipStreamWaitEvent(c->k1_side, c->k1_fork, 0));        // precede the second factorisation too
    int status[2] = {0, 0};
    for (int w = 0; w < 2; ++w) {
        PriorDev& p = c->prior[w];
        hipStream_t sw = w == SGPMP_PRIOR_SAMPLE ? st : c->k1_side;
        p.ks = 1. / (sigma_start[w] * sigma_start[w]);
        p.kg = sigma_goal[w] > 0. ? 1. / (sigma_goal[w] * sigma_goal[w]) : -1.;
        p.dt = dt; p.isotropic = 1; p.valid = 0; p.n_factor_modes = 0;
        HIPCHK(launch_prior_factor(n, c->dims.traj_len, dt, p.ks, p.kg, c->d_qc + (size_t)w * n * n, 1, p, sw));
    }
    // (both launches first: a device-to-host copy into pageable memory blocks the host until its stream is idle)
    HIPCHK(hipMemcpyAsync(&status[SGPMP_PRIOR_INIT], c->prior[SGPMP_PRIOR_INIT].status, sizeof(int), hipMemcpyDeviceToHost, c->k1_side));
    HIPCHK(hipMemcpyAsync(&status[SGPMP_PRIOR_SAMPLE], c->prior[SGPMP_PRIOR_SAMPLE].status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(c->k1_side));
    HIPCHK(hipStreamSynchronize(st));
    for (int w = 0; w < 2; ++w) {
        if (status[w] != 0) return fail(SGPMP_ENOTPD, "sgpmp_set_priors: prior precision matrix is not positive definite");
        int rcs;
        if ((rcs = upload_slab_prefix(c, c->prior[w], st)) != SGPMP_OK) return rcs;
        if ((rcs = upload_scan64(c, c->prior[w], st)) != SGPMP_OK) return rcs;
        c->prior[w].valid = 1;
    }
    return SGPMP_OK;
}

// MultiMPPrior.set_Sigma_invs (mp_priors_multi.py:125-128): per-mode block-tridiagonal precisions.
extern "C" int sgpmp_set_prior_blocks(sgpmp_ctx* c, int which, int n_modes, const double* D, const double* E,
                                      void* stream) {
    if (!c || (which != 0 && which != 1) || n_modes < 1 || !D || (c->dims.traj_len > 1 && !E))
        return fail(SGPMP_EINVAL, "sgpmp_set_prior_blocks: bad argument");
    const int n = c->dims.n_dof, T = c->dims.traj_len;
    const size_t dd = (size_t)c->d * c->d, nD = (size_t)n_modes * T * dd, nE = (size_t)n_modes * (T - 1) * dd;
    hipStream_t st = (hipStream_t)stream;
    PriorDev& p = c->prior[which];
    p.valid = 0;
    c->isw_ready = false;
    HIPCHK(hipStreamSynchronize(st));
    hipFree(p.G); hipFree(p.H); hipFree(p.G32); hipFree(p.H32); hipFree(p.Dm); hipFree(p.Em);
    p.G = p.H = p.Dm = p.Em = nullptr; p.G32 = p.H32 = nullptr;
    HIPCHK(hipMalloc(&p.G, sizeof(double) * nD));
    HIPCHK(hipMalloc(&p.H, sizeof(double) * nD));
    HIPCHK(hipMalloc(&p.G32, sizeof(float) * nD));
    HIPCHK(hipMalloc(&p.H32, sizeof(float) * nD));
    HIPCHK(hipMalloc(&p.Dm, sizeof(double) * nD));
    HIPCHK(hipMalloc(&p.Em, sizeof(double) * (nE ? nE : 1)));
    HIPCHK(hipMemcpyAsync(p.Dm, D, sizeof(double) * nD, hipMemcpyHostToDevice, st));
    if (nE) HIPCHK(hipMemcpyAsync(p.Em, E, sizeof(double) * nE, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(p.status, 0, sizeof(int), st));
    p.isotropic = 0;
    p.n_factor_modes = n_modes;
    HIPCHK(launch_prior_factor_blocks(n, T, n_modes, p.Dm, p.Em, p, st));
    int status = 0;
    HIPCHK(hipMemcpyAsync(&status, p.status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (status != 0)
        return fail(SGPMP_ENOTPD, "sgpmp_set_prior_blocks: a precision matrix is not positive definite");
    p.valid = 1;
    return SGPMP_OK;
}

// Quadratic forms (x_r - mu_m)^T Sigma_m^-1 (x_r - mu_m), m = r % n_modes, for MultiMPPrior.log_prob.
extern "C" int sgpmp_prior_quadform(sgpmp_ctx* c, int which, const void* x, int64_t rows, const void* means,
                                    int n_modes, double* out, void* stream) {
    if (!c || (which != 0 && which != 1) || rows < 0 || n_modes < 1 || (rows > 0 && (!x || !means || !out)))
        return fail(SGPMP_EINVAL, "sgpmp_prior_quadform: bad argument");
    const PriorDev& p = c->prior[which];
    if (!p.valid) return fail(SGPMP_ESTATE, "sgpmp_prior_quadform: prior not set");
    if (p.n_factor_modes > 0 && n_modes > p.n_factor_modes)
        return fail(SGPMP_EINVAL, "sgpmp_prior_quadform: more modes than precision matrices");
    HIPCHK(launch_prior_quadform(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, rows, n_modes, x, means, p, out,
                                 (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_get_prior(sgpmp_ctx* c, int which, double* blocks, double* G, double* H) {
    if (!c || (which != 0 && which != 1)) return fail(SGPMP_EINVAL, "sgpmp_get_prior: bad argument");
    const PriorDev& p = c->prior[which];
    if (!p.valid) return fail(SGPMP_ESTATE, "sgpmp_get_prior: prior not set");
    const size_t dd = (size_t)c->d * c->d, T = c->dims.traj_len;
    const size_t modes = p.n_factor_modes > 0 ? p.n_factor_modes : 1;     // per-mode factors: [modes][T][d][d]
    HIPCHK(hipDeviceSynchronize());
    // (after sgpmp_set_prior_blocks these are still the blocks of the shared closed-form prior of the last sgpmp_set_prior:
    // what the step's importance-sampling term keeps using)
    if (blocks) HIPCHK(hipMemcpy(blocks, p.blocks, sizeof(double) * 4 * dd, hipMemcpyDeviceToHost));
    if (G) HIPCHK(hipMemcpy(G, p.G, sizeof(double) * modes * T * dd, hipMemcpyDeviceToHost));
    if (H) HIPCHK(hipMemcpy(H, p.H, sizeof(double) * modes * T * dd, hipMemcpyDeviceToHost));
    return SGPMP_OK;
}

static int upload_as_dtype(sgpmp_ctx* c, const double* host, size_t count, const void** dev_out) {
    void* dev = nullptr;
    HIPCHK(hipMalloc(&dev, count * c->esz));
    if (c->dims.dtype == SGPMP_F64) {
        HIPCHK(hipMemcpy(dev, host, count * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp(count);
        for (size_t i = 0; i < count; ++i) tmp[i] = (float)host[i];
        HIPCHK(hipMemcpy(dev, tmp.data(), count * 4, hipMemcpyHostToDevice));
    }
    c->owned.push_back(dev);
    *dev_out = dev;
    return SGPMP_OK;
}

extern "C" int sgpmp_set_costs(sgpmp_ctx* c, const sgpmp_cost_desc* descs, int n_desc) {
    if (!c || (n_desc > 0 && !descs)) return fail(SGPMP_EINVAL, "sgpmp_set_costs: null argument");
    if (n_desc < 0 || n_desc > SGPMP_MAX_TERMS)
        return fail(SGPMP_EINVAL, "sgpmp_set_costs: at most 8 cost terms are supported");
    CostProgram prog;
    std::memset(&prog, 0, sizeof(prog));
    prog.n_terms = n_desc;
    const int d = c->d;
    for (int i = 0; i < n_desc; ++i) {
        const sgpmp_cost_desc& s = descs[i];
        CostTerm& t = prog.terms[i];
        t.kind = s.kind;
        t.flags = s.flags;
        if (!(s.sigma > 0.)) return fail(SGPMP_EINVAL, "sgpmp_set_costs: sigma must be positive");
        t.K = 1. / (s.sigma * s.sigma);                           // field_factor.py:16, unary_factor.py:19
        int rc;
        switch (s.kind) {
            case SGPMP_COST_GP:
                if (!(s.dt > 0.)) return fail(SGPMP_EINVAL, "sgpmp_set_costs: GP term needs dt > 0");
                t.dt = s.dt;
                t.c11 = 12. * std::pow(s.dt, -3.);                // gp_factor.py:45-47
                t.c12 = -6. * std::pow(s.dt, -2.);
                t.c22 = 4. * std::pow(s.dt, -1.);
                if (s.flags & SGPMP_FLAG_GP_START) {
                    if (!s.data || !(s.sigma2 > 0.))
                        return fail(SGPMP_EINVAL, "sgpmp_set_costs: CostGP needs start state and sigma_start");
                    t.K2 = 1. / (s.sigma2 * s.sigma2);
                    if ((rc = upload_as_dtype(c, (const double*)s.data, d, &t.dev_data)) != SGPMP_OK) return rc;
                }
                break;
            case SGPMP_COST_GOAL_PRIOR:
                if (!s.data || s.dim0 < 1 || s.dim1 < 1)
                    return fail(SGPMP_EINVAL, "sgpmp_set_costs: goal prior needs goals [G,d] and nppg*S");
                t.dim0 = s.dim0;
                t.rows_per_goal = s.dim1;
                if ((rc = upload_as_dtype(c, (const double*)s.data, (size_t)s.dim0 * d, &t.dev_data)) != SGPMP_OK)
                    return rc;
                break;
            case SGPMP_COST_GRID:
                if (!s.data || s.dim0 < 1 || s.dim1 < 1 || !(s.p0 > 0.))
                    return fail(SGPMP_EINVAL, "sgpmp_set_costs: grid term needs a device grid and cell size");
                if (c->dims.n_dof < 2)
                    return fail(SGPMP_EINVAL, "sgpmp_set_costs: grid term needs n_dof >= 2");
                t.dev_data = s.data;
                t.dim0 = s.dim0; t.dim1 = s.dim1;
                t.inv_cell = 1. / s.p0;                           // obst_map.py:172
                t.off_x = s.p1; t.off_y = s.p2;
                break;
            case SGPMP_COST_SPHERES:
            case SGPMP_COST_SELF:
                if (s.num_interpolate < 0 || s.num_interpolate > SGPMP_MAX_INTERP)
                    return fail(SGPMP_EINVAL, "sgpmp_set_costs: num_interpolate must be in [0, 8]");
                t.n_interp = s.num_interpolate;
                t.interp_lo = s.interp_lo; t.interp_hi = s.interp_hi;
                for (int a = 0; a < s.num_interpolate; ++a) t.alpha[a] = s.alpha[a];
                if (s.kind == SGPMP_COST_SELF) {
                    if (!(s.sigma2 > 0.)) return fail(SGPMP_EINVAL, "sgpmp_set_costs: self field needs margin > 0");
                    t.K2 = 1. / (-(s.sigma2 * s.sigma2) * 2.);    // fields.py:124
                }
                prog.needs_fk = 1;
                break;
            case SGPMP_COST_EE_GOAL:
                if (!s.data) return fail(SGPMP_EINVAL, "sgpmp_set_costs: EE goal needs a target frame");
                std::memcpy(t.target, s.data, sizeof(double) * 16);
                t.w_pos = s.p0; t.w_rot = s.p1;
                prog.n_ee += 1;
                break;
            default:
                return fail(SGPMP_EINVAL, "sgpmp_set_costs: unknown cost kind");
        }
    }
    c->h_prog = prog;
    c->have_costs = true;
    c->prog_dirty = true;
    return SGPMP_OK;
}

static void rpy_to_R(const double rpy[3], double R[9]) {
    const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]), cp = std::cos(rpy[1]), sp = std::sin(rpy[1]),
                 cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

// Link positions of the chain for joint vector q, in double, on the host (setup-time analysis only).
static void host_fk_points(const ChainDev& ch, const double* q, double (*pos)[3]) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    pos[0][0] = pos[0][1] = pos[0][2] = 0.;
    for (int j = 0; j < ch.n_joints; ++j) {
        const JointDev& J = ch.j[j];
        double Rn[9];
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3] * J.t[0] + R[r * 3 + 1] * J.t[1] + R[r * 3 + 2] * J.t[2];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc)
                Rn[r * 3 + cc] = R[r * 3] * J.R[cc] + R[r * 3 + 1] * J.R[3 + cc] + R[r * 3 + 2] * J.R[6 + cc];
        if (J.revolute) {
            const double s = std::sin(q[J.qidx]), co = std::cos(q[J.qidx]);
            for (int r = 0; r < 3; ++r) {
                const double a = Rn[r * 3], b = Rn[r * 3 + 1];
                Rn[r * 3] = a * co + b * s;
                Rn[r * 3 + 1] = b * co - a * s;
            }
        }
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        for (int i = 0; i < 3; ++i) pos[j + 1][i] = p[i];
    }
}

// Chain analysis for the register-resident FK path of the cost sweep (see FkPlan).  The link
// geometry is probed at pseudo-random joint vectors: links whose frames coincide for every probe
// are merged (multiplicity), and pairs whose distance never changes are rigid and become constants.
// Both are identities of the sums in reference fields.py:79,86,124, not approximations.
static void analyse_chain(ChainDev& ch) {
    FkPlan& pl = ch.plan;
    std::memset(&pl, 0, sizeof(pl));
    const int L = ch.n_links, ML = SGPMP_MAX_LINKS, K = 24;
    int nrev = 0;
    bool rev_first = true;
    for (int j = 0; j < ch.n_joints; ++j) {
        if (ch.j[j].revolute) { if (ch.j[j].qidx != j) rev_first = false; ++nrev; }
    }
    pl.fast = rev_first ? 1 : 0;
    std::vector<double> d2min((size_t)L * L, 1e300), d2max((size_t)L * L, 0.), d2sum((size_t)L * L, 0.);
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < K; ++k) {
        double q[SGPMP_MAX_JOINTS], pos[SGPMP_MAX_LINKS][3];
        for (int i = 0; i < nrev; ++i) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            q[i] = ((double)(lcg >> 11) / 9007199254740992.0 * 2. - 1.) * 3.0;
        }
        host_fk_points(ch, q, pos);
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < i; ++j) {
                double d2 = 0.;
                for (int a = 0; a < 3; ++a) d2 += (pos[i][a] - pos[j][a]) * (pos[i][a] - pos[j][a]);
                const size_t e = (size_t)i * L + j;
                d2min[e] = std::min(d2min[e], d2); d2max[e] = std::max(d2max[e], d2); d2sum[e] += d2;
            }
    }
    // representatives: the lowest-index link of each coincident cluster
    int rep[SGPMP_MAX_LINKS];
    for (int i = 0; i < L; ++i) {
        rep[i] = i;
        for (int j = 0; j < i; ++j)
            if (d2max[(size_t)i * L + j] < 1e-20) { rep[i] = rep[j]; break; }
    }
    for (int i = 0; i < L; ++i) pl.mult[rep[i]] += 1.f;
    for (int i = 0; i < L; ++i) {
        if (rep[i] == i) ++pl.n_rep;
        pl.diag += (rep[i] == i) ? (double)pl.mult[i] * pl.mult[i] : 0.;
    }
    for (int i = 0; i < L; ++i) {
        if (rep[i] != i) continue;
        for (int j = 0; j < i; ++j) {
            if (rep[j] != j) continue;
            const size_t e = (size_t)i * L + j;
            const double w = 2. * pl.mult[i] * pl.mult[j];
            if (d2max[e] - d2min[e] <= 1e-12 * std::max(1., d2max[e])) {      // rigid pair
                pl.cpair_w[pl.n_cpairs] = w;
                pl.cpair_d2[pl.n_cpairs] = d2sum[e] / K;
                ++pl.n_cpairs;
            } else {
                pl.wpair[i * ML + j] = (float)w;
            }
        }
    }
}

extern "C" int sgpmp_set_fk(sgpmp_ctx* c, const sgpmp_joint* chain, int n_joints) {
    if (!c || !chain) return fail(SGPMP_EINVAL, "sgpmp_set_fk: null argument");
    if (n_joints < 1 || n_joints > SGPMP_MAX_JOINTS)
        return fail(SGPMP_EINVAL, "sgpmp_set_fk: 1..16 joints supported");
    ChainDev ch;
    std::memset(&ch, 0, sizeof(ch));
    ch.n_joints = n_joints;
    ch.n_links = n_joints + 1;
    int q = 0;
    for (int j = 0; j < n_joints; ++j) {
        rpy_to_R(chain[j].rpy, ch.j[j].R);
        for (int i = 0; i < 3; ++i) ch.j[j].t[i] = chain[j].xyz[i];
        ch.j[j].revolute = chain[j].revolute ? 1 : 0;
        ch.j[j].qidx = chain[j].revolute ? q++ : -1;
    }
    if (q != c->dims.n_dof)
        return fail(SGPMP_EINVAL, "sgpmp_set_fk: number of revolute joints must equal n_dof");
    for (int j = 0; j < n_joints; ++j) {
        for (int i = 0; i < 9; ++i) ch.Rf[j][i] = (float)ch.j[j].R[i];
        for (int i = 0; i < 3; ++i) ch.tf[j][i] = (float)ch.j[j].t[i];
    }
    analyse_chain(ch);
    // does build-time generated code exist for exactly this chain? (chain_code_generated.h)
    if (n_joints == ChainCode_panda::NJ) {
        bool same = true;
        for (int j = 0; j < n_joints && same; ++j) {
            const double* g = ChainCode_panda::joints[j];
            for (int i = 0; i < 3; ++i) same = same && chain[j].rpy[i] == g[i] && chain[j].xyz[i] == g[3 + i];
            same = same && ((chain[j].revolute != 0) == (g[6] != 0.));
        }
        if (same) ch.plan.codegen_id = 1;
    }
    c->h_chain = ch;
    HIPCHK(hipMemcpy(c->d_chain, &ch, sizeof(ch), hipMemcpyHostToDevice));
    c->have_chain = true;
    c->prog_dirty = true;
    return SGPMP_OK;
}

// Straight-line code for the chain of the last sgpmp_set_fk, compiled at run time (chain_rtc.hip): `struct_src` is the text of
// `struct ChainCode_rt { ... };` as csrc/gen/chain_codegen.py emits it for that chain (gen_chain("rt", chain)).  The library
// compiles its fused sampler + sweep launch and its chunked sweep around it with hiprtc (lazily, per sphere-field type; code
// objects are cached in memory and on disk), CHECKS the code against the chain -- link positions at pseudo-random joint
// vectors against the host's fp64 forward kinematics, link / pair tables against the host analysis -- and from then on
// dispatches this chain like the built-in one.  On any failure (no libhiprtc, kernel sources not found, code that does not
// match the chain) the chain keeps the FkPlan register / generic sweep and the call says why.
extern "C" int sgpmp_set_fk_codegen(sgpmp_ctx* c, const char* struct_src) {
    if (!c || !struct_src) return fail(SGPMP_EINVAL, "sgpmp_set_fk_codegen: null argument");
    if (!c->have_chain) return fail(SGPMP_ESTATE, "sgpmp_set_fk_codegen: no chain (sgpmp_set_fk first)");
    if (c->h_chain.plan.codegen_id == 1) return SGPMP_OK;        // (the built-in chain needs none)
    if (c->dims.dtype != SGPMP_F32 || c->dims.n_dof > 7 || !c->h_chain.plan.fast)
        return fail(SGPMP_EINVAL, "sgpmp_set_fk_codegen: chain kernels are fp32, n_dof <= 7, revolute joints first");
    RtcChain* rc = nullptr;
    if (const char* e = rtc_chain_get(struct_src, c->dims.n_dof, &rc)) return fail(SGPMP_EINVAL, std::string("sgpmp_set_fk_codegen: ") + e);
    int ft = SGPMP_FIELD_RBF;
    for (int i = 0; i < c->h_prog.n_terms; ++i)
        if (c->h_prog.terms[i].kind == SGPMP_COST_SPHERES) ft = c->h_prog.terms[i].flags & 15;
    int mismatch = 0;
    if (const char* e = rtc_verify(rc, c->h_chain, ft, &mismatch))
        return fail(mismatch ? SGPMP_EINVAL : SGPMP_ESTATE, std::string("sgpmp_set_fk_codegen: ") + e);
    c->h_chain.plan.codegen_id = 2;
    c->h_chain.rtc = rc;
    HIPCHK(hipMemcpy(c->d_chain, &c->h_chain, sizeof(c->h_chain), hipMemcpyHostToDevice));
    return SGPMP_OK;
}

// Compile-only check (needs hiprtc, no device): does this chain code build into the chain kernels for sphere-field type
// `field_type`?  *code_bytes (may be NULL): size of the gfx950 code object.  Nothing is loaded or cached.
extern "C" int sgpmp_fk_codegen_compile(const char* struct_src, int field_type, int64_t* code_bytes) {
    if (!struct_src || field_type < 0 || field_type > 2) return fail(SGPMP_EINVAL, "sgpmp_fk_codegen_compile: bad argument");
    std::vector<char> err(4200);
    const long long r = rtc_compile_check_c(struct_src, field_type, err.data(), err.size());
    if (code_bytes) *code_bytes = r > 0 ? r : 0;
    if (r < 0) return fail(SGPMP_ESTATE, std::string("sgpmp_fk_codegen_compile: ") + err.data());
    return SGPMP_OK;
}

// codegen_id: 0 the chain runs on run-time constants (FkPlan register path / generic path), 1 the code built with the library
// (Panda), 2 code compiled at run time; compile_s / compiled / from_cache: seconds spent in hiprtc and code objects compiled /
// taken from the disk cache for this chain so far.  Any pointer may be NULL.
extern "C" int sgpmp_fk_codegen_info(sgpmp_ctx* c, int* codegen_id, double* compile_s, int* compiled, int* from_cache) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_fk_codegen_info: null context");
    if (codegen_id) *codegen_id = c->have_chain ? c->h_chain.plan.codegen_id : 0;
    rtc_stats(c->have_chain && c->h_chain.plan.codegen_id == 2 ? (const RtcChain*)c->h_chain.rtc : nullptr, compile_s, compiled, from_cache);
    return SGPMP_OK;
}

// Resolve point counts (they depend on the chain) and push the program to the device.
static int finalize_program(sgpmp_ctx* c) {
    if (!c->have_costs) return fail(SGPMP_ESTATE, "cost program not set (sgpmp_set_costs)");
    if (!c->prog_dirty) return SGPMP_OK;
    CostProgram& p = c->h_prog;
    if ((p.needs_fk || p.n_ee) && !c->have_chain)
        return fail(SGPMP_ESTATE, "link-distance / end-effector cost terms need an FK chain (sgpmp_set_fk)");
    for (int i = 0; i < p.n_terms; ++i) {
        CostTerm& t = p.terms[i];
        if (t.kind != SGPMP_COST_SPHERES && t.kind != SGPMP_COST_SELF) continue;
        const int L = c->h_chain.n_links;
        int extra = 0;
        if (t.n_interp > 0) {
            if (t.interp_lo < 0 || t.interp_hi > L - 1 || t.interp_lo > t.interp_hi)
                return fail(SGPMP_EINVAL, "link_interpolate_range outside the link table");
            extra = t.n_interp * (t.interp_hi - t.interp_lo);
        }
        t.n_points = L + extra;
        if (t.n_points > SGPMP_MAX_POINTS) return fail(SGPMP_EINVAL, "too many link points (max 32)");
        if (t.kind == SGPMP_COST_SELF) {          // q-independent part of the LxL sum (see FkPlan)
            const FkPlan& pl = c->h_chain.plan;
            t.selfc = pl.diag;
            for (int k = 0; k < pl.n_cpairs; ++k) t.selfc += pl.cpair_w[k] * std::exp(t.K2 * pl.cpair_d2[k]);
        }
    }
    HIPCHK(hipMemcpy(c->d_prog, &p, sizeof(p), hipMemcpyHostToDevice));
    c->prog_dirty = false;
    return SGPMP_OK;
}

extern "C" int sgpmp_sample(sgpmp_ctx* c, int which, uint64_t seed, uint64_t draw, const void* means,
                            int n_modes, int mode_offset, int n_samples, const void* eps, int eps_modes,
                            int eps_mode_offset, void* out, void* stream) {
    if (!c || !means || !out || (which != 0 && which != 1) || n_modes < 0 || n_samples < 1)
        return fail(SGPMP_EINVAL, "sgpmp_sample: bad argument");
    if (!c->prior[which].valid) return fail(SGPMP_ESTATE, "sgpmp_sample: prior not set");
    if (eps && (eps_modes < 1 || eps_mode_offset < 0 || eps_mode_offset + n_modes > eps_modes))
        return fail(SGPMP_EINVAL, "sgpmp_sample: eps mode window out of range");
    if (n_modes == 0) return SGPMP_OK;
    HIPCHK(launch_sample(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, c->prior[which], seed, draw, means,
                         n_modes, mode_offset, n_samples, eps, eps_modes, eps_mode_offset, out,
                         (hipStream_t)stream, c->tg));
    return SGPMP_OK;
}

static int check_spheres(sgpmp_ctx* c, const void* spheres, int n_spheres) {
    for (int i = 0; i < c->h_prog.n_terms; ++i)
        if (c->h_prog.terms[i].kind == SGPMP_COST_SPHERES && (!spheres || n_spheres < 1))
            return fail(SGPMP_EINVAL, "LinkDistanceField cost needs obstacle_spheres");
    return SGPMP_OK;
}

extern "C" int sgpmp_noise(sgpmp_ctx* c, uint64_t seed, uint64_t draw, int n_modes, int mode_offset, int n_samples, void* out,
                           void* stream) {
    if (!c || !out || n_modes < 0 || mode_offset < 0 || n_samples < 0) return fail(SGPMP_EINVAL, "sgpmp_noise: bad argument");
    HIPCHK(launch_noise(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, n_modes, mode_offset, n_samples, seed, draw, out,
                        (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_cost_eval(sgpmp_ctx* c, const void* trajs, int64_t batch, int64_t batch_offset,
                               const void* spheres, int n_spheres, const void* is_weights,
                               int rows_per_particle, void* costs, double* costs64, void* stream) {
    if (!c || batch < 0 || (!trajs && batch > 0)) return fail(SGPMP_EINVAL, "sgpmp_cost_eval: bad argument");
    int rc;
    if ((rc = finalize_program(c)) != SGPMP_OK) return rc;
    if ((rc = check_spheres(c, spheres, n_spheres)) != SGPMP_OK) return rc;
    if (batch == 0) return SGPMP_OK;
    HIPCHK(launch_cost(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, c->h_prog, c->d_chain,
                       c->h_chain, trajs, batch, batch_offset, spheres, n_spheres, is_weights,
                       rows_per_particle, c->prior[SGPMP_PRIOR_SAMPLE].dt, costs, costs64,
                       (hipStream_t)stream, c->tg, &c->last_cost_kernel));
    return SGPMP_OK;
}

extern "C" int sgpmp_is_weights(sgpmp_ctx* c, const void* means, int n_particles, double temperature,
                                void* out, void* stream) {
    if (!c || !means || !out || n_particles < 0) return fail(SGPMP_EINVAL, "sgpmp_is_weights: bad argument");
    if (!c->prior[SGPMP_PRIOR_SAMPLE].valid) return fail(SGPMP_ESTATE, "sgpmp_is_weights: sampling prior not set");
    HIPCHK(launch_is_weights(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, c->prior[SGPMP_PRIOR_SAMPLE], means,
                             n_particles, temperature, out, nullptr, (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_update(sgpmp_ctx* c, const void* costs, int costs_dtype, const void* samples, void* means,
                            double temperature, double step_size, void* weights, void* grad, void* means_prev,
                            double* stats, void* stream) {
    if (c) c->isw_ready = false;                                 // the means change behind the prepared IS weights
    if (!c || !costs || !samples || !means) return fail(SGPMP_EINVAL, "sgpmp_update: null argument");
    if (costs_dtype != SGPMP_F64 && costs_dtype != c->dims.dtype)
        return fail(SGPMP_EINVAL, "sgpmp_update: costs must be fp64 or the context dtype");
    if (!(temperature > 0.)) return fail(SGPMP_EINVAL, "sgpmp_update: temperature must be positive");
    HIPCHK(launch_update(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, c->dims.num_particles,
                         c->dims.num_samples, costs, costs_dtype, samples, means, temperature, step_size,
                         weights, grad, means_prev, stats, (hipStream_t)stream));
    return SGPMP_OK;
}

// The step's end-effector goal term goes INTO update_kernel (update_common.h: EeFold) when it is the only one and the kernel's
// scratch has room: one launch less per iteration (the term is a few hundred flops per trajectory; as a launch of its own it
// cost 6 us of the reference's Panda example's 24).  -> the term, or null: ee_goal_kernel in front of update_kernel as before.
static const CostTerm* ee_term_to_fold(const sgpmp_ctx* c) {
    if (c->tg.no_ee_fold || c->h_prog.n_ee != 1) return nullptr;
    if (!update_ee_fold_fits(c->dims.dtype, c->dims.n_dof, c->dims.traj_len, c->dims.num_samples)) return nullptr;
    for (int i = 0; i < c->h_prog.n_terms; ++i)
        if (c->h_prog.terms[i].kind == SGPMP_COST_EE_GOAL) return &c->h_prog.terms[i];
    return nullptr;
}

// Buffers the fused launch and update_kernel share across steps (allocated once, by the first fp32 step).
// Particles whose previous update spread its weight over more than S / 4 rows get partials (a fused wave pays ~6 % for them,
// the update reads ceil(S / 8) rows instead of nnz).  Everything here is a function of stream-ordered device state -- the
// per-particle counts -- so a run is reproducible, equal as one optimize(K) call or K calls, sharded or not.  (Round 4 armed
// the partials from a pinned host word read without synchronisation: the step at which a run switched from gathered rows to
// partials, equal to 1e-6 only, depended on host timing -- advisor finding, round 4.)
static int dense_buffers(sgpmp_ctx* c, FusedDenseHost* d, double temperature, bool chain_code_step) {
    const sgpmp_dims& D = c->dims;
    std::memset(d, 0, sizeof(*d));
    d->threshold = (unsigned)(D.num_samples / 4); d->temperature = temperature;
    // rows of a store-free step: kept for particles that wanted more rows than update_kernel regenerates in one round --
    // and for every particle that gets partials (they re-read the rows): never above the partials' threshold
    d->store_threshold = d->threshold < 4u ? d->threshold : 4u;
    d->particles_total = D.num_particles;
    d->particles_global = D.num_particles_global > 0 ? D.num_particles_global : D.num_particles;
    if (D.dtype != SGPMP_F32 || D.num_particles < 1) return SGPMP_OK;
    if (!c->d_nnz) {
        const size_t P = (size_t)D.num_particles;
        HIPCHK(hipMalloc(&c->d_nnz, P * sizeof(unsigned)));
        HIPCHK(hipMemset(c->d_nnz, 0, P * sizeof(unsigned)));
    }
    d->nnz = c->d_nnz;
    if (chain_code_step && !c->tg.no_dense_partials && c->M % 4 == 0) {
        if (!c->d_part)
            HIPCHK(hipMalloc(&c->d_part, (size_t)D.num_particles * (size_t)((D.num_samples + 7) / 8) * (size_t)(c->M + 4) * sizeof(float)));
        d->part = c->d_part;
    }
    return SGPMP_OK;
}

// Diagnostic (synchronous): how many particles' last update spread its weight over more than S / 4 samples -- the
// particles for which the NEXT fused launch leaves softmax partials (dense-weight regime).  -1 before the buffers exist.
extern "C" int sgpmp_dense_particles(sgpmp_ctx* c, int64_t* count, int64_t* armed_steps) {
    if (!c || !count) return fail(SGPMP_EINVAL, "sgpmp_dense_particles: null argument");
    *count = -1;
    if (armed_steps) *armed_steps = c->dense_armed_steps;
    if (!c->d_nnz) return SGPMP_OK;
    HIPCHK(hipDeviceSynchronize());
    std::vector<unsigned> nnz((size_t)c->dims.num_particles);
    HIPCHK(hipMemcpy(nnz.data(), c->d_nnz, nnz.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    int64_t k = 0;
    for (unsigned v : nnz) k += v > (unsigned)(c->dims.num_samples / 4) ? 1 : 0;
    *count = k;
    return SGPMP_OK;
}

// The per-particle row counts are part of a run's state (they decide, per particle, how the next step's update forms its
// sum: gathered / regenerated rows or partials -- equal to 1e-6, not bit for bit): a checkpoint carries them
// (StochGPMP.state_dict), reset() clears them.  Synchronous.  get: out [P] (zeros before the first fp32 step);
// set: in [P], or NULL = all zero.
extern "C" int sgpmp_row_counts_get(sgpmp_ctx* c, uint32_t* out) {
    if (!c || (!out && c->dims.num_particles > 0)) return fail(SGPMP_EINVAL, "sgpmp_row_counts_get: null argument");
    const size_t P = (size_t)c->dims.num_particles;
    if (P == 0) return SGPMP_OK;
    if (!c->d_nnz) { std::memset(out, 0, P * sizeof(uint32_t)); return SGPMP_OK; }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, c->d_nnz, P * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SGPMP_OK;
}
extern "C" int sgpmp_row_counts_set(sgpmp_ctx* c, const uint32_t* in) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_row_counts_set: null context");
    const size_t P = (size_t)c->dims.num_particles;
    if (P == 0 || c->dims.dtype != SGPMP_F32) return SGPMP_OK;
    HIPCHK(hipDeviceSynchronize());
    if (!c->d_nnz) HIPCHK(hipMalloc(&c->d_nnz, P * sizeof(unsigned)));
    if (in) HIPCHK(hipMemcpy(c->d_nnz, in, P * sizeof(uint32_t), hipMemcpyHostToDevice));
    else HIPCHK(hipMemset(c->d_nnz, 0, P * sizeof(unsigned)));
    return SGPMP_OK;
}
// reset()'s clear: stream-ordered, no host synchronisation (a receding-horizon loop resets every control cycle -- advisor
// finding, round 5: the synchronous set stalled the host on all streams each time).  Before the first fp32 step there is
// nothing to clear (dense_buffers allocates the counts zeroed).
extern "C" int sgpmp_row_counts_clear(sgpmp_ctx* c, void* stream) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_row_counts_clear: null context");
    if (!c->d_nnz || c->dims.num_particles == 0) return SGPMP_OK;
    HIPCHK(hipMemsetAsync(c->d_nnz, 0, (size_t)c->dims.num_particles * sizeof(unsigned), (hipStream_t)stream));
    return SGPMP_OK;
}
extern "C" long long sgpmp_store_free_steps(sgpmp_ctx* c) { return c ? c->store_free_steps : 0; }
extern "C" long long sgpmp_multi_iteration_launches(sgpmp_ctx* c) { return c ? c->multi_iteration_launches : 0; }

// ---- two-chain steps (StepPipe) ----------------------------------------------------------------------------
static int pipe_first_half(const sgpmp_ctx* c) {
    const long long s = (c->tg.pipe_split >= 1 && c->tg.pipe_split <= 15) ? c->tg.pipe_split : 8;
    return (int)((long long)c->dims.num_particles * s / 16);
}

static int pipe_fork(sgpmp_ctx* c, hipStream_t st) {
    StepPipe& q = c->pipe;
    if (q.forked) return SGPMP_OK;
    HIPCHK(hipEventRecord(q.fork_ev, st));                       // everything the caller enqueued so far ...
    for (int h = 0; h < 2; ++h) HIPCHK(hipStreamWaitEvent(q.side[h], q.fork_ev, 0));   // ... precedes both chains
    q.forked = true;
    return SGPMP_OK;
}

static int pipe_join(sgpmp_ctx* c, hipStream_t st) {
    StepPipe& q = c->pipe;
    if (!q.forked) return SGPMP_OK;
    for (int h = 0; h < 2; ++h) {
        HIPCHK(hipEventRecord(q.join_ev[h], q.side[h]));
        HIPCHK(hipStreamWaitEvent(st, q.join_ev[h], 0));
    }
    if (q.last_stats) HIPCHK(launch_stats_add(q.last_stats, q.stats2, st));   // the step's statistics: both halves
    q.last_stats = nullptr;
    q.forked = false;
    return SGPMP_OK;
}

extern "C" int sgpmp_pipeline_begin(sgpmp_ctx* c, void* stream) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_pipeline_begin: null context");
    StepPipe& q = c->pipe;
    if (q.active) return fail(SGPMP_ESTATE, "sgpmp_pipeline_begin: already begun");
    (void)stream;
    if (!q.side[0]) {
        for (int h = 0; h < 2; ++h) {
            HIPCHK(hipStreamCreateWithFlags(&q.side[h], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&q.join_ev[h], hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(&q.fork_ev, hipEventDisableTiming));
        HIPCHK(hipMalloc(&q.stats2, sizeof(double) * SGPMP_STAT_SHARDS * 4));
    }
    q.active = true;
    return SGPMP_OK;
}

extern "C" int sgpmp_pipeline_end(sgpmp_ctx* c, void* stream) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_pipeline_end: null context");
    if (!c->pipe.active) return SGPMP_OK;
    c->pipe.active = false;
    return pipe_join(c, (hipStream_t)stream);
}

// One step as two half-range launch sequences on the chains' own streams (see StepPipe).  The caller has checked
// that both halves qualify for the fused launch.
static int step_split(sgpmp_ctx* c, uint64_t seed, uint64_t draw, char* means, char* samples, char* costs,
                      char* weights, char* grad, char* means_prev, const void* spheres, int n_spheres,
                      double temperature, double step_size, double* stats, int flags, hipStream_t st) {
    const sgpmp_dims& D = c->dims;
    const int P = D.num_particles, S = D.num_samples, P0 = pipe_first_half(c);
    const PriorDev& pr = c->prior[SGPMP_PRIOR_SAMPLE];
    const size_t w = c->esz, M = (size_t)c->M, W = (size_t)(D.traj_len + 1) * c->d;
    int rc;
    // (Tried: chain 0's first launch as two quarter-range launches with chain 1 starting behind the first, so that the
    // chains are out of phase from the first iteration on -- no gain; a call's fixed cost of ~0.1 ms is the fill
    // and drain of the two-stage schedule itself, about half a sampler + sweep launch.)
    if ((rc = pipe_fork(c, st)) != SGPMP_OK) return rc;
    const bool prepared = (flags & SGPMP_STEP_MEANS_KEPT) && c->isw_ready && c->isw_means == (const void*)means &&
                          c->isw_temperature == temperature;
    c->isw_ready = false;
    // multi-GPU: each chain accumulates into its block of a ring slot; the all-reduce (communicator's stream) waits
    // for both update kernels, adds the blocks and writes the sums over all ranks into the caller's `stats`
    const bool reduce = c->comm && stats;
    double* slots[2] = {stats, stats ? c->pipe.stats2 : nullptr};
    hipEvent_t k4_done[2] = {nullptr, nullptr};
    bool isw_next[2] = {false, false};
    if (reduce)
        COMMCHK(comm_step_begin2(c->comm, c->pipe.side[0], c->pipe.side[1], &slots[0], &slots[1], &k4_done[0], &k4_done[1]));
    FusedDenseHost dense;                                        // (once per step: both halves share the buffers)
    if ((rc = dense_buffers(c, &dense, temperature, c->h_prog.needs_fk != 0)) != SGPMP_OK) return rc;
    dense.nostore = (flags & SGPMP_STEP_NO_SAMPLES) ? 1 : 0;
    if (dense.part) c->dense_armed_steps += 1;
    for (int h = 0; h < 2; ++h) {
        const size_t off = h ? (size_t)P0 : 0;
        const int Ph = h ? P - P0 : P0;
        hipStream_t sh = c->pipe.side[h];
        double* slot = slots[h];
        char* mu = means + off * M * w;
        char* X = samples + off * S * M * w;
        char* isw = (char*)c->d_isw + off * W * w;
        char* cs = costs ? costs + off * S * w : nullptr;
        double* c64 = c->d_costs64 + off * S;
        if (!prepared) HIPCHK(launch_is_weights(D.dtype, D.n_dof, D.traj_len, pr, mu, Ph, temperature, isw, slot, sh));
        bool launched = false;
        char* wh = weights ? weights + off * S * w : nullptr;
        char* gh = grad ? grad + off * M * w : nullptr;
        char* mph = means_prev ? means_prev + off * M * w : nullptr;
        FusedDenseHost dh = dense;
        if (dh.part) dh.part += off * (size_t)((S + 7) / 8) * (size_t)(c->M + 4);
        if (dh.nnz) dh.nnz += off;
        dh.tail_done = c->d_done + h; dh.tail_acc = c->d_tail_acc + (size_t)h * SGPMP_STAT_SHARDS * 4; dh.stats_out = slot;
        dh.weights = wh; dh.grad = gh; dh.means_prev = mph; dh.step_size = step_size;
        bool armed = false, tail_ran = false;
        RegenHost rgh;
        HIPCHK(launch_fused_step(D.dtype, D.n_dof, D.traj_len, pr, c->h_prog, c->h_chain, seed, draw, mu, Ph,
                                 D.particle_offset + (int)off, S, X, spheres, n_spheres, isw, slot, cs, c64, sh, c->tg,
                                 &c->last_cost_kernel, &launched, &dh, &armed, &rgh, &tail_ran));
        if (!launched) return fail(SGPMP_ESTATE, "sgpmp_step: a half of a pipelined step did not qualify for the fused launch");
        c->last_step_launches = 1;
        if (tail_ran) {                                          // the launch updated its particles itself (fused_planar_seg.inc: seg_update)
            isw_next[h] = true;
            if (h == 0) c->store_free_steps += 1;
            if (k4_done[h]) HIPCHK(hipEventRecord(k4_done[h], sh));
            continue;
        }
        const CostTerm* eet = ee_term_to_fold(c);
        const EeFoldHost eeh = {eet, c->d_chain, cs};
        for (int i = 0; !eet && i < c->h_prog.n_terms; ++i)
            if (c->h_prog.terms[i].kind == SGPMP_COST_EE_GOAL) {
                HIPCHK(launch_ee_goal(D.dtype, D.n_dof, D.traj_len, c->h_prog.terms[i], c->d_chain, X,
                                      (long long)Ph * S, cs, c64, sh));
                c->last_step_launches += 1;
            }
        HIPCHK(launch_update(D.dtype, D.n_dof, D.traj_len, Ph, S, c64, SGPMP_F64, X, mu, temperature, step_size, wh, gh, mph,
                             slot, sh, c->tg.comm_packet_event ? k4_done[h] : nullptr, &pr, isw, &isw_next[h], nullptr,
                             armed ? dh.part : nullptr, dh.nnz, dh.threshold, &rgh, eet ? &eeh : nullptr));
        if (h == 0 && rgh.recipe != 0) c->store_free_steps += 1;
        if (!c->tg.comm_packet_event && k4_done[h]) HIPCHK(hipEventRecord(k4_done[h], sh));
        c->last_step_launches += 1;
    }
    c->isw_ready = isw_next[0] && isw_next[1]; c->isw_means = means; c->isw_temperature = temperature;
    if (reduce) COMMCHK(comm_step_end(c->comm, stats, true));
    c->pipe.last_stats = reduce ? nullptr : stats;
    c->pipe.split_steps += 1;
    return SGPMP_OK;
}

extern "C" long long sgpmp_pipeline_split_steps(sgpmp_ctx* c) { return c ? c->pipe.split_steps : 0; }
extern "C" int sgpmp_last_step_launches(sgpmp_ctx* c) { return c ? c->last_step_launches : 0; }

extern "C" int sgpmp_step(sgpmp_ctx* c, uint64_t seed, uint64_t draw, const void* eps, int eps_modes,
                          int eps_mode_offset, void* means, void* samples, void* costs, void* weights,
                          void* grad, void* means_prev, const void* spheres, int n_spheres, double temperature,
                          double step_size, double* stats, int flags, void* stream) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_step: null context");
    HT_START();
    if (c->dims.num_particles == 0) {
        // A rank whose shard is empty (more ranks than particles) has no kernels to run, but the per-step
        // statistics all-reduce is a collective: it contributes a zeroed slot, or the other ranks' all-reduce
        // never completes.
        // The collectives of a step, in the order the ranks WITH particles issue them (below: the per-goal mean
        // statistics behind the update kernel, then the cost statistics) -- a communicator matches collectives by
        // their order of issue, not by their buffers.
        if (c->comm && c->ms_buf) {                              // (per-goal mean statistics: all zeros from here)
            hipStream_t side = comm_side_stream(c->comm);
            HIPCHK(hipEventRecord(c->ms_ready, (hipStream_t)stream));
            HIPCHK(hipStreamWaitEvent(side, c->ms_ready, 0));
            HIPCHK(hipMemsetAsync(c->ms_buf, 0, mode_stats_count(c) * sizeof(double), side));
            COMMCHK(comm_allreduce_f64(c->comm, c->ms_buf, mode_stats_count(c), side));
        }
        if (c->comm && stats) {
            hipStream_t st0 = (hipStream_t)stream;
            if (c->pipe.active) { int rcj = pipe_join(c, st0); if (rcj != SGPMP_OK) return rcj; }
            double* slot = nullptr;
            hipEvent_t done = nullptr;
            COMMCHK(comm_step_begin(c->comm, st0, &slot, &done));
            HIPCHK(hipMemsetAsync(slot, 0, sizeof(double) * SGPMP_STAT_SHARDS * 4, st0));
            HIPCHK(hipEventRecord(done, st0));
            COMMCHK(comm_step_end(c->comm, stats, false));
        }
        return SGPMP_OK;
    }
    if (!means || !samples) return fail(SGPMP_EINVAL, "sgpmp_step: null argument");
    if (!c->prior[SGPMP_PRIOR_SAMPLE].valid) return fail(SGPMP_ESTATE, "sgpmp_step: sampling prior not set");
    // Per-mode sampling precisions (MultiMPPrior.set_Sigma_invs on the planner's sampling distribution, mp_priors_multi.py:125-128
    // reached through planner._sample_dist): particle p is sampled from ITS factor (sample_dense_kernel: d x d blocks on the
    // matrix cores) while the importance-sampling term keeps the precision captured at reset -- the reference's
    // `self.Sigma_inv = self._sample_dist.Sigma_inv` (planner.py:226) is not updated by set_Sigma_invs (planner.py:233-236) --
    // i.e. the shared closed-form blocks (Qinv, ks, kg) sgpmp_set_prior left in the same PriorDev.  Such steps run the sampler,
    // the sweep and the update as separate launches (the fused launches need the shared isotropic factor).
    if (c->prior[SGPMP_PRIOR_SAMPLE].n_factor_modes > 0 && c->prior[SGPMP_PRIOR_SAMPLE].n_factor_modes < c->dims.num_particles)
        return fail(SGPMP_ESTATE, "sgpmp_step: per-mode precisions (sgpmp_set_prior_blocks) must cover every particle of the context");
    if (!(temperature > 0.)) return fail(SGPMP_EINVAL, "sgpmp_step: temperature must be positive");
    int rc;
    if ((rc = finalize_program(c)) != SGPMP_OK) return rc;
    if ((rc = check_spheres(c, spheres, n_spheres)) != SGPMP_OK) return rc;
    const sgpmp_dims& D = c->dims;
    const int P = D.num_particles, S = D.num_samples;
    if (eps && (eps_modes < 1 || eps_mode_offset < 0 || eps_mode_offset + P > eps_modes))
        return fail(SGPMP_EINVAL, "sgpmp_step: eps particle window out of range");
    hipStream_t st = (hipStream_t)stream;
    const PriorDev& pr = c->prior[SGPMP_PRIOR_SAMPLE];
    if (c->pipe.active) {
        // both halves big enough to fill the chip on their own (256 workgroups of 4 items of 8 rows) and fused
        const int P0 = pipe_first_half(c);
        // (fused_planar_seg_kernel: a workgroup is 16 waves, one per particle and 64 samples -- a half must still
        // bring a workgroup for every CU, or two half-empty launches take turns: 40.3 k against 44.4 k it/s at config 2)
        const bool seg = planar_seg_step(D.dtype, D.n_dof, D.traj_len, pr, c->h_prog, c->h_chain, P, D.particle_offset, S,
                                         n_spheres, c->tg);
        // (fp32 steps only: an fp64 step's launch is 0.5 - 0.8 ms of vector arithmetic at two waves per SIMD with a per-workgroup
        // prologue -- two half-size launches side by side measured 6 % SLOWER than one, and the update is 2 % of the step)
        const bool split = !eps && !c->profiling && !c->tg.no_step_pipeline && !c->ms_buf && D.dtype == SGPMP_F32 &&
                           (long long)(P0 < P - P0 ? P0 : P - P0) * S >= (seg ? 256 * 64 : 256 * 4 * 8) &&
                           fused_step_eligible(D.dtype, D.n_dof, D.traj_len, pr, c->h_prog, c->h_chain, P0, D.particle_offset,
                                               S, n_spheres, c->tg) &&
                           fused_step_eligible(D.dtype, D.n_dof, D.traj_len, pr, c->h_prog, c->h_chain, P - P0,
                                               D.particle_offset + P0, S, n_spheres, c->tg);
        if (split)
            return step_split(c, seed, draw, (char*)means, (char*)samples, (char*)costs, (char*)weights, (char*)grad,
                              (char*)means_prev, spheres, n_spheres, temperature, step_size, stats, flags, st);
        if ((rc = pipe_join(c, st)) != SGPMP_OK) return rc;      // an ordinary step: after the chains
    }
    HT(0);                                                       // checks + the pipeline's split decision
    StepEvents* se = nullptr;
    if (c->profiling) {
        c->events.emplace_back();
        se = &c->events.back();
        for (auto& e : se->ev) HIPCHK(hipEventCreate(&e));
        for (bool& h : se->has) h = true;
        HIPCHK(hipEventRecord(se->ev[0], st));
    }
    // multi-GPU: the step's statistics accumulate in a context-owned ring slot; their all-reduce (side
    // stream, enqueued at the end of the step) writes the sums over all ranks into the caller's `stats`
    double* acc_stats = stats;
    hipEvent_t k4_done = nullptr;
    if (c->comm && stats) COMMCHK(comm_step_begin(c->comm, st, &acc_stats, &k4_done));
    // One fused launch (importance-sampling weights + sampler + cost sweep) when the step qualifies
    // (in-kernel noise, fp32 Panda-type program); else K5, the sampler and the sweep one after the other
    bool fused = !eps && samples &&
                 fused_step_eligible(D.dtype, D.n_dof, D.traj_len, pr, c->h_prog, c->h_chain, P, D.particle_offset, S,
                                     n_spheres, c->tg);
    // K5 is skipped when the previous step's update kernel already prepared the weights for exactly these
    // means (the caller vouches with SGPMP_STEP_MEANS_KEPT that nothing else wrote them since); the fused
    // launch then zeroes the statistics itself
    HT(1);                                                       // eligibility
    const bool prepared = (flags & SGPMP_STEP_MEANS_KEPT) && c->isw_ready && c->isw_means == means &&
                          c->isw_temperature == temperature;
    c->isw_ready = false;
    if (!prepared)
        HIPCHK(launch_is_weights(D.dtype, D.n_dof, D.traj_len, pr, means, P, temperature, c->d_isw, acc_stats, st));
    if (se) { HIPCHK(hipEventRecord(se->ev[1], st)); se->has[0] = !prepared; }
    c->last_step_launches = prepared ? 0 : 1;
    FusedDenseHost dense;                                        // what the launch and the update share per particle (row counts, partials)
    std::memset(&dense, 0, sizeof(dense));
    bool partials = false, tail_ran = false;                     // tail_ran: the launch also updated its particles (fused_planar_seg.inc: seg_update)
    RegenHost regen;                                             // store-free step: how the update regenerates rows
    std::memset(&regen, 0, sizeof(regen));
    const CostTerm* eet = nullptr;                               // the end-effector goal term update_kernel evaluates itself (fused steps)
    if (fused) {
        if (se) { HIPCHK(hipEventRecord(se->ev[2], st)); se->has[1] = false; }   // (fused: the whole launch is booked on the sweep)
        if ((rc = dense_buffers(c, &dense, temperature, c->h_prog.needs_fk != 0)) != SGPMP_OK) return rc;
        // (per-goal mean statistics and the profiler read nothing of the samples either: they do not stand in the way)
        dense.nostore = (flags & SGPMP_STEP_NO_SAMPLES) ? 1 : 0;
        if (!c->ms_buf) {                                        // (the per-step mean statistics want update_kernel's snapshot of the new means)
            dense.tail_done = c->d_done; dense.tail_acc = c->d_tail_acc; dense.stats_out = acc_stats;
            dense.weights = weights; dense.grad = grad; dense.means_prev = means_prev; dense.step_size = step_size;
            dense.tail_iters = c->tail_iters_next;
        }
        HIPCHK(launch_fused_step(D.dtype, D.n_dof, D.traj_len, pr, c->h_prog, c->h_chain, seed, draw, means, P,
                                 D.particle_offset, S, samples, spheres, n_spheres, c->d_isw, acc_stats, costs,
                                 c->d_costs64, st, c->tg, &c->last_cost_kernel, &fused, &dense, &partials, &regen, &tail_ran));
        if (fused) { c->last_step_launches += 1; if (partials) c->dense_armed_steps += 1; if (regen.recipe != 0 || tail_ran) c->store_free_steps += 1; }
        if (dense.tail_iters > 1) {                              // (sgpmp_optimize checked that this step's launch carries its update)
            if (!tail_ran) return fail(SGPMP_ESTATE, "sgpmp_step: the launch of several iterations did not run (internal)");
            c->store_free_steps += dense.tail_iters - 1;
            c->multi_iteration_launches += 1;
        }
        eet = fused ? ee_term_to_fold(c) : nullptr;
        for (int i = 0; fused && !eet && i < c->h_prog.n_terms; ++i)
            if (c->h_prog.terms[i].kind == SGPMP_COST_EE_GOAL) {
                HIPCHK(launch_ee_goal(D.dtype, D.n_dof, D.traj_len, c->h_prog.terms[i], c->d_chain, samples,
                                      (long long)P * S, costs, c->d_costs64, st));
                c->last_step_launches += 1;
            }
    }
    if (!fused) {
        // (two-launch steps record their row counts too: sgpmp.h's contract is "the counts of the last in-step update", and a
        // later fused step or a checkpoint must not see those of an older one -- advisor finding, round 5)
        if ((rc = dense_buffers(c, &dense, temperature, false)) != SGPMP_OK) return rc;
        HIPCHK(launch_sample(D.dtype, D.n_dof, D.traj_len, pr, seed, draw, means, P, D.particle_offset, S, eps,
                             eps_modes, eps_mode_offset, samples, st, c->tg, prepared ? acc_stats : nullptr));
        if (se) HIPCHK(hipEventRecord(se->ev[2], st));
        HIPCHK(launch_cost(D.dtype, D.n_dof, D.traj_len, c->h_prog, c->d_chain, c->h_chain,
                           samples, (long long)P * S, (long long)D.particle_offset * S, spheres, n_spheres,
                           c->d_isw, S, pr.dt, costs, c->d_costs64, st, c->tg, &c->last_cost_kernel));
        c->last_step_launches += 2;
    }
    HT(2);                                                       // the fused launch (or sampler + sweep)
    if (se) HIPCHK(hipEventRecord(se->ev[3], st));
    // (the update also prepares the NEXT step's importance-sampling weights -- unless, as a kernel of its own, the new
    // means do not fit its LDS beside the weights: launch_update decides)
    bool isw_written = tail_ran;
    if (tail_ran) {
        if (k4_done) HIPCHK(hipEventRecord(k4_done, st));
    } else {
        // per-goal mean statistics (sgpmp_set_step_mode_stats): the update kernel also leaves a snapshot of the new
        // means for the side stream; a snapshot is reused two steps later -- by then its reduction has long finished
        // (host-side query; the stream wait is the never-taken fallback)
        const int slot = (int)(c->ms_step & 1);
        if (c->ms_buf && c->ms_used[slot] && hipEventQuery(c->ms_read[slot]) != hipSuccess)
            HIPCHK(hipStreamWaitEvent(st, c->ms_read[slot], 0));
        const EeFoldHost eeh = {eet, c->d_chain, costs};
        HIPCHK(launch_update(D.dtype, D.n_dof, D.traj_len, P, S, c->d_costs64, SGPMP_F64, samples, means,
                             temperature, step_size, weights, grad, means_prev, acc_stats, st,
                             c->tg.comm_packet_event ? k4_done : nullptr, &pr, c->d_isw,
                             &isw_written, c->ms_buf ? c->ms_snap[slot] : nullptr,
                             (fused && partials) ? dense.part : nullptr, dense.nnz, dense.threshold,
                             (fused && regen.recipe != 0) ? &regen : nullptr, eet ? &eeh : nullptr));
        if (!c->tg.comm_packet_event && k4_done) HIPCHK(hipEventRecord(k4_done, st));
        c->last_step_launches += 1;
        if (c->ms_buf) {
            if ((rc = step_mode_stats(c, slot, st)) != SGPMP_OK) return rc;
            c->ms_step += 1;
        }
    }
    HT(3);                                                       // the update launch
#ifdef SGPMP_HOST_TIMING
    g_htn += 1;
#endif
    c->isw_ready = isw_written; c->isw_means = means; c->isw_temperature = temperature;
    if (se) { HIPCHK(hipEventRecord(se->ev[4], st)); se->has[3] = !tail_ran; }
    // multi-GPU: sum the statistics over all ranks on the side stream (never gates the next step)
    if (c->comm && stats) COMMCHK(comm_step_end(c->comm, stats, false));
    return SGPMP_OK;
}

// The loop of StochGPMP.optimize itself (planner.py:289-299: `for opt_step in range(opt_iters)`), on this side of the C ABI:
// K x sgpmp_step with the bookkeeping the Python host did per iteration (round-5 verdict, item 4a) -- the draw counter,
// the alternating statistics slot, SGPMP_STEP_MEANS_KEPT from the second step on, SGPMP_STEP_NO_SAMPLES for all but the last,
// the last step's pre-update means into the tensor optimize() returns, the two-chain bracket around the call.
extern "C" int sgpmp_optimize(sgpmp_ctx* c, int opt_iters, uint64_t seed, uint64_t draw0, void* means, void* samples,
                              void* costs, void* weights, void* grad, void* means_prev_scratch, void* means_prev_last,
                              const void* spheres, int n_spheres, double temperature, double step_size,
                              double* stats_pair, int first_slot, int flags, void* stream) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_optimize: null context");
    if (opt_iters < 1) return fail(SGPMP_EINVAL, "sgpmp_optimize: opt_iters must be at least 1");
    const bool piped = (flags & SGPMP_OPT_PIPELINE) && opt_iters >= 2;
    int rc = SGPMP_OK;
    // A planar problem whose store-free steps carry their update inside the launch (S = 64: BASELINE configs[1]) runs ALL of
    // the call's store-free iterations in ONE launch (fused_planar_seg.inc: PERSIST) -- a particle's workgroup needs nothing from
    // outside itself between two iterations, so what the K - 2 launch boundaries cost (4 of an iteration's 20 us there) goes.
    // Same arithmetic on the same numbers: bit-identical.  (Not with a communicator, per-step statistics of the means or the
    // step profiler: each wants something per iteration from the host side.)
    int inner = 0;
    if ((flags & SGPMP_OPT_STORE_FREE) && opt_iters >= 3 && !c->comm && !c->ms_buf && !c->profiling &&
        means && samples && c->prior[SGPMP_PRIOR_SAMPLE].valid && finalize_program(c) == SGPMP_OK) {
        const sgpmp_dims& D = c->dims;
        const PriorDev& pr = c->prior[SGPMP_PRIOR_SAMPLE];
        // (such a step is never split over the two chains: a half would have to bring 256 workgroups, i.e. S = 64 x 512 particles
        // -- checked all the same, on the pipeline's own criterion)
        const int P = D.num_particles, P0 = pipe_first_half(c);
        const bool may_split = piped && !c->tg.no_step_pipeline && (long long)(P0 < P - P0 ? P0 : P - P0) * D.num_samples >= 256 * 64;
        if (!may_split && !(pr.n_factor_modes > 0) &&
            planar_persist_step(D.dtype, D.n_dof, D.traj_len, pr, c->h_prog, c->h_chain, P, D.particle_offset, D.num_samples, n_spheres, c->tg))
            inner = opt_iters - 1;
    }
    if (piped && (rc = sgpmp_pipeline_begin(c, stream)) != SGPMP_OK) return rc;
    // (a launch runs persist_max_iters iterations at most -- 2048: ~25 ms at BASELINE configs[1], far below anything a driver
    // would call a hang; a longer call takes several such launches, a single left-over iteration an ordinary store-free step)
    const long long cap = c->tg.persist_max_iters >= 2 ? c->tg.persist_max_iters : 2048;
    for (int k = 0; k < opt_iters; ++k) {
        const bool last = k == opt_iters - 1;
        if (inner - k >= 2) {
            // iterations k .. k + n - 1 (draws draw0 + k ...) in one launch; no statistics of theirs are formed
            const int n = (int)((long long)(inner - k) < cap ? (long long)(inner - k) : cap);
            c->tail_iters_next = n;
            rc = sgpmp_step(c, seed, draw0 + (uint64_t)k, nullptr, 0, 0, means, samples, costs, weights, grad, means_prev_scratch,
                            spheres, n_spheres, temperature, step_size, nullptr,
                            ((k > 0 || (flags & SGPMP_STEP_MEANS_KEPT)) ? SGPMP_STEP_MEANS_KEPT : 0) | SGPMP_STEP_NO_SAMPLES, stream);
            c->tail_iters_next = 0;
            if (rc != SGPMP_OK) break;
            k += n - 1;
            continue;
        }
        const int f = ((k > 0 || (flags & SGPMP_STEP_MEANS_KEPT)) ? SGPMP_STEP_MEANS_KEPT : 0) |
                      ((!last && (flags & SGPMP_OPT_STORE_FREE)) ? SGPMP_STEP_NO_SAMPLES : 0);
        double* st = stats_pair ? stats_pair + (size_t)((first_slot + k) & 1) * SGPMP_STAT_SHARDS * 4 : nullptr;
        rc = sgpmp_step(c, seed, draw0 + (uint64_t)k, nullptr, 0, 0, means, samples, costs, weights, grad,
                        last ? means_prev_last : means_prev_scratch, spheres, n_spheres, temperature, step_size, st, f, stream);
        if (rc != SGPMP_OK) break;
    }
    if (piped) {                                                 // (always closed: the chains must rejoin `stream` even after an error)
        const int rc2 = sgpmp_pipeline_end(c, stream);
        if (rc == SGPMP_OK) rc = rc2;
    }
    return rc;
}

extern "C" int sgpmp_fk(sgpmp_ctx* c, const void* q, int64_t batch, void* frames, void* stream) {
    if (!c || !q || !frames || batch < 0) return fail(SGPMP_EINVAL, "sgpmp_fk: bad argument");
    if (!c->have_chain) return fail(SGPMP_ESTATE, "sgpmp_fk: FK chain not set");
    HIPCHK(launch_fk(c->dims.dtype, c->dims.n_dof, c->d_chain, c->h_chain.n_links, q, batch, frames,
                     (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_grid_lookup(sgpmp_ctx* c, int term, const void* xy, int64_t batch, void* out, void* stream) {
    if (!c || !xy || !out || batch < 0) return fail(SGPMP_EINVAL, "sgpmp_grid_lookup: bad argument");
    if (!c->have_costs || term < 0 || term >= c->h_prog.n_terms || c->h_prog.terms[term].kind != SGPMP_COST_GRID)
        return fail(SGPMP_EINVAL, "sgpmp_grid_lookup: term is not a grid term");
    HIPCHK(launch_grid_lookup(c->dims.dtype, c->h_prog.terms[term], xy, batch, out, (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_field_eval(sgpmp_ctx* c, int term, const void* frames, int64_t batch, int n_links,
                                const void* spheres, int n_spheres, void* out, void* stream) {
    if (!c || !frames || !out || batch < 0 || n_links < 1)
        return fail(SGPMP_EINVAL, "sgpmp_field_eval: bad argument");
    if (!c->have_costs || term < 0 || term >= c->h_prog.n_terms)
        return fail(SGPMP_EINVAL, "sgpmp_field_eval: bad term index");
    CostTerm t = c->h_prog.terms[term];
    if (t.kind != SGPMP_COST_SPHERES && t.kind != SGPMP_COST_SELF && t.kind != SGPMP_COST_EE_GOAL)
        return fail(SGPMP_EINVAL, "sgpmp_field_eval: term is not a link field");
    if (t.kind == SGPMP_COST_SPHERES && (!spheres || n_spheres < 1))
        return fail(SGPMP_EINVAL, "LinkDistanceField cost needs obstacle_spheres");
    int extra = 0;
    if (t.n_interp > 0) {
        if (t.interp_lo < 0 || t.interp_hi > n_links - 1 || t.interp_lo > t.interp_hi)
            return fail(SGPMP_EINVAL, "link_interpolate_range outside the link table");
        extra = t.n_interp * (t.interp_hi - t.interp_lo);
    }
    t.n_points = n_links + extra;
    if (t.n_points > SGPMP_MAX_POINTS) return fail(SGPMP_EINVAL, "too many link points (max 32)");
    HIPCHK(launch_field_eval(c->dims.dtype, t, frames, batch, n_links, spheres, n_spheres, out,
                             (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_link_distances(sgpmp_ctx* c, const void* frames, int64_t batch, int n_links, const void* spheres,
                                    int n_spheres, int mode, double buffer, void* out, void* stream) {
    if (!c || batch < 0 || n_links < 1 || (batch > 0 && (!frames || !out)) || mode < 0 || mode > 2 ||
        (spheres && n_spheres < 1))
        return fail(SGPMP_EINVAL, "sgpmp_link_distances: bad argument");
    HIPCHK(launch_link_dist(c->dims.dtype, frames, batch, n_links, spheres, spheres ? n_spheres : n_links, mode, buffer,
                            out, (hipStream_t)stream));
    return SGPMP_OK;
}

extern "C" int sgpmp_field_grad(sgpmp_ctx* c, int term, const void* q, int64_t batch, const void* spheres,
                                int n_spheres, void* value, void* grad, void* stream) {
    if (!c || !q || !grad || batch < 0) return fail(SGPMP_EINVAL, "sgpmp_field_grad: bad argument");
    if (!c->have_costs || term < 0 || term >= c->h_prog.n_terms)
        return fail(SGPMP_EINVAL, "sgpmp_field_grad: bad term index");
    if (!c->have_chain) return fail(SGPMP_EINVAL, "sgpmp_field_grad: no FK chain (sgpmp_set_fk)");
    int rc;
    if ((rc = finalize_program(c)) != SGPMP_OK) return rc;
    CostTerm t = c->h_prog.terms[term];
    if (t.kind == SGPMP_COST_SPHERES) {
        if ((t.flags & 15) == SGPMP_FIELD_OCCUPANCY)
            return fail(SGPMP_EINVAL, "sgpmp_field_grad: the occupancy count has no gradient (rbf and sdf sphere fields do)");
        if (!spheres || n_spheres < 1) return fail(SGPMP_EINVAL, "LinkDistanceField cost needs obstacle_spheres");
    } else if (t.kind == SGPMP_COST_EE_GOAL) {
        HIPCHK(launch_ee_grad(c->dims.dtype, c->dims.n_dof, t, c->d_chain, q, batch, 0, 1, 0, value, grad,
                              (hipStream_t)stream));
        return SGPMP_OK;
    } else if (t.kind != SGPMP_COST_SELF) {
        return fail(SGPMP_EINVAL, "sgpmp_field_grad: term is not a smooth link field");
    }
    if (t.n_points > SGPMP_MAX_POINTS) return fail(SGPMP_EINVAL, "too many link points (max 32)");
    HIPCHK(launch_field_grad(c->dims.dtype, c->dims.n_dof, t, c->d_chain, c->h_chain.n_joints, q, batch, 0,
                             spheres, n_spheres, value, grad, (hipStream_t)stream));
    return SGPMP_OK;
}

// ---------------------------------------------------------------------------------- GPMP (Gauss-Newton)
// Fill the kernel arguments from the cost program: GP term (+ start factor), goal prior, smooth link
// fields.  Anything else in the cost list has no linear system here.
static int gpmp_args(sgpmp_ctx* c, GpmpArgs& a, int* field_terms) {
    const sgpmp_dims& D = c->dims;
    std::memset(&a, 0, sizeof(a));
    a.n = D.n_dof; a.T = D.traj_len; a.P = D.num_particles; a.p_offset = D.particle_offset;
    bool have_gp = false;
    for (int i = 0; i < c->h_prog.n_terms; ++i) {
        const CostTerm& t = c->h_prog.terms[i];
        switch (t.kind) {
            case SGPMP_COST_GP:
                if (have_gp) return fail(SGPMP_EINVAL, "GPMP: more than one GP term");
                have_gp = true;
                a.dt = t.dt; a.Kgp = t.K; a.c11 = t.c11; a.c12 = t.c12; a.c22 = t.c22;
                if (t.flags & SGPMP_FLAG_GP_START) { a.Ks = t.K2; a.start = t.dev_data; }
                break;
            case SGPMP_COST_GOAL_PRIOR:
                if (a.Kg > 0.) return fail(SGPMP_EINVAL, "GPMP: more than one goal prior");
                a.Kg = t.K; a.goals = t.dev_data; a.rows_per_goal = D.num_particles_per_goal > 0 ? D.num_particles_per_goal : 1;
                break;
            case SGPMP_COST_SPHERES:
                if ((t.flags & 15) == SGPMP_FIELD_OCCUPANCY)
                    return fail(SGPMP_EINVAL, "GPMP: the occupancy sphere field has no Jacobian (rbf and sdf do)");
                /* fall through */
            case SGPMP_COST_EE_GOAL:                      // (a field row on the last waypoint only)
            case SGPMP_COST_SELF:
                if (a.n_fields == 4) return fail(SGPMP_EINVAL, "GPMP: more than 4 link-field terms");
                field_terms[a.n_fields] = i;
                a.f[a.n_fields].K = t.K;
                a.n_fields += 1;
                break;
            default:
                return fail(SGPMP_EINVAL, "GPMP: cost term without a linear system (occupancy grid)");
        }
    }
    if (!have_gp) return fail(SGPMP_EINVAL, "GPMP: the cost list needs a CostGP term");
    if (a.T < 2 || a.T > SGPMP_MAX_T_GPMP) return fail(SGPMP_EINVAL, "GPMP: traj_len must be in [2, 128]");
    return SGPMP_OK;
}

static int gpmp_alloc(sgpmp_ctx* c, const GpmpArgs& a) {
    const size_t PT = (size_t)a.P * (a.T - 1);
    for (int k = 0; k < a.n_fields; ++k) {
        if (!c->d_fval[k]) HIPCHK(hipMalloc(&c->d_fval[k], PT * c->esz));
        if (!c->d_fgrad[k]) HIPCHK(hipMalloc(&c->d_fgrad[k], PT * a.n * c->esz));
    }
    if (!c->d_gscratch) HIPCHK(hipMalloc(&c->d_gscratch, (size_t)a.P * a.T * 2 * 256 * sizeof(double)));
    if (!c->d_diag) HIPCHK(hipMalloc(&c->d_diag, (size_t)a.T * 2 * a.n * sizeof(double)));
    if (!c->d_gstatus) { HIPCHK(hipMalloc(&c->d_gstatus, sizeof(int))); HIPCHK(hipMemset(c->d_gstatus, 0, sizeof(int))); }
    return SGPMP_OK;
}

extern "C" int sgpmp_gpmp_linearize(sgpmp_ctx* c, const void* means, const void* spheres, int n_spheres,
                                    double* diag_sum, void* stream) {
    if (!c || !means) return fail(SGPMP_EINVAL, "sgpmp_gpmp_linearize: bad argument");
    int rc;
    if ((rc = finalize_program(c)) != SGPMP_OK) return rc;
    GpmpArgs a;
    int ft[4];
    if ((rc = gpmp_args(c, a, ft)) != SGPMP_OK) return rc;
    if (a.n_fields > 0 && !c->have_chain) return fail(SGPMP_ESTATE, "GPMP: link fields need an FK chain");
    if ((rc = gpmp_alloc(c, a)) != SGPMP_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const long long B = (long long)a.P * (a.T - 1);
    for (int k = 0; k < a.n_fields; ++k) {
        const CostTerm& t = c->h_prog.terms[ft[k]];
        if (t.kind == SGPMP_COST_SPHERES && (!spheres || n_spheres < 1))
            return fail(SGPMP_EINVAL, "LinkDistanceField cost needs obstacle_spheres");
        if (t.kind == SGPMP_COST_EE_GOAL) {
            // CostGoal (cost_functions.py:323-337): one row, on the last waypoint -- the other waypoints' entries
            // of this term's value / gradient arrays are zero rows of the linear system
            HIPCHK(hipMemsetAsync(c->d_fval[k], 0, (size_t)B * c->esz, st));
            HIPCHK(hipMemsetAsync(c->d_fgrad[k], 0, (size_t)B * a.n * c->esz, st));
            HIPCHK(launch_ee_grad(c->dims.dtype, a.n, t, c->d_chain, means, a.P, a.T, a.T - 1, a.T - 2, c->d_fval[k],
                                  c->d_fgrad[k], st));
        } else
        HIPCHK(launch_field_grad(c->dims.dtype, a.n, t, c->d_chain, c->h_chain.n_joints, means, B, a.T, spheres,
                                 n_spheres, c->d_fval[k], c->d_fgrad[k], st));
        a.f[k].val = c->d_fval[k];
        a.f[k].grad = c->d_fgrad[k];
    }
    if (diag_sum) HIPCHK(launch_gpmp_diag(c->dims.dtype, a, diag_sum, st));
    return SGPMP_OK;
}

extern "C" int sgpmp_gpmp_solve(sgpmp_ctx* c, void* means, const double* diag_sum, double delta,
                                double step_size, void* d_theta, void* costs, void* stream) {
    if (c) c->isw_ready = false;
    if (!c || !means || !(delta >= 0.)) return fail(SGPMP_EINVAL, "sgpmp_gpmp_solve: bad argument");
    int rc;
    if ((rc = finalize_program(c)) != SGPMP_OK) return rc;
    GpmpArgs a;
    int ft[4];
    if ((rc = gpmp_args(c, a, ft)) != SGPMP_OK) return rc;
    if (!c->d_gscratch) return fail(SGPMP_ESTATE, "sgpmp_gpmp_solve: call sgpmp_gpmp_linearize first");
    for (int k = 0; k < a.n_fields; ++k) { a.f[k].val = c->d_fval[k]; a.f[k].grad = c->d_fgrad[k]; }
    a.delta = delta; a.diag_sum = diag_sum; a.step_size = step_size; a.scratch = c->d_gscratch;
    a.status = c->d_gstatus;
    a.inv_particles = 1.0 / (double)(c->dims.num_particles_global > 0 ? c->dims.num_particles_global : a.P);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(launch_gpmp_solve(c->dims.dtype, a, means, d_theta, costs, st, c->tg.gpmp_cholesky != 0));
    int status = 0;                                   // synchronous, like sgpmp_set_prior: GPMP is not the hot path
    HIPCHK(hipMemcpyAsync(&status, c->d_gstatus, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (status) {
        HIPCHK(hipMemsetAsync(c->d_gstatus, 0, sizeof(int), st));
        return fail(SGPMP_ENOTPD, "GPMP: normal matrix not positive definite");
    }
    return SGPMP_OK;
}

// ---------------------------------------------------------------------------------- timing helpers
extern "C" int sgpmp_event_create(void** ev) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void*)e;
    return SGPMP_OK;
}
extern "C" int sgpmp_event_record(void* ev, void* stream) {
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return SGPMP_OK;
}
extern "C" int sgpmp_event_elapsed_ms(void* start, void* stop, float* ms) {
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SGPMP_OK;
}
extern "C" int sgpmp_event_destroy(void* ev) {
    HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return SGPMP_OK;
}
extern "C" int sgpmp_profile_enable(sgpmp_ctx* c, int on) {
    if (!c) return fail(SGPMP_EINVAL, "sgpmp_profile_enable: null ctx");
    c->profiling = on != 0;
    return SGPMP_OK;
}
extern "C" int sgpmp_profile_read(sgpmp_ctx* c, double* ms4, int64_t* launches) {
    if (!c || !ms4) return fail(SGPMP_EINVAL, "sgpmp_profile_read: null argument");
    for (int k = 0; k < 4; ++k) ms4[k] = 0.;
    for (auto& se : c->events) {
        HIPCHK(hipEventSynchronize(se.ev[4]));
        for (int k = 0; k < 4; ++k) {
            if (!se.has[k]) continue;                    // (an interval without a kernel only holds event latency)
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, se.ev[k], se.ev[k + 1]));
            ms4[k] += ms;
        }
    }
    if (launches) *launches = (int64_t)c->events.size();
    for (auto& se : c->events)
        for (auto& e : se.ev) hipEventDestroy(e);
    c->events.clear();
    return SGPMP_OK;
}
