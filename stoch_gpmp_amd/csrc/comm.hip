// Multi-GPU collectives of the StochGPMP path behind the C ABI: RCCL over xGMI, one process per GPU.
//
// Particles never interact (reference planner.py:263-275 reduces over the samples of ONE particle),
// so the data path has no collective.  What crosses GPUs:
//   * once per iteration, an all-reduce (sum) of the [SGPMP_STAT_SHARDS][4] fp64 statistics that K4
//     accumulates -- the reference's print_info statistic (planner.py:668-672) made global.  It runs
//     on a SIDE stream chained with events, so it never gates the next iteration's kernels and no
//     Python runs in the loop (sgpmp_step enqueues it itself when a communicator is attached);
//   * on request, an all-gather of the particle means.
//
// librccl is loaded with dlopen at sgpmp_comm_init (no link-time dependency: a single-GPU user never
// touches it, and inside a PyTorch process the already-loaded librccl.so.1 is the one that is used).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "sgpmp_internal.h"

struct RcclApi {
    void* handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
    ncclResult_t (*GetVersion)(int*);
};

static RcclApi g_rccl = {};
static std::string g_rccl_name;                              // what load_rccl bound (sgpmp_comm_library)
#ifndef SGPMP_TEST_HOOKS
#define SGPMP_TEST_HOOKS 0
#endif

// Name of the collective library this process bound ("" before the first communicator); "librccl.so.1" etc. in the
// product library -- a test-hooks build may report the stand-in it was told to load.
const char* comm_library_name() { return g_rccl_name.c_str(); }
int comm_test_hooks() { return SGPMP_TEST_HOOKS; }

static const char* load_rccl() {
    if (g_rccl.handle) return nullptr;
    void* h = nullptr;
    // SGPMP_RCCL_LIB: a library with RCCL's entry points to bind instead (tests/fake_rccl: the shared-memory test double
    // that lets several ranks share ONE GPU, which real RCCL refuses -- the only way the N > 1 protocol runs on a 1-GPU box)
    // -- compiled ONLY into tests/fake_rccl/libsgpmp_testhooks.so (-DSGPMP_TEST_HOOKS, csrc/Makefile `testhooks`): the
    // product library ignores the variable, so a leaked environment cannot route its collectives anywhere but RCCL.
#if SGPMP_TEST_HOOKS
    if (const char* over = getenv("SGPMP_RCCL_LIB")) {
        h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!h) return "SGPMP_RCCL_LIB is set but cannot be loaded";
        g_rccl_name = over;
    }
#endif
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (h) break;
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) g_rccl_name = name;
    }
    if (!h) return "librccl.so.1 not found (dlopen)";
#define SYM(field, name)                                            \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);          \
    if (!g_rccl.field) return "librccl: missing symbol " name;
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(CommCount, "ncclCommCount")
    SYM(CommUserRank, "ncclCommUserRank")
    SYM(GetVersion, "ncclGetVersion")
#undef SYM
    g_rccl.handle = h;
    return nullptr;
}

#define SGPMP_COMM_RING 8
struct SgpmpComm {
    ncclComm_t comm;
    int rank, world;
    hipStream_t side;                                        // the all-reduce runs here
    hipEvent_t produced;                                     // main stream: statistics complete
    std::vector<std::pair<double*, hipEvent_t>> reduced;     // per statistics buffer: all-reduce complete
    // sgpmp_step's own statistics: K4 accumulates into a context-owned ring slot, the all-reduce reads the
    // slot and writes the caller's buffer.  A slot comes round again SGPMP_COMM_RING steps later, so the
    // check that its all-reduce has finished is a host-side event query (practically always true) and the
    // main stream carries no wait; the "statistics complete" event rides on K4's own dispatch packet.
    double* ring;                                            // [SGPMP_COMM_RING][2][SGPMP_STAT_SHARDS][4]
    // (second block of a slot, second event: the other particle half of a two-chain step, api.hip StepPipe)
    hipEvent_t ring_produced[SGPMP_COMM_RING], ring_produced2[SGPMP_COMM_RING], ring_reduced[SGPMP_COMM_RING];
    bool ring_used[SGPMP_COMM_RING];
    unsigned long long step;
};

// The events only ever order streams of this device (hipStreamWaitEvent), never the host: a device-scope
// release is enough and spares the system-scope cache flush of a default event record.
static const unsigned kEventFlags = hipEventDisableTiming | hipEventReleaseToDevice;

static hipEvent_t* reduced_event(SgpmpComm* c, double* stats, bool create) {
    for (auto& p : c->reduced)
        if (p.first == stats) return &p.second;
    if (!create) return nullptr;
    if (c->reduced.size() >= 8) {
        // buffers come and go (reset()): keep the table small -- but only entries whose all-reduce has COMPLETED
        // may go (a later sgpmp_stats_wait for an evicted buffer finds no event and does not wait)
        for (size_t i = 0; i < c->reduced.size();) {
            if (hipEventQuery(c->reduced[i].second) == hipSuccess) {
                hipEventDestroy(c->reduced[i].second);
                c->reduced.erase(c->reduced.begin() + (long)i);
            } else ++i;
        }
    }
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, kEventFlags) != hipSuccess) return nullptr;
    c->reduced.emplace_back(stats, e);
    return &c->reduced.back().second;
}

const char* comm_unique_id(unsigned char* out128) {
    if (const char* e = load_rccl()) return e;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return nullptr;
}

const char* comm_create(const unsigned char* id128, int world, int rank, SgpmpComm** out) {
    if (const char* e = load_rccl()) return e;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    SgpmpComm* c = new SgpmpComm();
    c->rank = rank; c->world = world; c->comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return g_rccl.GetErrorString(r); }
    c->side = nullptr; c->produced = nullptr; c->ring = nullptr;
    for (int i = 0; i < SGPMP_COMM_RING; ++i) {
        c->ring_produced[i] = c->ring_produced2[i] = c->ring_reduced[i] = nullptr;
        c->ring_used[i] = false;
    }
    bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->produced, kEventFlags) == hipSuccess &&
              hipMalloc(&c->ring, sizeof(double) * SGPMP_COMM_RING * 2 * SGPMP_STAT_SHARDS * 4) == hipSuccess;
    for (int i = 0; ok && i < SGPMP_COMM_RING; ++i)
        ok = hipEventCreateWithFlags(&c->ring_produced[i], kEventFlags) == hipSuccess &&
             hipEventCreateWithFlags(&c->ring_produced2[i], kEventFlags) == hipSuccess &&
             hipEventCreateWithFlags(&c->ring_reduced[i], kEventFlags) == hipSuccess;
    c->step = 0;
    if (!ok) {
        comm_destroy(c);                                     // (frees whatever was created: all handles start null)
        return "comm_create: cannot create the side stream / events / ring";
    }
    *out = c;
    return nullptr;
}

void comm_destroy(SgpmpComm* c) {
    if (!c) return;
    if (c->side) hipStreamSynchronize(c->side);
    for (auto& p : c->reduced) hipEventDestroy(p.second);
    if (c->produced) hipEventDestroy(c->produced);
    for (int i = 0; i < SGPMP_COMM_RING; ++i) {
        if (c->ring_produced[i]) hipEventDestroy(c->ring_produced[i]);
        if (c->ring_produced2[i]) hipEventDestroy(c->ring_produced2[i]);
        if (c->ring_reduced[i]) hipEventDestroy(c->ring_reduced[i]);
    }
    hipFree(c->ring);
    if (c->side) hipStreamDestroy(c->side);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    delete c;
}

// What the communicator ITSELF says (not what the caller passed to comm_create): a bench line quoting these
// numbers proves that RCCL saw that many ranks.
const char* comm_info(const SgpmpComm* c, int* world, int* rank, int* version) {
    int w = 0, r = 0, v = 0;
    ncclResult_t rc = g_rccl.CommCount(c->comm, &w);
    if (rc == ncclSuccess) rc = g_rccl.CommUserRank(c->comm, &r);
    if (rc == ncclSuccess) rc = g_rccl.GetVersion(&v);
    if (rc != ncclSuccess) return g_rccl.GetErrorString(rc);
    if (world) *world = w;
    if (rank) *rank = r;
    if (version) *version = v;
    return nullptr;
}

int comm_rank(const SgpmpComm* c) { return c->rank; }
int comm_world(const SgpmpComm* c) { return c->world; }

// `stream` has just produced `stats`: sum them over all ranks, in place, on the side stream.
const char* comm_allreduce_stats(SgpmpComm* c, double* stats, hipStream_t stream) {
    hipEvent_t* done = reduced_event(c, stats, true);
    if (!done) return "comm_allreduce_stats: cannot create an event";
    if (hipEventRecord(c->produced, stream) != hipSuccess) return "hipEventRecord failed";
    if (hipStreamWaitEvent(c->side, c->produced, 0) != hipSuccess) return "hipStreamWaitEvent failed";
    const ncclResult_t r = g_rccl.AllReduce(stats, stats, (size_t)SGPMP_STAT_SHARDS * 4, ncclDouble, ncclSum,
                                            c->comm, c->side);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    if (hipEventRecord(*done, c->side) != hipSuccess) return "hipEventRecord failed";
    return nullptr;
}

// The same for any buffer of doubles (the per-goal mean statistics): in place, on the side stream.
const char* comm_allreduce_f64(SgpmpComm* c, double* buf, size_t count, hipStream_t stream) {
    hipEvent_t* done = reduced_event(c, buf, true);
    if (!done) return "comm_allreduce_f64: cannot create an event";
    if (hipEventRecord(c->produced, stream) != hipSuccess) return "hipEventRecord failed";
    if (hipStreamWaitEvent(c->side, c->produced, 0) != hipSuccess) return "hipStreamWaitEvent failed";
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, c->side);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    if (hipEventRecord(*done, c->side) != hipSuccess) return "hipEventRecord failed";
    return nullptr;
}

hipStream_t comm_side_stream(SgpmpComm* c) { return c->side; }

// `buf` has just been all-reduced on the side stream by a caller that enqueued the collective itself
// (sgpmp_step's per-goal mean statistics): remember the event sgpmp_stats_wait(buf) must honour.
const char* comm_mark_reduced(SgpmpComm* c, double* buf) {
    hipEvent_t* done = reduced_event(c, buf, true);
    if (!done) return "comm_mark_reduced: cannot create an event";
    if (hipEventRecord(*done, c->side) != hipSuccess) return "hipEventRecord failed";
    return nullptr;
}

// ---- the per-step path of sgpmp_step ------------------------------------------------------------------
// Begin a step: the ring slot K5 will zero and K4 will accumulate into, and the event to attach to K4's
// dispatch.  If the slot's previous all-reduce (SGPMP_COMM_RING steps ago) were still running, `stream`
// is made to wait for it -- never observed, the query is the common path.
const char* comm_step_begin(SgpmpComm* c, hipStream_t stream, double** slot, hipEvent_t* k4_done) {
    const int r = (int)(c->step % SGPMP_COMM_RING);
    if (c->ring_used[r] && hipEventQuery(c->ring_reduced[r]) != hipSuccess)
        if (hipStreamWaitEvent(stream, c->ring_reduced[r], 0) != hipSuccess) return "hipStreamWaitEvent failed";
    *slot = c->ring + (size_t)r * 2 * SGPMP_STAT_SHARDS * 4;
    *k4_done = c->ring_produced[r];
    return nullptr;
}

// The same for a step that runs as two particle-half chains (api.hip, StepPipe): each chain gets its own block
// of the slot and its own "statistics complete" event.
const char* comm_step_begin2(SgpmpComm* c, hipStream_t s0, hipStream_t s1, double** slot0, double** slot1,
                             hipEvent_t* done0, hipEvent_t* done1) {
    const int r = (int)(c->step % SGPMP_COMM_RING);
    if (c->ring_used[r] && hipEventQuery(c->ring_reduced[r]) != hipSuccess)
        if (hipStreamWaitEvent(s0, c->ring_reduced[r], 0) != hipSuccess ||
            hipStreamWaitEvent(s1, c->ring_reduced[r], 0) != hipSuccess) return "hipStreamWaitEvent failed";
    *slot0 = c->ring + (size_t)r * 2 * SGPMP_STAT_SHARDS * 4;
    *slot1 = *slot0 + SGPMP_STAT_SHARDS * 4;
    *done0 = c->ring_produced[r];
    *done1 = c->ring_produced2[r];
    return nullptr;
}

// End a step (K4 launched with ring_produced[r] as its stop event): sum the slot over all ranks into the
// caller's `stats` on the side stream.
const char* comm_step_end(SgpmpComm* c, double* stats, bool two_halves) {
    const int r = (int)(c->step % SGPMP_COMM_RING);
    double* slot = c->ring + (size_t)r * 2 * SGPMP_STAT_SHARDS * 4;
    if (hipStreamWaitEvent(c->side, c->ring_produced[r], 0) != hipSuccess) return "hipStreamWaitEvent failed";
    if (two_halves) {
        if (hipStreamWaitEvent(c->side, c->ring_produced2[r], 0) != hipSuccess) return "hipStreamWaitEvent failed";
        if (launch_stats_add(slot, slot + SGPMP_STAT_SHARDS * 4, c->side) != hipSuccess) return "statistics add failed";
    }
    const ncclResult_t rc = g_rccl.AllReduce(slot, stats,
                                             (size_t)SGPMP_STAT_SHARDS * 4, ncclDouble, ncclSum, c->comm, c->side);
    if (rc != ncclSuccess) return g_rccl.GetErrorString(rc);
    if (hipEventRecord(c->ring_reduced[r], c->side) != hipSuccess) return "hipEventRecord failed";
    c->ring_used[r] = true;
    ++c->step;
    return nullptr;
}

// Before `stream` touches `stats` again (K5 zeroes it; a reader copies it): wait -- on the stream, not
// on the host -- for the all-reduce that may still be using it.  stats == NULL: every pending one.
const char* comm_stats_wait(SgpmpComm* c, double* stats, hipStream_t stream) {
    for (auto& p : c->reduced)
        if (!stats || p.first == stats)
            if (hipStreamWaitEvent(stream, p.second, 0) != hipSuccess) return "hipStreamWaitEvent failed";
    // the steps' own all-reduces complete in order on the side stream: the newest one covers them all
    if (c->step > 0) {
        const int r = (int)((c->step - 1) % SGPMP_COMM_RING);
        if (hipStreamWaitEvent(stream, c->ring_reduced[r], 0) != hipSuccess) return "hipStreamWaitEvent failed";
    }
    return nullptr;
}

// [count] elements of `bytes_per_elem` from every rank -> [world * count] on every rank, on `stream`.
const char* comm_allgather(SgpmpComm* c, const void* send, void* recv, size_t bytes, hipStream_t stream) {
    const ncclResult_t r = g_rccl.AllGather(send, recv, bytes, ncclUint8, c->comm, stream);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    return nullptr;
}
