// TEST DOUBLE for librccl -- the nine entry points csrc/comm.hip binds -- for SEVERAL PROCESSES SHARING ONE GPU.
//
// The build boxes have one MI355X: real RCCL refuses two ranks on one device, so the library's N > 1 protocol (ring
// slots, event chaining, two-chain steps feeding one all-reduce, empty shards, the per-goal mean statistics, all-gather)
// could never meet a second rank.  This shim implements the collectives over POSIX shared memory, STREAM-ORDERED like
// the real ones: device -> pinned host copy, a host function on the stream (deposit, barrier, reduce in rank order,
// barrier), pinned host -> device copy; successive collectives of a communicator are chained by an event, whatever
// stream they are issued on (as NCCL chains them).  Every wait has a time-out: a rank that never issues its collective
// fails the test instead of hanging the box.  libsgpmp.so loads it when SGPMP_RCCL_LIB points here (tests only).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclUint8 = 1, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

#define FAKE_MAX_RANKS 8
#define FAKE_MAX_BYTES (4u << 20)
#define FAKE_SLOTS 16
#define FAKE_TIMEOUT_S 60

struct Shm {
    std::atomic<unsigned> ready;                 // rank 0 has initialised the segment
    std::atomic<unsigned> count, gen;            // barrier
    unsigned long long meta[FAKE_MAX_RANKS][3];  // (sequence number, kind, bytes) of the collective each rank is in
    unsigned char data[FAKE_MAX_RANKS][FAKE_MAX_BYTES];
};

struct ncclComm {
    int rank, world;
    Shm* shm;
    char name[64];
    void* hin[FAKE_SLOTS];
    void* hout[FAKE_SLOTS];
    size_t cap[FAKE_SLOTS];
    unsigned long long seq;
    hipEvent_t last;                             // completion of the previous collective of this communicator
    bool have_last;
};
typedef ncclComm* ncclComm_t;

static void die(const char* what) {
    std::fprintf(stderr, "fake_rccl: %s\n", what);
    std::fflush(stderr);
    std::_Exit(86);
}

static void barrier(ncclComm* c) {
    Shm* s = c->shm;
    const unsigned g = s->gen.load(std::memory_order_acquire);
    if (s->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->world) {
        s->count.store(0, std::memory_order_relaxed);
        s->gen.fetch_add(1, std::memory_order_acq_rel);
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->gen.load(std::memory_order_acquire) == g) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(FAKE_TIMEOUT_S))
            die("time-out in a collective: a rank never arrived (ordering / participation bug in the caller)");
    }
}

struct Job { ncclComm* c; int slot; size_t bytes; size_t count; int kind; unsigned long long seq; };   // kind 0: all-reduce (double sum), 1: all-gather

static void host_step(void* p) {
    Job* j = (Job*)p;
    ncclComm* c = j->c;
    std::memcpy(c->shm->data[c->rank], c->hin[j->slot], j->bytes);
    c->shm->meta[c->rank][0] = j->seq; c->shm->meta[c->rank][1] = (unsigned long long)j->kind; c->shm->meta[c->rank][2] = j->bytes;
    barrier(c);
    // a communicator pairs collectives by ORDER OF ISSUE: every rank must be in the same one (real RCCL hangs or
    // reduces unrelated buffers into each other when they are not)
    for (int r = 0; r < c->world; ++r)
        if (c->shm->meta[r][0] != j->seq || c->shm->meta[r][1] != (unsigned long long)j->kind || c->shm->meta[r][2] != j->bytes) {
            std::fprintf(stderr, "fake_rccl: rank %d is in collective #%llu (kind %d, %zu bytes) but rank %d in #%llu (kind %llu, %llu bytes)\n",
                         c->rank, j->seq, j->kind, j->bytes, r, c->shm->meta[r][0], c->shm->meta[r][1], c->shm->meta[r][2]);
            die("ranks issued different collectives in the same position");
        }
    if (j->kind == 0) {
        double* out = (double*)c->hout[j->slot];
        for (size_t i = 0; i < j->count; ++i) {
            double v = 0.;
            for (int r = 0; r < c->world; ++r) v += ((const double*)c->shm->data[r])[i];    // rank order: same bits on every rank
            out[i] = v;
        }
    } else {
        for (int r = 0; r < c->world; ++r) std::memcpy((char*)c->hout[j->slot] + (size_t)r * j->bytes, c->shm->data[r], j->bytes);
    }
    barrier(c);
    delete j;
}

static ncclResult_t collective(ncclComm* c, const void* send, void* recv, size_t bytes, size_t count, int kind, hipStream_t st) {
    if (bytes > FAKE_MAX_BYTES) return ncclInvalidArgument;
    const unsigned long long seq = c->seq++;
    const int slot = (int)(seq % FAKE_SLOTS);
    const size_t out_bytes = kind == 0 ? bytes : bytes * c->world;
    if (c->cap[slot] < out_bytes) {
        if (c->hin[slot]) { (void)hipHostFree(c->hin[slot]); (void)hipHostFree(c->hout[slot]); }
        if (hipHostMalloc(&c->hin[slot], out_bytes) != hipSuccess || hipHostMalloc(&c->hout[slot], out_bytes) != hipSuccess)
            return ncclUnhandledCudaError;
        c->cap[slot] = out_bytes;
    }
    if (c->have_last && hipStreamWaitEvent(st, c->last, 0) != hipSuccess) return ncclUnhandledCudaError;   // chain, as NCCL does
    if (hipMemcpyAsync(c->hin[slot], send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(st, host_step, new Job{c, slot, bytes, count, kind, seq}) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, c->hout[slot], out_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(c->last, st) != hipSuccess) return ncclUnhandledCudaError;
    c->have_last = true;
    return ncclSuccess;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id->internal, 0, sizeof(id->internal));
    std::snprintf(id->internal, sizeof(id->internal), "/sgpmp_fake_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    std::memset(c, 0, sizeof(*c));
    c->rank = rank; c->world = nranks;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0) {
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(FAKE_TIMEOUT_S)) return ncclSystemError;
        }
        struct stat sb;
        while (fstat(fd, &sb) == 0 && (size_t)sb.st_size < sizeof(Shm)) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    c->shm = (Shm*)mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->shm == MAP_FAILED) return ncclSystemError;
    if (rank == 0) { c->shm->count.store(0); c->shm->gen.store(0); c->shm->ready.store(1, std::memory_order_release); }
    else {
        const auto t0 = std::chrono::steady_clock::now();
        while (c->shm->ready.load(std::memory_order_acquire) != 1) {
            std::this_thread::yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(FAKE_TIMEOUT_S)) return ncclSystemError;
        }
    }
    if (hipEventCreateWithFlags(&c->last, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    if (rank == 0) shm_unlink(c->name);          // every rank has it mapped: the name can go (nothing is left behind if a rank dies)
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < FAKE_SLOTS; ++i) if (c->hin[i]) { (void)hipHostFree(c->hin[i]); (void)hipHostFree(c->hout[i]); }
    (void)hipEventDestroy(c->last);
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
    if (dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;
    return collective(c, send, recv, count * sizeof(double), count, 0, st);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t dt, ncclComm_t c, hipStream_t st) {
    if (dt != ncclUint8) return ncclInvalidArgument;
    return collective(c, send, recv, sendcount, sendcount, 1, st);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) { case ncclSuccess: return "success"; case ncclUnhandledCudaError: return "fake_rccl: HIP error";
                 case ncclSystemError: return "fake_rccl: shared-memory set-up failed"; case ncclInvalidArgument: return "fake_rccl: invalid argument";
                 default: return "fake_rccl: internal error"; }
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
ncclResult_t ncclGetVersion(int* v) { *v = 1; return ncclSuccess; }      // (1: "this is the test double")
}
