// A stand-in for the HIP RUNTIME LIBRARY (not for its headers: the real <hip/hip_runtime.h> is used) that lets the HOST side of
// libsgpmp.so -- api.hip, comm.hip: context bookkeeping, ring slots, event tables, two-chain stream choreography -- run on a
// machine without a GPU under AddressSanitizer / UBSan.  TEST INFRASTRUCTURE (tests/test_cpu_host.py).
//   * device memory is host memory from malloc: the sanitizer sees every out-of-bounds or use-after-free access the host code
//     or the stub launchers (stub_launchers.cpp: they touch exactly the byte ranges the real kernels would) make;
//   * streams execute at enqueue time; an event is "complete" once recorded; hipEventQuery answers hipErrorNotReady for the
//     first query of every recorded event when STUB_EVENT_LAG=1, so the fall-back waits of the host code run too.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

namespace {
struct StubEvent { bool recorded = false; int queries = 0; };
struct StubStream { int id; };
std::set<void*> g_live_dev;            // device allocations: freeing anything else is an error of the code under test
std::set<StubEvent*> g_live_ev;
std::set<StubStream*> g_live_st;
int g_next_stream = 1;
bool lag() { const char* e = getenv("STUB_EVENT_LAG"); return e && *e == '1'; }
void check_stream(hipStream_t s) {
    if (s && !g_live_st.count((StubStream*)s)) { std::fprintf(stderr, "stub_hip: use of a destroyed / unknown stream\n"); std::abort(); }
}
StubEvent* ev(hipEvent_t e) {
    if (!g_live_ev.count((StubEvent*)e)) { std::fprintf(stderr, "stub_hip: use of a destroyed / unknown event\n"); std::abort(); }
    return (StubEvent*)e;
}
}  // namespace

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub error"; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); g_live_dev.insert(*p); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    if (!g_live_dev.erase(p)) { std::fprintf(stderr, "stub_hip: hipFree of a pointer that is not a live allocation\n"); std::abort(); }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); g_live_dev.insert(*p); return hipSuccess; }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { check_stream(st); std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { check_stream(st); std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { auto* x = new StubStream{g_next_stream++}; g_live_st.insert(x); *s = (hipStream_t)x; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) {
    if (!g_live_st.erase((StubStream*)s)) { std::fprintf(stderr, "stub_hip: double destroy of a stream\n"); std::abort(); }
    delete (StubStream*)s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { check_stream(s); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { check_stream(s); (void)ev(e); return hipSuccess; }
hipError_t hipLaunchHostFunc(hipStream_t s, hipHostFn_t fn, void* arg) { check_stream(s); fn(arg); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { auto* x = new StubEvent(); g_live_ev.insert(x); *e = (hipEvent_t)x; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
    if (!g_live_ev.erase((StubEvent*)e)) { std::fprintf(stderr, "stub_hip: double destroy of an event\n"); std::abort(); }
    delete (StubEvent*)e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { check_stream(s); StubEvent* x = ev(e); x->recorded = true; x->queries = 0; return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) {
    StubEvent* x = ev(e);
    if (lag() && x->recorded && x->queries++ == 0) return hipErrorNotReady;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { (void)ev(e); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { (void)ev(a); (void)ev(b); *ms = 0.01f; return hipSuccess; }

// leak check at exit: every stream / event / device buffer the library made must be gone after sgpmp_destroy
int stub_hip_live_objects(void) { return (int)(g_live_dev.size() + g_live_ev.size() + g_live_st.size()); }
}
