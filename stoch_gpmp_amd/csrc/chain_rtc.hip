// Run-time compiled chain kernels: the fused sampler + sweep launch and the chunked sweep for ANY serial chain.
//
// The reference takes any FK callable (cost_functions.py:39,51-52).  The fast launches (fused_step.inc) need the chain as
// straight-line, constant-folded code -- `struct ChainCode_<name>`, what gen/chain_codegen.py emits -- and the library
// carries that code for the Panda only (chain_code_generated.h, built with the library).  For every other chain the host
// hands the generated struct to sgpmp_set_fk_codegen at set-up time and this file compiles the SAME kernel source
// (cost_device.h + fused_step.inc, read from the csrc directory next to the library) around it with hiprtc, for gfx950,
// one field type at a time, on first use.  Code objects are cached per (chain code, field type, kernel sources) in this
// process and on disk.  libhiprtc is loaded with dlopen: where it is absent the chain stays on the generic sweep.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <hip/hip_ext.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include <fcntl.h>
#include <unistd.h>
#include "rng.h"
#include "sgpmp_internal.h"

#ifndef SGPMP_CSRC_DEFAULT
#define SGPMP_CSRC_DEFAULT ""
#endif
#if __has_include("build_extra.h")
#include "build_extra.h"                     // SGPMP_BUILD_EXTRA as a properly escaped literal (Makefile, gen/build_extra.py)
#endif
#ifndef SGPMP_BUILD_EXTRA
#define SGPMP_BUILD_EXTRA ""                 // the EXTRA compile flags of this build (Makefile): the run-time compiler gets the same -D's
#endif
#if __has_include("rtc_sources_hash.h")
#include "rtc_sources_hash.h"                // SGPMP_RTC_SOURCES_HASH: FNV-1a of the kernel sources this library was built from (Makefile)
#endif

namespace {

struct HiprtcApi {
    void* handle = nullptr;
    bool tried = false;
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char**) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
    hiprtcResult (*Version)(int*, int*) = nullptr;       // optional
};
HiprtcApi g_rtc;
std::mutex g_mu;

const char* load_hiprtc() {
    if (g_rtc.handle) return nullptr;
    if (g_rtc.tried) return "libhiprtc not available";
    g_rtc.tried = true;
    void* h = nullptr;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return "libhiprtc not found (dlopen)";
#define SYM(field, name)                                             \
    g_rtc.field = (decltype(g_rtc.field))dlsym(h, name);             \
    if (!g_rtc.field) return "libhiprtc: missing symbol " name;
    SYM(CreateProgram, "hiprtcCreateProgram")
    SYM(CompileProgram, "hiprtcCompileProgram")
    SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
    SYM(GetProgramLog, "hiprtcGetProgramLog")
    SYM(GetCodeSize, "hiprtcGetCodeSize")
    SYM(GetCode, "hiprtcGetCode")
    SYM(DestroyProgram, "hiprtcDestroyProgram")
#undef SYM
    g_rtc.Version = (decltype(g_rtc.Version))dlsym(h, "hiprtcVersion");
    g_rtc.handle = h;
    return nullptr;
}

// every file the run-time translation unit includes (a change of any of them changes the cache key)
const char* const kSources[] = {"sgpmp_internal.h", "rng.h", "update_common.h", "cost_device.h", "cost_sweep_kernel.inc",
                                "cost_sweep_dual.inc", "fused_step.inc", "../../include/sgpmp.h"};

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// The csrc directory: SGPMP_CSRC_DIR, else "csrc" next to this shared object, else where the library was built.
std::string csrc_dir() {
    if (const char* e = getenv("SGPMP_CSRC_DIR")) return e;
    Dl_info info;
    if (dladdr((const void*)&csrc_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.rfind('/');
        p = (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
        if (file_exists(p + "/fused_step.inc")) return p;
    }
    return SGPMP_CSRC_DEFAULT;
}

uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

// The directory code objects are cached in, or "" (no cache).  Whatever is found there is LOADED INTO THE GPU, so the
// directory must be this user's own and closed to everybody else: owner = getuid(), a real directory (not a symlink), no
// group / other permission bits -- otherwise another user of the machine could pre-seed a code object (round-4 advisor
// finding: the /tmp fallback was used unchecked).  A directory that fails the check is not used at all.
std::string cache_dir() {
    std::string d;
    if (const char* e = getenv("SGPMP_RTC_CACHE")) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/sgpmp";
    else if (const char* h = getenv("HOME")) d = std::string(h) + "/.cache/sgpmp";
    else d = "/tmp/sgpmp-rtc-" + std::to_string((long)getuid());
    if (d == "0" || d == "off") return "";
    // (mkdir -p, two levels are enough for the defaults)
    const size_t k = d.rfind('/');
    if (k != std::string::npos && k > 0) mkdir(d.substr(0, k).c_str(), 0700);
    mkdir(d.c_str(), 0700);
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & (S_IRWXG | S_IRWXO)) != 0) return "";
    return d;
}

// FNV-1a over the contents of the kernel sources in `dir`, in kSources order (what gen/rtc_hash.py computed at build time)
bool sources_hash(const std::string& dir, uint64_t* out, std::string* missing) {
    uint64_t h = 14695981039346656037ull;
    for (const char* f : kSources) {
        std::ifstream in(dir + "/" + f, std::ios::binary);
        if (!in) { if (missing) *missing = f; return false; }
        std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        h = fnv1a(h, buf.data(), buf.size());
    }
    *out = h;
    return true;
}

// The kernel sources on disk must be the ones this library was built from: the host side passes CostArgs / FlatProg /
// FusedArgs BY VALUE in layouts fixed at build time, and a kernel compiled from edited sources (or for a library loaded
// through SGPMP_LIB_PATH from another tree) would read them with another layout -- silent memory corruption (round-4
// advisor finding).  Returns nullptr when they match (or the development override SGPMP_RTC_ALLOW_EDITED_SOURCES=1 is set).
const char* check_sources(const std::string& dir, uint64_t* hash_out) {
    static thread_local std::string msg;
    std::string missing;
    uint64_t h = 0;
    if (!sources_hash(dir, &h, &missing)) { msg = "kernel source missing: " + missing; return msg.c_str(); }
    *hash_out = h;
#ifdef SGPMP_RTC_SOURCES_HASH
    if (h != SGPMP_RTC_SOURCES_HASH) {
        const char* ov = getenv("SGPMP_RTC_ALLOW_EDITED_SOURCES");
        if (!(ov && *ov == '1')) {
            msg = "the kernel sources in " + dir + " are not the ones this libsgpmp.so was built from (content hash differs): "
                  "rebuild the library, or point SGPMP_CSRC_DIR at its own csrc/";
            return msg.c_str();
        }
    }
#endif
    return nullptr;
}

}  // namespace

struct RtcChain {
    std::string struct_src;      // `struct ChainCode_rt { ... };` as gen/chain_codegen.py emits it
    int n_dof = 0;
    uint64_t key = 0;            // of struct_src
    hipModule_t mod[3] = {nullptr, nullptr, nullptr};
    hipFunction_t fused[3] = {nullptr, nullptr, nullptr}, sweep[3] = {nullptr, nullptr, nullptr}, probe = nullptr;
    hipFunction_t fused_rag[3] = {nullptr, nullptr, nullptr};   // ... for S, T off the launch's grid of 8 x 16 (fused_step.inc: RAG)
    hipFunction_t fused_small[3] = {nullptr, nullptr, nullptr}, fused_small_rag[3] = {nullptr, nullptr, nullptr};   // ... for steps of few items: one workgroup per item (fused_step.inc: LAT)
    bool tried[3] = {false, false, false};
    std::string err;             // why a compilation failed (sgpmp_fk_codegen_info)
    double compile_s = 0.;       // seconds spent in hiprtc for this chain (0: every code object came from the cache)
    int from_cache = 0, compiled = 0;
};

static std::map<uint64_t, RtcChain*> g_chains;      // process-wide: contexts of one process share the modules

static std::string translation_unit(const RtcChain& c) {
    std::ostringstream tu;
    tu << "// run-time translation unit of libsgpmp.so (chain_rtc.hip)\n"
          "#define SGPMP_RTC 1\n"
          "using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; "
          "using __hip_internal::uint64_t;\n"
          "#include \"sgpmp_internal.h\"\n#include \"rng.h\"\n#include \"update_common.h\"\n"
          "#pragma clang diagnostic ignored \"-Wunused-variable\"\n"
       << c.struct_src
       << "\nstatic_assert(ChainCode_rt::NREP >= 1 && ChainCode_rt::NREP <= SGPMP_MAX_LINKS && ChainCode_rt::NJ <= SGPMP_MAX_JOINTS && "
          "ChainCode_rt::N >= 1 && ChainCode_rt::N <= 7 && ChainCode_rt::NPAIR >= 0 && ChainCode_rt::NPAIR <= SGPMP_MAX_LINKS * SGPMP_MAX_LINKS / 2, "
          "\"chain code outside the library's link / joint / pair tables\");\n"
       << "using ChainCode_panda = ChainCode_rt;       // (the generic kernels' ChainOf<> default; never instantiated here)\n"
          "#include \"cost_device.h\"\n#include \"cost_sweep_kernel.inc\"\n#include \"cost_sweep_dual.inc\"\n"
          "#include \"fused_step.inc\"\n"
          "extern \"C\" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SGPMP_FUSED_WAVES, SGPMP_FUSED_WAVES)))\n"
          "sgpmp_rtc_fused(const CostArgs<float> a, const FlatProg<float> F, const FusedArgs s) {\n"
          "    chunked_body<ChainCode_rt::N, ChainCode_rt, SGPMP_RTC_FT, false>(a, F, s);\n}\n"
          "extern \"C\" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SGPMP_FUSED_WAVES, SGPMP_FUSED_WAVES)))\n"
          "sgpmp_rtc_fused_rag(const CostArgs<float> a, const FlatProg<float> F, const FusedArgs s) {\n"
          "    chunked_body<ChainCode_rt::N, ChainCode_rt, SGPMP_RTC_FT, false, true>(a, F, s);\n}\n"
          "extern \"C\" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))\n"
          "sgpmp_rtc_fused_small(const CostArgs<float> a, const FlatProg<float> F, const FusedArgs s) {\n"
          "    chunked_body<ChainCode_rt::N, ChainCode_rt, SGPMP_RTC_FT, false, false, true>(a, F, s);\n}\n"
          "extern \"C\" __global__ void __launch_bounds__(256)\n"
          "sgpmp_rtc_fused_small_rag(const CostArgs<float> a, const FlatProg<float> F, const FusedArgs s) {\n"
          "    chunked_body<ChainCode_rt::N, ChainCode_rt, SGPMP_RTC_FT, false, true, true>(a, F, s);\n}\n"
          "extern \"C\" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SGPMP_FUSED_WAVES, SGPMP_FUSED_WAVES)))\n"
          "sgpmp_rtc_sweep(const CostArgs<float> a, const FlatProg<float> F, const FusedArgs s) {\n"
          "    chunked_body<ChainCode_rt::N, ChainCode_rt, SGPMP_RTC_FT, true>(a, F, s);\n}\n"
          // what the host checks the generated code against: link positions at given joint vectors and the link / pair tables
          "extern \"C\" __global__ void sgpmp_rtc_probe(const float* __restrict__ q, int K, float* __restrict__ pos, float* __restrict__ tab) {\n"
          "    using CC = ChainCode_rt;\n"
          "    const int k = blockIdx.x * blockDim.x + threadIdx.x;\n"
          "    if (k < K) {\n"
          "        float qq[CC::N], P[CC::NREP][3];\n"
          "        for (int i = 0; i < CC::N; ++i) qq[i] = q[k * CC::N + i];\n"
          "        fk_cg<float, CC>(qq, P);\n"
          "        for (int l = 0; l < CC::NREP; ++l) for (int x = 0; x < 3; ++x) pos[(k * CC::NREP + l) * 3 + x] = P[l][x];\n"
          "    }\n"
          "    if (k == 0) {\n"
          "        float msum = 0.f, wsum = 0.f; int nstat = 0;\n"
          "        for (int l = 0; l < CC::NREP; ++l) { msum += CC::mult(l); nstat += CC::is_static(l) ? 1 : 0; tab[8 + l] = (float)CC::rep_link(l); }\n"
          "        for (int p = 0; p < CC::NPAIR; ++p) wsum += CC::pair_w(p);\n"
          "        tab[0] = (float)CC::N; tab[1] = (float)CC::NJ; tab[2] = (float)CC::NREP; tab[3] = (float)CC::NPAIR;\n"
          "        tab[4] = msum; tab[5] = wsum; tab[6] = (float)nstat; tab[7] = 0.f;\n"
          "    }\n}\n";
    return tu.str();
}

// hiprtc: translation unit -> gfx950 code object.  Needs no device.
// *compiler_verdict (may be null): true iff the failure is hiprtc's judgement of the SOURCE -- final for the process -- and not an
// environment problem (no hiprtc, no program object) that a later call may find repaired.
static bool compile_tu(const std::string& tu, const std::string& dir, int ft, std::vector<char>& code, std::string& err, double* secs,
                       bool* compiler_verdict = nullptr) {
    if (compiler_verdict) *compiler_verdict = false;
    if (const char* e = load_hiprtc()) { err = e; return false; }
    hiprtcProgram prog;
    if (g_rtc.CreateProgram(&prog, tu.c_str(), "sgpmp_chain_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        err = "hiprtcCreateProgram failed";
        return false;
    }
    const std::string inc = "-I" + dir, ftd = "-DSGPMP_RTC_FT=" + std::to_string(ft),
                      rounds = "-DSGPMP_PHILOX_ROUNDS=" + std::to_string(SGPMP_PHILOX_ROUNDS);
    std::vector<std::string> extra;                          // the library's own EXTRA -D flags: kernels and host agree on every build parameter
    { std::istringstream es(SGPMP_BUILD_EXTRA); for (std::string tok; es >> tok;) if (tok.rfind("-D", 0) == 0 && tok.find("SGPMP_PHILOX_ROUNDS") == std::string::npos) extra.push_back(tok); }
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", inc.c_str(), ftd.c_str(), rounds.c_str()};
    for (const std::string& e : extra) opts.push_back(e.c_str());
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const hiprtcResult r = g_rtc.CompileProgram(prog, (int)opts.size(), opts.data());
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (secs) *secs += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        g_rtc.GetProgramLogSize(prog, &ls);
        std::string log(ls + 1, '\0');
        if (ls) g_rtc.GetProgramLog(prog, &log[0]);
        err = "hiprtc: compilation of the chain kernels failed:\n" + log.substr(0, 4000);
        g_rtc.DestroyProgram(&prog);
        if (compiler_verdict) *compiler_verdict = true;
        return false;
    }
    size_t cs = 0;
    g_rtc.GetCodeSize(prog, &cs);
    code.resize(cs);
    g_rtc.GetCode(prog, code.data());
    g_rtc.DestroyProgram(&prog);
    return true;
}

// Compile-only check of chain code (no device, no cache): does `struct_src` build into the chain kernels?  -> bytes of code
// object, or -1 with the reason in `err`.
long long rtc_compile_check(const char* struct_src, int ft, std::string& err) {
    std::lock_guard<std::mutex> lk(g_mu);
    const std::string dir = csrc_dir();
    if (dir.empty() || !file_exists(dir + "/fused_step.inc")) { err = "kernel sources not found (csrc/ next to libsgpmp.so; SGPMP_CSRC_DIR)"; return -1; }
    uint64_t srch = 0;
    if (const char* e = check_sources(dir, &srch)) { err = e; return -1; }
    RtcChain tmp;
    tmp.struct_src = struct_src;
    std::vector<char> code;
    if (!compile_tu(translation_unit(tmp), dir, ft, code, err, nullptr)) return -1;
    return (long long)code.size();
}

// Compile (or fetch from the disk cache) the code object of chain `c` for sphere-field type `ft`; load it as a module.
static bool build_module(RtcChain& c, int ft) {
    if (c.mod[ft]) return true;
    // (only a COMPILER verdict is final for the process: missing sources, an unusable cache directory or a failed module
    // load may be repaired -- SGPMP_CSRC_DIR, a rebuilt library -- and are retried on the next call)
    if (c.tried[ft]) return false;
    const std::string dir = csrc_dir();
    if (dir.empty() || !file_exists(dir + "/fused_step.inc")) {
        c.err = "kernel sources not found (looked for csrc/fused_step.inc next to libsgpmp.so; set SGPMP_CSRC_DIR)";
        return false;
    }
    uint64_t srch = 0;
    if (const char* e = check_sources(dir, &srch)) { c.err = e; return false; }
    const std::string tu = translation_unit(c);
    // cache key: the translation unit, the field type, the build parameters (incl. the EXTRA flags), the run-time compiler's
    // version and the CONTENT of every included file
    uint64_t key = fnv1a(14695981039346656037ull, tu.data(), tu.size());
    int rv[2] = {0, 0};
    if (!load_hiprtc() && g_rtc.Version) g_rtc.Version(&rv[0], &rv[1]);
    const int params[5] = {ft, SGPMP_PHILOX_ROUNDS, SGPMP_ABI_VERSION, rv[0], rv[1]};
    key = fnv1a(key, params, sizeof(params));
    key = fnv1a(key, SGPMP_BUILD_EXTRA, sizeof(SGPMP_BUILD_EXTRA));
    key = fnv1a(key, &srch, sizeof(srch));
    char name[64];
    std::snprintf(name, sizeof(name), "chain-%016llx-ft%d.hsaco", (unsigned long long)key, ft);
    const std::string cdir = cache_dir();
    const std::string cpath = cdir.empty() ? std::string() : cdir + "/" + name;
    std::vector<char> code;
    if (!cpath.empty()) {
        // the cached object is trusted like the directory: opened without following a link, then checked ON THE OPEN DESCRIPTOR --
        // a regular file of this user's, not writable by anyone else (advisor finding, round 5: the directory check alone does not
        // cover a file planted while the directory was still open to others)
        const int fd = open(cpath.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        if (fd >= 0) {
            struct stat fst;
            if (fstat(fd, &fst) == 0 && S_ISREG(fst.st_mode) && fst.st_uid == getuid() && (fst.st_mode & (S_IWGRP | S_IWOTH)) == 0 &&
                fst.st_size > 0 && fst.st_size < (off_t)(256u << 20)) {
                code.resize((size_t)fst.st_size);
                size_t got = 0;
                while (got < code.size()) {
                    const ssize_t k = read(fd, code.data() + got, code.size() - got);
                    if (k <= 0) break;
                    got += (size_t)k;
                }
                if (got != code.size()) code.clear();
            }
            close(fd);
            if (!code.empty()) c.from_cache += 1;
        }
    }
    if (code.empty()) {
        bool verdict = false;
        if (!compile_tu(tu, dir, ft, code, c.err, &c.compile_s, &verdict)) { c.tried[ft] = verdict; return false; }
        c.compiled += 1;
        if (!cpath.empty()) {                     // write-then-rename: concurrent processes never see half a file
            const std::string tmp = cpath + "." + std::to_string((long)getpid());
            std::ofstream f(tmp, std::ios::binary);
            if (f) { f.write(code.data(), (std::streamsize)code.size()); f.close(); std::rename(tmp.c_str(), cpath.c_str()); }
        }
    }
    hipModule_t m = nullptr;
    if (hipModuleLoadData(&m, code.data()) != hipSuccess) { c.err = "hipModuleLoadData failed for the chain kernels"; (void)hipGetLastError(); return false; }
    hipFunction_t f1 = nullptr, f2 = nullptr, f3 = nullptr, f4 = nullptr, f5 = nullptr, f6 = nullptr;
    if (hipModuleGetFunction(&f1, m, "sgpmp_rtc_fused") != hipSuccess || hipModuleGetFunction(&f2, m, "sgpmp_rtc_sweep") != hipSuccess ||
        hipModuleGetFunction(&f3, m, "sgpmp_rtc_probe") != hipSuccess || hipModuleGetFunction(&f4, m, "sgpmp_rtc_fused_rag") != hipSuccess ||
        hipModuleGetFunction(&f5, m, "sgpmp_rtc_fused_small") != hipSuccess || hipModuleGetFunction(&f6, m, "sgpmp_rtc_fused_small_rag") != hipSuccess) {
        c.err = "chain code object lacks its kernels";
        (void)hipGetLastError();
        hipModuleUnload(m);
        return false;
    }
    c.mod[ft] = m; c.fused[ft] = f1; c.sweep[ft] = f2; c.fused_rag[ft] = f4; c.fused_small[ft] = f5; c.fused_small_rag[ft] = f6;
    if (!c.probe) c.probe = f3;
    return true;
}

// Register chain code (idempotent per source text); *out stays valid for the life of the process.
const char* rtc_chain_get(const char* struct_src, int n_dof, RtcChain** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t len = std::strlen(struct_src);
    if (len < 32 || !std::strstr(struct_src, "struct ChainCode_rt")) return "chain code must define `struct ChainCode_rt` (gen/chain_codegen.py gen_chain('rt', chain))";
    const uint64_t key = fnv1a(fnv1a(14695981039346656037ull, struct_src, len), &n_dof, sizeof(n_dof));
    auto it = g_chains.find(key);
    if (it == g_chains.end()) {
        RtcChain* c = new RtcChain();
        c->struct_src.assign(struct_src, len);
        c->n_dof = n_dof; c->key = key;
        it = g_chains.emplace(key, c).first;
    }
    *out = it->second;
    return nullptr;
}

// The kernel for (field type, fused launch | stand-alone sweep), compiled on first use; null when unavailable (c->err says why).
hipFunction_t rtc_kernel(RtcChain* c, int ft, bool sweep, bool rag, bool small) {
    if (!c || ft < 0 || ft > 2) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!build_module(*c, ft)) return nullptr;
    return sweep ? c->sweep[ft] : small ? (rag ? c->fused_small_rag[ft] : c->fused_small[ft]) : rag ? c->fused_rag[ft] : c->fused[ft];
}

const char* rtc_error(const RtcChain* c) { return c ? c->err.c_str() : ""; }
void rtc_stats(const RtcChain* c, double* compile_s, int* compiled, int* from_cache) {
    if (compile_s) *compile_s = c ? c->compile_s : 0.;
    if (compiled) *compiled = c ? c->compiled : 0;
    if (from_cache) *from_cache = c ? c->from_cache : 0;
}

hipError_t rtc_launch(hipFunction_t f, unsigned blocks, unsigned dyn_lds, hipStream_t stream, void** args, hipEvent_t done) {
    return hipExtModuleLaunchKernel(f, blocks * 256u, 1, 1, 256, 1, 1, dyn_lds, stream, args, nullptr, nullptr, done, 0);
}

// Does the generated code describe THIS chain?  Link positions of the distinct links at pseudo-random joint vectors against
// the host's double-precision FK of the chain given to sgpmp_set_fk, and the link / pair tables against the host analysis.
// *mismatch (may be null): 1 when the failure is "this code is not this chain's" (the caller's SGPMP_EINVAL), 0 when the chain
// kernels are merely unavailable (no compiler, no sources, a failed launch: SGPMP_ESTATE).
const char* rtc_verify(RtcChain* c, const ChainDev& ch, int ft_hint, int* mismatch) {
    if (mismatch) *mismatch = 0;
    if (!rtc_kernel(c, ft_hint, false)) return c->err.c_str();
    const int K = 16, N = c->n_dof, ML = SGPMP_MAX_LINKS;
    std::vector<float> q((size_t)K * N);
    uint64_t lcg = 0x2545F4914F6CDD1Dull;
    for (auto& v : q) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; v = (float)(((double)(lcg >> 11) / 9007199254740992.0 * 2. - 1.) * 2.8); }
    float *dq = nullptr, *dpos = nullptr, *dtab = nullptr;
    const size_t npos = (size_t)K * ML * 3, ntab = 8 + ML;
    if (hipMalloc(&dq, q.size() * 4) != hipSuccess || hipMalloc(&dpos, npos * 4) != hipSuccess || hipMalloc(&dtab, ntab * 4) != hipSuccess)
        return "hipMalloc failed (chain verification)";
    hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dtab, 0, ntab * 4);
    int Kc = K;
    void* args[] = {&dq, &Kc, &dpos, &dtab};
    hipError_t e = hipModuleLaunchKernel(c->probe, 1, 1, 1, 64, 1, 1, 0, nullptr, args, nullptr);
    std::vector<float> pos(npos), tab(ntab);
    if (e == hipSuccess) e = hipMemcpy(tab.data(), dtab, ntab * 4, hipMemcpyDeviceToHost);
    const int nrep = e == hipSuccess ? (int)tab[2] : 0;
    if (e == hipSuccess && nrep > 0 && nrep <= ML) e = hipMemcpy(pos.data(), dpos, (size_t)K * nrep * 3 * 4, hipMemcpyDeviceToHost);
    hipFree(dq); hipFree(dpos); hipFree(dtab);
    if (e != hipSuccess) return "chain verification kernel failed";
    const FkPlan& pl = ch.plan;
    double msum = 0., wsum = 0.;
    int npair = 0;
    for (int i = 0; i < ML; ++i) msum += pl.mult[i];
    for (int i = 0; i < ML * ML; ++i) if (pl.wpair[i] != 0.f) { wsum += pl.wpair[i]; ++npair; }
    static thread_local std::string msg;
    if ((int)tab[0] != N || (int)tab[1] != ch.n_joints || nrep != pl.n_rep || (int)tab[3] != npair ||
        std::fabs(tab[4] - msum) > 1e-3 || std::fabs(tab[5] - wsum) > 1e-3) {
        if (mismatch) *mismatch = 1;
        msg = "generated chain code does not match the chain of sgpmp_set_fk (tables: N " + std::to_string((int)tab[0]) + "/" + std::to_string(N) +
              ", joints " + std::to_string((int)tab[1]) + "/" + std::to_string(ch.n_joints) + ", distinct links " + std::to_string(nrep) + "/" +
              std::to_string(pl.n_rep) + ", q-dependent pairs " + std::to_string((int)tab[3]) + "/" + std::to_string(npair) + ")";
        return msg.c_str();
    }
    // positions: host FK in double (same recursion as api.hip host_fk_points)
    for (int k = 0; k < K; ++k) {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0}, hp[SGPMP_MAX_LINKS][3];
        hp[0][0] = hp[0][1] = hp[0][2] = 0.;
        for (int j = 0; j < ch.n_joints; ++j) {
            const JointDev& J = ch.j[j];
            double Rn[9];
            for (int r = 0; r < 3; ++r) p[r] += R[r * 3] * J.t[0] + R[r * 3 + 1] * J.t[1] + R[r * 3 + 2] * J.t[2];
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) Rn[r * 3 + cc] = R[r * 3] * J.R[cc] + R[r * 3 + 1] * J.R[3 + cc] + R[r * 3 + 2] * J.R[6 + cc];
            if (J.revolute) {
                const double s = std::sin((double)q[(size_t)k * N + J.qidx]), co = std::cos((double)q[(size_t)k * N + J.qidx]);
                for (int r = 0; r < 3; ++r) { const double a = Rn[r * 3], b = Rn[r * 3 + 1]; Rn[r * 3] = a * co + b * s; Rn[r * 3 + 1] = b * co - a * s; }
            }
            for (int i = 0; i < 9; ++i) R[i] = Rn[i];
            for (int i = 0; i < 3; ++i) hp[j + 1][i] = p[i];
        }
        for (int l = 0; l < nrep; ++l) {
            const int link = (int)tab[8 + l];
            if (link < 0 || link > ch.n_joints) { if (mismatch) *mismatch = 1; return "generated chain code names a link outside the chain"; }
            for (int x = 0; x < 3; ++x)
                if (std::fabs((double)pos[((size_t)k * nrep + l) * 3 + x] - hp[link][x]) > 2e-4) {    // (v_sin / v_cos: ~1e-6 per joint)
                    if (mismatch) *mismatch = 1;
                    msg = "generated chain code does not reproduce the forward kinematics of the chain of sgpmp_set_fk (link " + std::to_string(link) + ")";
                    return msg.c_str();
                }
        }
    }
    return nullptr;
}

long long rtc_compile_check_c(const char* struct_src, int field_type, char* err, size_t err_len) {
    std::string e;
    const long long r = rtc_compile_check(struct_src, field_type, e);
    if (err && err_len) { std::strncpy(err, e.c_str(), err_len - 1); err[err_len - 1] = 0; }
    return r;
}
