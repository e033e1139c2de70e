// Internal declarations shared by the HIP translation units of libsgpmp.so (gfx950 only).
#pragma once
#ifndef __HIPCC_RTC__                        // (hiprtc brings the runtime header and the fixed-width types itself)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "../../include/sgpmp.h"

#define SGPMP_TILE 16                       // state blocks padded to one 16x16 MFMA tile
#define SGPMP_MAX_D (2 * SGPMP_MAX_DOF)
#define SGPMP_MAX_LINKS (SGPMP_MAX_JOINTS + 1)
#define SGPMP_MAX_POINTS 32                 // links + interpolated points per field term
#define SGPMP_WAVE 64

// ---------------------------------------------------------------------------------- prior factor
// Device-resident result of K1 for one prior (INIT or SAMPLE).
struct PriorDev {
    double* blocks;    // [4][d][d]  D0, D, Dlast, E
    double* G;         // [T][d][d]
    double* H;         // [T][d][d]
    double* iso64;     // [T][8]  g11 g21 g22 h11 h12 h21 h22 0   (valid when isotropic)
    float* iso32;      // [T][8]
    float* iso32p;     // [T][8]  g11 g21 | h11 h21 | h12 h22 | g22 0: the same numbers in the pair order of rng.h scan_step2 (fused_step_kernel)
    float* slabpre;    // [5][T][4]  isotropic priors, n = 2, 3: prefix products H_t .. H_{start} of the scan's 2 x 2 propagators from the
                       //            start of t's segment, built by the host in fp64 (table 2: the in-chunk segments of fused_planar.inc;
                       //            tables 3, 4: segments of 8 and of 16 waypoints, fused_planar_seg.inc; 0, 1: unused since round 5)
    double* scan64;    // [T][7][4]  fp64 contexts, isotropic priors: the Kogge-Stone tables of fused_step_f64_kernel (cost_sweep_kernel.inc:
                       //            GenArgs64) -- A_t^(r) = H_t .. H_{t - 2^r + 1}, r = 0 .. 5 (zero where t - 2^r lies before t's pass of 64
                       //            waypoints), and C_t = H_t .. H_{start of the pass}; built by the host in fp64 (api.hip: upload_scan64)
    double* Qinv;      // [d][d]   one-step GP precision of this prior
    float* G32;        // [T][d][d] fp32 copies for the dense sampler
    float* H32;
    int* status;       // 0 ok, 1 not PD / non-finite
    double ks, kg;     // 1/sigma_start^2, 1/sigma_goal^2 (kg < 0: not goal-directed)
    double dt;
    int isotropic;
    int valid;
    // per-mode precisions (MultiMPPrior.set_Sigma_invs): n_factor_modes > 0 -> G/H/G32/H32 are
    // [modes][T][d][d] and Dm/Em hold the given blocks; 0 -> one shared closed-form factor
    int n_factor_modes;
    double* Dm;        // [modes][T][d][d]
    double* Em;        // [modes][T-1][d][d]
};

// ---------------------------------------------------------------------------------- cost program
struct CostTerm {
    int kind;
    int flags;
    double K;             // term weight: GP 1/sigma_gp^2, others 1/sigma^2
    double K2;            // GP: 1/sigma_start^2 ; SELF: -1/(2 margin^2)
    double dt;
    double c11, c12, c22; // GP: 12/dt^3, -6/dt^2, 4/dt
    const void* dev_data; // GP: start [d]; GOAL_PRIOR: goals [G,d]; GRID: grid [dim0,dim1]  (ctx dtype)
    int dim0, dim1;
    long long rows_per_goal;   // GOAL_PRIOR: nppg * S
    double inv_cell, off_x, off_y;
    double selfc;         // SELF: q-independent part of the LxL sum (diagonal, coincident and rigid pairs)
    int n_points;         // links + interpolated points
    int n_interp, interp_lo, interp_hi;
    double alpha[SGPMP_MAX_INTERP];
    double target[16];    // EE_GOAL: target end-effector frame
    double w_pos, w_rot;  // EE_GOAL
};

struct CostProgram {
    int n_terms;
    int needs_fk;         // has link-position fields (SPHERES / SELF) evaluated inside the sweep
    int n_ee;             // EE_GOAL terms (evaluated by ee_goal_kernel after the sweep)
    CostTerm terms[SGPMP_MAX_TERMS];
};

struct JointDev {
    double R[9];          // fixed rotation of the joint origin (RPY)
    double t[3];          // fixed translation
    int revolute;
    int qidx;
};

// Result of the host-side chain analysis (api.hip: analyse_chain).  Link frames that always
// coincide are merged into one representative point with a multiplicity, and link pairs whose
// distance does not depend on q are folded into a constant -- both exact rewrites of the sums in
// fields.py:79,86,124 that the register-resident FK path of the cost sweep exploits.
struct FkPlan {
    int fast;                                   // chain is revolute-first: register path usable
    int codegen_id;                             // 0 none; 1 = ChainCode_panda (chain_code_generated.h); 2 = compiled at run time (ChainDev::rtc)
    int n_rep;                                  // distinct link positions
    float mult[SGPMP_MAX_LINKS];                // multiplicity of link l if representative, else 0
    float wpair[SGPMP_MAX_LINKS * SGPMP_MAX_LINKS];   // [i*ML+j], i>j: 2 m_i m_j if q-dependent, else 0
    int n_cpairs;                               // q-independent representative pairs (host use)
    double cpair_w[SGPMP_MAX_LINKS * SGPMP_MAX_LINKS / 2];
    double cpair_d2[SGPMP_MAX_LINKS * SGPMP_MAX_LINKS / 2];
    double diag;                                // sum_l m_l^2  (i == j and coincident terms)
};

struct ChainDev {
    int n_joints;
    int n_links;          // n_joints + 1
    JointDev j[SGPMP_MAX_JOINTS];
    float Rf[SGPMP_MAX_JOINTS][9];              // fp32 copies of the joint constants
    float tf[SGPMP_MAX_JOINTS][3];
    FkPlan plan;
    void* rtc;                                  // HOST: RtcChain* of this chain's run-time compiled kernels (chain_rtc.hip), or null
};

// ---------------------------------------------------------------------------------- debug toggles
// Development switches (A/B of kernel variants, tests of the fallback paths).  Read from the
// SGPMP_* environment variables ONCE, when a context is created; `sgpmp_set_option` changes them on
// a live context.  Nothing on the launch path calls getenv().
struct SgpmpToggles {
    int force_generic_fk;     // SGPMP_FORCE_GENERIC_FK     link positions through LDS, any chain
    int no_flat_program;      // SGPMP_NO_FLAT_PROGRAM      interpret the term list instead of named fields
    int no_chain_codegen;     // SGPMP_NO_CHAIN_CODEGEN     run-time chain constants instead of generated code
    int no_dual_sweep;        // SGPMP_NO_DUAL_SWEEP        one trajectory per wave
    int k3_no_one;            // SGPMP_K3_NO_ONE            no single-pass specialisation
    int k3_no_lds_prefetch;   // SGPMP_K3_NO_LDS_PREFETCH   register-held prefetch
    int no_small_sampler;     // SGPMP_NO_SMALL_SAMPLER     standard sampler for tiny launches
    int no_fused_step;        // SGPMP_NO_FUSED_STEP        K2 and K3 as separate kernels inside sgpmp_step
    int no_chunked_sweep;     // SGPMP_NO_CHUNKED_SWEEP     64-lane-pass two-trajectory sweeps instead of the chunked one
    int no_step_pipeline;     // SGPMP_NO_STEP_PIPELINE     sgpmp_pipeline_begin .. _end run their steps as one chain
    int gpmp_cholesky;        // SGPMP_GPMP_CHOLESKY        GPMP solve by round 3's block Cholesky through LDS instead of the register-resident block-Thomas kernel
    int no_dense_partials;    // SGPMP_NO_DENSE_PARTIALS    update_kernel re-reads all rows with weight even when the weights are spread (round 3)
    int comm_packet_event;    // SGPMP_COMM_PACKET_EVENT    statistics all-reduce chained by the update kernel's own stop event (hipExtLaunchKernelGGL) instead of a plain event record behind it: +16 us instead of +9 us per iteration at one rank on this round's boxes (round 2's boxes had it the other way round)
    int no_planar_tail;       // SGPMP_NO_PLANAR_TAIL       store-free steps of S = 64 planar problems: update_kernel (+ regeneration, if planar_store_free) instead of the update inside fused_planar_seg_kernel
    int planar_store_free;    // SGPMP_PLANAR_STORE_FREE    store-free steps (SGPMP_STEP_NO_SAMPLES) also for fused_planar_seg_kernel: measured SLOWER at config 2 (the launch saves 3.8 us, the update's regeneration costs 5.2: 42.4 k -> 40.0 k it/s, profiles/r05), hence opt-in
    int no_planar_seg;        // SGPMP_NO_PLANAR_SEG        planar one-launch step as fused_planar_kernel (8 samples per wave through an LDS tile) even where fused_planar_seg_kernel (lane = sample, wave = time segment) applies
    int no_ee_fold;           // SGPMP_NO_EE_FOLD           the step's end-effector goal term by a launch of ee_goal_kernel in front of update_kernel (rounds 1-4) instead of inside it
    int f64_fields_f32;       // SGPMP_F64_FIELDS_F32       fp64 steps (fused_step_f64_kernel) evaluate the LINK fields -- forward kinematics, self-distance and sphere fields -- on the packed-fp32 code of the fp32 launches, from the fp64 waypoint rounded to fp32; samples, means, GP / goal-prior / importance-sampling terms stay fp64.  Opt-in: the collision part of a cost then carries fp32's ~1e-6 relative error (~1e-9 of a total cost at the reference's hyper-parameters)
    int no_persist_planar;    // SGPMP_NO_PERSIST_PLANAR    sgpmp_optimize runs the store-free iterations of a planar S = 64 problem as one launch each (round 5) instead of ONE launch for all of them (fused_planar_seg.inc: PERSIST)
    int no_small_step;        // SGPMP_NO_SMALL_STEP        small steps through fused_step_kernel (one wave per item) instead of fused_step_small_kernel (one workgroup per item)
    long long persist_max_iters;  // SGPMP_PERSIST_MAX_ITERS    iterations ONE launch of fused_planar_seg_kernel<.., PERSIST> runs at most (0: 2048 -- ~25 ms at BASELINE configs[1]: a launch must stay far below the driver's hang detection); longer calls take several such launches
    long long small_step_items;   // SGPMP_SMALL_STEP_ITEMS     items (groups of 8 samples) up to which a step counts as small (0: default 512 -- two workgroups per CU -- for shapes on the launch's 8 x 16 grid, 256 for the others)
    long long store_free_min_bytes;   // SGPMP_STORE_FREE_MIN_BYTES  a store-free step that REGENERATES rows in update_kernel is taken when one waypoint of all the step's samples (P S 2n floats) has at least this many bytes (0: the measured break-even, SGPMP_STORE_FREE_BREAK_EVEN; 1: always)
    long long pipe_split;     // SGPMP_PIPE_SPLIT           first chain's share of the particles in 16ths (0 = default 8)
    long long k3_blocks;      // SGPMP_K3_BLOCKS            workgroup cap of the dual sweep (0: default)
};

#ifndef __HIPCC_RTC__     // host-side declarations: not part of the run-time (hiprtc) translation unit of chain_rtc.hip
// ---------------------------------------------------------------------------------- collectives (comm.hip)
// RCCL communicator of one context; every function returns NULL on success or a static error string.
struct SgpmpComm;
const char* comm_unique_id(unsigned char* out128);
const char* comm_create(const unsigned char* id128, int world, int rank, SgpmpComm** out);
void comm_destroy(SgpmpComm* c);
int comm_rank(const SgpmpComm* c);
int comm_world(const SgpmpComm* c);
const char* comm_library_name();
int comm_test_hooks();
const char* comm_info(const SgpmpComm* c, int* world, int* rank, int* version);   // asked of RCCL itself
const char* comm_allreduce_stats(SgpmpComm* c, double* stats, hipStream_t stream);
const char* comm_allreduce_f64(SgpmpComm* c, double* buf, size_t count, hipStream_t stream);
hipStream_t comm_side_stream(SgpmpComm* c);
const char* comm_mark_reduced(SgpmpComm* c, double* buf);
const char* comm_step_begin(SgpmpComm* c, hipStream_t stream, double** slot, hipEvent_t* k4_done);
const char* comm_step_begin2(SgpmpComm* c, hipStream_t s0, hipStream_t s1, double** slot0, double** slot1,
                             hipEvent_t* done0, hipEvent_t* done1);
const char* comm_step_end(SgpmpComm* c, double* stats, bool two_halves);
const char* comm_stats_wait(SgpmpComm* c, double* stats, hipStream_t stream);
const char* comm_allgather(SgpmpComm* c, const void* send, void* recv, size_t bytes, hipStream_t stream);

// ---------------------------------------------------------------------------------- run-time chain kernels (chain_rtc.hip)
struct RtcChain;
const char* rtc_chain_get(const char* struct_src, int n_dof, RtcChain** out);       // null on success, else the reason
hipFunction_t rtc_kernel(RtcChain* c, int ft, bool sweep, bool rag = false, bool small = false); // compiled on first use; null: unavailable (rag: the fused launch for S, T off its 8 x 16 grid)
hipError_t rtc_launch(hipFunction_t f, unsigned blocks, unsigned dyn_lds, hipStream_t stream, void** args, hipEvent_t done);
const char* rtc_verify(RtcChain* c, const ChainDev& chain, int field_type_hint, int* mismatch = nullptr);   // generated code == this chain?  (*mismatch: 1 = no, 0 = kernels unavailable)
const char* rtc_error(const RtcChain* c);
void rtc_stats(const RtcChain* c, double* compile_s, int* compiled, int* from_cache);
long long rtc_compile_check_c(const char* struct_src, int field_type, char* err, size_t err_len);   // compile only: bytes, or -1

// ---------------------------------------------------------------------------------- launchers
// (defined in the .hip files; all asynchronous on `stream`)
hipError_t launch_prior_factor(int n, int T, double dt, double ks, double kg, const double* d_qc_inv,
                               int isotropic, PriorDev out, hipStream_t stream);

hipError_t launch_prior_factor_blocks(int n, int T, int n_modes, const double* d_D, const double* d_E,
                                      PriorDev out, hipStream_t stream);
hipError_t launch_prior_quadform(int dtype, int n, int T, long long rows, int n_modes, const void* x,
                                 const void* means, const PriorDev& prior, double* out, hipStream_t stream);

hipError_t launch_sample(int dtype, int n, int T, const PriorDev& prior, uint64_t seed, uint64_t draw,
                         const void* means, int n_modes, int mode_offset, int n_samples,
                         const void* eps, int eps_modes, int eps_mode_offset, void* out,
                         hipStream_t stream, const SgpmpToggles& tg, double* zero_stats = nullptr);

hipError_t launch_noise(int dtype, int n, int T, int n_modes, int mode_offset, int S, uint64_t seed, uint64_t draw, void* out,
                        hipStream_t stream);

hipError_t launch_cost(int dtype, int n, int T, const CostProgram& h_prog, const ChainDev* d_chain,
                       const ChainDev& h_chain, const void* trajs, long long batch,
                       long long batch_offset, const void* spheres, int n_spheres,
                       const void* is_weights, int rows_per_particle, double is_dt, void* costs,
                       double* costs64, hipStream_t stream, const SgpmpToggles& tg, const char** picked);

// What the fused launch and the update behind it agree on, per particle, through the count of rows that carried weight in
// the particle's PREVIOUS update (a device word: stream-ordered, no host round trip, the same in every run):
//  * dense-weight regime (FusedArgs::part): particles above `threshold` get softmax partials from the launch;
//  * store-free steps (FusedArgs::nostore): only particles above `store_threshold` have their rows written.
struct FusedDenseHost {
    float* part;                  // [P][ceil(S / 8)][4 + T d], or null (not an fp32 chain-code step, switched off)
    unsigned* nnz;                // [P] rows with weight in each particle's previous update
    unsigned threshold;           // partials for particles with nnz above it
    double temperature;
    int nostore;                  // the caller does not need this step's samples (SGPMP_STEP_NO_SAMPLES) and the update can regenerate rows
    unsigned store_threshold;     // ... rows are then stored for particles with nnz above it only
    int particles_total;          // particles of the whole step (a pipelined step launches halves): the size a regenerating store-free step is judged on
    int particles_global;         // ... of all ranks (sgpmp_dims::num_particles_global): the size the small-step launch is judged on
    // the update INSIDE fused_planar_seg_kernel (store-free steps, S = 64; fused_planar_seg.inc: seg_update) -- what update_kernel
    // would have been given; tail_done == null: not offered (per-step mean statistics, ...)
    unsigned* tail_done;          // finished-particle counter of this launch (zero between launches)
    double* tail_acc;             // [SGPMP_STAT_SHARDS][4] statistics accumulators (zero between launches)
    double* stats_out;            // the step's statistics buffer or null
    void* weights; void* grad; void* means_prev;   // K4's optional outputs (context dtype)
    double step_size;
    int tail_iters;               // > 1: the launch runs that many store-free iterations itself (fused_planar_seg.inc: PERSIST; sgpmp_optimize)
};
// bytes of one waypoint of all samples of a step above which a regenerating store-free step is faster than a storing one
// (cost_sweep.hip: launch_fused_step has the measurement)
#define SGPMP_STORE_FREE_BREAK_EVEN 2800000LL
// The step's end-effector goal term evaluated by update_kernel itself (update_common.h: EeFold) instead of ee_goal_kernel
struct EeFoldHost {
    const CostTerm* term;         // the SGPMP_COST_EE_GOAL term
    const ChainDev* d_chain;      // DEVICE chain
    void* costs;                  // [P][S] cost output of the step (context dtype) or null
};
// How update_kernel regenerates the rows a store-free step did not write (update_common.h: RegenArgs)
struct RegenHost {
    int recipe;                   // 0: all rows are in memory; 1: fused_step_kernel's rows; 2: fused_planar_seg_kernel's (segments of L)
    int L;
    uint64_t seed, draw;
    int mode_offset;              // global index of the launch's particle 0
    const float* coef;            // recipe 1: PriorDev::iso32p; 2: PriorDev::iso32
    const float* pre;             // recipe 2: the segment table of PriorDev::slabpre
    unsigned store_threshold;
};
int update_regen_rows(int dtype, int n, int T, int S, int recipe);   // rows per round update_kernel can regenerate; 0: not this shape
// K2 + K3 fused (cost_sweep.hip / fused_step.inc): launches only when the step qualifies (*launched)
hipError_t launch_fused_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog,
                             const ChainDev& h_chain, uint64_t seed, uint64_t draw, const void* means, int P,
                             int mode_offset, int S, void* samples, const void* spheres, int n_spheres,
                             const void* isw, double* zero_stats, void* costs, double* costs64,
                             hipStream_t stream, const SgpmpToggles& tg, const char** picked, bool* launched,
                             const FusedDenseHost* dense = nullptr, bool* partials_armed = nullptr, RegenHost* regen = nullptr,
                             bool* tail_ran = nullptr);   // *tail_ran: the launch also updated its particles (no update_kernel behind it)
// the recipe update_kernel would need to regenerate this step's rows (0: the step's launch cannot run store-free)
int fused_step_regen_recipe(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                            int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg, int* seg_len);
bool planar_seg_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                     int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg);
// ... and would a store-free step of it run its update inside the launch (one launch per iteration; what PERSIST extends)?
bool planar_tail_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                      int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg);
// ... and can one launch run SEVERAL such steps (FusedDenseHost::tail_iters > 1)?
bool planar_persist_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                         int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg);
// does the step qualify for the fused launch? (same conditions, no launch)
bool fused_step_eligible(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog,
                         const ChainDev& h_chain, int P, int mode_offset, int S, int n_spheres,
                         const SgpmpToggles& tg);

hipError_t launch_is_weights(int dtype, int n, int T, const PriorDev& prior, const void* means,
                             int n_particles, double temperature, void* out, double* zero_stats,
                             hipStream_t stream);

bool update_ee_fold_fits(int dtype, int n, int T, int S);
hipError_t launch_update(int dtype, int n, int T, int P, int S, const void* costs, int costs_dtype,
                         const void* samples, void* means, double temperature, double step_size,
                         void* weights, void* grad, void* means_prev, double* stats,
                         hipStream_t stream, hipEvent_t done = nullptr, const PriorDev* isw_prior = nullptr,
                         void* isw_next = nullptr, bool* isw_written = nullptr, void* means_copy = nullptr,
                         const float* part = nullptr, unsigned* nnz = nullptr, unsigned nnz_threshold = 0,
                         const RegenHost* regen = nullptr, const EeFoldHost* ee = nullptr);

hipError_t launch_stats_add(double* dst, const double* src, hipStream_t stream);
hipError_t launch_mode_stats(int dtype, int n, int T, int P, long long p_offset, int nppg, int G, const void* means,
                             double* out, hipStream_t stream);
hipError_t launch_ee_goal(int dtype, int n, int T, const CostTerm& term, const ChainDev* d_chain,
                          const void* trajs, long long batch, void* costs, double* costs64,
                          hipStream_t stream);
hipError_t launch_ee_grad(int dtype, int n, const CostTerm& term, const ChainDev* d_chain, const void* q,
                          long long batch, int traj_T, long long out_stride, long long out_offset, void* value,
                          void* grad, hipStream_t stream);
hipError_t launch_field_grad(int dtype, int n, const CostTerm& term, const ChainDev* d_chain, int n_joints,
                             const void* q, long long batch, int traj_T, const void* spheres, int n_spheres,
                             void* value, void* grad, hipStream_t stream);
// GPMP (gpmp.hip)
#define SGPMP_MAX_T_GPMP 128
struct GpmpFieldK {               // one link-field term of the cost list
    double K;                     // 1 / sigma^2
    const void* val;              // [P, T-1]   field values        (context dtype)
    const void* grad;             // [P, T-1, n] d field / d q       (context dtype)
};

struct GpmpArgs {
    int n, T, P;
    long long p_offset;           // global index of particle 0 (goal lookup)
    double dt, Kgp, c11, c12, c22;   // cost GP term: Q^-1 = Kgp [[c11, c12], [c12, c22]] (x) I_n
    double Ks;                    // start prior weight (0: no start factor)
    const void* start;            // [d] context dtype
    double Kg;                    // goal prior weight (0: none)
    const void* goals;            // [G, d]
    long long rows_per_goal;      // particles per goal
    int n_fields;
    GpmpFieldK f[4];
    double delta;
    const double* diag_sum;       // [T*d] sum over all particles of the FIELD part of diag(A^T K A), or null
    double inv_particles;         // 1 / (global particle count)
    double step_size;
    double* scratch;              // [P][T][2][256]
    int* status;                  // set to 1 when a pivot is not positive
};

hipError_t launch_gpmp_diag(int dtype, const GpmpArgs& a, double* diag_sum, hipStream_t stream);
hipError_t launch_gpmp_solve(int dtype, const GpmpArgs& a, void* means, void* d_theta, void* costs,
                             hipStream_t stream, bool cholesky = false);
hipError_t launch_link_dist(int dtype, const void* frames, long long batch, int n_links, const void* spheres,
                            int n_other, int mode, double buffer, void* out, hipStream_t stream);
hipError_t launch_fk(int dtype, int n, const ChainDev* d_chain, int n_links, const void* q,
                     long long batch, void* frames, hipStream_t stream);
hipError_t launch_grid_lookup(int dtype, const CostTerm& term, const void* xy, long long batch,
                              void* out, hipStream_t stream);
hipError_t launch_field_eval(int dtype, const CostTerm& term, const void* frames, long long batch,
                             int n_links, const void* spheres, int n_spheres, void* out,
                             hipStream_t stream);
#endif  // !__HIPCC_RTC__
