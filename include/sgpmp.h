/*
 * sgpmp.h -- C ABI of libsgpmp.so: the MI355X (gfx950) StochGPMP inner loop.
 *
 * The reference (anindex/stoch_gpmp) is pure Python and has no FFI of its own; its boundary for this
 * path is the Python API of `stoch_gpmp.planner.StochGPMP` and `stoch_gpmp.costs.*`.  Every entry
 * point below therefore cites the reference Python function whose work it performs; the Python
 * classes in `stoch_gpmp_amd/` keep the reference's names/signatures and forward here via ctypes
 * (the binding a reference maintainer would add is shown in INTEGRATION.md).
 *
 * Conventions
 *   - All tensor memory is owned by the caller (torch tensors used as containers); the library
 *     borrows raw DEVICE pointers for the duration of the stream operation it enqueues.  The
 *     library owns only small factor / descriptor / scratch buffers inside the context.
 *   - `dtype`: SGPMP_F32 or SGPMP_F64 is the element type of every `void*` tensor argument of a
 *     context.  Prior factors are always computed in fp64.
 *   - Layouts are dense row-major: means [P,T,d], samples [P,S,T,d], costs/weights [P,S],
 *     d = 2*n_dof = (positions, velocities)  (reference planner.py:52, 215, 227).
 *   - Every call returns 0 on success or a negative SGPMP_E* code; `sgpmp_last_error()` returns a
 *     thread-local message.  All launches are asynchronous on the caller's `stream`
 *     (a hipStream_t passed as void*; NULL = default stream) except where noted.
 *   - A context is bound to the device current at creation and is not thread-safe.
 */
#ifndef SGPMP_H
#define SGPMP_H

#ifndef __HIPCC_RTC__          /* (the run-time compiler of the chain kernels has the fixed-width types built in) */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* 2: sgpmp_step gained `flags`, sgpmp_set_priors / pipeline_* / comm_* appeared (round 2); 3: round 3 (see git log);
 * 4: sgpmp_comm_library, sgpmp_set_fk_codegen / _info / _compile, sgpmp_dense_particles (round 4);
 * 5: SGPMP_STEP_NO_SAMPLES, sgpmp_row_counts_get / _set, sgpmp_store_free_steps, sgpmp_step honours per-mode sampling
 *    precisions; the options of the retired experiments are gone (round 5);
 * 6: sgpmp_optimize (the K-loop of optimize() behind the ABI), sgpmp_row_counts_clear, sgpmp_noise, sgpmp_multi_iteration_launches;
 *    fp64 contexts draw the fp32 noise stream, widened (round 6).
 * The Python binding refuses any other value at load time. */
#define SGPMP_ABI_VERSION 6

enum { SGPMP_F32 = 0, SGPMP_F64 = 1 };
enum { SGPMP_PRIOR_INIT = 0, SGPMP_PRIOR_SAMPLE = 1 };

enum {
    SGPMP_OK = 0,
    SGPMP_EINVAL = -1,     /* bad argument / unsupported size  (Python: AssertionError/ValueError) */
    SGPMP_ENOTPD = -2,     /* prior precision not positive definite (Python: ValueError, as torch's
                              MultivariateNormal raises in the reference, README.md:35)            */
    SGPMP_EHIP = -3,       /* HIP runtime error                                                     */
    SGPMP_ESTATE = -4      /* call order violated (e.g. step before set_prior)                      */
};

/* Cost-term kinds: one per reference Cost class on the StochGPMP path. */
enum {
    SGPMP_COST_GP = 1,          /* CostGP.eval / CostGPTrajectory.eval   cost_functions.py:128-146,202-215 */
    SGPMP_COST_GOAL_PRIOR = 2,  /* CostGoalPrior.eval                    cost_functions.py:376-388         */
    SGPMP_COST_GRID = 3,        /* CostCollision + ObstacleMap           cost_functions.py:247-261, obst_map.py:164-185 */
    SGPMP_COST_SPHERES = 4,     /* CostCollision + LinkDistanceField     fields.py:63-86                   */
    SGPMP_COST_SELF = 5,        /* CostCollision + LinkSelfDistanceField fields.py:114-124                 */
    SGPMP_COST_EE_GOAL = 6      /* CostGoal + EESE3DistanceField         cost_functions.py:282-321, fields.py:130-153
                                   (SE3_distance itself is third-party and un-vendored: DESIGN.md)      */
};
enum { SGPMP_FIELD_RBF = 0, SGPMP_FIELD_SDF = 1, SGPMP_FIELD_OCCUPANCY = 2 };
#define SGPMP_FLAG_GP_START 1      /* GP term includes the start-state unary factor (CostGP)  */
#define SGPMP_FLAG_SDF_CLAMP 16    /* LinkDistanceField(clamp_sdf=True)                       */
#define SGPMP_FLAG_EE_SQUARE 32    /* EESE3DistanceField(square=True)                         */

#define SGPMP_MAX_TERMS 8
#define SGPMP_MAX_JOINTS 16
#define SGPMP_MAX_DOF 8            /* state blocks are handled as one 16x16 MFMA tile        */
#define SGPMP_MAX_INTERP 8
#define SGPMP_STAT_SHARDS 64       /* statistics buffers are double[SGPMP_STAT_SHARDS][4]; sum over shards */

typedef struct sgpmp_dims {
    int32_t n_dof;                 /* n;  d = 2n                                                  */
    int32_t traj_len;              /* T                                                           */
    int32_t num_particles;         /* particles held by THIS context (local shard)                */
    int32_t particle_offset;       /* global index of local particle 0 (multi-GPU sharding)       */
    int32_t num_particles_global;  /* G * nppg                                                    */
    int32_t num_samples;           /* S                                                           */
    int32_t num_goals;             /* G (1 when not goal-directed)                                */
    int32_t num_particles_per_goal;/* nppg; goal of global particle p is p / nppg                 */
    int32_t dtype;                 /* SGPMP_F32 | SGPMP_F64                                       */
    int32_t reserved;
} sgpmp_dims;

typedef struct sgpmp_cost_desc {
    int32_t kind;                  /* SGPMP_COST_*                                                */
    int32_t flags;                 /* GP: SGPMP_FLAG_GP_START; SPHERES: SGPMP_FIELD_* | SGPMP_FLAG_SDF_CLAMP */
    double sigma;                  /* GP: sigma_gp; GOAL_PRIOR: sigma_goal_prior; collision: sigma_coll */
    double sigma2;                 /* GP: sigma_start; SELF: margin                               */
    double dt;                     /* GP: time step                                               */
    const void* data;              /* GP: HOST double[d] start state; GOAL_PRIOR: HOST double[G*d]
                                      goal states; GRID: DEVICE grid [dim0,dim1] in ctx dtype;
                                      EE_GOAL: HOST double[16] target frame (row-major 4x4), with
                                      p0 = w_pos, p1 = w_rot                                      */
    int32_t dim0, dim1;            /* GOAL_PRIOR: G, nppg*S of the cost (cost_functions.py:379);
                                      GRID: map.shape[0], map.shape[1]                            */
    double p0, p1, p2;             /* GRID: cell_size, c_offset[0], c_offset[1] (obst_map.py:129-140) */
    int32_t num_interpolate;       /* fields.py:32,94                                             */
    int32_t interp_lo, interp_hi;  /* link_interpolate_range                                      */
    int32_t reserved;
    double alpha[SGPMP_MAX_INTERP];/* interpolation weights torch.linspace(0,1,K+2)[1:K+1] (fields.py:69) */
} sgpmp_cost_desc;

/* One joint of a serial URDF chain: H_child = H_parent * Trans(xyz) * RPY(rpy) * Rz(q).
 * The FK callable of the reference (cost_functions.py:39,51-52) is third-party; see DESIGN.md. */
typedef struct sgpmp_joint {
    double rpy[3];
    double xyz[3];
    int32_t revolute;              /* 1: consumes the next joint coordinate; 0: fixed             */
    int32_t reserved;
} sgpmp_joint;

typedef struct sgpmp_ctx sgpmp_ctx;

int sgpmp_abi_version(void);
/* Rounds of the Philox4x32 bijection behind the in-kernel noise (csrc/rng.h, build parameter SGPMP_PHILOX_ROUNDS: 10 =
 * Random123's default, 7 = its documented Crush-resistant minimum): the CPU restatement of the stream
 * (oracle/native_noise.py) must be run with the same count. */
int sgpmp_philox_rounds(void);
const char* sgpmp_last_error(void);

/* ---- context ---------------------------------------------------------------------------------- */
/* StochGPMP.__init__ / reset bookkeeping (planner.py:20-82,181-227). Synchronous. */
int sgpmp_create(const sgpmp_dims* dims, sgpmp_ctx** out);
void sgpmp_destroy(sgpmp_ctx* ctx);

/* Development switches (kernel-variant A/B, tests of the fallback paths).  Every switch has an
 * environment variable SGPMP_<NAME> that is read ONCE, in sgpmp_create; this call changes a switch
 * on a live context.  Names: force_generic_fk, no_flat_program, no_chain_codegen, no_dual_sweep,
 * k3_no_one, k3_no_lds_prefetch, no_small_sampler, no_fused_step, no_chunked_sweep, no_step_pipeline, comm_packet_event,
 * gpmp_cholesky (GPMP solve by round 3's LDS block-Cholesky kernel), no_dense_partials (dense-weight regime: update_kernel re-reads every row with weight, as in round 3),
 * no_planar_seg (planar one-launch step through the LDS tile, fused_planar_kernel, even where the lane-per-sample launch applies),
 * planar_store_free (store-free planar steps of ANY sample count by regenerating rows in update_kernel: bit-identical, measured slower
 * than storing at BASELINE configs[1]; problems with 64 samples per particle run store-free by default, with the update inside the launch),
 * no_planar_tail (those steps with update_kernel behind the launch instead),
 * no_persist_planar (sgpmp_optimize runs the store-free iterations of such a problem one launch each, as round 5 did, instead of
 * all of them in ONE launch: bit-identical, 1.4 x slower per iteration at BASELINE configs[1]),
 * persist_max_iters (count: iterations one such launch runs at most; 0: 2048 -- a longer call takes several launches),
 * no_ee_fold (the step's end-effector goal term by a launch of ee_goal_kernel in front of update_kernel, as in rounds 1-4, instead of
 * inside update_kernel: same numbers, one launch more),
 * no_small_step (steps of up to small_step_items items -- groups of 8 samples; default 512, two per CU, 256 for shapes off the
 * launch's 8 x 16 grid -- go out as
 * fused_step_small_kernel, one WORKGROUP per item with its four waves on the item's time chunks side by side, bit-identical to
 * the one-wave-per-item launch and about twice as fast where every wave sits alone on its SIMD; 1: always fused_step_kernel),
 * small_step_items (count),
 * store_free_min_bytes (a store-free step that regenerates rows in update_kernel is taken when one waypoint of all the step's
 * samples -- P S 2n floats -- has at least this many bytes; 0: the measured break-even of 2.8 MB, below which it ran 4 .. 20 %
 * slower than a storing step; 1: always),
 * f64_fields_f32 (opt-in, fp64 contexts: the one-launch step evaluates the LINK fields -- forward kinematics, self-distance and
 * sphere fields -- on the fp32 launches' packed code from the fp64 waypoint rounded to fp32; noise, recurrence, samples, means and
 * the GP / goal-prior / importance-sampling terms stay fp64.  The collision part of a cost then carries fp32's ~1e-6 relative
 * error -- ~1e-9 of a total cost at the reference's hyper-parameters -- and the step runs 1.7 x as fast: 0.46 against 0.79 ms at BASELINE configs[2]'s shape; the occupancy COUNT is then taken on
 * fp32 link positions too: a point within fp32 rounding of a sphere's surface may count differently, one quantum 1 / sigma_coll^2),
 * pipe_split (1..15) and k3_blocks (count).  (The launches that measured slower -- tail_update, small_step, planar_slabs,
 * wave_groups, fused_pipe -- were removed in round 5; DESIGN.md 8 keeps their numbers and the commit that last held them.)
 * No reference counterpart. */
int sgpmp_set_option(sgpmp_ctx* ctx, const char* name, long long value);
/* Name of the cost-sweep kernel the dispatcher chose at the last sgpmp_cost_eval / sgpmp_step
 * (static string; "" before the first launch).  For bench.py's roofline label and the tests that
 * must prove which kernel they exercised. */
const char* sgpmp_last_cost_kernel(sgpmp_ctx* ctx);

/* K1. GP-prior precision blocks + reverse block-Cholesky, fp64, one launch.
 * Replaces GPFactor.calc_phi/calc_Q_inv (gp_factor.py:36-52), UnaryFactor.K (unary_factor.py:19),
 * MultiMPPrior.get_const_vel_covariance (mp_priors_multi.py:170-202) and the precision->scale_tril
 * conversion torch performs on every MultiMPPrior.update_dist (mp_priors_multi.py:100-110).
 * sigma_goal < 0: not goal-directed.  qc_inv: HOST double[n*n] or NULL (= I/sigma_gp^2).
 * Synchronous (reads back the positive-definiteness flag): returns SGPMP_ENOTPD on failure. */
int sgpmp_set_prior(sgpmp_ctx* ctx, int which, double dt, double sigma_start, double sigma_gp,
                    double sigma_goal, const double* qc_inv, void* stream);
/* The two priors of StochGPMP.reset (planner.py:204-226: [0] initialisation, [1] sampling; isotropic Q_c) in one
 * call: K1 is a single wave for ~1.2 ms, so the two factorisations run concurrently (the second on a stream of the
 * context) and the call synchronises once.  sigma_goal[w] < 0: not goal-directed.  Same results and errors as two
 * sgpmp_set_prior calls. */
int sgpmp_set_priors(sgpmp_ctx* ctx, double dt, const double* sigma_start, const double* sigma_gp,
                     const double* sigma_goal, void* stream);

/* MultiMPPrior.set_Sigma_invs (mp_priors_multi.py:125-128): one precision matrix PER MODE, given by its
 * blocks -- D HOST double[n_modes][T][d][d] (diagonal blocks), E HOST double[n_modes][T-1][d][d] with
 * E[m][t] = Sigma_m^-1[t+1-block, t-block]; anything outside the block-tridiagonal band must be zero
 * (the caller checks).  K1 factors every mode (one workgroup each, fp64 MFMA); sgpmp_sample then uses
 * mode m's factor for mean m, sgpmp_prior_quadform its precision.  The planner loop: with which = SGPMP_PRIOR_SAMPLE and
 * n_modes = the context's particle count, sgpmp_step samples particle p from factor p (two-launch route: the dense sampler
 * on the matrix cores, then the sweep) while its importance-sampling term -- and sgpmp_is_weights -- keep the shared
 * closed-form precision of the last sgpmp_set_prior, exactly as the reference does after
 * planner._sample_dist.set_Sigma_invs(...) (planner.py:226,233-236: `self.Sigma_inv` is captured at reset).  A later
 * sgpmp_set_prior(s) returns to the shared factor.
 * Synchronous; SGPMP_ENOTPD when a matrix is not positive definite. */
int sgpmp_set_prior_blocks(sgpmp_ctx* ctx, int which, int n_modes, const double* D, const double* E,
                           void* stream);

/* MultiMPPrior.log_prob (mp_priors_multi.py:209-210 -> torch MultivariateNormal.log_prob): the quadratic
 * forms q_r = (x_r - mu_m)^T Sigma_m^-1 (x_r - mu_m), m = r % n_modes, for x DEVICE [rows, T*d] and
 * means DEVICE [n_modes, T*d] in ctx dtype -> out DEVICE double[rows]  (log_prob = -q/2 - M/2 log 2pi
 * + 1/2 log det Sigma_m^-1; the log-determinant is a by-product of K1's factor, see sgpmp_get_prior). */
int sgpmp_prior_quadform(sgpmp_ctx* ctx, int which, const void* x, int64_t rows, const void* means,
                         int n_modes, double* out, void* stream);

/* Test/inspection hook: copy K1's outputs to HOST buffers (any may be NULL). Synchronous.
 *   blocks  double[4*d*d]  D_0, D_interior, D_last, E = Sigma_inv[(i+1)-block, i-block]
 *   G, H    double[T*d*d]  scan coefficients:  y_t = G_t eps_t + H_t y_{t-1}  (= scale_tril @ eps);
 *           after sgpmp_set_prior_blocks: double[n_modes*T*d*d], and `blocks` are those of the shared prior of the
 *           last sgpmp_set_prior (what the step's importance-sampling term keeps using).
 *   G_t = B_t^-1 is lower triangular and Sigma^-1 = L_inv^T L_inv with diag blocks B_t, so
 *   log det Sigma^-1 = -2 sum_t sum_i log G_t[i][i]. */
int sgpmp_get_prior(sgpmp_ctx* ctx, int which, double* blocks, double* G, double* H);

/* CostComposite(cost_list, FK) compiled to a device cost program (cost_functions.py:34-58). */
int sgpmp_set_costs(sgpmp_ctx* ctx, const sgpmp_cost_desc* descs, int n_desc);
int sgpmp_set_fk(sgpmp_ctx* ctx, const sgpmp_joint* chain, int n_joints);
/* Any serial chain on the fast launches (the reference takes any FK callable, cost_functions.py:39,51-52).  The fused
 * sampler + sweep launch and the chunked sweep need the chain as straight-line, constant-folded code; the library is built
 * with that code for the Panda only.  For another chain the host generates `struct ChainCode_rt { ... };` with
 * stoch_gpmp_amd/csrc/gen/chain_codegen.py (gen_chain("rt", chain): the same generator that wrote the built-in code) and
 * hands the text over AFTER sgpmp_set_fk; the library compiles the same kernel sources around it with hiprtc for gfx950
 * (dlopen; lazily per sphere-field type; cached in memory and under $SGPMP_RTC_CACHE | ~/.cache/sgpmp), verifies the code
 * against the chain of sgpmp_set_fk (forward kinematics at random joint vectors, link and pair tables) and dispatches the chain
 * like the built-in one.  fp32 contexts, n_dof <= 7, revolute joints first.  On failure -- SGPMP_ESTATE: hiprtc or the
 * kernel sources (csrc/ next to the library, or $SGPMP_CSRC_DIR) unavailable; SGPMP_EINVAL: the code is not this chain's --
 * the chain keeps the slower run-time-constant kernels; nothing else changes.  sgpmp_set_fk resets it. */
int sgpmp_set_fk_codegen(sgpmp_ctx* ctx, const char* chain_struct_source);
/* *codegen_id: 0 run-time constants, 1 built-in code (Panda), 2 compiled at run time; seconds spent in hiprtc and the code
 * objects compiled / found in the disk cache for the current chain.  Any pointer may be NULL. */
int sgpmp_fk_codegen_info(sgpmp_ctx* ctx, int* codegen_id, double* compile_s, int* compiled, int* from_cache);
/* Compile-only check of chain code (hiprtc; no device, no context, nothing cached or loaded): SGPMP_OK and the size of the
 * gfx950 code object, or SGPMP_ESTATE with the compiler's log in sgpmp_last_error(). */
int sgpmp_fk_codegen_compile(const char* chain_struct_source, int field_type, int64_t* code_bytes);

/* ---- kernels ---------------------------------------------------------------------------------- */
/* K2. MultiMPPrior.sample (mp_priors_multi.py:204-207): out[m,s,:] = means[m,:] + scale_tril @ eps.
 * `n_modes` mean trajectories [n_modes,T,d], `n_samples` draws each -> out [n_modes,n_samples,T,d].
 * eps == NULL: counter-based Philox noise keyed by (seed, draw, mode_offset+m, s, element).
 * eps != NULL (parity mode): eps is laid out like torch's randn(n_samples, eps_modes, T*d)
 * (multivariate_normal.py:250-253) and mode m reads column eps_mode_offset + m. */
int sgpmp_sample(sgpmp_ctx* ctx, int which, uint64_t seed, uint64_t draw, const void* means,
                 int n_modes, int mode_offset, int n_samples, const void* eps, int eps_modes,
                 int eps_mode_offset, void* out, void* stream);

/* The noise itself: out[s][m][:] = the eps of sample s of mode m (global particle mode_offset + m) at draw `draw`, in the layout
 * of torch's randn(n_samples, n_modes, T*d) (multivariate_normal.py:250-253; element t*d + k: position noise of dof k at
 * waypoint t, t*d + n + k: its velocity noise) and the context's dtype -- exactly the values sgpmp_sample / sgpmp_step /
 * sgpmp_optimize draw for eps == NULL.  One counter-based stream serves fp32 and fp64 contexts (Philox4x32 + the fp32
 * Box-Muller, widened for fp64: csrc/rng.h), keyed on (seed, draw, global particle, sample, waypoint pair, dof).  What it is
 * for: feeding the SAME eps to the reference's algorithm (oracle/ref_equiv.py: TrajPrior.sample(eps=...)) -- the hardware's
 * log2 / sin / cos are approximations that a CPU restatement (oracle/native_noise.py) reproduces to an ulp of fp32, not bit
 * for bit.  The reference draws from torch's sequential generator and has no counterpart. */
int sgpmp_noise(sgpmp_ctx* ctx, uint64_t seed, uint64_t draw, int n_modes, int mode_offset, int n_samples, void* out,
                void* stream);

/* K3. CostComposite.eval (cost_functions.py:47-58): trajs [B,T,d] -> costs [B].
 * Row b of the batch is global row batch_offset + b (for CostGoalPrior's goal lookup).
 * spheres: DEVICE [n_spheres,4] (cx,cy,cz,r) in ctx dtype = observation['obstacle_spheres'].
 * is_weights: NULL, or DEVICE [B/rows_per_particle, T+1, d] importance-sampling weights (K5) whose
 * inner product with (x_0, e_0..e_{T-2}, x_{T-1}) is added (planner.py:233-236).
 * costs (ctx dtype) and costs64 (double) may each be NULL. */
int sgpmp_cost_eval(sgpmp_ctx* ctx, const void* trajs, int64_t batch, int64_t batch_offset,
                    const void* spheres, int n_spheres, const void* is_weights,
                    int rows_per_particle, void* costs, double* costs64, void* stream);

/* K5. Importance-sampling weights  a = temperature * blkdiag(K_s, Q^-1.., K_g) A mu  per particle
 * (the factored form of Sigma_inv @ mu, planner.py:226,235-236). out: DEVICE [P,T+1,d] ctx dtype. */
int sgpmp_is_weights(sgpmp_ctx* ctx, const void* means, int n_particles, double temperature,
                     void* out, void* stream);

/* K4. StochGPMP._update_distribution (planner.py:263-275): softmax(-costs/temperature) over S,
 * grad = sum_s w (x - mu), means += step_size * grad.  costs_dtype: SGPMP_F64 or the ctx dtype.
 * weights [P,S] and grad [P,T,d] in ctx dtype (may be NULL). means_prev [P,T,d] (may be NULL)
 * receives the PRE-update means, which is what optimize() returns (planner.py:252-253).
 * stats: DEVICE double[SGPMP_STAT_SHARDS][4] or NULL; each workgroup adds (sum of costs, min cost,
 * 1, -) of its particle to one shard -- the consumer sums the shards (sharding avoids serialising a
 * thousand atomics on one address). */
int sgpmp_update(sgpmp_ctx* ctx, const void* costs, int costs_dtype, const void* samples,
                 void* means, double temperature, double step_size, void* weights, void* grad,
                 void* means_prev, double* stats, void* stream);
/* Diagnostic (synchronous): the number of particles whose last update (inside sgpmp_step) spread its weight over more than
 * S / 4 samples.  For those the next step's fused launch leaves softmax partials of every 8 rows and the update adds S / 8
 * partials instead of re-reading the rows (planner.py:263-275 is a softmax; with the reference's hyper-parameters it is
 * one-hot and the count is 0).  -1: no fp32 step has run yet.  Which particles get partials is decided on the device, per
 * particle, from the row count its previous update left (stream-ordered: the same in every run; round 4 armed the partials
 * from a host word read without synchronisation).  *armed_steps (may be NULL): steps launched with the partials buffer so far. */
int sgpmp_dense_particles(sgpmp_ctx* ctx, int64_t* count, int64_t* armed_steps);
/* Those per-particle row counts are state of a run (they decide per particle how the NEXT update forms its sum -- gathered /
 * regenerated rows, or partials: equal to 1e-6, not bit for bit): StochGPMP.state_dict carries them, reset() clears them.
 * Synchronous.  get: out HOST uint32[P] (zeros before the first fp32 step); set: in HOST uint32[P], or NULL = all zero. */
int sgpmp_row_counts_get(sgpmp_ctx* ctx, uint32_t* out);
int sgpmp_row_counts_set(sgpmp_ctx* ctx, const uint32_t* in);
/* reset()'s clear (planner.py:181-227 starts a fresh problem): all zero, asynchronous on `stream`, no host synchronisation. */
int sgpmp_row_counts_clear(sgpmp_ctx* ctx, void* stream);
/* steps of this context that ran store-free so far (SGPMP_STEP_NO_SAMPLES honoured; tests and bench.py report it) */
long long sgpmp_store_free_steps(sgpmp_ctx* ctx);
/* launches of this context that ran SEVERAL iterations each (sgpmp_optimize, planar problems with 64 samples per particle) */
long long sgpmp_multi_iteration_launches(sgpmp_ctx* ctx);

/* One body of the loop at planner.py:289-299 for the context's particle shard:
 * K5 -> K2 -> K3 -> K4 on `stream` (K2 + K3 as ONE launch when the configuration qualifies, see
 * csrc/fused_step.inc; K5 folded into the previous step's K4 under SGPMP_STEP_MEANS_KEPT).  samples [P,S,T,d] is written (state_samples of the iteration);
 * costs [P,S] ctx dtype may be NULL. means updated in place. stats (DEVICE
 * double[SGPMP_STAT_SHARDS][4] or NULL) is zeroed at the start of the step and holds this step's
 * sharded sums afterwards; with a communicator attached (sgpmp_comm_init) the step also enqueues
 * their all-reduce on the side stream, so `stats` then holds the sums over ALL ranks once
 * sgpmp_stats_wait has been honoured. */
int sgpmp_step(sgpmp_ctx* ctx, uint64_t seed, uint64_t draw, const void* eps, int eps_modes,
               int eps_mode_offset, void* means, void* samples, void* costs, void* weights,
               void* grad, void* means_prev, const void* spheres, int n_spheres, double temperature,
               double step_size, double* stats, int flags, void* stream);
/* flags of sgpmp_step: */
#define SGPMP_STEP_MEANS_KEPT 1    /* the caller guarantees that `means` still holds exactly what this context's
                                      previous sgpmp_step left there: the importance-sampling weights that step's
                                      update kernel prepared for them are then used, and the K5 launch is skipped
                                      (on every path: the fused launch, or else the sampler, zeroes `stats`).
                                      Without the flag (or after anything else wrote the means) K5 runs. */
#define SGPMP_STEP_NO_SAMPLES 2    /* the caller will not read `samples` of THIS step (iterations 1 .. K - 1 of
                                      optimize(opt_iters = K): the reference returns the last iteration's tensors only,
                                      planner.py:289-317).  Where the step runs as one fused launch (fused_step_kernel,
                                      fused_planar_seg_kernel) it then does not write them -- 470 MB per launch at BASELINE
                                      configs[2] -- and update_kernel REGENERATES the rows that carry weight from their noise
                                      keys (one row per particle with the reference's one-hot weights), bit for bit what the
                                      launch would have stored: means, costs, weights, gradient come out identical to a
                                      storing step's.  (fused_planar_seg_kernel with 64 samples per particle: the workgroup
                                      holds all samples of its particle in registers and runs the update itself -- one
                                      launch per iteration, bit-identical to update_kernel.)  Rows of particles whose previous update spread its weight over more
                                      than 4 samples are still written (`samples` must be a valid buffer).  A permission,
                                      not a demand: steps on other paths store as always, and so do steps too small for the
                                      regeneration to pay (option store_free_min_bytes: below 2.8 MB of samples per waypoint a
                                      regenerating step measured 4 .. 20 % slower than a storing one).  After such a step
                                      `samples` holds rows of earlier steps. */

/* The loop of planner.py:289-299 itself (`for opt_step in range(opt_iters)`), when its iterations follow each
 * other without the caller looking at the buffers in between: bracket the sgpmp_step calls of one optimize() with
 *     sgpmp_pipeline_begin(ctx, stream);  K x sgpmp_step(..., stream);  sgpmp_pipeline_end(ctx, stream);
 * Particles are independent (every reduction of planner.py:263-275 runs over the samples of ONE particle), so
 * inside the bracket the context runs each step as TWO launch sequences, one per half of its particle range, on two
 * streams of its own: while one half's update kernel (latency-bound) runs, the other half's sampler + sweep launch
 * keeps the chip busy, across iterations.  Every buffer passed to the steps belongs to the context until
 * sgpmp_pipeline_end has returned, which makes `stream` wait for both sequences (stream-ordered like any other
 * call: no host synchronisation); results are those of unbracketed steps, bit for bit.  Steps that do not qualify
 * (not the fused launch, parity-mode noise, a communicator attached, fewer than 2 x 8192 trajectories) run as
 * usual inside the bracket.  Switch: SGPMP_NO_STEP_PIPELINE / "no_step_pipeline". */
int sgpmp_pipeline_begin(sgpmp_ctx* ctx, void* stream);
int sgpmp_pipeline_end(sgpmp_ctx* ctx, void* stream);
/* StochGPMP.optimize's loop (planner.py:289-299) as ONE call: opt_iters x sgpmp_step with in-kernel noise on `stream`,
 * draw counters draw0 .. draw0 + opt_iters - 1; step k accumulates its statistics in stats_pair[(first_slot + k) & 1]
 * (DEVICE double[2][SGPMP_STAT_SHARDS][4], or NULL); steps 0 .. K - 2 leave their pre-update means in means_prev_scratch,
 * the last one in means_prev_last (either may be NULL) -- the tensor optimize() hands out (planner.py:252-253).
 * flags: SGPMP_STEP_MEANS_KEPT speaks for the FIRST step (the later ones follow this call's own steps and always carry it);
 * SGPMP_OPT_STORE_FREE gives steps 0 .. K - 2 SGPMP_STEP_NO_SAMPLES (the reference returns the last iteration's tensors
 * only); SGPMP_OPT_PIPELINE brackets the call with sgpmp_pipeline_begin / _end when opt_iters >= 2.  Results are those of
 * the same sgpmp_step calls made one by one, bit for bit; the first failing step's status is returned (the bracket is
 * closed first).  No host synchronisation.
 * Under SGPMP_OPT_STORE_FREE, opt_iters >= 3, a planar problem with 64 samples per particle (n = 2, time segments of 8: the
 * store-free step carries its update inside the launch) runs steps 0 .. K - 2 as ONE launch (csrc/fused_planar_seg.inc:
 * PERSIST; not with a communicator, the step profiler or per-step mean statistics).  Every output is the same, bit for bit,
 * with one exception: the statistics of steps 0 .. K - 2 are not formed -- only the LAST step's slot of stats_pair,
 * (first_slot + K - 1) & 1, is written by such a call.  Option no_persist_planar. */
#define SGPMP_OPT_PIPELINE 4
#define SGPMP_OPT_STORE_FREE 8
int sgpmp_optimize(sgpmp_ctx* ctx, int opt_iters, uint64_t seed, uint64_t draw0, void* means, void* samples,
                   void* costs, void* weights, void* grad, void* means_prev_scratch, void* means_prev_last,
                   const void* spheres, int n_spheres, double temperature, double step_size,
                   double* stats_pair, int first_slot, int flags, void* stream);
/* Kernels the last sgpmp_step enqueued for its particle range (per chain when it ran as two): 2 = fused sampler + sweep,
 * then update_kernel (the default); 3-4 = separate kernels; 1 = a store-free planar step whose launch also updated its
 * particles (fused_planar_seg.inc: seg_update). */
int sgpmp_last_step_launches(sgpmp_ctx* ctx);
/* how many steps of this context ran as two chains so far (tests and bench.py report it) */
long long sgpmp_pipeline_split_steps(sgpmp_ctx* ctx);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) -------------------------------------------- */
/* The reference is single-process and has no counterpart; these calls carry out SURVEY.md 8(e):
 * particles are sharded by contiguous ranges (sgpmp_dims.particle_offset), the data path needs no
 * exchange, and the per-iteration statistics are summed over ranks.  librccl is loaded with dlopen
 * by the first of these calls; single-GPU users never need it.
 *
 * sgpmp_comm_unique_id: rank 0 obtains the 128-byte RCCL id and hands it to the other ranks by any
 *   host-side channel (the Python host uses torch.distributed's store).
 * sgpmp_comm_init: collective over all `world_size` ranks (ncclCommInitRank); attaches the
 *   communicator to the context.  From then on sgpmp_step all-reduces its `stats` by itself. */
int sgpmp_comm_unique_id(unsigned char* out128);
int sgpmp_comm_init(sgpmp_ctx* ctx, const unsigned char* id128, int world_size, int rank);
int sgpmp_comm_destroy(sgpmp_ctx* ctx);
/* What the attached communicator itself reports (ncclCommCount / ncclCommUserRank / ncclGetVersion), so that a
 * multi-GPU bench line can prove that RCCL saw N ranks: *world = 0 when no communicator is attached.
 * rccl_version: NCCL_VERSION_CODE of the loaded librccl (e.g. 22205), 0 if unavailable.  Any pointer may be NULL. */
int sgpmp_comm_info(sgpmp_ctx* ctx, int* world, int* rank, int* rccl_version);
/* Name of the collective library this process bound at its first sgpmp_comm_unique_id / sgpmp_comm_init ("" before):
 * "librccl.so.1" (or one of its fallback names) in the product library, always.  *test_hooks (may be NULL) = 1 only in
 * tests/fake_rccl/libsgpmp_testhooks.so, the build of comm.hip that honours SGPMP_RCCL_LIB (a stand-in that lets several
 * ranks share one GPU in the tests); the product library never reads that variable.  Static string. */
const char* sgpmp_comm_library(int* test_hooks);
/* Sum stats (DEVICE double[SGPMP_STAT_SHARDS][4], produced on `stream`) over all ranks, in place, on
 * the context's side stream: returns at once and never makes `stream` wait. (planner.py:668-672's
 * statistic over all particles of all GPUs.) */
int sgpmp_allreduce_stats(sgpmp_ctx* ctx, double* stats, void* stream);
/* Make `stream` wait (stream-side, not host-side) for the pending all-reduce of `stats`
 * (NULL: of every statistics buffer) -- call before reading or overwriting it. */
int sgpmp_stats_wait(sgpmp_ctx* ctx, double* stats, void* stream);
/* Per-goal statistics of the particle means -- the "weighted-mean / covariance statistics" of the trajectory
 * distribution's modes (a mode = the particles of one goal, p = g * nppg + k, planner.py:215); what a covariance
 * adaptation through MultiMPPrior.set_Sigma_invs (mp_priors_multi.py:125-128) or a mode summary consumes.
 * out DEVICE double[G][T*d + 1][2]:  [g][m] = (sum, sum of squares) of mu_p[m] over THIS context's particles of goal g,
 * [g][T*d] = (their number, 0).  sgpmp_mode_stats zeroes `out` and accumulates on `stream` (no atomics: bitwise
 * reproducible run to run for a given sharding; across shardings the additions are ordered differently -- equal to rounding).
 * sgpmp_allreduce_f64: sum any buffer of doubles over all ranks, in place, on the context's side stream (like
 * sgpmp_allreduce_stats; honour sgpmp_stats_wait(buf) before reading).
 * sgpmp_set_step_mode_stats(buf): from now on EVERY sgpmp_step produces them by itself, once per iteration: its update
 * kernel leaves a snapshot of the new means, the side stream reduces it per goal into `buf` and all-reduces the sums
 * over the ranks -- nothing is added to the steps' own stream but one event record; sgpmp_mode_stats_wait makes
 * `stream` wait for the newest one.  NULL switches it off.  (Steps then run as one chain and with update_kernel.) */
int sgpmp_mode_stats(sgpmp_ctx* ctx, const void* means, double* out, void* stream);
int sgpmp_allreduce_f64(sgpmp_ctx* ctx, double* buf, int64_t count, void* stream);
int sgpmp_set_step_mode_stats(sgpmp_ctx* ctx, double* buf);
int sgpmp_mode_stats_wait(sgpmp_ctx* ctx, void* stream);
/* All ranks' particle means: local [P_local,T,d] -> all [P_global,T,d] on every rank (ncclAllGather on
 * `stream`; equal shards only). */
int sgpmp_allgather_means(sgpmp_ctx* ctx, const void* local_means, void* all_means, void* stream);

/* ---- standalone field / FK ops (the planner <-> cost seam) ----------------------- */
/* FK callable: q [B,n] -> link frames [B,L,4,4], L = 1 + n_joints (cost_functions.py:51-52). */
int sgpmp_fk(sgpmp_ctx* ctx, const void* q, int64_t batch, void* frames, void* stream);
/* ObstacleMap.compute_cost (obst_map.py:164-185): X [B,2] -> values [B]; uses cost term `term`. */
int sgpmp_grid_lookup(sgpmp_ctx* ctx, int term, const void* xy, int64_t batch, void* out,
                      void* stream);
/* LinkDistanceField / LinkSelfDistanceField.compute_cost on explicit frames [B,L,4,4] -> [B]
 * using cost term `term` (its field type, margin, interpolation). */
int sgpmp_field_eval(sgpmp_ctx* ctx, int term, const void* frames, int64_t batch, int n_links,
                     const void* spheres, int n_spheres, void* out, void* stream);

/* LinkDistanceField.distances / compute_collision / compute_distance (fields.py:40-61; spheres DEVICE [n_spheres,4]:
 * D[l][o] = |p_l - c_o| - r_o) and LinkSelfDistanceField's (fields.py:100-112; spheres NULL: D[i][j] = |p_i - p_j|)
 * on link frames [B,L,4,4] in ctx dtype.  mode 0: out [B,L,n_spheres or L] = D;  mode 1: out [B] = 1 where any
 * D < buffer (self: over link pairs i - j >= 2, torch.tril(.., diagonal=-2)), else 0;  mode 2: out [B] = sum D. */
int sgpmp_link_distances(sgpmp_ctx* ctx, const void* frames, int64_t batch, int n_links, const void* spheres,
                         int n_spheres, int mode, double buffer, void* out, void* stream);

/* FieldFactor.get_error(calc_jacobian=True) (factors/field_factor.py:28-38): value [B] (may be NULL) and
 * gradient d value / d q [B,n] of link-field term `term` at joint configurations q [B,n], with the
 * context's FK chain -- the analytic form of the reference's torch.autograd.grad through FK and the
 * field (the reference's H is MINUS this gradient).  SPHERES with the rbf type; SPHERES with the sdf type
 * (fields.py:79-83: value = max over (link point, sphere) of r - dist, optionally clamped at 0; gradient = that of
 * the arg-max pair, first maximum in (point, sphere) order as torch's max keeps it, zero where the clamp is
 * active -- what autograd returns through sdf.max(-1)[0].max(-1)[0]); SELF; EE_GOAL (end-effector SE(3) distance:
 * position part u . (z_j x (p - o_j)), rotation part z_j . axis; what CostGoal.get_linear_system needs,
 * cost_functions.py:323-337).  The occupancy count has no gradient -> SGPMP_EINVAL.   */
int sgpmp_field_grad(sgpmp_ctx* ctx, int term, const void* q, int64_t batch, const void* spheres,
                     int n_spheres, void* value, void* grad, void* stream);

/* ---- GPMP: the reference's Gauss-Newton planner (planner.py:352-661) ------- */
/* First half of GPMP._step (planner.py:580-581, cost.get_linear_system): evaluates every smooth link
 * field of the cost list and its Jacobian at waypoints 1..T-1 of the particle means [P,T,d] (kept in
 * the context), and -- when `diag_sum` [T*d] is given -- the sum over THIS context's particles of the
 * field part of diag(A^T K A), which the trust-region damping averages over all particles
 * (planner.py:618-622; all-reduce it across ranks before sgpmp_gpmp_solve when particles are sharded).
 * The cost list may hold one CostGP, one CostGoalPrior and up to 4 rbf- or sdf-sphere / self-distance /
 * end-effector-goal fields (CostGoal: one row, on the last waypoint). */
int sgpmp_gpmp_linearize(sgpmp_ctx* ctx, const void* means, const void* spheres, int n_spheres,
                         double* diag_sum, void* stream);
/* Second half (planner.py:583-603): per particle, assemble the block-tridiagonal normal equations
 * (A^T K A + damping) d_theta = A^T K b, solve them by block Cholesky, means += step_size * d_theta.
 * diag_sum NULL: damping delta * I (trust_region=False); else delta * diag(mean_p A^T K A).
 * d_theta [P,T,d] and costs [P] (= b^T K b at the linearisation point, planner.py:642-644) may be NULL.
 * Synchronous; SGPMP_ENOTPD if a pivot is not positive (torch raises from cholesky in the reference). */
int sgpmp_gpmp_solve(sgpmp_ctx* ctx, void* means, const double* diag_sum, double delta, double step_size,
                     void* d_theta, void* costs, void* stream);

/* Kernel timing helper for bench.py: elapsed ms between two events recorded on `stream`
 * (HIP events on the stream the kernels run on). */
int sgpmp_event_create(void** ev);
int sgpmp_event_record(void* ev, void* stream);
int sgpmp_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on stop */
int sgpmp_event_destroy(void* ev);
/* Per-kernel accumulated device time (ms) measured with events inside sgpmp_step when enabled:
 * out double[4] = K5, K2, K3, K4; `launches` = steps accumulated.  Enabling inserts events only. */
int sgpmp_profile_enable(sgpmp_ctx* ctx, int on);
int sgpmp_profile_read(sgpmp_ctx* ctx, double* ms4, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* SGPMP_H */
