// K3 -- composite cost sweep over the trajectory batch: trajs [B,T,d] -> costs [B].
//
// Replaces CostComposite.eval (costs/cost_functions.py:47-58) and everything it calls:
//   CostGP / CostGPTrajectory.eval   cost_functions.py:128-146, 202-215  (+ gp_factor.py:54-67,
//                                    unary_factor.py:21-22)
//   CostGoalPrior.eval               cost_functions.py:376-388
//   CostCollision.eval               cost_functions.py:247-261 + field_factor.py:18-40 with
//       ObstacleMap.get_collisions   envs/obst_map.py:164-182
//       LinkDistanceField            costs/fields.py:63-86   (rbf / sdf / occupancy)
//       LinkSelfDistanceField        costs/fields.py:114-124
//   the FK callable                  cost_functions.py:51-52 (URDF chain)
//   the importance-sampling term     planner.py:233-236, in the factored form
//       x^T Sigma^-1 mu = (A x)^T [blkdiag(K_s, Q^-1.., K_g) A mu],  A x = (x_0, e_0.., x_{T-1})
//
// Mapping: ONE WAVE PER TRAJECTORY, ONE LANE PER WAYPOINT (64 waypoints per pass).  A lane loads its
// own d-vector (the wave reads one contiguous 64*d*w-byte span), gets x_{t-1} from its neighbour
// lane, evaluates every per-waypoint term, and the wave reduces the 64 partial costs in fp64.
//
// Per-term constants travel in the kernel-argument segment (ProgK, by value) and per-joint constants
// are read through the constant address space, so both are scalar loads into SGPRs; read through a
// plain global pointer they become loop-invariant VECTOR loads that pin >200 VGPRs.
//
// Link fields have two paths:
//   register path (FKMODE = number of joints): revolute-first chains without interpolated points.
//     The link positions stay in registers (every loop over links is unrolled), coincident links
//     are merged and rigid link pairs folded into a constant by the host analysis (FkPlan), exp is
//     one v_exp_f32 (exp2 with log2(e) folded into the constant) and sin/cos are native in fp32.
//   generic path (FKMODE = -1): any chain / interpolated points; link positions of the lane's
//     waypoint live in LDS (structure-of-arrays, one column per lane: bank-conflict free) because
//     the loops index them dynamically.
#include <cstdlib>
#include <cstring>

#include "chain_code_generated.h"
#include "rng.h"
#include "sgpmp_internal.h"
#include "update_common.h"
#include <hip/hip_ext.h>

#include "cost_device.h"

// ---- host side of the kernel-argument structs
template <typename real>
static TermK<real> make_termk(const CostTerm& s) {
    TermK<real> k;
    k.kind = s.kind; k.flags = s.flags;
    k.K = (real)s.K; k.K2 = (real)s.K2; k.dt = (real)s.dt;
    k.c11 = (real)s.c11; k.c12 = (real)s.c12; k.c22 = (real)s.c22; k.selfc = (real)s.selfc;
    k.inv_cell = (real)s.inv_cell; k.off_x = (real)s.off_x; k.off_y = (real)s.off_y;
    k.dev_data = s.dev_data; k.dim0 = s.dim0; k.dim1 = s.dim1; k.rows_per_goal = s.rows_per_goal;
    k.n_points = s.n_points; k.n_interp = s.n_interp; k.interp_lo = s.interp_lo; k.interp_hi = s.interp_hi;
    for (int a = 0; a < SGPMP_MAX_INTERP; ++a) k.alpha[a] = (real)s.alpha[a];
    return k;
}

template <typename real>
static ProgK<real> make_progk(const CostProgram& p) {
    ProgK<real> k;
    k.n_terms = p.n_terms; k.needs_fk = p.needs_fk;
    for (int i = 0; i < SGPMP_MAX_TERMS; ++i) k.t[i] = make_termk<real>(p.terms[i]);
    return k;
}

template <typename real>
static bool make_flat(const CostProgram& p, FlatProg<real>& f) {
    std::memset(&f, 0, sizeof(f));
    for (int i = 0; i < p.n_terms; ++i) {
        const TermK<real> k = make_termk<real>(p.terms[i]);
        int* has = nullptr;
        TermK<real>* slot = nullptr;
        switch (k.kind) {
            case SGPMP_COST_GP: has = &f.has_gp; slot = &f.gp; break;
            case SGPMP_COST_GOAL_PRIOR: has = &f.has_goal; slot = &f.goal; break;
            case SGPMP_COST_GRID: has = &f.has_grid; slot = &f.grid; break;
            case SGPMP_COST_SELF: has = &f.has_self; slot = &f.self; break;
            case SGPMP_COST_SPHERES: has = &f.has_sph; slot = &f.sph; f.sph_index = i; break;
            case SGPMP_COST_EE_GOAL: continue;      // evaluated by ee_goal_kernel after the sweep
            default: return false;
        }
        if (*has) return false;                      // a second term of this kind: not flat
        *has = 1;
        *slot = k;
    }
    return true;
}

#include "cost_sweep_kernel.inc"
#include "cost_sweep_dual.inc"
#include "fused_step.inc"
#include "fused_planar.inc"
#include "fused_planar_seg.inc"

// Does a step qualify for a fused launch?  1: chain-code program (fused_step.inc), 2: program without forward
// kinematics (fused_planar.inc), 0: no.
static int fused_step_kind(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog,
                           const ChainDev& h_chain, int P, int mode_offset, int S, int n_spheres,
                           const SgpmpToggles& tg) {
    using CCp = ChainCode_panda;
    if (tg.no_fused_step || !prior.isotropic) return 0;
    if (dtype == SGPMP_F64) {
        // fp64 contexts: sampler + sweep as fused_step_f64_kernel (cost_sweep_kernel.inc: GEN) -- one wave per trajectory, lane =
        // waypoint, the recurrence as a scan over the lanes; FLAT programs on the positions themselves (n = 2, 3) or on the chain
        // code built with the library
        if (T < 2 || P < 1 || S < 1 || !prior.scan64 || tg.no_flat_program) return 0;
        if ((long long)P * S + (long long)mode_offset * S >= (1LL << 31)) return 0;
        FlatProg<double> F;
        if (!make_flat<double>(h_prog, F)) return 0;
        if (F.has_goal && F.goal.rows_per_goal % S != 0) return 0;
        if (F.has_gp && prior.dt != F.gp.dt) return 0;
        for (int i = 0; i < h_prog.n_terms; ++i)
            if (h_prog.terms[i].n_interp > 0) return 0;
        if (!h_prog.needs_fk) {
            if (F.has_self || F.has_sph || (n != 2 && n != 3)) return 0;
            return 3;
        }
        if (tg.no_chain_codegen || tg.force_generic_fk || !h_chain.plan.fast || h_chain.plan.codegen_id != 1 || n != CCp::N || F.has_grid) return 0;
        return 3;
    }
    if (dtype != SGPMP_F32) return 0;
    // (S: the chain-code launch masks the rows of a particle's last group of 8 -- round 4; the planar launches want whole groups)
    // (T: the chain-code launch masks the columns and cost lanes past T in the last chunk of 16 -- T even: 16-byte rows)
    if (T < 2 || T % 2 != 0 || P < 1 || S < 1) return 0;
    const bool ragged = S % SGPMP_FUSED_SPW != 0 || T % SGPMP_FUSED_TC != 0;
    if ((long long)P * S + (long long)mode_offset * S >= (1LL << 31)) return 0;
    FlatProg<float> F;
    if (tg.no_flat_program || !make_flat<float>(h_prog, F)) return 0;
    if (F.has_goal && (F.goal.rows_per_goal % S != 0 || F.goal.dim0 > SGPMP_FUSED_GOALS)) return 0;   // (a particle has one goal)
    if (F.has_gp && (float)prior.dt != F.gp.dt) return 0;                 // IS term and GP factors share Phi
    if (!h_prog.needs_fk && h_prog.n_ee == 0) {
        if (ragged) return 0;
        // no link fields: GP / goal prior / occupancy grid on the positions themselves
        if (F.has_self || F.has_sph || (n != 2 && n != 3) || T > SGPMP_PLANAR_TMAX) return 0;
        if (F.has_grid && n < 2) return 0;
        return 2;
    }
    if (tg.no_dual_sweep || tg.no_chain_codegen || tg.force_generic_fk) return 0;
    if (!h_chain.plan.fast || F.has_grid) return 0;
    if (h_chain.plan.codegen_id == 1) { if (n != CCp::N) return 0; }         // the chain built with the library
    else if (h_chain.plan.codegen_id != 2 || !h_chain.rtc || n > 7) return 0;   // ... or compiled at run time (sgpmp_set_fk_codegen)
    if (n_spheres > SGPMP_FUSED_SPH) return 0;
    for (int i = 0; i < h_prog.n_terms; ++i)
        if (h_prog.terms[i].n_interp > 0) return 0;
    if (h_chain.plan.codegen_id == 2 &&
        !rtc_kernel((RtcChain*)h_chain.rtc, F.has_sph ? (F.sph.flags & 15) : SGPMP_FIELD_RBF, false)) return 0;   // (compiled on first use)
    return 1;
}

bool fused_step_eligible(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog,
                         const ChainDev& h_chain, int P, int mode_offset, int S, int n_spheres,
                         const SgpmpToggles& tg) {
    return fused_step_kind(dtype, n, T, prior, h_prog, h_chain, P, mode_offset, S, n_spheres, tg) != 0;
}

// Shape of the lane-per-sample launch (fused_planar_seg.inc): waypoints per wave, 0 when (S, T, n) does not fit.
// Never a function of the particle count.
static int planar_seg_len(int n, int T, int S, const SgpmpToggles& tg) {
    if (tg.no_planar_seg || S % 64 != 0) return 0;
    const int L = T <= 128 ? 8 : 16;
    if (T % L != 0 || T / L > 16 || (L == 16 && n != 2)) return 0;
    return L;
}

// Does the step run as fused_planar_seg_kernel (1024-thread workgroups, one per particle and 64 samples)?
bool planar_seg_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                     int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg) {
    return fused_step_kind(dtype, n, T, prior, h_prog, h_chain, P, mode_offset, S, n_spheres, tg) == 2 &&
           planar_seg_len(n, T, S, tg) != 0;
}

static size_t planar_seg_lds(int n, int T, int L) {
    const int G = T / L;
    return (size_t)G * 64 * (16 * n + 8) + (size_t)G * 64 * 20 * sizeof(float) + (size_t)G * 16;
}
bool planar_tail_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                      int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg) {
    if (!planar_seg_step(dtype, n, T, prior, h_prog, h_chain, P, mode_offset, S, n_spheres, tg)) return false;
    const int L = planar_seg_len(n, T, S, tg);
    return S == 64 && prior.isotropic && !tg.no_planar_tail && (long long)P * S / 64 <= (1LL << 20) &&
           (size_t)T * 2 * n * sizeof(float) + planar_seg_lds(n, T, L) <= 160 * 1024;
}
// ... and can ONE launch run several of them (PERSIST)?  Instantiated for n = 2 with segments of 8 waypoints -- BASELINE
// configs[1]'s shape: 110 vector registers.  Segments of 16 and n = 3 hold 32 / 48 waypoint values per lane: their single-step
// launches use 118 / 114 of the 128 registers a 1024-thread workgroup's waves can have, and the loop's few carried values
// pushed 21 / 19 registers into scratch (tools/audit_asm_loads.py refuses scratch in these kernels) -- those shapes keep one
// launch per iteration.
bool planar_persist_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                         int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg) {
    return !tg.no_persist_planar && n == 2 && planar_seg_len(n, T, S, tg) == 8 &&
           planar_tail_step(dtype, n, T, prior, h_prog, h_chain, P, mode_offset, S, n_spheres, tg);
}

// Which rows would update_kernel have to regenerate if this step ran store-free?  1: fused_step_kernel's, 2:
// fused_planar_seg_kernel's (*seg_len waypoints per segment), 0: the step's launch has no store-free form (the tile launch
// fused_planar_kernel, the two-launch paths) or its costs are not complete inside the launch (ee_goal_kernel reads the rows).
int fused_step_regen_recipe(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog, const ChainDev& h_chain,
                            int P, int mode_offset, int S, int n_spheres, const SgpmpToggles& tg, int* seg_len) {
    if (seg_len) *seg_len = 0;
    const int kind = fused_step_kind(dtype, n, T, prior, h_prog, h_chain, P, mode_offset, S, n_spheres, tg);
    if (kind == 0 || kind == 3 || h_prog.n_ee > 0) return 0;
    if (kind == 1) return 1;
    // (the lane-per-sample planar launch: built, bit-identical, and measured slower store-free at config 2 -- its update kernel is
    // not hidden under another chain's launch, and regenerating a row costs it more than the launch saves: opt-in)
    if (!tg.planar_store_free) return 0;
    const int L = planar_seg_len(n, T, S, tg);
    if (!L || (long long)P * S / 64 > (1LL << 20)) return 0;
    if (seg_len) *seg_len = L;
    return 2;
}

// K2 + K3 in one launch when the step qualifies; *launched says whether it did.
hipError_t launch_fused_step(int dtype, int n, int T, const PriorDev& prior, const CostProgram& h_prog,
                             const ChainDev& h_chain, uint64_t seed, uint64_t draw, const void* means, int P,
                             int mode_offset, int S, void* samples, const void* spheres, int n_spheres,
                             const void* isw, double* zero_stats, void* costs, double* costs64,
                             hipStream_t stream, const SgpmpToggles& tg, const char** picked, bool* launched,
                             const FusedDenseHost* dense, bool* partials_armed, RegenHost* regen, bool* tail_ran) {
    *launched = false;
    if (tail_ran) *tail_ran = false;
    if (partials_armed) *partials_armed = false;
    if (regen) std::memset(regen, 0, sizeof(*regen));
    using CCp = ChainCode_panda;
    const int kind = (!samples || !isw) ? 0 : fused_step_kind(dtype, n, T, prior, h_prog, h_chain, P, mode_offset, S, n_spheres, tg);
    if (kind == 0) return hipSuccess;
    if (kind == 3) {
        FlatProg<double> F;
        make_flat<double>(h_prog, F);
        const ProgK<double> PK = make_progk<double>(h_prog);
        auto log2x = [](long long v) { int s = 0; while ((1LL << s) < v && s < 62) ++s; return (1LL << s) == v ? s : -1; };
        CostArgs<double> a;
        a.T = T; a.chain = nullptr; a.n_links = h_chain.n_links; a.trajs = (const double*)samples;
        a.batch = (long long)P * S; a.batch_offset = (long long)mode_offset * S;
        a.spheres = (const double*)spheres; a.n_spheres = n_spheres;
        a.isw = (const double*)isw; a.rows_per_particle = S; a.is_dt = prior.dt;
        a.costs = (double*)costs; a.costs64 = costs64;
        a.rpp_shift = log2x(S); a.rpg_shift = F.has_goal ? log2x(F.goal.rows_per_goal) : -1;
        GenArgs64 g;
        g.coef = prior.iso64; g.scan = prior.scan64; g.means = (const double*)means; g.samples = (double*)samples;
        g.seed = seed; g.draw = draw; g.mode_offset = mode_offset; g.S = S; g.zero_stats = zero_stats;
        const size_t lds = (size_t)T * SGPMP_SCAN64_ROW * sizeof(double);
        // (few trajectories -- BASELINE configs[0] has 64 -- : one wave per workgroup, so that every wave gets a CU of its own, and
        // the lanes read their rows of the scan table straight from memory: staging 14 KB per workgroup for one or two items is a
        // string of dependent round trips a latency-bound launch cannot hide)
        const int block = a.batch <= 2048 ? 64 : 256;
        const bool scan_lds = lds <= 48 * 1024 && a.batch > 2048;
        long long blocks = (a.batch + block / 64 - 1) / (block / 64);
        const long long cap = 256LL * 32;
        if (blocks > cap) blocks = cap;
        const bool mixed = tg.f64_fields_f32 && h_prog.needs_fk && n_spheres <= SGPMP_SPH_LDS;
        // (+ the chain launches' store tile: a pass of 64 waypoints x d doubles per wave -- cost_sweep_kernel.inc)
        const bool store_tile = scan_lds && h_prog.needs_fk;
        const unsigned dyn = scan_lds ? (unsigned)(lds + (store_tile ? (size_t)(block / 64) * 64 * 2 * n * sizeof(double) : 0)) : 0u;
        // (SCAN_: how a big launch reads its staged scan table -- cost_sweep_kernel.inc: in one batch, or round by round)
#define F64_LAUNCH(K_, N_, FK_, SCAN_) do { if (scan_lds) hipLaunchKernelGGL((K_<N_, FK_, SCAN_>), dim3((unsigned)blocks), dim3(block), dyn, stream, a, PK, F, g); \
                                            else hipLaunchKernelGGL((K_<N_, FK_, 0>), dim3((unsigned)blocks), dim3(block), dyn, stream, a, PK, F, g); } while (0)
        if (h_prog.needs_fk && mixed) F64_LAUNCH(fused_step_f64_mixed_kernel, CCp::N, 1000, 2);
        else if (h_prog.needs_fk) F64_LAUNCH(fused_step_f64_kernel, CCp::N, 1000, 1);
        else if (n == 2) F64_LAUNCH(fused_step_f64_kernel, 2, 0, 2);
        else F64_LAUNCH(fused_step_f64_kernel, 3, 0, 2);
#undef F64_LAUNCH
        if (picked) *picked = mixed ? "fused_step_f64_mixed_kernel" : "fused_step_f64_kernel";
        *launched = true;
        return hipGetLastError();
    }
    FlatProg<float> F;
    make_flat<float>(h_prog, F);
    const long long batch = (long long)P * S, batch_offset = (long long)mode_offset * S;
    auto log2_exact = [](long long v) { int s = 0; while ((1LL << s) < v && s < 62) ++s; return (1LL << s) == v ? s : -1; };
    CostArgs<float> a;
    a.T = T; a.chain = nullptr; a.n_links = h_chain.n_links; a.trajs = (const float*)samples;
    a.batch = batch; a.batch_offset = batch_offset; a.spheres = (const float*)spheres; a.n_spheres = n_spheres;
    a.isw = (const float*)isw; a.rows_per_particle = S; a.is_dt = (float)prior.dt;
    a.costs = (float*)costs; a.costs64 = costs64;
    a.rpp_shift = log2_exact(S);
    a.rpg_shift = F.has_goal ? log2_exact(F.goal.rows_per_goal) : -1;
    FusedArgs fs;
    fs.coef = prior.iso32; fs.coefp = prior.iso32p; fs.means = (const float*)means; fs.samples = (float*)samples;
    fs.seed = seed; fs.draw = draw; fs.mode_offset = mode_offset; fs.S = S;
    fs.gpp = (S + SGPMP_FUSED_SPW - 1) / SGPMP_FUSED_SPW; fs.gpp_shift = log2_exact(fs.gpp);
    fs.nitems = (long long)P * fs.gpp;
    fs.zero_stats = zero_stats;
    fs.part = nullptr; fs.nnz_prev = nullptr; fs.nnz_threshold = 0u; fs.inv_temperature = 0.f;
    fs.nostore = 0; fs.store_threshold = 0u;
    std::memset(&fs.tail, 0, sizeof(fs.tail));
    if (dense && dense->nnz) fs.nnz_prev = dense->nnz;
    // softmax partials for the dense-weight regime of the update: chain-code launch whose costs are complete inside it
    if (kind == 1 && dense && dense->part && dense->nnz && h_prog.n_ee == 0 && !tg.no_dense_partials && (T * 2 * n) % 4 == 0) {
        fs.part = dense->part; fs.nnz_threshold = dense->threshold;
        fs.inv_temperature = (float)(1. / dense->temperature);
        if (partials_armed) *partials_armed = true;
    }
    // store-free step: the caller does not read this step's samples and the update behind the launch can regenerate rows
    if (dense && dense->nostore && dense->nnz && regen) {
        int L = 0;
        const int recipe = fused_step_regen_recipe(dtype, n, T, prior, h_prog, h_chain, P, mode_offset, S, n_spheres, tg, &L);
        // ... and regenerating pays: update_kernel's regeneration is a dependent chain of ~6.5 us per particle (T = 64) that a small
        // step cannot hide, while what the launch saves grows with the bytes it does not write.  Measured break-even on MI355X
        // (tools/store_free_sizes.py, profiles/r05/store_free_sizes.txt: Panda, S = 64 .. 512, T = 32 and 64, P = 16 .. 2048):
        // 176 MB of samples per step at T = 64, ~88 MB at T = 32 -- i.e. 2.75 MB per waypoint; below it a store-free step ran
        // 4 .. 20 % SLOWER than a storing one, so the step stores (same results either way: the choice is a function of the shape).
        const long long waypoint_bytes = (long long)dense->particles_total * S * 2 * n * (long long)sizeof(float);
        const long long min_bytes = tg.store_free_min_bytes > 0 ? tg.store_free_min_bytes : SGPMP_STORE_FREE_BREAK_EVEN;
        if (recipe != 0 && waypoint_bytes >= min_bytes && update_regen_rows(dtype, n, T, S, recipe) > 0) {
            fs.nostore = 1; fs.store_threshold = dense->store_threshold;
            regen->recipe = recipe; regen->L = L; regen->seed = seed; regen->draw = draw; regen->mode_offset = mode_offset;
            regen->coef = recipe == 1 ? prior.iso32p : prior.iso32;
            regen->pre = recipe == 2 ? prior.slabpre + (size_t)(L == 8 ? 3 : 4) * T * 4 : nullptr;
            regen->store_threshold = dense->store_threshold;
        }
    }
    const long long nitems = kind == 1 ? fs.nitems : batch / SGPMP_FUSED_SPW;
    long long blocks = (nitems + 3) / 4;
    // one item per wave measured fastest at config 3 (4096 workgroups 0.216 ms/iteration, 2048: 0.219,
    // 1024: 0.227): the per-workgroup set-up is small and the hardware dispatcher balances better than
    // a grid-stride loop; the loop stays for batches beyond 2^20 items and for the k3_blocks switch
    long long cap = 1LL << 18;
    if (tg.k3_blocks > 0) cap = tg.k3_blocks;
    if (blocks > cap) blocks = cap;
    if (kind == 2) {
        // lane = sample, wave = time segment (fused_planar_seg.inc) where the shape allows; picked from (S, T, n) alone
        {
            const int L = planar_seg_len(n, T, S, tg), G = L ? T / L : 0;
            const long long wgs = batch / 64;
            if (L && wgs <= (1LL << 20)) {
                size_t lds = planar_seg_lds(n, T, L);
                fs.gpp = S / 64; fs.gpp_shift = log2_exact(fs.gpp);
                // Store-free step of a problem whose particles have exactly one workgroup's 64 samples: the UPDATE runs inside the
                // launch (seg_update) -- no sample store, no update_kernel, no regeneration: one launch per iteration
                const bool upd = dense && dense->nostore && dense->tail_done && tail_ran && S == 64 && prior.isotropic && !tg.no_planar_tail &&
                                 (size_t)T * 2 * n * sizeof(float) + lds <= 160 * 1024;
                if (upd) {
                    SegTail& t = fs.tail;
                    t.done = dense->tail_done; t.acc = dense->tail_acc; t.stats_out = dense->stats_out;
                    t.means = (float*)const_cast<void*>(means); t.weights = (float*)dense->weights; t.grad = (float*)dense->grad;
                    t.means_prev = (float*)dense->means_prev; t.isw_next = (float*)const_cast<void*>(isw); t.nnz_out = dense->nnz;
                    t.Qinv = prior.Qinv; t.ks = prior.ks; t.kg = prior.kg; t.dt = prior.dt;
                    t.temperature = dense->temperature; t.step_size = dense->step_size; t.P = P;
                    t.iters = dense->tail_iters > 1 ? dense->tail_iters : 1;
                    if (t.iters > 1 && !(n == 2 && L == 8)) return hipErrorInvalidValue;    // (planar_persist_step: the caller asks first)
                    if (t.iters > 1) { t.done = nullptr; t.stats_out = nullptr; }    // (several iterations in this launch: no statistics)
                    fs.nostore = 1; fs.store_threshold = 0xffffffffu;
                    fs.zero_stats = nullptr;                      // (the launch's last particle writes the statistics)
                    lds += (size_t)T * 2 * n * sizeof(float);
                    if (regen) regen->recipe = 0;                 // (nothing left for update_kernel to regenerate: there is no update_kernel)
                    *tail_ran = true;
                }
                const float* tab = prior.slabpre + (size_t)(L == 8 ? 3 : 4) * T * 4;
                const bool persist = upd && fs.tail.iters > 1;
#define SEG_LAUNCH(NN, LL) do { if (upd) hipLaunchKernelGGL((fused_planar_seg_kernel<NN, LL, true>), dim3((unsigned)wgs), dim3(64 * G), (unsigned)lds, stream, a, F, fs, tab); \
                                 else hipLaunchKernelGGL((fused_planar_seg_kernel<NN, LL, false>), dim3((unsigned)wgs), dim3(64 * G), (unsigned)lds, stream, a, F, fs, tab); } while (0)
                if (persist) hipLaunchKernelGGL((fused_planar_seg_kernel<2, 8, true, true>), dim3((unsigned)wgs), dim3(64 * G), (unsigned)lds, stream, a, F, fs, tab);
                else if (n == 2) { if (L == 8) SEG_LAUNCH(2, 8); else SEG_LAUNCH(2, 16); }
                else SEG_LAUNCH(3, 8);
#undef SEG_LAUNCH
                if (picked) *picked = "fused_planar_seg_kernel";
                *launched = true;
                return hipGetLastError();
            }
        }
        if (n == 2) hipLaunchKernelGGL((fused_planar_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, stream, a, F, fs);
        else hipLaunchKernelGGL((fused_planar_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, stream, a, F, fs);
        if (picked) *picked = "fused_planar_kernel";
        *launched = true;
        return hipGetLastError();
    }
    const int ft = F.has_sph ? (F.sph.flags & 15) : SGPMP_FIELD_RBF;
    // a SMALL step -- fewer items than SIMDs: every wave of the one-wave-per-item launch would sit alone on its SIMD for as long as
    // one item takes one wave (~20 us) -- goes out with one WORKGROUP per item instead, its four waves on the item's chunks side
    // by side (fused_step.inc: LAT; same samples and costs, bit for bit).  Up to two workgroups per CU for shapes on the launch's
    // 8 x 16 grid (253 registers, 57 KB of LDS: two per SIMD set), one for the others (the masked instantiation needs 262 registers):
    // beyond, a second round of workgroups costs what the other launch does in one (tools/small_step_sizes.py).
    const bool srag = S % SGPMP_FUSED_SPW != 0 || T % SGPMP_FUSED_TC != 0;           // (the instantiation fused_step_kernel would take)
    const long long small_items = tg.small_step_items > 0 ? tg.small_step_items : srag ? 256 : 512;
    // (judged on the WHOLE problem -- all ranks' particles, both halves of a pipelined step -- so that a shard takes the launch its
    // unsharded run takes)
    const long long items_global = (long long)(dense && dense->particles_global > 0 ? dense->particles_global : P) * fs.gpp;
    const bool small = !tg.no_small_step && items_global <= small_items && (T + SGPMP_FUSED_TC - 1) / SGPMP_FUSED_TC <= 16;
    if (h_chain.plan.codegen_id == 2) {           // this chain's kernels were compiled at run time (chain_rtc.hip)
        hipFunction_t f = rtc_kernel((RtcChain*)h_chain.rtc, ft, false, srag, small);
        if (!f) return hipSuccess;                // (not launched: the caller takes the two-launch path)
        void* args[] = {&a, &F, &fs};
        const hipError_t e = rtc_launch(f, small ? (unsigned)fs.nitems : (unsigned)blocks, (unsigned)fused_wave_dyn_lds(n_spheres, F.has_goal ? F.goal.dim0 : 0), stream, args, nullptr);
        if (picked) *picked = small ? "fused_step_small_kernel (run-time chain code)" : "fused_step_kernel (run-time chain code)";
        *launched = e == hipSuccess;
        return e;
    }
#ifndef SGPMP_FUSED_EXTRA_LDS   // occupancy diagnostic: unused bytes per workgroup, e.g. 8000 -> four workgroups per CU instead of five
#define SGPMP_FUSED_EXTRA_LDS 0
#endif
    // the launch sizes the sphere / state tables (dynamic LDS): 30.7 KB of tiles + these per workgroup, five workgroups per CU while
    // they stay under 2 KB
    const size_t dyn = fused_wave_dyn_lds(n_spheres, F.has_goal ? F.goal.dim0 : 0);
    // S, T off the launch's grid of 8 rows x 16 waypoints: the instantiation with the masks (fused_step.inc: RAG)
    if (small) {
#define SMALL_LAUNCH(FT_, RAG_) hipLaunchKernelGGL((fused_step_small_kernel<CCp::N, CCp, FT_, RAG_>), dim3((unsigned)fs.nitems), dim3(256), (unsigned)dyn, stream, a, F, fs)
        if (ft == SGPMP_FIELD_RBF) { if (srag) SMALL_LAUNCH(SGPMP_FIELD_RBF, true); else SMALL_LAUNCH(SGPMP_FIELD_RBF, false); }
        else if (ft == SGPMP_FIELD_SDF) { if (srag) SMALL_LAUNCH(SGPMP_FIELD_SDF, true); else SMALL_LAUNCH(SGPMP_FIELD_SDF, false); }
        else { if (srag) SMALL_LAUNCH(SGPMP_FIELD_OCCUPANCY, true); else SMALL_LAUNCH(SGPMP_FIELD_OCCUPANCY, false); }
#undef SMALL_LAUNCH
        if (picked) *picked = "fused_step_small_kernel";
        *launched = true;
        return hipGetLastError();
    }
#define FUSED_LAUNCH(FT_, RAG_) hipExtLaunchKernelGGL((fused_step_kernel<CCp::N, CCp, FT_, RAG_>), dim3((unsigned)blocks), dim3(256), (unsigned)dyn + SGPMP_FUSED_EXTRA_LDS, stream, (hipEvent_t) nullptr, (hipEvent_t) nullptr, 0u, a, F, fs)
    const bool rag = S % SGPMP_FUSED_SPW != 0 || T % SGPMP_FUSED_TC != 0;
    if (ft == SGPMP_FIELD_RBF) { if (rag) FUSED_LAUNCH(SGPMP_FIELD_RBF, true); else FUSED_LAUNCH(SGPMP_FIELD_RBF, false); }
    else if (ft == SGPMP_FIELD_SDF) { if (rag) FUSED_LAUNCH(SGPMP_FIELD_SDF, true); else FUSED_LAUNCH(SGPMP_FIELD_SDF, false); }
    else { if (rag) FUSED_LAUNCH(SGPMP_FIELD_OCCUPANCY, true); else FUSED_LAUNCH(SGPMP_FIELD_OCCUPANCY, false); }
#undef FUSED_LAUNCH
    if (picked) *picked = "fused_step_kernel";
    *launched = true;
    return hipGetLastError();
}

template <typename real>
static hipError_t cost_dispatch(int n, int T, const CostProgram& h_prog, const ChainDev* d_chain,
                                const ChainDev& h_chain, const real* trajs, long long batch,
                                long long batch_offset, const real* spheres, int n_spheres,
                                const real* isw, int rows_per_particle, double is_dt, real* costs,
                                double* costs64, hipStream_t stream, const SgpmpToggles& tg, const char** picked) {
    const int n_links = h_chain.n_links;
    const char* unused_name;
    if (!picked) picked = &unused_name;
    constexpr bool f64 = sizeof(real) == 8;
    if (batch + batch_offset >= (1LL << 31)) return hipErrorInvalidValue;   // row indices are 32-bit
    CostArgs<real> a;
    a.T = T; a.chain = d_chain; a.n_links = n_links; a.trajs = trajs;
    a.batch = batch; a.batch_offset = batch_offset; a.spheres = spheres; a.n_spheres = n_spheres;
    a.isw = isw; a.rows_per_particle = rows_per_particle > 0 ? rows_per_particle : 1;
    a.is_dt = (real)is_dt; a.costs = costs; a.costs64 = costs64;
    const ProgK<real> P = make_progk<real>(h_prog);
    int block = 256;
    size_t lds = 0;
    const bool fk = h_prog.needs_fk != 0;
    // register path: revolute-first chain, no interpolated points, an instantiated (n, joints) pair
    bool reg = fk && h_chain.plan.fast && !tg.force_generic_fk;
    for (int i = 0; i < h_prog.n_terms; ++i)
        if (h_prog.terms[i].n_interp > 0) reg = false;
    const int nj = n_links - 1;
    if (fk && !reg) {
        int max_pts = n_links;
        for (int i = 0; i < h_prog.n_terms; ++i)
            if (h_prog.terms[i].n_points > max_pts) max_pts = h_prog.terms[i].n_points;
        lds = (size_t)max_pts * 3 * block * sizeof(real);
        while (lds > 64 * 1024 && block > 64) { block >>= 1; lds >>= 1; }
    }
    const int wpb = block / 64;
    long long blocks = (batch + wpb - 1) / wpb;
    const long long cap = 256LL * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    FlatProg<real> F;
    const bool flat = make_flat<real>(h_prog, F) && !tg.no_flat_program;
    auto log2_exact = [](long long v) { int s = 0; while ((1LL << s) < v && s < 62) ++s; return (1LL << s) == v ? s : -1; };
    a.rpp_shift = log2_exact(a.rows_per_particle);
    a.rpg_shift = (flat && F.has_goal) ? log2_exact(F.goal.rows_per_goal) : -1;
    const bool cg_static = h_chain.plan.codegen_id == 1 && n == ChainCode_panda::N;
    const bool cg_rtc = h_chain.plan.codegen_id == 2 && h_chain.rtc && sizeof(real) == 4 && n <= 7;
    if (reg && (cg_static || cg_rtc) && !tg.no_chain_codegen) {
        if constexpr (sizeof(real) == 4) {
            // two trajectories per wave on packed fp32 math (cost_sweep_dual.inc) when rows pair up
            const bool pairs_ok = (batch_offset % 2 == 0) && (!isw || a.rows_per_particle % 2 == 0) &&
                                  (!F.has_goal || F.goal.rows_per_goal % 2 == 0);
            const int ft = F.has_sph ? (F.sph.flags & 15) : SGPMP_FIELD_RBF;
            const bool sph_ok = !F.has_sph || n_spheres <= SGPMP_SPH_LDS;
            // chunked sweep (fused_step.inc, phase C alone on existing samples): 8 rows of one particle and goal
            // per wave, 16 waypoints per chunk -- fewer instructions per waypoint than the 64-lane passes below
            const bool chunk_ok = flat && !F.has_grid && sph_ok && !tg.no_dual_sweep && !tg.no_chunked_sweep &&
                                  batch % SGPMP_FUSED_SPW == 0 && batch_offset % SGPMP_FUSED_SPW == 0 &&
                                  T % SGPMP_FUSED_TC == 0 && n_spheres <= SGPMP_FUSED_SPH &&
                                  (!isw || a.rows_per_particle % SGPMP_FUSED_SPW == 0) &&
                                  (!F.has_goal || (F.goal.rows_per_goal % SGPMP_FUSED_SPW == 0 && F.goal.dim0 <= SGPMP_FUSED_GOALS)) &&
                                  (!F.has_gp || a.is_dt == F.gp.dt || !isw);
            if (chunk_ok) {
                FusedArgs fs;
                std::memset(&fs, 0, sizeof(fs));
                fs.gpp = 1; fs.gpp_shift = 0;
                long long cblocks = (batch / SGPMP_FUSED_SPW + 3) / 4;
                long long ccap = 1LL << 18;
                if (tg.k3_blocks > 0) ccap = tg.k3_blocks;
                if (cblocks > ccap) cblocks = ccap;
                const int cft = F.has_sph ? (F.sph.flags & 15) : SGPMP_FIELD_RBF;
                hipFunction_t rf = cg_rtc ? rtc_kernel((RtcChain*)h_chain.rtc, cft, true) : nullptr;
                if (cg_rtc && rf) {                       // the chain's own kernels, compiled at run time (chain_rtc.hip)
                    void* args[] = {&a, &F, &fs};
                    const hipError_t e = rtc_launch(rf, (unsigned)cblocks, 0u, stream, args, nullptr);
                    *picked = "cost_sweep_chunked_kernel (run-time chain code)";
                    return e;
                }
                if (!cg_rtc) {                            // (cg_rtc without a kernel: FkPlan register path below)
                if (cft == SGPMP_FIELD_RBF)
                    hipLaunchKernelGGL((cost_sweep_chunked_kernel<ChainCode_panda::N, ChainCode_panda, SGPMP_FIELD_RBF>),
                                       dim3((unsigned)cblocks), dim3(256), 0, stream, a, F, fs);
                else if (cft == SGPMP_FIELD_SDF)
                    hipLaunchKernelGGL((cost_sweep_chunked_kernel<ChainCode_panda::N, ChainCode_panda, SGPMP_FIELD_SDF>),
                                       dim3((unsigned)cblocks), dim3(256), 0, stream, a, F, fs);
                else
                    hipLaunchKernelGGL((cost_sweep_chunked_kernel<ChainCode_panda::N, ChainCode_panda, SGPMP_FIELD_OCCUPANCY>),
                                       dim3((unsigned)cblocks), dim3(256), 0, stream, a, F, fs);
                *picked = "cost_sweep_chunked_kernel";
                return hipGetLastError();
                }
            }
            if (cg_static && flat && !F.has_grid && pairs_ok && sph_ok && !tg.no_dual_sweep) {
                long long pblocks = ((batch + 1) / 2 + 3) / 4;
                long long pcap = 256LL * 20;          // 20 workgroups per CU (4 resident): measured optimum (tools/k3_grid_sweep.sh)
                if (tg.k3_blocks > 0) pcap = tg.k3_blocks;
                if (pblocks > pcap) pblocks = pcap;
                if (pblocks < 1) pblocks = 1;
                const bool pf = T <= 64 && T % 2 == 0 && !tg.k3_no_one && !tg.k3_no_lds_prefetch;
                const bool one = T <= 64 && !tg.k3_no_one;
                const bool pf_multi = T > 64 && T % 2 == 0 && !tg.k3_no_lds_prefetch;
                *picked = pf ? "cost_sweep_dual_pf_kernel" : pf_multi ? "cost_sweep_dual_pf_multi_kernel"
                                                                      : "cost_sweep_dual_kernel";
#define DUAL_LAUNCH(FT)                                                                                    \
                if (pf)                                                                                    \
                    hipLaunchKernelGGL((cost_sweep_dual_pf_kernel<ChainCode_panda::N, ChainCode_panda, FT>),          \
                                       dim3((unsigned)pblocks), dim3(256), 0, stream, a, F);               \
                else if (pf_multi)                                                                         \
                    hipLaunchKernelGGL((cost_sweep_dual_pf_multi_kernel<ChainCode_panda::N, ChainCode_panda, FT>),    \
                                       dim3((unsigned)pblocks), dim3(256), 0, stream, a, F);               \
                else if (one && FT == SGPMP_FIELD_RBF)                                                     \
                    hipLaunchKernelGGL((cost_sweep_dual_kernel<ChainCode_panda::N, ChainCode_panda, true, SGPMP_FIELD_RBF>), \
                                       dim3((unsigned)pblocks), dim3(256), 0, stream, a, F);               \
                else                                                                                       \
                    hipLaunchKernelGGL((cost_sweep_dual_kernel<ChainCode_panda::N, ChainCode_panda, false, FT>),      \
                                       dim3((unsigned)pblocks), dim3(256), 0, stream, a, F)
                if (ft == SGPMP_FIELD_RBF) { DUAL_LAUNCH(SGPMP_FIELD_RBF); }
                else if (ft == SGPMP_FIELD_SDF) { DUAL_LAUNCH(SGPMP_FIELD_SDF); }
                else { DUAL_LAUNCH(SGPMP_FIELD_OCCUPANCY); }
#undef DUAL_LAUNCH
                return hipGetLastError();
            }
        }
        if (cg_static) {
        *picked = f64 ? "cost_sweep_kernel<f64, generated chain>" : "cost_sweep_kernel<f32, generated chain>";
        if (flat)
            hipLaunchKernelGGL((cost_sweep_kernel<real, ChainCode_panda::N, 1000, true>), dim3((unsigned)blocks),
                               dim3(256), 0, stream, a, P, F);
        else
            hipLaunchKernelGGL((cost_sweep_kernel<real, ChainCode_panda::N, 1000, false>), dim3((unsigned)blocks),
                               dim3(256), 0, stream, a, P, F);
        return hipGetLastError();
        }
    }
#define COST_REG(NN, NJJ)                                                                          \
    if (reg && n == NN && nj == NJJ) {                                                             \
        *picked = f64 ? "cost_sweep_kernel<f64, register FK>" : "cost_sweep_kernel<f32, register FK>"; \
        hipLaunchKernelGGL((cost_sweep_kernel<real, NN, NJJ, false>), dim3((unsigned)blocks), dim3(256), 0, \
                           stream, a, P, F);                                                       \
        return hipGetLastError();                                                                  \
    }
    COST_REG(7, 10) COST_REG(7, 7) COST_REG(6, 6) COST_REG(3, 3) COST_REG(2, 2)
#undef COST_REG
    if (reg) {               // no instantiation for this (n, joints): fall back to the generic path
        int max_pts = n_links;
        lds = (size_t)max_pts * 3 * block * sizeof(real);
        while (lds > 64 * 1024 && block > 64) { block >>= 1; lds >>= 1; }
        blocks = (batch + block / 64 - 1) / (block / 64);
        if (blocks > cap) blocks = cap;
    }
    *picked = fk ? (f64 ? "cost_sweep_kernel<f64, generic FK>" : "cost_sweep_kernel<f32, generic FK>")
                 : (f64 ? "cost_sweep_kernel<f64, no FK>" : "cost_sweep_kernel<f32, no FK>");
#define COST_CASE(NN)                                                                              \
    case NN:                                                                                       \
        if (fk)                                                                                    \
            hipLaunchKernelGGL((cost_sweep_kernel<real, NN, -1, false>), dim3((unsigned)blocks),   \
                               dim3(block), lds, stream, a, P, F);                                 \
        else if (flat)                                                                             \
            hipLaunchKernelGGL((cost_sweep_kernel<real, NN, 0, true>), dim3((unsigned)blocks),     \
                               dim3(block), 0, stream, a, P, F);                                   \
        else                                                                                       \
            hipLaunchKernelGGL((cost_sweep_kernel<real, NN, 0, false>), dim3((unsigned)blocks),    \
                               dim3(block), 0, stream, a, P, F);                                   \
        break;
    switch (n) {
        COST_CASE(1) COST_CASE(2) COST_CASE(3) COST_CASE(4) COST_CASE(5) COST_CASE(6) COST_CASE(7) COST_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef COST_CASE
    return hipGetLastError();
}

hipError_t launch_cost(int dtype, int n, int T, const CostProgram& h_prog, const ChainDev* d_chain,
                       const ChainDev& h_chain, const void* trajs, long long batch,
                       long long batch_offset, const void* spheres, int n_spheres,
                       const void* is_weights, int rows_per_particle, double is_dt, void* costs,
                       double* costs64, hipStream_t stream, const SgpmpToggles& tg, const char** picked) {
    hipError_t e;
    if (dtype == SGPMP_F64)
        e = cost_dispatch<double>(n, T, h_prog, d_chain, h_chain, (const double*)trajs, batch,
                                  batch_offset, (const double*)spheres, n_spheres,
                                  (const double*)is_weights, rows_per_particle, is_dt, (double*)costs,
                                  costs64, stream, tg, picked);
    else
        e = cost_dispatch<float>(n, T, h_prog, d_chain, h_chain, (const float*)trajs, batch,
                                 batch_offset, (const float*)spheres, n_spheres, (const float*)is_weights,
                                 rows_per_particle, is_dt, (float*)costs, costs64, stream, tg, picked);
    // end-effector goal terms act on the last waypoint only: one thread per trajectory, added to
    // the costs the sweep has just written (same stream)
    for (int i = 0; i < h_prog.n_terms && e == hipSuccess; ++i)
        if (h_prog.terms[i].kind == SGPMP_COST_EE_GOAL)
            e = launch_ee_goal(dtype, n, T, h_prog.terms[i], d_chain, trajs, batch, costs, costs64, stream);
    return e;
}

// ---------------------------------------------------------------------------------- EE goal term
// CostGoal + EESE3DistanceField (cost_functions.py:308-321, fields.py:146-150): K * dist^2 on the last
// waypoint, dist = SE3_distance(H_ee, H_target).  SE3_distance is third-party (torch_robotics) and
// un-vendored; this build defines it as  w_pos |p - p*| + w_rot angle(R*^T R)  (DESIGN.md).
// (EeTarget, se3_distance and the row's cost ee_goal_add live in update_common.h: update_kernel evaluates the term itself when it
// is the step's only end-effector goal -- one launch less per iteration)
template <typename real>
__global__ void ee_goal_kernel(int n, int T, const ChainDev* __restrict__ ch, const real* __restrict__ trajs,
                               long long batch, EeTarget<real> tg, real* __restrict__ costs,
                               double* __restrict__ costs64) {
    __shared__ real jr[SGPMP_MAX_JOINTS * SGPMP_EE_JR];
    __shared__ int ji[SGPMP_MAX_JOINTS * 2];
    ee_stage_chain<real>(ch, jr, ji, threadIdx.x, blockDim.x);
    const int nj = ch->n_joints;
    __syncthreads();
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double add = ee_goal_add<real>(jr, ji, nj, trajs + ((size_t)b * T + (T - 1)) * 2 * n, tg);
    if (costs) costs[b] = (real)((double)costs[b] + add);
    if (costs64) costs64[b] += add;
}

hipError_t launch_ee_goal(int dtype, int n, int T, const CostTerm& term, const ChainDev* d_chain,
                          const void* trajs, long long batch, void* costs, double* costs64,
                          hipStream_t stream) {
    const int block = 128;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((ee_goal_kernel<double>), dim3(grid), dim3(block), 0, stream, n, T, d_chain,
                           (const double*)trajs, batch, make_ee_target<double>(term), (double*)costs, costs64);
    else
        hipLaunchKernelGGL((ee_goal_kernel<float>), dim3(grid), dim3(block), 0, stream, n, T, d_chain,
                           (const float*)trajs, batch, make_ee_target<float>(term), (float*)costs, costs64);
    return hipGetLastError();
}

// Value and joint-space gradient of the end-effector field (EESE3DistanceField.compute_cost, fields.py:146-150, at
// this build's SE(3) distance d = w_pos |p - p*| + w_rot theta, theta = angle(R*^T R)): what
// FieldFactor.get_error(calc_jacobian=True) (field_factor.py:34-38) obtains with torch.autograd.grad for
// CostGoal.get_linear_system (cost_functions.py:323-337).  For a revolute joint j (axis z_j through o_j, world frame)
//   d|p - p*| / dq_j = u . (z_j x (p - o_j)),  u = (p - p*) / |p - p*|
//   d theta / dq_j   = z_j . a,                a = axis of E = R R*^T = vee(E - E^T) / |vee(E - E^T)|
// (dR/dq_j = [z_j]x R, tr([z]x E) = -z . vee(E - E^T), vee(E - E^T) = 2 sin(theta) a).  Both directions are taken as
// 0 where they are undefined (|p - p*| = 0, theta = 0 or pi).  traj_T = 0: q is [B,n]; traj_T = T: q is a
// trajectory batch [B,T,2n] and the configuration is its LAST waypoint.  value[b * out_stride + out_offset],
// grad[(b * out_stride + out_offset) * n + k].
template <typename real>
__global__ void ee_grad_kernel(int n, int traj_T, const ChainDev* __restrict__ ch, const real* __restrict__ q,
                               long long batch, EeTarget<real> tg, long long out_stride, long long out_offset,
                               real* __restrict__ value, real* __restrict__ grad) {
    using O = RealOps<real>;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const real* qb = traj_T > 0 ? q + ((size_t)b * traj_T + (traj_T - 1)) * 2 * n : q + (size_t)b * n;
    real Z[SGPMP_MAX_JOINTS][3], Oj[SGPMP_MAX_JOINTS][3];
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    const int nj = ch->n_joints;
    for (int j = 0; j < nj; ++j) {
        const JointDev& J = ch->j[j];
        real F[9], tt[3], Rn[9];
        for (int i = 0; i < 9; ++i) F[i] = (real)J.R[i];
        for (int i = 0; i < 3; ++i) tt[i] = (real)J.t[i];
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3 + 0] * tt[0] + R[r * 3 + 1] * tt[1] + R[r * 3 + 2] * tt[2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rn[r * 3 + c] = R[r * 3 + 0] * F[c] + R[r * 3 + 1] * F[3 + c] + R[r * 3 + 2] * F[6 + c];
        for (int r = 0; r < 3; ++r) { Z[j][r] = Rn[r * 3 + 2]; Oj[j][r] = p[r]; }     // joint axis and origin
        if (J.revolute) {
            real sn, cs;
            O::sincos_(qb[J.qidx], &sn, &cs);
            for (int r = 0; r < 3; ++r) {
                const real aa = Rn[r * 3 + 0], bb = Rn[r * 3 + 1];
                Rn[r * 3 + 0] = aa * cs + bb * sn;
                Rn[r * 3 + 1] = bb * cs - aa * sn;
            }
        }
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    // distance and its directions
    const real dx = p[0] - tg.p[0], dy = p[1] - tg.p[1], dz = p[2] - tg.p[2];
    const real dpos = O::sqrt_(dx * dx + dy * dy + dz * dz);
    real tr = 0;
    for (int i = 0; i < 9; ++i) tr += R[i] * tg.R[i];
    const real cth = fmin(fmax((tr - (real)1) * (real)0.5, (real)-1), (real)1);
    const real dist = tg.w_pos * dpos + tg.w_rot * acos(cth);
    real E[9];                                                             // E = R R*^T
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            E[r * 3 + c] = R[r * 3 + 0] * tg.R[c * 3 + 0] + R[r * 3 + 1] * tg.R[c * 3 + 1] + R[r * 3 + 2] * tg.R[c * 3 + 2];
    real w[3] = {E[7] - E[5], E[2] - E[6], E[3] - E[1]};
    const real wn = O::sqrt_(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const bool interior = cth > (real)-1 && cth < (real)1 && wn > (real)0;   // (acos has no slope to offer at +-1)
    const real outer = tg.square ? (real)2 * dist : (real)1;                 // d f / d dist
    const real su = dpos > (real)0 ? outer * tg.w_pos / dpos : (real)0;
    const real sa = interior ? outer * tg.w_rot / wn : (real)0;
    const real gp[3] = {su * dx, su * dy, su * dz};                          // d f / d p
    const real ta[3] = {sa * w[0], sa * w[1], sa * w[2]};                    // d f / d (rotation vector)
    const size_t o = (size_t)(b * out_stride + out_offset);
    if (value) value[o] = tg.square ? dist * dist : dist;
    real* gb = grad + o * n;
    for (int k = 0; k < n; ++k) gb[k] = 0;
    for (int j = 0; j < nj; ++j) {
        if (!ch->j[j].revolute) continue;
        const real rx = p[0] - Oj[j][0], ry = p[1] - Oj[j][1], rz = p[2] - Oj[j][2];
        const real cx = Z[j][1] * rz - Z[j][2] * ry, cy = Z[j][2] * rx - Z[j][0] * rz, cz = Z[j][0] * ry - Z[j][1] * rx;
        gb[ch->j[j].qidx] += gp[0] * cx + gp[1] * cy + gp[2] * cz + Z[j][0] * ta[0] + Z[j][1] * ta[1] + Z[j][2] * ta[2];
    }
}

hipError_t launch_ee_grad(int dtype, int n, const CostTerm& term, const ChainDev* d_chain, const void* q,
                          long long batch, int traj_T, long long out_stride, long long out_offset, void* value,
                          void* grad, hipStream_t stream) {
    const int block = 64;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((ee_grad_kernel<double>), dim3(grid), dim3(block), 0, stream, n, traj_T, d_chain,
                           (const double*)q, batch, make_ee_target<double>(term), out_stride, out_offset,
                           (double*)value, (double*)grad);
    else
        hipLaunchKernelGGL((ee_grad_kernel<float>), dim3(grid), dim3(block), 0, stream, n, traj_T, d_chain,
                           (const float*)q, batch, make_ee_target<float>(term), out_stride, out_offset,
                           (float*)value, (float*)grad);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------- standalone ops
// FK callable: q [B,n] -> frames [B,L,4,4]  (cost_functions.py:51-52).
template <typename real>
__global__ void fk_frames_kernel(int n, const ChainDev* __restrict__ ch, const real* __restrict__ q,
                                 long long batch, real* __restrict__ frames) {
    using O = RealOps<real>;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int L = ch->n_joints + 1;
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    real* out = frames + (size_t)b * L * 16;
    auto emit = [&](int l) {
        real* o = out + l * 16;
        for (int r = 0; r < 3; ++r) {
            o[r * 4 + 0] = R[r * 3 + 0]; o[r * 4 + 1] = R[r * 3 + 1]; o[r * 4 + 2] = R[r * 3 + 2];
            o[r * 4 + 3] = p[r];
        }
        o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
    };
    emit(0);
    for (int j = 0; j < ch->n_joints; ++j) {
        const JointDev& J = ch->j[j];
        real F[9], tt[3], Rn[9];
        for (int i = 0; i < 9; ++i) F[i] = (real)J.R[i];
        for (int i = 0; i < 3; ++i) tt[i] = (real)J.t[i];
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3 + 0] * tt[0] + R[r * 3 + 1] * tt[1] + R[r * 3 + 2] * tt[2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rn[r * 3 + c] = R[r * 3 + 0] * F[c] + R[r * 3 + 1] * F[3 + c] + R[r * 3 + 2] * F[6 + c];
        if (J.revolute) {
            real s, c;
            O::sincos_(q[(size_t)b * n + J.qidx], &s, &c);
            for (int r = 0; r < 3; ++r) {
                const real aa = Rn[r * 3 + 0], bb = Rn[r * 3 + 1];
                Rn[r * 3 + 0] = aa * c + bb * s;
                Rn[r * 3 + 1] = bb * c - aa * s;
            }
        }
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        emit(j + 1);
    }
}

hipError_t launch_fk(int dtype, int n, const ChainDev* d_chain, int n_links, const void* q,
                     long long batch, void* frames, hipStream_t stream) {
    const int block = 128;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((fk_frames_kernel<double>), dim3(grid), dim3(block), 0, stream, n, d_chain,
                           (const double*)q, batch, (double*)frames);
    else
        hipLaunchKernelGGL((fk_frames_kernel<float>), dim3(grid), dim3(block), 0, stream, n, d_chain,
                           (const float*)q, batch, (float*)frames);
    return hipGetLastError();
}

template <typename real>
__global__ void grid_lookup_kernel(const TermK<real> tm, const real* __restrict__ xy, long long batch,
                                   real* __restrict__ out) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    out[b] = grid_value<real>(tm, xy[2 * b], xy[2 * b + 1]);
}

hipError_t launch_grid_lookup(int dtype, const CostTerm& term, const void* xy, long long batch,
                              void* out, hipStream_t stream) {
    const int block = 256;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((grid_lookup_kernel<double>), dim3(grid), dim3(block), 0, stream,
                           make_termk<double>(term), (const double*)xy, batch, (double*)out);
    else
        hipLaunchKernelGGL((grid_lookup_kernel<float>), dim3(grid), dim3(block), 0, stream,
                           make_termk<float>(term), (const float*)xy, batch, (float*)out);
    return hipGetLastError();
}

// LinkDistanceField / LinkSelfDistanceField.compute_cost on explicit frames [B,L,4,4] -> [B].
template <typename real>
__global__ void field_eval_kernel(const TermK<real> tm, const real* __restrict__ frames, long long batch,
                                  int n_links, const real* __restrict__ sph, int n_sph,
                                  real* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    real* col = reinterpret_cast<real*>(lds_raw) + threadIdx.x;
    const int stride = blockDim.x;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long bb = b < batch ? b : batch - 1;
    const real* f = frames + (size_t)bb * n_links * 16;
    for (int l = 0; l < n_links; ++l)
        for (int c = 0; c < 3; ++c) col[(l * 3 + c) * stride] = f[l * 16 + c * 4 + 3];
    if (tm.n_interp > 0) add_interp_points<real>(tm, n_links, col, stride);
    real v;
    if (tm.kind == SGPMP_COST_SPHERES) v = spheres_field<real>(tm, tm.n_points, col, stride, sph, n_sph);
    else v = self_field<real>(tm, tm.n_points, col, stride);
    if (b < batch) out[b] = v;
}

// LinkDistanceField.distances / compute_collision / compute_distance (fields.py:40-61) and the same three of
// LinkSelfDistanceField (fields.py:100-112) on explicit frames [B,L,4,4]; no interpolated points there.
//   other = spheres [O,4]:  D[l][o] = |p_l - c_o| - r_o                       (n_other = O)
//   other = null (self):    D[i][j] = |p_i - p_j|                             (n_other = L)
//   mode 0: out [B,L,n_other] = D;  mode 1: out [B] = any(D < buffer) -- self: over the pairs i - j >= 2 only
//   (torch.tril(.., diagonal=-2), fields.py:106);  mode 2: out [B] = sum of D.
// One thread per (configuration, link) in mode 0, per configuration otherwise.
template <typename real>
__global__ void link_dist_kernel(const real* __restrict__ frames, long long batch, int n_links,
                                 const real* __restrict__ sph, int n_other, int mode, real buffer,
                                 real* __restrict__ out) {
    using O = RealOps<real>;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = mode == 0 ? batch * n_links : batch;
    if (i >= total) return;
    const long long b = mode == 0 ? i / n_links : i;
    const real* f = frames + (size_t)b * n_links * 16;
    const int l0 = mode == 0 ? (int)(i - b * n_links) : 0, l1 = mode == 0 ? l0 + 1 : n_links;
    real acc = 0;
    bool hit = false;
    for (int l = l0; l < l1; ++l) {
        const real px = f[l * 16 + 3], py = f[l * 16 + 7], pz = f[l * 16 + 11];
        for (int o = 0; o < n_other; ++o) {
            real cx, cy, cz, r = 0;
            if (sph) { cx = sph[o * 4]; cy = sph[o * 4 + 1]; cz = sph[o * 4 + 2]; r = sph[o * 4 + 3]; }
            else { cx = f[o * 16 + 3]; cy = f[o * 16 + 7]; cz = f[o * 16 + 11]; }
            const real dx = px - cx, dy = py - cy, dz = pz - cz;
            const real dist = O::sqrt_(dx * dx + dy * dy + dz * dz) - r;
            if (mode == 0) out[(size_t)i * n_other + o] = dist;
            acc += dist;
            if (dist < buffer && (sph || l - o >= 2)) hit = true;
        }
    }
    if (mode == 1) out[i] = hit ? (real)1 : (real)0;
    if (mode == 2) out[i] = acc;
}

hipError_t launch_link_dist(int dtype, const void* frames, long long batch, int n_links, const void* spheres,
                            int n_other, int mode, double buffer, void* out, hipStream_t stream) {
    const long long total = mode == 0 ? batch * n_links : batch;
    if (total <= 0) return hipSuccess;
    const int block = 64;
    const unsigned grid = (unsigned)((total + block - 1) / block);
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((link_dist_kernel<double>), dim3(grid), dim3(block), 0, stream, (const double*)frames, batch,
                           n_links, (const double*)spheres, n_other, mode, buffer, (double*)out);
    else
        hipLaunchKernelGGL((link_dist_kernel<float>), dim3(grid), dim3(block), 0, stream, (const float*)frames, batch,
                           n_links, (const float*)spheres, n_other, mode, (float)buffer, (float*)out);
    return hipGetLastError();
}

// EESE3DistanceField.compute_cost on explicit frames [B,L,4,4]: distance of the LAST link frame.
template <typename real>
__global__ void ee_field_kernel(EeTarget<real> tg, const real* __restrict__ frames, long long batch, int n_links,
                                real* __restrict__ out) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const real* f = frames + ((size_t)b * n_links + (n_links - 1)) * 16;
    real R[9], p[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = f[r * 4 + c];
        p[r] = f[r * 4 + 3];
    }
    real dist = se3_distance<real>(R, p, tg);
    if (tg.square) dist = dist * dist;
    out[b] = dist;
}

// ---------------------------------------------------------------------------------- field Jacobians
// FieldFactor.get_error(calc_jacobian=True) (field_factor.py:28-38): the field value at a batch of
// joint configurations q [B,n] and its gradient with respect to q, analytically instead of through
// autograd.  The field is a function of the link positions p_l(q); with g_l = df/dp_l,
//   df/dq_j = z_j . sum_{l after joint j} (p_l - o_j) x g_l = z_j . (M_j - o_j x F_j),
// F_j = sum g_l, M_j = sum p_l x g_l over the links behind revolute joint j (axis z_j through o_j),
// accumulated from the end effector backwards.  Interpolated link points (fields.py:68-74) hand their
// gradient to the two links they lie between.  One thread per configuration; any joint order.
// NJ > 0: chain length known at compile time and no interpolated points -- every loop unrolls and the
// per-thread arrays live in registers (NJ = 0 is the generic form, whose arrays go to scratch).
template <typename real, int NJ>
__global__ void __launch_bounds__(64)
field_grad_kernel(const ChainDev* __restrict__ ch, TermK<real> tm, int n, const real* __restrict__ q,
                  long long batch, int traj_T, const real* __restrict__ spheres, int n_spheres,
                  real* __restrict__ value, real* __restrict__ grad) {
    using O = RealOps<real>;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    // traj_T = 0: q is [B, n].  traj_T = T: q is a trajectory batch [P, T, 2n] and configuration b is
    // waypoint 1 + b % (T-1) of trajectory b / (T-1) (the waypoint range of CostCollision)
    const real* qb = traj_T > 0 ? q + ((size_t)(b / (traj_T - 1)) * traj_T + 1 + b % (traj_T - 1)) * 2 * n
                                : q + (size_t)b * n;
    const int nj = NJ > 0 ? NJ : ch->n_joints;
    constexpr int MP = NJ > 0 ? NJ + 1 : SGPMP_MAX_POINTS;
    constexpr int MJ = NJ > 0 ? NJ : SGPMP_MAX_JOINTS;
    real P[MP][3], G[MP][3];
    real Z[MJ][3], Oj[MJ][3];
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    P[0][0] = P[0][1] = P[0][2] = 0;
#pragma unroll
    for (int j = 0; j < nj; ++j) {
        const JointDev& J = ch->j[j];
        real F[9], tt[3], Rn[9];
        for (int i = 0; i < 9; ++i) F[i] = (real)J.R[i];
        for (int i = 0; i < 3; ++i) tt[i] = (real)J.t[i];
        for (int r = 0; r < 3; ++r) p[r] += R[r * 3 + 0] * tt[0] + R[r * 3 + 1] * tt[1] + R[r * 3 + 2] * tt[2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rn[r * 3 + c] = R[r * 3 + 0] * F[c] + R[r * 3 + 1] * F[3 + c] + R[r * 3 + 2] * F[6 + c];
        for (int r = 0; r < 3; ++r) { Z[j][r] = Rn[r * 3 + 2]; Oj[j][r] = p[r]; }     // joint axis and origin
        if (J.revolute) {
            real sn, cs;
            O::sincos_(qb[J.qidx], &sn, &cs);
            for (int r = 0; r < 3; ++r) {
                const real aa = Rn[r * 3 + 0], bb = Rn[r * 3 + 1];
                Rn[r * 3 + 0] = aa * cs + bb * sn;
                Rn[r * 3 + 1] = bb * cs - aa * sn;
            }
        }
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        for (int r = 0; r < 3; ++r) P[j + 1][r] = p[r];
    }
    const int n_links = nj + 1;
    int npts = n_links;
    if (NJ == 0 && tm.n_interp > 0)
        for (int i = tm.interp_lo; i < tm.interp_hi; ++i)
            for (int a = 0; a < tm.n_interp; ++a, ++npts)
                for (int r = 0; r < 3; ++r) P[npts][r] = P[i][r] + (P[i + 1][r] - P[i][r]) * tm.alpha[a];
#pragma unroll
    for (int l = 0; l < npts; ++l) G[l][0] = G[l][1] = G[l][2] = 0;
    real val = 0;
    if (tm.kind == SGPMP_COST_SPHERES && (tm.flags & 15) == SGPMP_FIELD_SDF) {
        // sdf (fields.py:79-83): value = max over (point, sphere) of r - dist (clamped from above at 0 with
        // clamp_sdf).  The reference differentiates it by autograd (field_factor.py:35): the gradient is that of
        // the arg-max pair -- torch's max(-1)[0].max(-1)[0] keeps the FIRST maximum, spheres within a point
        // first, then points, hence point-major order and a strict comparison here -- i.e. -(p - c)/dist on that
        // point, and exactly zero where the clamp is active (clamp(max=0) passes the gradient for sdf <= 0).
        const bool clampv = (tm.flags & SGPMP_FLAG_SDF_CLAMP) != 0;
        real best = (real)-__builtin_inff();
        int bl = 0;
        real bg[3] = {0, 0, 0};
#pragma unroll
        for (int l = 0; l < npts; ++l)
            for (int o = 0; o < n_spheres; ++o) {
                const real dx = P[l][0] - spheres[o * 4], dy = P[l][1] - spheres[o * 4 + 1], dz = P[l][2] - spheres[o * 4 + 2];
                const real dist = O::sqrt_(dx * dx + dy * dy + dz * dz);
                real sd = spheres[o * 4 + 3] - dist;
                const bool cut = clampv && sd > (real)0;
                if (cut) sd = 0;
                if (sd > best) {
                    best = sd; bl = l;
                    const real w = cut ? (real)0 : (real)-1 / dist;
                    bg[0] = w * dx; bg[1] = w * dy; bg[2] = w * dz;
                }
            }
        val = best;
#pragma unroll
        for (int l = 0; l < npts; ++l)
            if (l == bl) { G[l][0] = bg[0]; G[l][1] = bg[1]; G[l][2] = bg[2]; }
    } else if (tm.kind == SGPMP_COST_SPHERES) {              // rbf (occupancy is refused by the host)
        for (int o = 0; o < n_spheres; ++o) {
            const real cx = spheres[o * 4], cy = spheres[o * 4 + 1], cz = spheres[o * 4 + 2], rr = spheres[o * 4 + 3];
            const real ir2 = (real)1 / (rr * rr);
#pragma unroll
            for (int l = 0; l < npts; ++l) {
                const real dx = P[l][0] - cx, dy = P[l][1] - cy, dz = P[l][2] - cz;
                const real e = O::exp_((real)-0.5 * (dx * dx + dy * dy + dz * dz) * ir2);
                val += e;
                const real w = -e * ir2;
                G[l][0] += w * dx; G[l][1] += w * dy; G[l][2] += w * dz;
            }
        }
    } else {                                                 // SELF: the full L x L sum of fields.py:124
        // = diagonal (value 1 each, no gradient) + twice the strict lower triangle
        val = (real)npts;
#pragma unroll
        for (int i = 1; i < npts; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const real dx = P[i][0] - P[j][0], dy = P[i][1] - P[j][1], dz = P[i][2] - P[j][2];
                const real e = O::exp_((dx * dx + dy * dy + dz * dz) * tm.K2);
                val += (real)2 * e;
                const real w = (real)4 * tm.K2 * e;
                G[i][0] += w * dx; G[i][1] += w * dy; G[i][2] += w * dz;
                G[j][0] -= w * dx; G[j][1] -= w * dy; G[j][2] -= w * dz;
            }
    }
    if (NJ == 0 && tm.n_interp > 0) {                                   // interpolated points -> their two links
        int m = n_links;
        for (int i = tm.interp_lo; i < tm.interp_hi; ++i)
            for (int a = 0; a < tm.n_interp; ++a, ++m)
                for (int r = 0; r < 3; ++r) {
                    G[i][r] += ((real)1 - tm.alpha[a]) * G[m][r];
                    G[i + 1][r] += tm.alpha[a] * G[m][r];
                }
    }
    if (value) value[b] = val;
    real Fs[3] = {0, 0, 0}, Ms[3] = {0, 0, 0};
    real* gb = grad + (size_t)b * n;
    for (int k = 0; k < n; ++k) gb[k] = 0;
#pragma unroll
    for (int j = nj - 1; j >= 0; --j) {
        const int l = j + 1;
        Fs[0] += G[l][0]; Fs[1] += G[l][1]; Fs[2] += G[l][2];
        Ms[0] += P[l][1] * G[l][2] - P[l][2] * G[l][1];
        Ms[1] += P[l][2] * G[l][0] - P[l][0] * G[l][2];
        Ms[2] += P[l][0] * G[l][1] - P[l][1] * G[l][0];
        if (ch->j[j].revolute) {
            const real ox = Oj[j][0], oy = Oj[j][1], oz = Oj[j][2];
            const real tx = Ms[0] - (oy * Fs[2] - oz * Fs[1]);
            const real ty = Ms[1] - (oz * Fs[0] - ox * Fs[2]);
            const real tz = Ms[2] - (ox * Fs[1] - oy * Fs[0]);
            gb[ch->j[j].qidx] += Z[j][0] * tx + Z[j][1] * ty + Z[j][2] * tz;
        }
    }
}

hipError_t launch_field_grad(int dtype, int n, const CostTerm& term, const ChainDev* d_chain, int n_joints,
                             const void* q, long long batch, int traj_T, const void* spheres, int n_spheres,
                             void* value, void* grad, hipStream_t stream) {
    const int block = 64;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
#define FG_LAUNCH(REAL, NJ)                                                                         \
    hipLaunchKernelGGL((field_grad_kernel<REAL, NJ>), dim3(grid), dim3(block), 0, stream, d_chain,  \
                       make_termk<REAL>(term), n, (const REAL*)q, batch, traj_T, (const REAL*)spheres, \
                       n_spheres, (REAL*)value, (REAL*)grad)
    const int nj = term.n_interp > 0 ? 0 : n_joints;       // specialised chain lengths: 10 (Panda), 7
    if (dtype == SGPMP_F64) {
        if (nj == 10) FG_LAUNCH(double, 10); else if (nj == 7) FG_LAUNCH(double, 7); else FG_LAUNCH(double, 0);
    } else {
        if (nj == 10) FG_LAUNCH(float, 10); else if (nj == 7) FG_LAUNCH(float, 7); else FG_LAUNCH(float, 0);
    }
#undef FG_LAUNCH
    return hipGetLastError();
}

hipError_t launch_field_eval(int dtype, const CostTerm& term, const void* frames, long long batch,
                             int n_links, const void* spheres, int n_spheres, void* out,
                             hipStream_t stream) {
    const int block = 64;
    const unsigned grid = (unsigned)((batch + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (term.kind == SGPMP_COST_EE_GOAL) {
        if (dtype == SGPMP_F64)
            hipLaunchKernelGGL((ee_field_kernel<double>), dim3(grid), dim3(block), 0, stream,
                               make_ee_target<double>(term), (const double*)frames, batch, n_links, (double*)out);
        else
            hipLaunchKernelGGL((ee_field_kernel<float>), dim3(grid), dim3(block), 0, stream,
                               make_ee_target<float>(term), (const float*)frames, batch, n_links, (float*)out);
        return hipGetLastError();
    }
    const size_t esz = dtype == SGPMP_F64 ? 8 : 4;
    const size_t lds = (size_t)term.n_points * 3 * block * esz;
    if (dtype == SGPMP_F64)
        hipLaunchKernelGGL((field_eval_kernel<double>), dim3(grid), dim3(block), lds, stream,
                           make_termk<double>(term), (const double*)frames, batch, n_links,
                           (const double*)spheres, n_spheres, (double*)out);
    else
        hipLaunchKernelGGL((field_eval_kernel<float>), dim3(grid), dim3(block), lds, stream,
                           make_termk<float>(term), (const float*)frames, batch, n_links,
                           (const float*)spheres, n_spheres, (float*)out);
    return hipGetLastError();
}
