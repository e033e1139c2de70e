"""Planner-level parity of the HIP path: StochGPMP.optimize() against the reference's captured
runs (identical seeds / noise) and against the oracle, plus size-independent properties at the
full BASELINE sizes.  Needs the MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch

from tests import scenarios as SC
from tests.hip_builders import hip_panda_planner, hip_planar_planner, hip_panda_cost

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
F64 = {"device": DEV, "dtype": torch.float64}
F32 = {"device": DEV, "dtype": torch.float32}



def rel_err(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / np.abs(b).max())


def planar_map(golden, ta):
    from stoch_gpmp_amd.envs.obst_map import ObstacleMap
    g = golden("g2_planar_e2e.npz")
    return ObstacleMap.from_grid(g["grid"], float(g["cell_size"]), tensor_args=ta)


# --------------------------------------------------------------------------- config 1, fp64
def test_config1_identical_seed_matches_reference_run(golden):
    """BASELINE config 1 (planar, 2 goals x 2 particles, S=16, T=64, fp64, seed 0): replaying the
    reference's noise stream from the same seed reproduces its particle means (1e-5 rel target)."""
    z = golden("g2_planar_e2e.npz")
    T, nppg, S, seed, n_iters = [int(v) for v in z["dims"]]
    pl = hip_planar_planner(SC.PLANAR, T, z["goals"], nppg, S, planar_map(golden, F64), F64,
                            seed=seed, noise='torch')
    assert rel_err(pl.particle_means, z["means_reset"]) < 1e-7
    for it in range(1, n_iters + 1):
        sp, cp, st, cs, costs, grad = pl.optimize()
        if it <= 3:
            assert rel_err(costs, z[f"costs_{it}"]) < 1e-7
            assert rel_err(grad, z[f"grad_{it}"]) < 1e-6
            assert rel_err(pl._weights.reshape(pl.num_particles, S), z[f"weights_{it}"]) < 1e-6
        if it == 1:
            assert rel_err(pl.state_samples[0, 0], z["samples_1_p0_s0"]) < 1e-7
            assert rel_err(pl.state_samples[3, 15], z["samples_1_p3_s15"]) < 1e-7
            assert rel_err(sp, z["ret_state_particles_1"]) < 1e-7          # pre-update means
            assert st.shape == (4, S, T, 2) and cs.shape == (4, S, T, 2) and cp.shape == (4, T, 2)
        if it in (1, 2, 3, 10):
            assert rel_err(pl.particle_means, z[f"means_{it}"]) < 1e-5
    assert rel_err(pl.particle_means, z["means_10"]) < 1e-6
    tr, ctl = pl.get_recent_samples()
    assert tr.shape == (4, S, T, 2) and ctl.shape == (4, S, T, 2)


def test_config1_committed_eps_matches_reference_run(golden):
    """Same run, but fed the committed eps tensors (no reliance on torch's generator)."""
    z = golden("g2_planar_e2e.npz")
    T, nppg, S, seed, n_iters = [int(v) for v in z["dims"]]
    pl = hip_planar_planner(SC.PLANAR, T, z["goals"], nppg, S, planar_map(golden, F64), F64,
                            seed=seed, noise='torch')
    for it in (1, 2, 3):
        eps = torch.as_tensor(z[f"eps_{it}"]).to(**F64)
        pl._draw_eps = lambda e=eps: e
        pl.optimize()
        assert rel_err(pl.particle_means, z[f"means_{it}"]) < 1e-6


def test_per_particle_sampling_precisions_inside_the_loop_match_reference_run(golden):
    """g10 (round-4 verdict, item 4): `planner._sample_dist` is a live MultiMPPrior over the planner's own context; after
    `planner._sample_dist.set_Sigma_invs(S)` -- one precision matrix per particle: the prior with its GP blocks scaled by
    1 + 0.1 p, built by get_const_vel_covariance -- the loop samples particle p from factor p (sample_dense_kernel: per-mode
    factorisation by K1 on the fp64 matrix cores) while the importance-sampling term keeps `planner.Sigma_inv`, exactly
    as the reference's run did (planner.py:218-237, mp_priors_multi.py:120-128).  Also: shards of the same problem agree
    with the unsharded run bit for bit, and reset() returns to the shared prior."""
    z = golden("g10_per_particle_precisions.npz")
    T, nppg, S, seed, n_iters = [int(v) for v in z["dims"]]

    def build(**kw):
        pl = hip_planar_planner(SC.PLANAR, T, z["goals"], nppg, S, planar_map(golden, F64), F64, seed=seed, noise='torch', **kw)
        return pl

    def per_particle(pl):
        sd = pl._sample_dist
        K_s, K_g, Q = pl.start_prior_sample.K, pl.multi_goal_prior_sample[0].K, pl.gp_prior_sample.Q_inv[0]
        f = z["gp_scale_per_particle"][pl.p0:pl.p1]
        new = torch.stack([sd.get_const_vel_covariance(pl.dt, K_s, Q * float(v), K_g) for v in f])
        assert new.shape == sd.Sigma_invs.shape == (pl.num_particles_local, T * 4, T * 4)
        return new

    pl = build()
    assert rel_err(pl.particle_means, z["means_reset"]) < 1e-7
    sd = pl._sample_dist
    assert sd.num_modes == 4 and rel_err(sd.Sigma_inv[:8, :12], z["Sigma_inv_planner_rows_0_8"]) < 1e-12
    new = per_particle(pl)
    assert rel_err(new[3, :8, :12], z["Sigma_invs_p3_rows_0_8"]) < 1e-12
    sd.set_Sigma_invs(new)
    assert torch.equal(sd.Sigma_invs, new) and rel_err(pl.Sigma_inv[:8, :12], z["Sigma_inv_planner_rows_0_8"]) < 1e-12
    full = []
    for it in (1, 2, 3):
        eps = torch.as_tensor(z[f"eps_{it}"]).to(**F64)
        pl._draw_eps = lambda e=eps: e
        costs = pl.optimize()[4]
        assert pl._engine.last_cost_kernel().startswith("cost_sweep_kernel")          # (sampler and sweep as separate launches)
        assert rel_err(costs, z[f"costs_{it}"]) < 1e-7
        assert rel_err(pl.particle_means, z[f"means_{it}"]) < 1e-7
        if it == 1:
            assert rel_err(pl.state_samples[3, 5], z["samples_1_p3_s5"]) < 1e-7
            assert rel_err(pl.state_samples[0, 0], z["samples_1_p0_s0"]) < 1e-7
        full.append(pl.particle_means.clone())
    # shards: every rank factors its own particles' precisions; same numbers as the unsharded run, bit for bit
    for r in range(2):
        sh = build(rank=r, world_size=2)
        sh._sample_dist.set_Sigma_invs(per_particle(sh))
        for it in (1, 2, 3):
            eps = torch.as_tensor(z[f"eps_{it}"]).to(**F64)
            sh._draw_eps = lambda e=eps: e
            sh.optimize()
            assert torch.equal(sh.particle_means, full[it - 1][sh.p0:sh.p1]), (r, it)
    # the in-kernel noise takes the same route (fp32 too), and reset() returns to the shared prior and the fused launch
    p32 = hip_planar_planner(SC.PLANAR, T, z["goals"], nppg, S, planar_map(golden, F32), F32, seed=seed)
    p32._sample_dist.set_Sigma_invs(per_particle(p32).to(**F32))
    p32.optimize(opt_iters=3)
    assert p32._engine.last_cost_kernel().startswith("cost_sweep_kernel") and torch.isfinite(p32.particle_means).all()
    p32.reset()
    p32.optimize(opt_iters=2)
    assert p32._engine.last_cost_kernel().startswith("fused_planar"), p32._engine.last_cost_kernel()
    # a precision outside the band, or not positive definite, is refused as torch refuses it in the reference
    bad = new.clone()
    bad[0, 0, -1] = bad[0, -1, 0] = 1.0
    with pytest.raises(ValueError):
        sd.set_Sigma_invs(bad)
    with pytest.raises(ValueError):
        sd.set_Sigma_invs(-new)


def test_soft_weights_const_vel_matches_reference_run(golden):
    z, g = golden("g2b_planar_constvel_soft.npz"), golden("g2_planar_e2e.npz")
    T, nppg, S, seed, n_iters = [int(v) for v in z["dims"]]
    dt, css, csg, scoll, sgp, sss, sgs, sgps = [float(v) for v in z["sigmas"]]
    c = dict(SC.PLANAR, start=list(z["start"]), dt=dt, cost_sigma_start=css, cost_sigma_gp=csg,
             sigma_coll=scoll, sigma_goal_prior=sgp, sigma_start_sample=sss, sigma_goal_sample=sgs,
             sigma_gp_sample=sgps)
    pl = hip_planar_planner(c, T, z["goals"], nppg, S, planar_map(golden, F64), F64,
                            initial_particle_means='const_vel', temperature=float(z["temperature"]),
                            seed=1, noise='torch')
    assert rel_err(pl.particle_means, z["means_reset"]) < 1e-14
    for it in (1, 2, 3):
        eps = torch.as_tensor(z[f"eps_{it}"]).to(**F64)
        pl._draw_eps = lambda e=eps: e
        _, _, _, _, costs, grad = pl.optimize()
        assert rel_err(costs, z[f"costs_{it}"]) < 1e-9
        assert rel_err(pl._weights.reshape(pl.num_particles, S), z[f"weights_{it}"]) < 1e-8
        assert rel_err(grad, z[f"grad_{it}"]) < 1e-8
        assert rel_err(pl.particle_means, z[f"means_{it}"]) < 1e-9


# --------------------------------------------------------------------------- Panda, fp64
@pytest.mark.parametrize("field_type", ["rbf", "sdf"])
def test_panda_small_matches_oracle_fp64(field_type):
    c = SC.PANDA
    T, nppg, S, iters = 16, 3, 6, 4
    sph = torch.as_tensor(SC.panda_spheres()).to(**F64)
    torch.manual_seed(3)
    ora = SC.oracle_panda_planner(c, T, nppg, S, field_type=field_type, seed=3)
    ora.draw_discarded()
    pl = hip_panda_planner(c, T, nppg, S, F64, field_type=field_type, seed=3, noise='torch')
    assert rel_err(pl.particle_means, ora.particle_means) < 1e-7
    for it in range(iters):
        st = torch.get_rng_state()
        costs_o, grad_o = ora.step(obstacle_spheres=sph.cpu())
        torch.set_rng_state(st)                          # HIP side replays the same draw
        _, _, _, _, costs, grad = pl.optimize(obstacle_spheres=sph)
        assert rel_err(costs, costs_o) < 1e-8
        assert rel_err(pl.particle_means, ora.particle_means) < 1e-6


def test_panda_default_native_noise_matches_oracle_on_the_restated_stream():
    """The DEFAULT mode (noise='philox', nothing fed from the host): the oracle planner driven by the
    CPU restatement of the in-kernel stream (oracle/native_noise.py, pinned by the Random123 vectors)
    must follow the HIP planner -- initial particle means, costs and means over the iterations."""
    from oracle.native_noise import native_eps
    c = SC.PANDA
    T, nppg, S, iters, n, seed = 16, 4, 6, 4, 7, 11
    sph = torch.as_tensor(SC.panda_spheres()).to(**F64)
    pl = hip_panda_planner(c, T, nppg, S, F64, seed=seed)                              # noise='philox'
    # The oracle is fed the eps the kernels drew, read back through sgpmp_noise (fp64 contexts draw the fp32 stream, widened:
    # the hardware's log2 / sin / cos are approximations the numpy restatement follows to an ulp of fp32, which the 1e-7 bound on
    # the costs below would see) -- and that eps is held to the restatement right here.
    eps0 = pl._engine.noise(seed, 0, 1, nppg).cpu()                                    # [nppg, G, M]
    assert float((eps0 - torch.from_numpy(native_eps(seed, 0, range(1), nppg, T, n, "float64"))).abs().max()) < 2e-6
    ora = SC.oracle_panda_planner(c, T, nppg, S, seed=seed, eps_init=eps0)
    assert rel_err(pl.particle_means, ora.particle_means) < 1e-7
    for it in range(iters):
        eps = pl._engine.noise(seed, 2 + it, nppg, S).cpu()                            # draw 1 is discarded
        assert float((eps - torch.from_numpy(native_eps(seed, 2 + it, range(nppg), S, T, n, "float64"))).abs().max()) < 2e-6
        costs_o, grad_o = ora.step(eps=eps, obstacle_spheres=sph.cpu())
        _, _, _, _, costs, grad = pl.optimize(obstacle_spheres=sph)
        assert rel_err(costs, costs_o) < 1e-7
        assert rel_err(pl.particle_means, ora.particle_means) < 1e-6


def test_panda_multigoal_long_horizon_matches_oracle_fp64():
    """BASELINE config 5 shape in miniature: Panda, 2 goals, T = 128 (two 64-waypoint passes per
    wave, goal lookup by particle // nppg), fp64, sdf field, against the oracle on the same noise."""
    c = SC.PANDA
    T, nppg, S = 128, 2, 5
    n = 7
    goals = [c["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n]
    sph = torch.as_tensor(SC.panda_spheres(num=7, seed=4)).to(**F64)
    ora = SC.oracle_panda_planner(c, T, nppg, S, field_type='sdf', seed=8, goals=goals)
    ora.draw_discarded()
    pl = hip_panda_planner(c, T, nppg, S, F64, field_type='sdf', seed=8, noise='torch', goals=goals)
    assert pl.num_particles == 4 and rel_err(pl.particle_means, ora.particle_means) < 1e-6
    for it in range(3):
        st = torch.get_rng_state()
        costs_o, _ = ora.step(obstacle_spheres=sph.cpu())
        torch.set_rng_state(st)
        _, _, _, _, costs, _ = pl.optimize(obstacle_spheres=sph)
        assert rel_err(costs, costs_o) < 1e-8
        assert rel_err(pl.particle_means, ora.particle_means) < 1e-6


# --------------------------------------------------------------------------- fp32 compute path
def test_planar_fp32_against_fp64_oracle_same_noise(golden):
    """fp32 kernels (prior factored in fp64) fed the oracle's noise, compared with the fp64 oracle:
    samples and costs agree to fp32 accuracy.  Because temperature=1 and costs are ~1e9-1e11 the
    update is an arg-min over samples, so particle means either agree to ~1e-4 or (rarely) pick a
    different sample; we require the former for the large majority and report the rest."""
    z = golden("g2_planar_e2e.npz")
    T, nppg, S = 64, 8, 32
    goals = z["goals"]
    om64 = planar_map(golden, F64)
    torch.manual_seed(21)
    ora = SC.oracle_planar_planner(SC.PLANAR, T, goals, nppg, S, z["grid"], float(z["cell_size"]),
                                   z["c_offset"], seed=21)
    ora.draw_discarded()
    om32 = planar_map(golden, F32)
    pl = hip_planar_planner(SC.PLANAR, T, goals, nppg, S, om32, F32, seed=21, noise='torch',
                            initial_particle_means=ora.particle_means.reshape(2, nppg, T, 4).to(**F32))
    P = pl.num_particles
    agree = np.ones(P, dtype=bool)
    for it in range(5):
        eps = torch.randn(S, P, T * 4, dtype=torch.float64)
        costs_o, _ = ora.step(eps=eps)
        pl._draw_eps = lambda e=eps: e.to(**F32)
        _, _, _, _, costs, _ = pl.optimize()
        scale = float(ora.state_samples.abs().max())
        assert float((pl.state_samples.cpu().double() - ora.state_samples).abs().max()) < 2e-5 * scale
        assert rel_err(costs, costs_o) < 5e-3
        d = (pl.particle_means.cpu().double() - ora.particle_means).abs().amax(dim=(1, 2))
        agree &= (d / ora.particle_means.abs().max()).numpy() < 1e-3
        # keep both sides on the same trajectory so later iterations stay comparable
        pl.particle_means.copy_(ora.particle_means.to(**F32))
    print(f"\n[fp32 parity] planar 16 x 32 x 64, reference noise: particles within 1e-3 over 5 iterations: {agree.mean():.4f}")
    assert agree.mean() >= 0.9, f"only {agree.mean():.2f} of the particles within 1e-3 in EVERY one of 5 iterations"


def _is_dot(samples, w, n, dt):
    """Importance-sampling inner product (A x) . w with A x = (x_0, e_0 .. e_{T-2}, x_{T-1}),
    e_t = x_{t+1} - Phi x_t (planner.py:233-236 in the factored form the kernels use); fp64 torch."""
    x = samples.double()
    pos, vel = x[..., :n], x[..., n:]
    e = torch.cat([pos[..., 1:, :] - pos[..., :-1, :] - dt * vel[..., :-1, :],
                   vel[..., 1:, :] - vel[..., :-1, :]], dim=-1)                      # [P,S,T-1,d]
    Ax = torch.cat([x[..., :1, :], e, x[..., -1:, :]], dim=-2)                       # [P,S,T+1,d]
    return (Ax * w.double().unsqueeze(1)).sum((-1, -2))


def _fp32_panda_run(T, nppg, S, iters, goals=None, field_type='rbf', n_sph=5, seed=13, expect_kernel=None,
                    fused=True):
    """Panda fp32 planner in its DEFAULT mode (in-kernel Philox noise) against the fp64 oracle driven by
    the CPU restatement of that stream.  Means are re-synchronised after every iteration so that each
    iteration is an independent trial of "does the fp32 path move every particle where the fp64
    reference moves it".  Returns per-iteration records."""
    from oracle.native_noise import native_eps
    c, n = SC.PANDA, 7
    G = 1 if goals is None else len(goals)
    P = G * nppg
    sph = torch.as_tensor(SC.panda_spheres(num=n_sph, seed=seed))
    eps0 = torch.from_numpy(native_eps(seed, 0, range(G), nppg, T, n, "float32")).double()
    ora = SC.oracle_panda_planner(c, T, nppg, S, seed=seed, eps_init=eps0, goals=goals, field_type=field_type)
    pl = hip_panda_planner(c, T, nppg, S, F32, seed=seed, goals=goals, field_type=field_type)
    if fused == "small":
        pl._engine.set_option("no_small_step", 0)        # the product's default for <= 512 items: one workgroup per item (conftest.py switches it off)
    elif fused != True:
        pl._engine.set_option("no_fused_step", 1)        # sampler and sweep as two launches
    if fused == "dual":
        pl._engine.set_option("no_chunked_sweep", 1)     # ... and the 64-lane-pass two-trajectory sweep
    assert rel_err(pl.particle_means, ora.particle_means) < 2e-5
    pl.particle_means.copy_(ora.particle_means.to(**F32))
    scale = float(ora.particle_means.abs().max())
    out = []
    for it in range(iters):
        eps = torch.from_numpy(native_eps(seed, 2 + it, range(P), S, T, n, "float32")).double()
        ora.particle_means.copy_(pl.particle_means.cpu().double())      # identical (fp32-representable) means
        ora.prior.set_mean(ora.particle_means.view(P, -1))
        costs_o, _ = ora.step(eps=eps, obstacle_spheres=sph)
        _, _, _, _, costs, _ = pl.optimize(obstacle_spheres=sph.to(**F32))
        if expect_kernel is not None:
            assert pl._engine.last_cost_kernel().startswith(expect_kernel), pl._engine.last_cost_kernel()
        samples_err = float((pl.state_samples.cpu().double() - ora.state_samples).abs().max())
        assert samples_err < 2e-5 * float(ora.state_samples.abs().max())
        # the fp32 kernel's arithmetic alone: fp64 sweep of the very same (fp32-rounded) samples
        e64 = pl.cost._engine(torch.float64, DEV)
        exact = e64.cost_eval(pl.state_samples.double().contiguous(), batch_offset=0,
                              spheres=sph.to(**F64).reshape(-1, 4).contiguous()).reshape(P, S).cpu()
        w = pl._engine.is_weights(pl._means_prev.contiguous(), pl.temperature)      # pre-update means
        exact = exact + _is_dot(pl.state_samples.cpu(), w.cpu(), n, c["dt"])
        c32 = costs.cpu().double()
        d = (pl.particle_means.cpu().double() - ora.particle_means).abs().amax(dim=(1, 2)) / scale
        out.append(dict(
            cost_rel=float(((c32 - costs_o).abs() / costs_o.abs()).max()),
            arith_rel=float(((c32 - exact).abs() / exact.abs()).max()),
            rounding_rel=float(((exact - costs_o).abs() / costs_o.abs()).max()),
            argmin_same=float((c32.argmin(1) == costs_o.argmin(1)).double().mean()),
            means_within_1e3=float((d < 1e-3).double().mean()), worst=float(d.max())))
    return out


def _report(tag, recs):
    frac = float(np.mean([r["means_within_1e3"] for r in recs]))
    print(f"\n[fp32 parity] {tag}: particles within 1e-3 of the fp64 means: {frac:.4f} "
          f"(per iteration {[round(r['means_within_1e3'], 4) for r in recs]}); same arg-min {np.mean([r['argmin_same'] for r in recs]):.4f}; "
          f"cost rel err vs oracle {max(r['cost_rel'] for r in recs):.2e} = kernel arithmetic "
          f"{max(r['arith_rel'] for r in recs):.2e} + fp32 rounding of the samples {max(r['rounding_rel'] for r in recs):.2e}")
    return frac


@pytest.mark.parametrize("fused,kernel", [(True, "fused_step_kernel"), ("small", "fused_step_small_kernel"),
                                          ("chunked", "cost_sweep_chunked_kernel"), ("dual", "cost_sweep_dual_pf_kernel")])
def test_panda_fp32_headline_kernel_means_match_fp64_oracle(fused, kernel):
    """north_star: fp32 trajectory means within 1e-3 of the reference CPU path.  The headline kernels
    (the fused sampler + sweep launch, and cost_sweep_dual_pf_kernel behind the separate sampler: even
    S, even T <= 64, rbf) over 4 iterations, 48 particles, native noise.  With temperature = 1 and
    costs of 1e9-1e11 the update is an arg-min over samples, so a particle agrees unless fp32 flips
    the arg-min; the agreeing fraction is printed and bounded."""
    recs = _fp32_panda_run(T=32, nppg=48, S=32, iters=4, expect_kernel=kernel, fused=fused)
    frac = _report(f"Panda 48x32x32 rbf ({kernel})", recs)
    assert max(r["cost_rel"] for r in recs) < 5e-3
    assert frac >= 0.99, frac                          # measured on MI355X: 1.0000 (two-launch kernels), 0.9965 (fused: one near-tie of 288)


@pytest.mark.parametrize("fused,kernel", [(True, "fused_step_kernel"), ("chunked", "cost_sweep_chunked_kernel"),
                                          ("dual", "cost_sweep_dual_pf_multi_kernel")])
def test_panda_fp32_config5_kernel_means_match_fp64_oracle(fused, kernel):
    """BASELINE config 5 in its stated precision and in miniature: 2 goals, T = 128 (eight 16-waypoint
    chunks of the fused launch / two passes of cost_sweep_dual_pf_multi_kernel with the carried
    neighbour waypoint), fp64 prior + fp32 cost path, GP + multi-goal prior + IS + self + sphere fields,
    against the fp64 oracle: costs and means."""
    n = 7
    goals = [SC.PANDA["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n]
    recs = _fp32_panda_run(T=128, nppg=6, S=16, iters=2, goals=goals, n_sph=7, expect_kernel=kernel, fused=fused)
    frac = _report(f"Panda 2 goals x 6 x 16 x 128 rbf ({kernel})", recs)
    assert max(r["cost_rel"] for r in recs) < 5e-3
    assert frac >= 0.99, frac                          # measured: 1.0000
    # T = 66 is not a multiple of the fused launch's 16-waypoint chunk: its last chunk holds two waypoints (round 4: the launch
    # masks the rest; before, such T always ran the two-launch path) -- against the fp64 oracle like every other shape
    recs = _fp32_panda_run(T=66, nppg=6, S=16, iters=2, goals=goals, n_sph=7, field_type='sdf',
                           expect_kernel="fused_step_kernel" if fused is True else "cost_sweep_dual_pf_multi_kernel", fused=fused)
    frac = _report("Panda 2 goals x 6 x 16 x 66 sdf", recs)
    assert max(r["cost_rel"] for r in recs) < 5e-3 and frac >= 0.99


def test_planar_fp32_means_match_fp64_oracle_native_noise(golden):
    """The planar twin of the test above (config 2's kernels: fp32 generic sweep with the grid lookup)."""
    from oracle.native_noise import native_eps
    z = golden("g2_planar_e2e.npz")
    T, nppg, S, n, seed = 64, 32, 32, 2, 5
    goals = z["goals"]
    G = len(goals)
    P = G * nppg
    eps0 = torch.from_numpy(native_eps(seed, 0, range(G), nppg, T, n, "float32")).double()
    ora = SC.oracle_planar_planner(SC.PLANAR, T, goals, nppg, S, z["grid"], float(z["cell_size"]),
                                   z["c_offset"], seed=seed, eps_init=eps0)
    pl = hip_planar_planner(SC.PLANAR, T, goals, nppg, S, planar_map(golden, F32), F32, seed=seed)
    scale = float(ora.particle_means.abs().max())
    fr, same = [], []
    for it in range(6):
        eps = torch.from_numpy(native_eps(seed, 2 + it, range(P), S, T, n, "float32")).double()
        ora.particle_means.copy_(pl.particle_means.cpu().double())
        ora.prior.set_mean(ora.particle_means.view(P, -1))
        costs_o, _ = ora.step(eps=eps)
        _, _, _, _, costs, _ = pl.optimize()
        assert rel_err(costs, costs_o) < 5e-3
        d = (pl.particle_means.cpu().double() - ora.particle_means).abs().amax(dim=(1, 2)) / scale
        fr.append(float((d < 1e-3).double().mean()))
        same.append(float((costs.cpu().argmin(1) == costs_o.argmin(1)).double().mean()))
    print(f"\n[fp32 parity] planar 64x32x64: particles within 1e-3: {np.mean(fr):.4f} {fr}; same arg-min {np.mean(same):.4f}")
    assert np.mean(fr) >= 0.99                         # measured: 1.0000 (arg-min identical for 99.7 %)


def test_planar_one_call_of_several_iterations_against_the_oracle(golden):
    """Round 6: optimize(opt_iters = K) of a planar problem with 64 samples per particle runs iterations 0 .. K - 2 in ONE launch
    (fused_planar_seg.inc: PERSIST) -- nothing of them is visible from outside, so the oracle meets the call's END state: K free
    iterations of the fp64 oracle from the same means on the restated noise stream (draws 2 .. 2 + K - 1), particle for particle.
    (fp32 against fp64 over K free iterations: a particle whose two best samples tie to fp32 rounding takes the other one and
    leaves -- the tests above count such flips; here: >= 97 % of the particles within 1e-3 after 6 iterations.)"""
    from oracle.native_noise import native_eps
    z = golden("g2_planar_e2e.npz")
    T, nppg, S, n, seed, K = 64, 16, 64, 2, 9, 6
    goals = z["goals"]
    G = len(goals)
    P = G * nppg
    eps0 = torch.from_numpy(native_eps(seed, 0, range(G), nppg, T, n, "float32")).double()
    ora = SC.oracle_planar_planner(SC.PLANAR, T, goals, nppg, S, z["grid"], float(z["cell_size"]),
                                   z["c_offset"], seed=seed, eps_init=eps0)
    pl = hip_planar_planner(SC.PLANAR, T, goals, nppg, S, planar_map(golden, F32), F32, seed=seed)
    ora.particle_means.copy_(pl.particle_means.cpu().double())
    ora.prior.set_mean(ora.particle_means.view(P, -1))
    scale = float(ora.particle_means.abs().max())
    _, _, _, _, costs, _ = pl.optimize(opt_iters=K)
    assert pl._engine.last_cost_kernel() == "fused_planar_seg_kernel" and pl._engine.multi_iteration_launches() == 1
    assert pl._engine.store_free_steps() == K - 1
    for it in range(K):
        eps = torch.from_numpy(native_eps(seed, 2 + it, range(P), S, T, n, "float32")).double()
        costs_o, _ = ora.step(eps=eps)
    d = (pl.particle_means.cpu().double() - ora.particle_means).abs().amax(dim=(1, 2)) / scale
    frac = float((d < 1e-3).double().mean())
    same = float((costs.cpu().argmin(1) == costs_o.argmin(1)).double().mean())
    print(f"\n[fp32 parity] planar {P} x {S} x {T}, one call of {K} iterations: particles within 1e-3 of the oracle's free run: {frac:.4f}; "
          f"same arg-min in the last iteration {same:.4f}; median departure {float(d.median()):.2e}")
    assert frac >= 0.97 and float(d.median()) < 1e-5


# --------------------------------------------------------------------------- BASELINE's stated sizes, directly
# The tests above meet the oracle in miniature and carry the result to the full sizes through kernel-vs-kernel
# identities.  These run the HIP planner AT the sizes BASELINE.json states and check its particles directly against the
# oracle (configs 2, 3 and config 5's share: EVERY particle, test_whole_population_parity_* below -- the handful-of-particles
# tests of rounds 3-4 for those configurations were strictly weaker and went in round 6; config 4: a shard's end particles): in-kernel noise is keyed on the GLOBAL particle index, so any particle of the big
# run is reproducible on its own -- the oracle gets oracle.native_noise.native_eps for exactly those indices and
# the same (fp32-representable) means, and must return the same costs [p, :] and the same updated means.
_PARITY_LOG = {}


def _record_parity(tag, rec):
    """Measured fractions -> gpurun_out/parity_full_size.json (copied to profiles/rNN/ and quoted by bench.py)."""
    import json
    import os
    _PARITY_LOG[tag] = rec
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        path = os.path.join(out, "parity_full_size.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old.update(_PARITY_LOG)
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)


def _check_subset(tag, pl, local_idx, oracle_step, iters, obs, expect_kernel):
    """Run `iters` iterations of the full-size HIP planner; after each, compare particles `local_idx` with
    oracle_step(means [k,T,d] fp64, global indices, draw) -> (costs [k,S], samples [k,S,T,d], new means [k,T,d])."""
    idx = torch.as_tensor(local_idx, device=DEV)
    scale = None
    trials = ok = flips = 0
    worst_cost = worst_samples = worst_means = 0.0
    for it in range(iters):
        draw = pl._draw
        mu = pl.particle_means[idx].cpu().double()
        costs_o, samples_o, means_o = oracle_step(mu, [pl.p0 + i for i in local_idx], draw)
        _, _, _, _, costs, _ = pl.optimize(**obs)
        assert pl._engine.last_cost_kernel().startswith(expect_kernel), pl._engine.last_cost_kernel()
        scale = scale or float(means_o.abs().max())
        x = pl.state_samples[idx].cpu().double()
        worst_samples = max(worst_samples, float((x - samples_o).abs().max() / samples_o.abs().max()))
        c32 = costs[idx].cpu().double()
        worst_cost = max(worst_cost, float(((c32 - costs_o).abs() / costs_o.abs()).max()))
        d = (pl.particle_means[idx].cpu().double() - means_o).abs().amax(dim=(1, 2)) / scale
        worst_means = max(worst_means, float(d.max()))
        for k in range(len(local_idx)):
            trials += 1
            if float(d[k]) < 1e-3:
                ok += 1
                continue
            # the update is an arg-min over samples (temperature 1, costs of 1e9..1e11): a particle can only differ
            # when the two best samples lie closer than the fp32 cost error, and then legitimately
            a, b = int(c32[k].argmin()), int(costs_o[k].argmin())
            gap = float((costs_o[k, a] - costs_o[k, b]).abs() / costs_o[k, b].abs())
            assert a != b and gap < 2e-5, f"{tag}: particle {local_idx[k]} off by {float(d[k]):.2e} without a near-tie (gap {gap:.2e})"
            flips += 1
    rec = dict(particles=[int(pl.p0 + i) for i in local_idx], iterations=iters, trials=trials,
               means_within_1e3=ok / trials, argmin_flips_on_near_ties=flips, cost_rel_max=worst_cost,
               samples_rel_max=worst_samples, means_rel_max=worst_means, kernel=expect_kernel)
    print(f"\n[full-size parity] {tag}: {rec}")
    _record_parity(tag, rec)
    assert worst_samples < 2e-5 and worst_cost < 5e-3, rec
    return rec


def _dense_panda_oracle(T, S, k, seed, sph, goals=None):
    """oracle_step for _check_subset on the dense reference-equivalent oracle (k particles of ONE goal)."""
    from oracle.native_noise import native_eps
    c, n = SC.PANDA, 7
    ora = SC.oracle_panda_planner(c, T, k, S, seed=seed, goals=goals,
                                  eps_init=torch.zeros(k, 1, T * 2 * n, dtype=torch.float64))

    def step(mu, gidx, draw):
        ora.particle_means.copy_(mu)
        ora.prior.set_mean(ora.particle_means.view(k, -1))
        eps = torch.from_numpy(native_eps(seed, draw, gidx, S, T, n, "float32")).double()
        costs, _ = ora.step(eps=eps, obstacle_spheres=sph)
        return costs, ora.state_samples.clone(), ora.particle_means.clone()
    return step


def test_config4_last_shard_particles_match_the_dense_oracle():
    """BASELINE configs[3]: 8192 particles sharded over 8 ranks; rank 7's shard (global particles 7168..8191) on this
    GPU, its first and last particle (8191 = the highest global noise key of the problem) against the dense oracle."""
    T, S, seed = 64, 128, 37
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    pl = hip_panda_planner(SC.PANDA, T, 8192, S, F32, seed=seed, rank=7, world_size=8)
    assert (pl.p0, pl.p1) == (7168, 8192)
    sub = [0, 1023]
    _check_subset("config 4: shard 7 of 8 of Panda 8192 x 128 x 64 fp32 (fused launch)", pl, sub,
                  _dense_panda_oracle(T, S, len(sub), seed, sph), 1, {"obstacle_spheres": sph.to(**F32)},
                  "fused_step_kernel")


# --------------------------------------------------------------------------- free-running fp32 parity
# SURVEY 8d defines parity as "max rel err of particle_means after K = 10 iterations".  The tests above hand the HIP
# means to the oracle before every iteration (each iteration an independent trial); these do NOT: the oracle gets the
# subset's means ONCE, before iteration 1, and from then on both sides run on their own -- the HIP planner in fp32 at
# the full BASELINE size, the oracle in fp64 on the restated noise of the same global particle indices -- for K = 10
# iterations.  Whatever fp32 does to a particle (rounding of means and samples, a flipped arg-min) stays in its means
# and is carried into the following iterations, as it would be in a user's run.
def _free_run(tag, pl, local_idx, set_means, oracle_step, K, obs, expect_kernel):
    """set_means(mu [k,T,d] fp64) once; then K times: oracle_step(global indices, draw) -> (costs [k,S], means [k,T,d])
    beside pl.optimize(opt_iters=1).  A particle TRACKS while its means stay within 1e-3 (north_star's fp32 bound, relative
    to the largest mean) of the oracle's.  Asserted every iteration: costs of tracking particles within 5e-3; a particle
    that stops tracking does so through an arg-min flip between two samples whose oracle costs lie within 2e-5 (the fp32
    cost error is ~4e-6) -- anything else fails.  Reported per iteration: the tracking fraction, and the first departure
    with the near-tie gap that explains it."""
    idx = torch.as_tensor(local_idx, device=DEV)
    k = len(local_idx)
    set_means(pl.particle_means[idx].cpu().double())
    gidx = [pl.p0 + i for i in local_idx]
    tracking = [True] * k
    per_iter, departures = [], []
    worst_cost = worst_means = 0.0
    scale = None
    for it in range(1, K + 1):
        draw = pl._draw
        costs_o, means_o = oracle_step(gidx, draw)
        _, _, _, _, costs, _ = pl.optimize(opt_iters=1, **obs)
        assert pl._engine.last_cost_kernel().startswith(expect_kernel), pl._engine.last_cost_kernel()
        scale = scale or float(means_o.abs().max())
        c32 = costs[idx].cpu().double()
        d = (pl.particle_means[idx].cpu().double() - means_o).abs().amax(dim=(1, 2)) / scale
        for j in range(k):
            if not tracking[j]:
                continue
            crel = float(((c32[j] - costs_o[j]).abs() / costs_o[j].abs()).max())
            worst_cost = max(worst_cost, crel)
            assert crel < 5e-3, f"{tag}: iteration {it}, particle {gidx[j]}: cost rel err {crel:.2e} while still tracking"
            if float(d[j]) < 1e-3:
                worst_means = max(worst_means, float(d[j]))
                continue
            a, b = int(c32[j].argmin()), int(costs_o[j].argmin())
            gap = float((costs_o[j, a] - costs_o[j, b]).abs() / costs_o[j, b].abs())
            assert a != b and gap < 2e-5, \
                f"{tag}: iteration {it}, particle {gidx[j]} left by {float(d[j]):.2e} without a near-tie (gap {gap:.2e})"
            tracking[j] = False
            departures.append(dict(iteration=it, particle=int(gidx[j]), near_tie_gap=gap, means_rel=float(d[j])))
        per_iter.append(sum(tracking) / k)
    rec = dict(particles=[int(g) for g in gidx], iterations=K, resynchronised=False,
               tracking_fraction_per_iteration=per_iter, means_within_1e3_after_K=per_iter[-1],
               first_departure=departures[0] if departures else None, departures=departures,
               cost_rel_max_while_tracking=worst_cost, means_rel_max_while_tracking=worst_means, kernel=expect_kernel)
    print(f"\n[free-running fp32 parity, K = {K}] {tag}: {rec}")
    _record_parity("free-running " + tag, rec)
    return rec


def test_config3_free_running_ten_iterations_against_the_dense_oracle():
    """BASELINE configs[2] at full size (Panda 1024 x 128 x 64, fp32, fused launch), particles 0, 5, 511, 1023, ten
    iterations without resynchronisation against the dense fp64 oracle (planner.py:289-299 restated)."""
    from oracle.native_noise import native_eps
    T, S, P, seed, n = 64, 128, 1024, 47, 7
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    pl = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=seed)
    sub = [0, 5, 511, 1023]
    ora = SC.oracle_panda_planner(SC.PANDA, T, len(sub), S, seed=seed,
                                  eps_init=torch.zeros(len(sub), 1, T * 2 * n, dtype=torch.float64))

    def set_means(mu):
        ora.particle_means.copy_(mu)
        ora.prior.set_mean(ora.particle_means.view(len(sub), -1))

    def step(gidx, draw):
        eps = torch.from_numpy(native_eps(seed, draw, gidx, S, T, n, "float32")).double()
        costs, _ = ora.step(eps=eps, obstacle_spheres=sph)
        return costs, ora.particle_means.clone()
    rec = _free_run("config 3: Panda 1024 x 128 x 64 fp32 (fused launch)", pl, sub, set_means, step, 10,
                    {"obstacle_spheres": sph.to(**F32)}, "fused_step_kernel")
    assert rec["tracking_fraction_per_iteration"][0] == 1.0


@pytest.mark.parametrize("shape", [(64, 5, 32), (50, 3, 20)], ids=["reference_panda_example_5x32x64", "ragged_3x20x50"])
def test_product_default_dispatch_small_problems_against_the_dense_oracle(shape, monkeypatch):
    """The product's DEFAULT dispatch against the oracle (round-5 verdict / advisor: tests/conftest.py sets SGPMP_NO_SMALL_STEP and
    SGPMP_STORE_FREE_MIN_BYTES for the whole session, so the oracle-facing tests above never run what a user of small problems
    gets).  Here every SGPMP_* variable is removed before the context is created (they are read once, in sgpmp_create): the
    reference's own Panda example size (panda_environment.py:29-32: 5 particles x 32 samples x 64 waypoints) and a ragged
    3 x 20 x 50 go out as fused_step_small_kernel -- one workgroup per item -- and EVERY particle is followed by the dense
    fp64 oracle on its restated noise, three free-running iterations (costs <= 5e-3, means <= 1e-3; _free_run asserts both),
    then the same problem inside one optimize(opt_iters=3) call (the C-side loop, storing at this size) bit for bit."""
    import os
    from oracle.native_noise import native_eps
    for k in [k for k in os.environ if k.startswith("SGPMP_") and k not in ("SGPMP_LIB_PATH", "SGPMP_RCCL_LIB", "SGPMP_RTC_CACHE")]:
        monkeypatch.delenv(k)
    T, P, S = shape
    seed, n = 61, 7
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    pl = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=seed)
    twin = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=seed)
    sub = list(range(P))
    ora = SC.oracle_panda_planner(SC.PANDA, T, P, S, seed=seed, eps_init=torch.zeros(P, 1, T * 2 * n, dtype=torch.float64))

    def set_means(mu):
        ora.particle_means.copy_(mu)
        ora.prior.set_mean(ora.particle_means.view(P, -1))

    def step(gidx, draw):
        eps = torch.from_numpy(native_eps(seed, draw, gidx, S, T, n, "float32")).double()
        costs, _ = ora.step(eps=eps, obstacle_spheres=sph)
        return costs, ora.particle_means.clone()
    rec = _free_run(f"default dispatch: Panda {P} x {S} x {T} fp32 (fused_step_small_kernel)", pl, sub, set_means, step, 3,
                    {"obstacle_spheres": sph.to(**F32)}, "fused_step_small_kernel")
    assert rec["tracking_fraction_per_iteration"][-1] == 1.0 and rec["means_rel_max_while_tracking"] < 1e-3
    assert pl._engine.store_free_steps() == 0
    out = twin.optimize(opt_iters=3, obstacle_spheres=sph.to(**F32))
    assert twin._engine.last_cost_kernel() == "fused_step_small_kernel" and twin._engine.store_free_steps() == 0   # (below the 2.8 MB bar)
    assert torch.equal(twin.particle_means, pl.particle_means) and torch.equal(out[4], pl._costs)
    assert torch.equal(twin.state_samples, pl.state_samples)


def test_off_grid_shape_free_running_ten_iterations_against_the_dense_oracle():
    """The masked instantiation of the fused launch (round 4: S = 100 is not a multiple of 8, T = 50 not of 16) the same way:
    Panda 256 x 100 x 50, fp32, particles 0, 7, 128, 255, ten iterations without resynchronisation against the dense fp64
    oracle."""
    from oracle.native_noise import native_eps
    T, S, P, seed, n = 50, 100, 256, 59, 7
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    pl = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=seed)
    sub = [0, 7, 128, 255]
    ora = SC.oracle_panda_planner(SC.PANDA, T, len(sub), S, seed=seed,
                                  eps_init=torch.zeros(len(sub), 1, T * 2 * n, dtype=torch.float64))

    def set_means(mu):
        ora.particle_means.copy_(mu)
        ora.prior.set_mean(ora.particle_means.view(len(sub), -1))

    def step(gidx, draw):
        eps = torch.from_numpy(native_eps(seed, draw, gidx, S, T, n, "float32")).double()
        costs, _ = ora.step(eps=eps, obstacle_spheres=sph)
        return costs, ora.particle_means.clone()
    rec = _free_run("off the grid: Panda 256 x 100 x 50 fp32 (fused launch, masked instantiation)", pl, sub, set_means, step, 10,
                    {"obstacle_spheres": sph.to(**F32)}, "fused_step_kernel")
    assert rec["tracking_fraction_per_iteration"][0] == 1.0


def test_config5_share_free_running_ten_iterations_against_the_banded_oracle():
    """BASELINE configs[4]'s per-GPU share (4 goals x 1024 x 256 x 128, shard 3 of 8), three particles, ten free iterations
    against oracle/banded_equiv.py (pinned to the dense oracle at 1e-9)."""
    from oracle import banded_equiv as B
    from oracle.native_noise import native_eps
    c, n = SC.PANDA, 7
    T, S, nppg, seed = 128, 256, 1024, 53
    goals = torch.tensor([g + [0.] * n for g in [c["goal_q"], [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5],
                                                 [0.9, -0.2, 0.4, -1.1, -0.3, 1.9, 0.8],
                                                 [-0.8, 0.1, 0.6, -2.4, 0.4, 2.6, -0.2]]], dtype=torch.float64)
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    pl = hip_panda_planner(c, T, nppg, S, F32, seed=seed, goals=goals.tolist(), rank=3, world_size=8)
    sub = [0, 255, 511]
    g = pl.p0 // nppg
    start = torch.tensor(c["start_q"] + [0.] * n, dtype=torch.float64)
    box = {}

    def set_means(mu):
        box["band"] = B.BandedPlanner(len(sub), S, T, c["dt"], n, start, goals[g:g + 1],
                                      B.panda_chunk_cost(c, T, S, goals[g:g + 1], "rbf"), c["step_size"], c["temperature"],
                                      c["sigma_start_sample"], c["sigma_goal_sample"], c["sigma_gp_sample"], mu.clone(), chunk=1)

    def step(gidx, draw):
        eps = torch.from_numpy(native_eps(seed, draw, gidx, S, T, n, "float32")).double()
        costs, _ = box["band"].step(eps, obstacle_spheres=sph)
        return costs, box["band"].particle_means.clone()
    rec = _free_run("config 5 share: shard 3 of 8 of Panda 4 goals x 1024 x 256 x 128 fp32 (fused launch)", pl, sub, set_means,
                    step, 10, {"obstacle_spheres": sph.to(**F32)}, "fused_step_kernel")
    assert rec["tracking_fraction_per_iteration"][0] == 1.0


@pytest.mark.parametrize("kind", ["panda", "panda_two_goals_sdf", "panda_scan_table_in_lds", "planar", "planar_two_passes",
                                  "planar_three_passes_ragged"])
def test_fused_f64_step_equals_the_two_launch_step(golden, kind):
    """fp64 contexts run sampler + sweep as ONE launch since round 6 (fused_step_f64_kernel: one wave per trajectory, lane =
    waypoint, the sampling recurrence as a Kogge-Stone scan over the lanes with 2 x 2 propagator products from a host-built
    table, x = mu + y stored and handed to the cost terms in registers).  Against the same planner with `no_fused_step`
    (sample_iso_kernel<double>'s serial recurrence, then cost_sweep_kernel<double>): same noise stream, another order of the
    same additions -- samples within 1e-13 of the largest sample, costs 1e-12, identical arg-mins, means 1e-13, through
    single steps and a pipelined call; T beyond one pass of 64 lanes (the carried state) and off it."""
    if kind.startswith("panda"):
        two = kind == "panda_two_goals_sdf"
        g = [SC.PANDA["goal_q"] + [0.] * 7, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * 7] if two else None
        # (beyond 2048 trajectories the launch runs 256-thread workgroups that stage the scan table in LDS; below, one-wave
        # workgroups whose lanes read their rows from memory)
        nppg, S = (24, 128) if kind == "panda_scan_table_in_lds" else (6, 24)
        mk = lambda: hip_panda_planner(SC.PANDA, 64, nppg, S, F64, seed=13, goals=g, field_type="sdf" if two else "rbf")   # noqa: E731
        obs = {"obstacle_spheres": torch.as_tensor(SC.panda_spheres(num=5, seed=3)).to(**F64)}
        two_launch = "cost_sweep_kernel<f64, generated chain>"
    else:
        T = {"planar": 64, "planar_two_passes": 128, "planar_three_passes_ragged": 150}[kind]
        goals = [[9., 6., 0., 0.], [9., -3., 0., 0.]]
        om = planar_map(golden, F64)
        mk = lambda: hip_planar_planner(SC.PLANAR, T, goals, 5, 20, om, F64, seed=13)     # noqa: E731
        obs = {}
        two_launch = "cost_sweep_kernel<f64, no FK>"
    a, b = mk(), mk()
    b._engine.set_option("no_fused_step", 1)
    for k in (1, 1, 3, 1):
        ra, rb = a.optimize(opt_iters=k, **obs), b.optimize(opt_iters=k, **obs)
        assert a._engine.last_cost_kernel() == "fused_step_f64_kernel" and b._engine.last_cost_kernel() == two_launch
        # (one launch + update against sampler + sweep + update; + K5 where the means were edited since the context's last step)
        assert a._engine.last_step_launches() in (2, 3) and b._engine.last_step_launches() in (3, 4)
        scale = float(b.state_samples.abs().max())
        assert float((a.state_samples - b.state_samples).abs().max()) <= 1e-13 * scale
        assert rel_err(a._costs, b._costs) < 1e-12 and torch.equal(a._costs.argmin(1), b._costs.argmin(1))
        assert float((a.particle_means - b.particle_means).abs().max()) <= 1e-13 * float(b.particle_means.abs().max())
        assert float((ra[0] - rb[0]).abs().max()) <= 1e-13 * float(rb[0].abs().max())
        sa, sb = a.global_stats(), b.global_stats()
        assert abs(sa[0] / sb[0] - 1) < 1e-11 and abs(sa[1] / sb[1] - 1) < 1e-11
        b.particle_means.copy_(a.particle_means)         # (keep the twins on one trajectory: rounding differences must not pile up)


def test_fused_f64_step_with_link_fields_in_fp32():
    """Option f64_fields_f32 (opt-in): an fp64 step whose LINK fields -- forward kinematics, self-distance, sphere fields -- run on
    the fp32 launches' packed code from the fp64 waypoint rounded to fp32, while noise, recurrence, samples, means, GP / goal /
    importance-sampling terms stay fp64.  Against the all-fp64 step from the same state: samples bit-identical, costs within the
    collision part's fp32 error (1e-6 of that part: <= 1e-7 of a total cost here), the same arg-mins, means to 1e-12 -- rbf, sdf
    and occupancy sphere fields."""
    for ft in ("rbf", "sdf", "occupancy"):
        sph = torch.as_tensor(SC.panda_spheres(num=6, seed=5)).to(**F64)
        a = hip_panda_planner(SC.PANDA, 64, 8, 32, F64, seed=19, field_type=ft)
        b = hip_panda_planner(SC.PANDA, 64, 8, 32, F64, seed=19, field_type=ft, f64_fields_f32=True)     # (= option / SGPMP_F64_FIELDS_F32)
        for it in range(3):
            a.optimize(obstacle_spheres=sph)
            b.optimize(obstacle_spheres=sph)
            assert a._engine.last_cost_kernel() == "fused_step_f64_kernel"
            assert b._engine.last_cost_kernel() == "fused_step_f64_mixed_kernel"
            assert torch.equal(a.state_samples, b.state_samples)
            assert rel_err(b._costs, a._costs) < 1e-7, (ft, it, rel_err(b._costs, a._costs))
            assert torch.equal(a._costs.argmin(1), b._costs.argmin(1))
            assert float((a.particle_means - b.particle_means).abs().max()) <= 1e-12 * float(a.particle_means.abs().max())


def test_config3_shape_fp64_free_running_against_the_dense_oracle():
    """north_star's fp64 clause -- trajectory means within 1e-5 relative -- AT configs[2]'s shape (Panda 1024 x 128 x 64; rounds
    1-4 measured it at config 1's 4 particles only): the fp64 context (sampler + generic sweep on the generated chain + update:
    two launches since round 6: fused_step_f64_kernel + update) against the dense fp64 oracle on the restated fp64 noise stream, four particles, five free iterations."""
    from oracle.native_noise import native_eps
    c, n = SC.PANDA, 7
    T, S, P, seed = 64, 128, 1024, 91
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    pl = hip_panda_planner(c, T, P, S, F64, seed=seed)
    sub = [0, 3, 512, 1023]
    idx = torch.as_tensor(sub, device=DEV)
    ora = SC.oracle_panda_planner(c, T, len(sub), S, seed=seed, eps_init=torch.zeros(len(sub), 1, T * 2 * n, dtype=torch.float64))
    ora.particle_means.copy_(pl.particle_means[idx].cpu())
    ora.prior.set_mean(ora.particle_means.view(len(sub), -1))
    scale = float(ora.particle_means.abs().max())
    worst = worst_cost = 0.0
    for it in range(5):
        # (the eps this step's launch draws for the four particles, read back from the library: sgpmp_noise)
        eps = torch.cat([pl._engine.noise(seed, pl._draw, 1, S, mode_offset=p) for p in sub], dim=1).cpu()
        costs_o, _ = ora.step(eps=eps, obstacle_spheres=sph)
        costs = pl.optimize(opt_iters=1, obstacle_spheres=sph.to(**F64))[4]
        assert pl._engine.last_cost_kernel() == "fused_step_f64_kernel"
        worst_cost = max(worst_cost, rel_err(costs[idx], costs_o))
        worst = max(worst, float((pl.particle_means[idx].cpu() - ora.particle_means).abs().max()) / scale)
    rec = {"particles": sub, "iterations": 5, "means_rel_err_max": worst, "cost_rel_err_max": worst_cost,
           "kernel": "fused_step_f64_kernel", "tolerance": 1e-5}
    print(f"\n[full-size parity, fp64] config 3 shape: {rec}")
    _record_parity("config 3 shape in fp64: Panda 1024 x 128 x 64 (sampler + generic sweep + update)", rec)
    assert worst < 1e-5 and worst_cost < 1e-7, rec


def test_config2_free_running_ten_iterations_against_the_dense_oracle(golden):
    """BASELINE configs[1] at full size (planar 256 x 64 x 128, fp32, fused_planar_seg_kernel), two particles of every goal,
    ten free iterations against the dense fp64 oracle."""
    from oracle.native_noise import native_eps
    z = golden("g2_planar_e2e.npz")
    T, nppg, S, n, seed = 128, 64, 64, 2, 59
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
    pl = hip_planar_planner(SC.PLANAR, T, goals, nppg, S, planar_map(golden, F32), F32, seed=seed)
    sub = [g * nppg + k for g in range(4) for k in ((7 * g) % nppg, nppg - 1 - g)]
    ora = SC.oracle_planar_planner(SC.PLANAR, T, goals, 2, S, z["grid"], float(z["cell_size"]), z["c_offset"],
                                   seed=seed, eps_init=torch.zeros(2, 4, T * 2 * n, dtype=torch.float64))

    def set_means(mu):
        ora.particle_means.copy_(mu)
        ora.prior.set_mean(ora.particle_means.view(len(sub), -1))

    def step(gidx, draw):
        eps = torch.from_numpy(native_eps(seed, draw, gidx, S, T, n, "float32")).double()
        costs, _ = ora.step(eps=eps)
        return costs, ora.particle_means.clone()
    rec = _free_run("config 2: planar 256 x 64 x 128 fp32 (fused_planar_seg_kernel)", pl, sub, set_means, step, 10, {},
                    "fused_planar_seg")
    assert rec["tracking_fraction_per_iteration"][0] == 1.0


def test_fused_launch_equals_the_two_launch_path_bitwise_at_every_shape():
    """Round 4 (`rng.h: scan_step`): every sampler kernel evaluates the scan recurrence in ONE explicit-fma order (the fused
    launch two-wide, `scan_step2`), so samples, costs and means of the fused launch equal the sampler + sweep pair's bit for
    bit at EVERY shape -- also where the stand-alone sampler is the small-problem kernel (before, hipcc's choice of which
    multiply to contract with which add made them differ by an ulp below full size)."""
    sph = torch.as_tensor(SC.panda_spheres(num=5)).to(**F32)
    for T, nppg, S in ((64, 40, 32), (32, 8, 8), (16, 3, 8), (48, 5, 24), (128, 6, 16)):
        a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=27)
        b = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=27)
        b._engine.set_option("no_fused_step", 1)
        for it in range(2):
            a.optimize(opt_iters=1, obstacle_spheres=sph)
            b.optimize(opt_iters=1, obstacle_spheres=sph)
            assert a._engine.last_cost_kernel() == "fused_step_kernel" and b._engine.last_cost_kernel() != "fused_step_kernel"
            assert torch.equal(a.state_samples, b.state_samples), (T, nppg, S, it)
            assert torch.equal(a._costs, b._costs) and torch.equal(a.particle_means, b.particle_means), (T, nppg, S, it)


def test_fused_launch_at_sample_counts_and_lengths_off_its_grid():
    """Round 4: the Panda launch masks the rows past S in a particle's last group of 8 and the columns / cost lanes past T in
    the last chunk of 16 waypoints (T even; before, such S or T ran the two-launch path at ~1.5 x the time).  Samples bit for bit those of the stand-alone sampler, costs those of the generic sweep to fp32
    rounding (another summation order), means the same; nothing is written past a particle's rows (the next particle's first
    rows take part in the same comparison)."""
    sph = torch.as_tensor(SC.panda_spheres(num=5)).to(**F32)
    for T, nppg, S in ((32, 5, 12), (64, 7, 30), (16, 3, 100), (32, 4, 7), (32, 6, 1),
                       (50, 5, 16), (18, 3, 24), (34, 4, 8), (100, 3, 8), (50, 6, 30), (2, 3, 8), (4, 3, 8), (6, 2, 16), (14, 2, 5)):
        a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=31)
        b = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=31)
        b._engine.set_option("no_fused_step", 1)
        for it in range(3):
            a.optimize(opt_iters=1, obstacle_spheres=sph)
            b.optimize(opt_iters=1, obstacle_spheres=sph)
            assert a._engine.last_cost_kernel() == "fused_step_kernel" and b._engine.last_cost_kernel() != "fused_step_kernel"
            assert a.state_samples.shape[1] == S
            assert torch.equal(a.state_samples, b.state_samples), (T, nppg, S, it)
            assert torch.allclose(a._costs, b._costs, rtol=2e-5, atol=0), (T, nppg, S, it)
            assert torch.equal(a._costs.argmin(1), b._costs.argmin(1)), (T, nppg, S, it)
            assert float((a.particle_means - b.particle_means).abs().max()) <= 1e-6 * float(b.particle_means.abs().max()), (T, nppg, S, it)
    # one optimize(opt_iters=K) call (two chains where the halves are big enough) = K single calls
    a = hip_panda_planner(SC.PANDA, 50, 200, 100, F32, seed=5)
    b = hip_panda_planner(SC.PANDA, 50, 200, 100, F32, seed=5)
    a.optimize(opt_iters=4, obstacle_spheres=sph)
    assert a._engine.pipeline_split_steps() > 0 if hasattr(a._engine, "pipeline_split_steps") else True
    for _ in range(4):
        b.optimize(opt_iters=1, obstacle_spheres=sph)
    assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a._costs, b._costs) and torch.equal(a.state_samples, b.state_samples)


# --------------------------------------------------------------------------- dense-weight regime of the update
@pytest.mark.parametrize("S", [64, 44])       # 44: the particle's last group of 8 is half empty
def test_dense_weight_update_adds_partials_of_the_fused_launch_instead_of_rereading_the_rows(S):
    """planner.py:263-275 is a softmax.  With the reference's hyper-parameters it is one-hot and update_kernel reads one row;
    at a temperature where many samples carry weight, round 3's update re-read every such row.  Now the fused launch leaves a
    softmax partial per 8 rows for the particles whose PREVIOUS update was spread (device-side count, no host round trip) and
    the update adds S / 8 partials: same means / gradient / weights as the row-reading update (`no_dense_partials`) to 1e-6,
    and both follow the fp64 oracle at that temperature."""
    from oracle.native_noise import native_eps
    n, T, nppg, seed = 7, 32, 24, 23
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=seed))
    found = None
    # (the importance-sampling term temperature * x^T Sigma^-1 mu grows with the temperature: with the reference's stiff
    # sampling prior the softmax stays one-hot at ANY temperature -- the weights spread only under a weak sampling prior)
    soft = dict(sigma_start_sample=1.0, sigma_goal_sample=1.0, sigma_gp_sample=30.0)
    for temp, extra in ((1e9, {}), (1e13, {}), (1e11, soft), (1e14, soft), (1e17, soft)):
        c = dict(SC.PANDA, temperature=temp, **extra)
        a = hip_panda_planner(c, T, nppg, S, F32, seed=seed)
        for _ in range(2):
            a.optimize(opt_iters=1, obstacle_spheres=sph.to(**F32))
        k = a._engine.dense_particles()
        nz = float((a._weights_buf != 0).sum()) / nppg
        print(f"\n[dense-weight regime] temperature {temp:g} {extra}: {k} of {nppg} particles spread their weight over more than "
              f"S / 4 samples ({nz:.1f} rows with weight per particle)")
        if k >= nppg // 2:
            found = (temp, c)
            break
    assert found is not None, "no temperature of the scan spreads the weights: the test needs another scenario"
    temp, c = found
    a = hip_panda_planner(c, T, nppg, S, F32, seed=seed)
    b = hip_panda_planner(c, T, nppg, S, F32, seed=seed)
    b._engine.set_option("no_dense_partials", 1)
    eps0 = torch.from_numpy(native_eps(seed, 0, range(1), nppg, T, n, "float32")).double()
    ora = SC.oracle_panda_planner(c, T, nppg, S, seed=seed, eps_init=eps0)
    scale = float(a.particle_means.abs().max())
    used = armed_before = 0
    for it in range(8):
        b.particle_means.copy_(a.particle_means)
        ora.particle_means.copy_(a.particle_means.cpu().double())
        ora.prior.set_mean(ora.particle_means.view(nppg, -1))
        eps = torch.from_numpy(native_eps(seed, 2 + it, range(nppg), S, T, n, "float32")).double()
        costs_o, grad_o = ora.step(eps=eps, obstacle_spheres=sph)
        _, _, _, _, ca, ga = a.optimize(opt_iters=1, obstacle_spheres=sph.to(**F32))
        _, _, _, _, cb, gb = b.optimize(opt_iters=1, obstacle_spheres=sph.to(**F32))
        assert a._engine.last_cost_kernel() == "fused_step_kernel" and torch.equal(ca, cb)
        assert float((a.particle_means - b.particle_means).abs().max()) <= 1e-6 * scale, it
        assert float((ga - gb).abs().max()) <= 1e-6 * max(float(gb.abs().max()), 1e-30) + 1e-7 * scale, it
        assert torch.allclose(a._weights, b._weights, rtol=0, atol=1e-7)
        # the oracle at this temperature: soft weights make the means a smooth function of the costs
        assert rel_err(ca, costs_o) < 5e-3
        assert float((a.particle_means.cpu().double() - ora.particle_means).abs().max()) < 1e-3 * scale, it
        used = max(used, a._engine.dense_particles())
    # which particles get partials is decided on the device from the row count the particle's previous update left (round 5:
    # a function of stream-ordered state -- round 4 armed the partials from a host word read without synchronisation, and
    # the step at which a run switched from rows to partials depended on host timing): the first iteration gathers rows,
    # every later one adds partials, in every run
    assert used >= nppg // 2 and a._engine.dense_armed_steps() == 8 and b._engine.dense_armed_steps() == 0
    assert b._engine.dense_particles() >= nppg // 2          # (the counts are kept either way; `b` just never uses partials)
    # ... so two runs of the same problem are the same run, bit for bit -- and so is the run cut into optimize() calls
    # of several iterations (store-free iterations inside: a spread particle's rows are written all the same)
    r1 = hip_panda_planner(c, T, nppg, S, F32, seed=seed)
    r2 = hip_panda_planner(c, T, nppg, S, F32, seed=seed)
    r3 = hip_panda_planner(c, T, nppg, S, F32, seed=seed, store_free=False)
    for _ in range(6):
        r1.optimize(opt_iters=1, obstacle_spheres=sph.to(**F32))
    r2.optimize(opt_iters=4, obstacle_spheres=sph.to(**F32))
    r2.optimize(opt_iters=2, obstacle_spheres=sph.to(**F32))
    r3.optimize(opt_iters=6, obstacle_spheres=sph.to(**F32))
    for r in (r2, r3):
        assert torch.equal(r1.particle_means, r.particle_means) and torch.equal(r1.state_samples, r.state_samples)
        assert torch.equal(r1._weights_buf, r._weights_buf) and torch.equal(r1._grad, r._grad)
    assert r2._engine.store_free_steps() == 4 and r3._engine.store_free_steps() == 0
    # one-hot weights (the reference's hyper-parameters): no particle ever asks for partials
    h = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=seed)
    for _ in range(6):
        h.optimize(opt_iters=1, obstacle_spheres=sph.to(**F32))
    assert h._engine.dense_particles() == 0 and int(h._engine.row_counts().max()) == 1


# --------------------------------------------------------------------------- whole-population parity at full size (round 5)
# Round-4 verdict, weak #1 / next #2: the full-size fp32 parity above follows 3-8 particles per configuration; here EVERY
# particle of the configuration is followed by the fp64 banded oracle (oracle/banded_equiv.py: the dense oracle's mathematics
# with the prior factored once -- pinned to the dense oracle, which is pinned to reference runs, at 1e-9) on the restated
# noise of its own global index: 2 re-synchronised iterations (the oracle is handed the HIP means before the step: every
# particle-iteration an independent trial) + 5 free-running ones (both sides on their own).  The population is cut into
# chunks of particles that worker threads step side by side (particles are independent: planner.py:263-275).
# SGPMP_LONG_PARITY=1: the whole-population tests at round 5's iteration counts (2 + 5 at config 3, 1 + 3 at config 5's share) and
# the fp32-against-fp64 twin over 120 iterations -- for the round's record (profiles/rNN), not for the driver's timed suite
_LONG_PARITY = bool(__import__("os").environ.get("SGPMP_LONG_PARITY"))


def _population_parity(tag, pl, make_band, n, obs_hip, obs_ora, expect_kernel, sync_iters=2, free_iters=5, pchunk=32, workers=32,
                       cost_quantum=None):
    """cost_quantum (planar problems): the weight 1 / sigma_coll^2 of one occupancy-grid count.  The grid lookup is a step
    function of the position (obst_map.py:173-174: floor(X / cell + offset)); a sample whose fp32 position lies within
    rounding of a cell boundary reads the neighbouring cell, and its cost differs from the fp64 oracle's by a whole number
    of quanta (1e10 at configs[1]) -- counted and reported as `grid_boundary_samples`, not as cost error."""
    import json
    import os
    import time
    from concurrent.futures import ThreadPoolExecutor
    from oracle.native_noise import native_eps
    t_start = time.perf_counter()
    P, S, T = pl.num_particles_local, pl.num_samples, pl.traj_len
    chunks = [(lo, min(P, lo + pchunk)) for lo in range(0, P, pchunk)]
    bands = [make_band(lo, hi, pl.particle_means[lo:hi].cpu().double()) for lo, hi in chunks]
    threads_before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // workers)))
    tracking = np.ones(P, dtype=bool)
    per_iter, flips_log, unexplained = [], [], []
    worst_cost = worst_samples = 0.0
    boundary = 0
    scale = float(pl.particle_means.abs().max())
    try:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for it in range(sync_iters + free_iters):
                sync = it < sync_iters
                draw = pl._draw
                mu_before = pl.particle_means.cpu().double()
                costs = pl.optimize(opt_iters=1, **obs_hip)[4].cpu().double()
                assert pl._engine.last_cost_kernel().startswith(expect_kernel), pl._engine.last_cost_kernel()
                x_hip = pl.state_samples.cpu()
                mu_after = pl.particle_means.cpu().double()

                def one(ci):
                    lo, hi = chunks[ci]
                    band = bands[ci]
                    if sync:
                        band.particle_means = mu_before[lo:hi].clone()
                    eps = torch.from_numpy(native_eps(pl.seed, draw, range(pl.p0 + lo, pl.p0 + hi), S, T, n, "float32")).double()
                    c_o, _ = band.step(eps, **obs_ora)
                    x_o = band.state_samples
                    xs = float((x_hip[lo:hi].double() - x_o).abs().max() / x_o.abs().max())
                    diff = costs[lo:hi] - c_o
                    rel = diff.abs() / c_o.abs()
                    on_boundary = 0
                    if cost_quantum is not None:
                        q = diff / cost_quantum
                        stepped = (rel > 1e-3) & (q.round() != 0) & ((q - q.round()).abs() < 0.02)
                        on_boundary = int(stepped.sum())
                        rel = torch.where(stepped, torch.zeros_like(rel), rel)
                    cr = rel.amax(dim=1)
                    d = (mu_after[lo:hi] - band.particle_means).abs().amax(dim=(1, 2)) / scale
                    a = costs[lo:hi].argmin(dim=1)
                    b = c_o.argmin(dim=1)
                    gap = ((c_o.gather(1, a[:, None]) - c_o.gather(1, b[:, None])).abs() / c_o.gather(1, b[:, None]).abs())[:, 0]
                    return lo, hi, xs, cr.numpy(), d.numpy(), (a != b).numpy(), gap.numpy(), on_boundary
                within = flips = 0
                for lo, hi, xs, cr, d, differ, gap, nb_ in pool.map(one, range(len(chunks))):
                    boundary += nb_
                    for j in range(hi - lo):
                        p = lo + j
                        if not (sync or tracking[p]):
                            continue                         # (left its twin through a documented near-tie: no longer comparable)
                        worst_cost = max(worst_cost, float(cr[j]))
                        if d[j] < 1e-3:
                            within += 1
                            continue
                        rec = {"iteration": it + 1, "particle": int(pl.p0 + p), "means_rel": float(d[j]), "near_tie_gap": float(gap[j]),
                               "resynchronised": sync}
                        if differ[j] and gap[j] < 2e-5:
                            flips += 1
                            flips_log.append(rec)
                        else:
                            unexplained.append(rec)
                        if not sync:
                            tracking[p] = False
                    if sync or tracking[lo:hi].all():
                        worst_samples = max(worst_samples, xs)
                compared = P if sync else int(tracking.sum()) + flips
                per_iter.append({"iteration": it + 1, "resynchronised": sync, "particles_compared": compared,
                                 "within_1e-3": within / max(compared, 1), "near_tie_flips": flips})
    finally:
        torch.set_num_threads(threads_before)
    rec = {"configuration": tag, "kernel": expect_kernel, "particles": P, "global_range": [int(pl.p0), int(pl.p1)],
           "samples": S, "traj_len": T, "oracle": "oracle/banded_equiv.py (fp64; pinned to the dense oracle at 1e-9)",
           "iterations": per_iter, "near_tie_flips": flips_log, "flip_rate_per_particle_iteration":
           len(flips_log) / float(P * (sync_iters + free_iters)), "unexplained_departures": unexplained,
           "largest_unexplained_gap": max([u["near_tie_gap"] for u in unexplained], default=None),
           "cost_rel_err_max": worst_cost, "samples_rel_err_max": worst_samples,
           "grid_boundary_samples": boundary if cost_quantum is not None else None,
           "sample_costs_compared": P * S * (sync_iters + free_iters),
           "still_tracking_after_free_run": int(tracking.sum()), "seconds": time.perf_counter() - t_start}
    print(f"\n[whole-population parity] {tag}: " + json.dumps({k: v for k, v in rec.items() if k not in ('near_tie_flips',)}))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        path = os.path.join(out, "parity_population.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old[tag] = rec
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    assert not unexplained, unexplained[:5]
    assert worst_cost < 5e-3 and worst_samples < 2e-5, (worst_cost, worst_samples)
    assert all(r["within_1e-3"] + r["near_tie_flips"] / max(r["particles_compared"], 1) == 1.0 for r in per_iter)
    return rec


def test_whole_population_parity_config3():
    """BASELINE configs[2]: all 1024 particles x 128 samples x 64 waypoints of the fused launch against the banded oracle."""
    from oracle import banded_equiv as B
    c, n = SC.PANDA, 7
    T, S, P, seed = 64, 128, 1024, 83
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    goal = torch.tensor([c["goal_q"] + [0.] * n], dtype=torch.float64)
    start = torch.tensor(c["start_q"] + [0.] * n, dtype=torch.float64)
    pl = hip_panda_planner(c, T, P, S, F32, seed=seed)
    cost = B.panda_chunk_cost(c, T, S, goal, "rbf")

    def make_band(lo, hi, mu):
        return B.BandedPlanner(hi - lo, S, T, c["dt"], n, start, goal, cost, c["step_size"], c["temperature"],
                               c["sigma_start_sample"], c["sigma_goal_sample"], c["sigma_gp_sample"], mu, chunk=8)
    # (1 re-synchronised + 3 free-running iterations: 4 096 particle-iterations per run; round 5's record of 2 + 5 -- 7 168, one
    # explained near-tie flip -- is profiles/r05/parity_population.json.  The oracle side costs the host ~18 s per iteration, and the
    # driver's GPU suite has a 1200 s limit.)
    rec = _population_parity("config 3: Panda 1024 x 128 x 64 fp32 (fused launch)", pl, make_band, n,
                             {"obstacle_spheres": sph.to(**F32)}, {"obstacle_spheres": sph}, "fused_step_kernel",
                             sync_iters=2 if _LONG_PARITY else 1, free_iters=5 if _LONG_PARITY else 3)
    assert rec["still_tracking_after_free_run"] >= 0.99 * P


def test_whole_population_parity_config2(golden):
    """BASELINE configs[1]: all 256 particles (4 goals x 64) x 64 samples x 128 waypoints of fused_planar_seg_kernel."""
    from oracle import banded_equiv as B
    z = golden("g2_planar_e2e.npz")
    c, n = SC.PLANAR, 2
    T, nppg, S, seed = 128, 64, 64, 85
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
    goals_t = torch.tensor(goals, dtype=torch.float64)
    start = torch.tensor(c["start"], dtype=torch.float64)
    pl = hip_planar_planner(c, T, goals, nppg, S, planar_map(golden, F32), F32, seed=seed)

    def make_band(lo, hi, mu):
        g = lo // nppg
        assert (hi - 1) // nppg == g
        cost = B.planar_chunk_cost(c, T, S, goals_t[g:g + 1], z["grid"], float(z["cell_size"]), z["c_offset"])
        return B.BandedPlanner(hi - lo, S, T, c["dt"], n, start, goals_t[g:g + 1], cost, c["step_size"], c["temperature"],
                               c["sigma_start_sample"], c["sigma_goal_sample"], c["sigma_gp_sample"], mu, chunk=32)
    rec = _population_parity("config 2: planar 256 x 64 x 128 fp32 (fused_planar_seg_kernel)", pl, make_band, n, {}, {},
                             "fused_planar_seg", pchunk=32, workers=8, cost_quantum=1. / c["sigma_coll"] ** 2)
    assert rec["grid_boundary_samples"] <= 1e-3 * rec["sample_costs_compared"]
    assert rec["still_tracking_after_free_run"] >= 0.98 * 256


def test_whole_population_parity_config5_share():
    """BASELINE configs[4]'s per-GPU share: all 512 particles of shard 3 of 8 (4 goals x 1024 x 256 samples x 128 waypoints;
    global particles 1536..2047, goal 1) -- 1 re-synchronised + 2 free-running iterations (each costs the host four
    config-3 iterations; round 5 ran 1 + 3: `profiles/r05/parity_population.json`)."""
    from oracle import banded_equiv as B
    c, n = SC.PANDA, 7
    T, S, nppg, seed = 128, 256, 1024, 87
    goals = torch.tensor([g + [0.] * n for g in [c["goal_q"], [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5],
                                                 [0.9, -0.2, 0.4, -1.1, -0.3, 1.9, 0.8],
                                                 [-0.8, 0.1, 0.6, -2.4, 0.4, 2.6, -0.2]]], dtype=torch.float64)
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    start = torch.tensor(c["start_q"] + [0.] * n, dtype=torch.float64)
    pl = hip_panda_planner(c, T, nppg, S, F32, seed=seed, goals=goals.tolist(), rank=3, world_size=8)
    assert (pl.p0, pl.p1) == (1536, 2048)
    g = pl.p0 // nppg
    cost = B.panda_chunk_cost(c, T, S, goals[g:g + 1], "rbf")

    def make_band(lo, hi, mu):
        return B.BandedPlanner(hi - lo, S, T, c["dt"], n, start, goals[g:g + 1], cost, c["step_size"], c["temperature"],
                               c["sigma_start_sample"], c["sigma_goal_sample"], c["sigma_gp_sample"], mu, chunk=2)
    rec = _population_parity("config 5 share: shard 3 of 8 of Panda 4 goals x 1024 x 256 x 128 fp32 (fused launch)", pl,
                             make_band, n, {"obstacle_spheres": sph.to(**F32)}, {"obstacle_spheres": sph}, "fused_step_kernel",
                             sync_iters=1, free_iters=3 if _LONG_PARITY else 2, pchunk=16, workers=32)
    assert rec["still_tracking_after_free_run"] >= 0.99 * 512


@pytest.mark.parametrize("config", ["config3", "config5_share"])
def test_fp32_population_follows_its_fp64_twin_on_identical_noise(config):
    """One noise stream serves both precisions (round 6), so an fp32 planner can be held to an fp64 planner of THIS library on
    identical eps at any size without a CPU oracle in the loop -- and the fp64 one-launch step is itself held to the dense
    oracle at 1e-9 (test_config3_shape_fp64_free_running_*, bench.py's parity leg).  BASELINE configs[2] at full size, all 1024
    particles, 12 re-synchronised iterations = 12 288 particle-iterations: samples equal to fp32 rounding, costs within 5e-3
    (measured ~1e-5), and every particle whose fp32 update leaves its fp64 twin (> 1e-3 of the means' scale) does so through a
    near tie of its two best samples (fp64 costs within 2e-5 of each other): an arg-min flip, or -- same arg-min -- a softmax
    weight the two share in fp64 and fp32 cannot resolve (found by this test's first run: 1 in 12 288) -- the fp32 path's only
    ways to differ.
    The flip rate is reported (profiles: ~1 in 7 000, as the banded-oracle population test measured)."""
    import json
    import os
    sph = torch.as_tensor(SC.panda_spheres(num=5))
    if config == "config3":
        T, S, P, seed, iters = 64, 128, 1024, 101, 120 if _LONG_PARITY else 12
        lo = hip_panda_planner(SC.PANDA, T, P, S, F32, seed=seed)
        hi = hip_panda_planner(SC.PANDA, T, P, S, F64, seed=seed)
    else:
        # BASELINE configs[4]'s per-GPU share (4 goals x 1024 x 256 x 128, shard 3 of 8): T = 128 is two passes of the fp64
        # launch's 64 lanes -- the carried state of its scan at full size
        T, S, P, seed, iters = 128, 256, 512, 103, 24 if _LONG_PARITY else 4
        goals = [g + [0.] * 7 for g in [SC.PANDA["goal_q"], [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5],
                                        [0.9, -0.2, 0.4, -1.1, -0.3, 1.9, 0.8], [-0.8, 0.1, 0.6, -2.4, 0.4, 2.6, -0.2]]]
        lo = hip_panda_planner(SC.PANDA, T, 1024, S, F32, seed=seed, goals=goals, rank=3, world_size=8)
        hi = hip_panda_planner(SC.PANDA, T, 1024, S, F64, seed=seed, goals=goals, rank=3, world_size=8)
        assert (lo.p0, lo.p1) == (1536, 2048)
    assert lo._draw == hi._draw
    flips, unexplained, worst_cost, worst_samples = [], [], 0.0, 0.0
    for it in range(iters):
        mu = hi.particle_means.float()                       # identical, fp32-representable means on both sides
        lo.particle_means.copy_(mu)
        hi.particle_means.copy_(mu.double())
        c_lo = lo.optimize(obstacle_spheres=sph.to(**F32))[4]
        c_hi = hi.optimize(obstacle_spheres=sph.to(**F64))[4]
        assert lo._engine.last_cost_kernel() == "fused_step_kernel" and hi._engine.last_cost_kernel() == "fused_step_f64_kernel"
        scale = float(hi.particle_means.abs().max())
        worst_samples = max(worst_samples, float((lo.state_samples.double() - hi.state_samples).abs().max()
                                                 / hi.state_samples.abs().max()))
        worst_cost = max(worst_cost, rel_err(c_lo.double(), c_hi))
        d = (lo.particle_means.double() - hi.particle_means).abs().amax(dim=(1, 2)) / scale
        off = torch.nonzero(d >= 1e-3).flatten().tolist()
        a_lo, a_hi = c_lo.argmin(1), c_hi.argmin(1)
        for p in off:
            a, b = int(a_lo[p]), int(a_hi[p])
            gap = float((c_hi[p, a] - c_hi[p, b]).abs() / c_hi[p, b].abs())
            # (the same arg-min on both sides can still move the means apart: two samples whose fp64 costs differ by O(1) of 1e9
            # SHARE the softmax weight in fp64, while fp32 -- cost resolution 64 at 1e9 -- sees a tie or a one-hot: a near tie
            # of the two best samples all the same)
            top2 = torch.topk(c_hi[p], 2, largest=False).values
            gap2 = float((top2[1] - top2[0]) / top2[0].abs())
            rec = {"iteration": it + 1, "particle": p, "near_tie_gap": gap if a != b else gap2, "same_arg_min": a == b, "means_rel": float(d[p])}
            (flips if ((a != b and gap < 2e-5) or gap2 < 2e-5) else unexplained).append(rec)
    rec = {"configuration": f"{config}: Panda {P} x {S} x {T}, fp32 fused launch against the fp64 one-launch step on identical noise",
           "particle_iterations": P * iters, "near_tie_flips": flips, "flip_rate_per_particle_iteration": len(flips) / float(P * iters),
           "unexplained_departures": unexplained, "cost_rel_err_max": worst_cost, "samples_rel_err_max": worst_samples}
    print("\n[fp32 vs fp64 twin, whole population] " + json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        path = os.path.join(out, "parity_fp32_vs_fp64_twin.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        if "configuration" in old:
            old = {}
        old[config] = rec
        json.dump(old, open(path, "w"), indent=1)
    assert not unexplained, unexplained[:5]
    assert worst_cost < 5e-3 and worst_samples < 2e-6, (worst_cost, worst_samples)
    assert len(flips) <= 0.002 * P * iters


# --------------------------------------------------------------------------- store-free iterations (round 5)
def _store_free_twins(build, calls, obs, expect_kernel, expect_store_free=True):
    """The same planner twice: `a` lets the iterations inside optimize(opt_iters = K) skip their sample stores (all but the
    call's last: SGPMP_STEP_NO_SAMPLES; update_kernel regenerates the rows that carry weight from their noise keys), `b`
    stores every iteration as rounds 1-4 did.  Everything a caller can see must be BIT-IDENTICAL after every call: the
    returned 6-tuple, particle_means, state_samples, weights, gradient, costs."""
    a, b = build(store_free=True), build(store_free=False)
    free = 0
    for k in calls:
        ra, rb = a.optimize(opt_iters=k, **obs), b.optimize(opt_iters=k, **obs)
        assert a._engine.last_cost_kernel() == expect_kernel, a._engine.last_cost_kernel()
        for i, (x, y) in enumerate(zip(ra, rb)):
            assert torch.equal(x, y), (k, i)
        assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a.state_samples, b.state_samples), k
        assert torch.equal(a._weights_buf, b._weights_buf) and torch.equal(a._grad, b._grad), k
        assert torch.equal(a._costs, b._costs) and torch.equal(a._means_prev, b._means_prev), k
        sa, sb = a.global_stats(), b.global_stats()
        assert abs(sa[0] / sb[0] - 1) < 1e-12 and abs(sa[1] / sb[1] - 1) < 1e-12
        free += k - 1
    assert a._engine.store_free_steps() == (free if expect_store_free else 0) and b._engine.store_free_steps() == 0
    return a, b


@pytest.mark.parametrize("kind", ["config3", "config3_one_chain", "config5_share", "off_grid", "two_goals_sdf", "small"])
def test_store_free_iterations_equal_storing_iterations_bitwise_panda(kind):
    """Verdict round 4, item 1: K iterations in one optimize() call, store-free against storing -- final particle_means,
    the returned 6-tuple and state_samples bit-identical -- at BASELINE configs[2] (two particle-half chains, and as one
    chain), config 5's share (T = 128: two regenerated Philox blocks per thread), a shape off the launch's 8 x 16 grid (the
    masked instantiation), two goals with the sdf field, and a problem too small for two chains."""
    sph = torch.as_tensor(SC.panda_spheres(num=5)).to(**F32)
    obs = {"obstacle_spheres": sph}
    calls = (10, 1, 3)
    if kind == "config3":
        build = lambda **kw: hip_panda_planner(SC.PANDA, 64, 1024, 128, F32, seed=61, **kw)                     # noqa: E731
    elif kind == "config3_one_chain":
        build = lambda **kw: hip_panda_planner(SC.PANDA, 64, 1024, 128, F32, seed=62, pipeline_steps=False, **kw)   # noqa: E731
        calls = (6, 2)
    elif kind == "config5_share":
        goals = [SC.PANDA["goal_q"] + [0.] * 7, [0.3, 0.1, 0.2, -1.2, 0.0, 1.8, 0.2] + [0.] * 7,
                 [-0.2, 0.4, 0.1, -1.8, 0.2, 2.2, 0.5] + [0.] * 7, [0.1, -0.1, 0.4, -2.0, -0.1, 1.6, 0.0] + [0.] * 7]
        build = lambda **kw: hip_panda_planner(SC.PANDA, 128, 1024, 256, F32, seed=63, goals=goals, rank=3, world_size=8, **kw)   # noqa: E731
        calls = (5, 2)
    elif kind == "off_grid":
        build = lambda **kw: hip_panda_planner(SC.PANDA, 50, 256, 100, F32, seed=64, **kw)                      # noqa: E731
    elif kind == "two_goals_sdf":
        goals = [SC.PANDA["goal_q"] + [0.] * 7, [0.3, 0.1, 0.2, -1.2, 0.0, 1.8, 0.2] + [0.] * 7]
        build = lambda **kw: hip_panda_planner(SC.PANDA, 32, 96, 64, F32, seed=65, goals=goals, field_type="sdf", **kw)   # noqa: E731
    else:
        build = lambda **kw: hip_panda_planner(SC.PANDA, 16, 3, 8, F32, seed=66, **kw)                           # noqa: E731
    a, _ = _store_free_twins(build, calls, obs, "fused_step_kernel")
    assert int(a._engine.row_counts().max()) <= 2          # (the reference's hyper-parameters: one-hot weights, at most a near-tie)


@pytest.mark.parametrize("nppg,G,S,T,n", [(64, 4, 64, 128, 2), (3, 2, 64, 64, 2), (5, 1, 64, 256, 2), (2, 2, 192, 16, 2), (3, 1, 64, 96, 3)])
def test_store_free_iterations_equal_storing_iterations_bitwise_planar(golden, nppg, G, S, T, n):
    """The same at BASELINE configs[1] (256 x 64 x 128: fused_planar_seg_kernel) and at the launch's other shapes: segments of
    16 waypoints (T = 256), several 64-sample blocks per particle, n = 3.  Two store-free forms:
      * S = 64 (a particle's samples are ONE workgroup's): the update runs INSIDE the launch (seg_update) -- the default: one
        launch per iteration, nothing stored, nothing regenerated;
      * any S (opt-in `planar_store_free`, measured slower): update_kernel regenerates its rows segment by segment -- zero-start
        recurrences, the chain over the segments' end states, the fix-up (rng.h seg_chain / seg_fixup)."""
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]][:G]
    c = SC.PLANAR
    if n == 3:
        c = dict(SC.PLANAR, n_dof=3, start=[-9., -9., 0.5, 0., 0., 0.])
        goals = [g[:2] + [0.3 * (i + 1), 0., 0., 0.] for i, g in enumerate(goals)]
    om = planar_map(golden, F32)

    def regen(**kw):
        pl = hip_planar_planner(c, T, goals, nppg, S, om, F32, seed=67, **kw)
        pl._engine.set_option("planar_store_free", 1)    # (opt-in: measured slower than storing at config 2)
        pl._engine.set_option("no_planar_tail", 1)
        return pl
    _store_free_twins(regen, (7, 1, 4), {}, "fused_planar_seg_kernel")
    # the default: S = 64 -> the update inside the launch; else this launch stores every iteration
    a, _ = _store_free_twins(lambda **kw: hip_planar_planner(c, T, goals, nppg, S, om, F32, seed=67, **kw), (5, 1, 3), {},
                             "fused_planar_seg_kernel", expect_store_free=(S == 64))
    if S == 64:
        a.optimize(opt_iters=3)
        assert a._engine.last_step_launches() == 2           # (the call's last iteration: launch + update_kernel)
        a.step(_samples_unread=True)
        assert a._engine.last_step_launches() == 1           # a store-free step: ONE launch
        # round 6: calls of three iterations and more run ALL their store-free iterations in one launch where the launch has that
        # form (n = 2, segments of 8 waypoints: BASELINE configs[1]'s shape) -- the calls of 5 and 3 above did
        assert a._engine.multi_iteration_launches() == (3 if (n == 2 and T <= 128) else 0)


@pytest.mark.parametrize("nppg,G,T", [(64, 4, 128), (3, 2, 64), (1, 1, 16)])
def test_all_store_free_iterations_in_one_launch_equal_one_launch_each(golden, nppg, G, T):
    """sgpmp_optimize hands iterations 0 .. K - 2 of a planar problem with 64 samples per particle to ONE launch
    (fused_planar_seg.inc: PERSIST -- a particle's workgroup loops over them, re-reading its own means and importance-sampling
    weights) against round 5's launch per iteration (option no_persist_planar): the returned tuple, means, samples, weights,
    gradient, costs, previous means, row counts and the statistics of the call's last step, bit for bit, over calls of several
    lengths (K = 2 has one store-free iteration: no such launch)."""
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]][:G]
    om = planar_map(golden, F32)
    a = hip_planar_planner(SC.PLANAR, T, goals, nppg, 64, om, F32, seed=71)
    b = hip_planar_planner(SC.PLANAR, T, goals, nppg, 64, om, F32, seed=71)
    b._engine.set_option("no_persist_planar", 1)
    if T == 64:
        # a launch runs `persist_max_iters` iterations at most (default 2048): here calls of 40 / 7 iterations take 39 = 5 x 7 + 4
        # and 6 store-free iterations as 6 launches and 1 launch -- the same bits again
        a._engine.set_option("persist_max_iters", 7)
    launches = 0
    for k in (3, 2, 40, 1, 7):
        ra, rb = a.optimize(opt_iters=k), b.optimize(opt_iters=k)
        for i, (x, y) in enumerate(zip(ra, rb)):
            assert torch.equal(x, y), (k, i)
        assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a.state_samples, b.state_samples), k
        assert torch.equal(a._weights_buf, b._weights_buf) and torch.equal(a._grad, b._grad), k
        assert torch.equal(a._costs, b._costs) and torch.equal(a._means_prev, b._means_prev), k
        assert (a._engine.row_counts() == b._engine.row_counts()).all()
        sa, sb = a.global_stats(), b.global_stats()
        assert abs(sa[0] / sb[0] - 1) < 1e-12 and abs(sa[1] / sb[1] - 1) < 1e-12
        launches += (0 if k < 3 else 1 if T != 64 else -(-(k - 1) // 7))
        assert a._engine.multi_iteration_launches() == launches and b._engine.multi_iteration_launches() == 0
    assert a._engine.store_free_steps() == b._engine.store_free_steps() == 2 + 1 + 39 + 0 + 6
    # a step of its own afterwards starts from the state the loop left (the weights of the next step were written inside it)
    ra, rb = a.step(), b.step()
    assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a.state_samples, b.state_samples)


def test_store_free_planar_update_inside_the_launch_with_soft_weights(golden):
    """seg_update with many samples carrying weight (a small workspace at temperature 20, the soft fixture's hyper-parameters):
    the weighted mean over the rows in ascending order, row by row through the wave's LDS block -- bit-identical to update_kernel
    -- and the per-particle row counts it leaves are update_kernel's."""
    soft = dict(SC.PLANAR, start=[-2.9, -2.9, 0., 0.], dt=0.5, cost_sigma_start=0.5, cost_sigma_gp=8., sigma_coll=0.4,
                sigma_goal_prior=2., sigma_start_sample=2., sigma_goal_sample=2., sigma_gp_sample=6.)
    goals = [[2.9, 2.6, 0., 0.], [2.9, -1.3, 0., 0.], [-1.3, 2.9, 0., 0.]]
    om = planar_map(golden, F32)
    build = lambda **kw: hip_planar_planner(soft, 64, goals, 7, 64, om, F32, seed=69, temperature=20., **kw)   # noqa: E731
    a, b = _store_free_twins(build, (6, 1, 4), {}, "fused_planar_seg_kernel")
    assert int(a._engine.row_counts().max()) > 8 and (a._engine.row_counts() == b._engine.row_counts()).all()


def test_store_free_permission_is_ignored_where_the_step_has_no_store_free_form(golden):
    """SGPMP_STEP_NO_SAMPLES is a permission: the tile launch of odd planar shapes, the two-launch paths, a program with an
    end-effector goal (ee_goal_kernel reads the rows) and fp64 contexts store as always -- and give what they always gave."""
    om = planar_map(golden, F32)
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.]]
    _store_free_twins(lambda **kw: hip_planar_planner(SC.PLANAR, 48, goals, 3, 24, om, F32, seed=68, **kw), (4, 2), {},
                      "fused_planar_kernel", expect_store_free=False)
    sph = torch.as_tensor(SC.panda_spheres(num=5)).to(**F64)
    _store_free_twins(lambda **kw: hip_panda_planner(SC.PANDA, 16, 4, 8, F64, seed=69, **kw), (3,), {"obstacle_spheres": sph},
                      "fused_step_f64_kernel", expect_store_free=False)


@pytest.mark.parametrize("temperature,extra", [(1e11, True), (1e14, True), (1e17, True), (1e13, False)])
def test_store_free_iterations_with_soft_weights(temperature, extra):
    """planner.py:263-275 is a softmax: at temperatures where several samples carry weight the update of a store-free
    iteration regenerates several rows (rounds of four), and particles whose previous update spread its weight over more than
    four rows have their rows WRITTEN as in a storing iteration (and, above S / 4, get the launch's softmax partials): whatever
    mix of the three a run takes, it is the mix the storing run takes -- every buffer bit-identical."""
    soft = dict(sigma_start_sample=1.0, sigma_goal_sample=1.0, sigma_gp_sample=30.0) if extra else {}
    c = dict(SC.PANDA, temperature=temperature, **soft)
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=23)).to(**F32)
    build = lambda **kw: hip_panda_planner(c, 32, 48, 64, F32, seed=71, **kw)                                 # noqa: E731
    a, _ = _store_free_twins(build, (6, 1, 5), {"obstacle_spheres": sph}, "fused_step_kernel")
    rc = a._engine.row_counts()
    print(f"\n[store-free, soft weights] temperature {temperature:g}: rows with weight per particle min {rc.min()} median "
          f"{int(np.median(rc))} max {rc.max()}")


def test_store_free_update_regenerates_any_number_of_rows():
    """The regeneration on its own, away from the thresholds: both planners get their row counts CLEARED before every step, so
    the store-free one never has a row in memory and regenerates every row that carries weight -- up to all S, in rounds of
    four with the sums carried through LDS -- while the storing one gathers them.  Same sums, bit for bit, for a Panda shape
    and a planar one (the segment recipe)."""
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=23)).to(**F32)
    c = dict(SC.PANDA, temperature=1e14, sigma_start_sample=1.0, sigma_goal_sample=1.0, sigma_gp_sample=30.0)
    a = hip_panda_planner(c, 32, 24, 64, F32, seed=73)
    b = hip_panda_planner(c, 32, 24, 64, F32, seed=73, store_free=False)
    most = 0
    for it in range(5):
        for pl in (a, b):
            pl._engine.set_row_counts(None)
        a.step(_samples_unread=True, obstacle_spheres=sph)
        b.step(obstacle_spheres=sph)
        assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a._grad, b._grad), it
        assert torch.equal(a._weights_buf, b._weights_buf) and torch.equal(a._costs, b._costs), it
        most = max(most, int(a._engine.row_counts().max()))
    assert a._engine.store_free_steps() == 5 and most > 16, most


@pytest.mark.parametrize("shape", ["example", "one_particle", "ragged", "long", "two_goals_sdf", "occupancy", "soft_weights",
                                   "store_free"])
def test_small_step_launch_equals_the_one_wave_per_item_launch_bitwise(shape):
    """A step of few items (the reference's own example: 5 particles x 32 samples x 64 waypoints, panda_environment.py:29-32) goes
    out as fused_step_small_kernel -- one WORKGROUP per item, its four waves on the item's chunks side by side, the recurrence's
    state handed from wave to wave -- instead of one wave per item (fused_step.inc: LAT).  Same counters, expressions and order
    of sums: samples, costs, means, weights, gradient bit-identical, for shapes on and off the 8 x 16 grid, T beyond one round of
    four chunks, several goals, every field type, spread weights (softmax partials re-read rows other waves stored) and
    store-free iterations (rows regenerated by update_kernel)."""
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=23)).to(**F32)
    kw, calls, cfg, args = {}, (1, 3, 1), SC.PANDA, (64, 5, 32)
    if shape == "one_particle":
        args = (16, 1, 8)
    elif shape == "ragged":
        args = (50, 7, 27)
    elif shape == "long":
        args = (176, 3, 40)                              # 11 chunks: three rounds, the last one short
    elif shape == "two_goals_sdf":
        kw = dict(goals=[SC.PANDA["goal_q"] + [0.] * 7, [0.3, 0.1, 0.2, -1.2, 0.0, 1.8, 0.2] + [0.] * 7], field_type="sdf")
        args = (32, 6, 16)
    elif shape == "occupancy":
        kw = dict(field_type="occupancy")
        args = (48, 4, 24)
    elif shape == "soft_weights":
        cfg = dict(SC.PANDA, temperature=1e14, sigma_start_sample=1.0, sigma_goal_sample=1.0, sigma_gp_sample=30.0)
        args, calls = (32, 6, 64), (1, 1, 4, 2)
    elif shape == "store_free":
        calls = (5, 1, 4)
    T, nppg, S = args
    a = hip_panda_planner(cfg, T, nppg, S, F32, seed=81, **kw)
    b = hip_panda_planner(cfg, T, nppg, S, F32, seed=81, **kw)
    a._engine.set_option("no_small_step", 0)             # (tests/conftest.py switches the small launch off for every other test)
    for k in calls:
        ra, rb = a.optimize(opt_iters=k, obstacle_spheres=sph), b.optimize(opt_iters=k, obstacle_spheres=sph)
        assert a._engine.last_cost_kernel() == "fused_step_small_kernel" and b._engine.last_cost_kernel() == "fused_step_kernel"
        for i, (x, y) in enumerate(zip(ra, rb)):
            assert torch.equal(x, y), (k, i)
        assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a.state_samples, b.state_samples), k
        assert torch.equal(a._weights_buf, b._weights_buf) and torch.equal(a._grad, b._grad) and torch.equal(a._costs, b._costs), k
    if shape == "soft_weights":
        assert a._engine.dense_armed_steps() > 0 and int(a._engine.row_counts().max()) > 16
    if shape == "store_free":
        assert a._engine.store_free_steps() == 7 and b._engine.store_free_steps() == 7
    # above the size bar the step is the one-wave-per-item launch again (512 items by default -- 256 off the 8 x 16 grid; here: 3)
    a._engine.set_option("small_step_items", 3)
    a.optimize(opt_iters=1, obstacle_spheres=sph)
    assert a._engine.last_cost_kernel() == ("fused_step_small_kernel" if shape == "one_particle" else "fused_step_kernel")


def test_small_step_launch_costs_bitwise_over_many_seeds():
    """The rare event the shapes above could miss: the small-step launch once ran the MASKED instantiation for every shape, and
    against the unmasked one-wave-per-item launch of an on-grid shape ~0.1 % of the costs came out one ulp apart (hipcc pairs the
    cost terms' multiplies and adds differently around the masks' branches).  Both launches now take the instantiation the shape
    calls for: 12 seeds x 3 iterations x 160 costs at the reference's example size, near and far obstacles -- every bit equal."""
    for far in (False, True):
        for seed in range(12):
            sph = torch.as_tensor(SC.panda_spheres(num=5, seed=seed)).to(**F32)
            if far:
                sph = sph.clone()
                sph[:, :3] += 50.0
            a = hip_panda_planner(SC.PANDA, 64, 5, 32, F32, seed=seed)
            b = hip_panda_planner(SC.PANDA, 64, 5, 32, F32, seed=seed)
            a._engine.set_option("no_small_step", 0)
            for it in range(3):
                a.optimize(obstacle_spheres=sph)
                b.optimize(obstacle_spheres=sph)
                assert a._engine.last_cost_kernel() == "fused_step_small_kernel" and b._engine.last_cost_kernel() == "fused_step_kernel"
                assert torch.equal(a._costs, b._costs) and torch.equal(a.particle_means, b.particle_means), (far, seed, it)


def test_store_free_steps_are_taken_where_they_pay(golden):
    """The permission is not an order: a step that would REGENERATE rows in update_kernel runs store-free only when the bytes it
    does not write outweigh the regeneration (2.8 MB per waypoint of all samples, measured: tools/store_free_sizes.py) -- a
    small problem stores, config 3 does not, and the results are the same bits either way.  (tests/conftest.py lowers the bar
    to 1 byte for every other test.)"""
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=23)).to(**F32)
    small = [hip_panda_planner(SC.PANDA, 32, 16, 64, F32, seed=77) for _ in range(2)]
    small[0]._engine.set_option("store_free_min_bytes", 0)            # the default
    for pl in small:
        pl.optimize(opt_iters=5, obstacle_spheres=sph)
    assert small[0]._engine.store_free_steps() == 0 and small[1]._engine.store_free_steps() == 4
    assert torch.equal(small[0].particle_means, small[1].particle_means) and torch.equal(small[0].state_samples, small[1].state_samples)
    big = hip_panda_planner(SC.PANDA, 64, 1024, 128, F32, seed=78)    # BASELINE configs[2]: 7.3 MB per waypoint
    big._engine.set_option("store_free_min_bytes", 0)
    big.optimize(opt_iters=3, obstacle_spheres=sph)
    assert big._engine.store_free_steps() == 2
    # the planar problems whose update runs inside the launch have nothing to regenerate: store-free at any size
    pl = hip_planar_planner(SC.PLANAR, 64, [[2.9, 2.6, 0., 0.]], 4, 64, planar_map(golden, F32), F32, seed=79)
    pl._engine.set_option("store_free_min_bytes", 0)
    pl.optimize(opt_iters=3)
    assert pl._engine.store_free_steps() == 2


def test_store_free_row_counts_travel_with_the_state():
    """The per-particle row counts steer the next step (partials / stored rows / regenerated rows): state_dict carries them,
    reset() clears them -- a resumed soft-weight run continues bit for bit, also when it resumes in the middle of what would
    have been one optimize() call."""
    c = dict(SC.PANDA, temperature=1e14, sigma_start_sample=1.0, sigma_goal_sample=1.0, sigma_gp_sample=30.0)
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=23)).to(**F32)
    mk = lambda: hip_panda_planner(c, 32, 24, 64, F32, seed=75)                                              # noqa: E731
    straight = mk()
    straight.optimize(opt_iters=3, obstacle_spheres=sph)
    straight.optimize(opt_iters=4, obstacle_spheres=sph)
    a = mk()
    a.optimize(opt_iters=3, obstacle_spheres=sph)
    sd = a.state_dict()
    assert int(sd['row_counts'].max()) > 16
    b = mk()
    assert int(b._engine.row_counts().max()) == 0
    b.load_state_dict(sd)
    b.optimize(opt_iters=4, obstacle_spheres=sph)
    assert torch.equal(b.particle_means, straight.particle_means) and torch.equal(b.state_samples, straight.state_samples)
    # without the counts the continuation differs in the last bits (rows gathered where the straight run added partials)
    sd2 = dict(sd)
    sd2.pop('row_counts')
    d = mk()
    d.load_state_dict(sd2)
    d.optimize(opt_iters=4, obstacle_spheres=sph)
    assert float((d.particle_means - straight.particle_means).abs().max()) <= 1e-5 * float(straight.particle_means.abs().max())
    b.reset()
    assert int(b._engine.row_counts().max()) == 0


# --------------------------------------------------------------------------- any serial chain on the fast launches
# The reference takes any FK callable (cost_functions.py:39,51-52).  The library is built with straight-line chain code for
# the Panda only; another chain's code is generated by the host (csrc/gen/chain_codegen.py) and compiled by the library at
# run time (csrc/chain_rtc.hip, hiprtc): these tests put two such robots on the fused launch and the chunked sweep and
# check them against the oracle running oracle/fk.py on the SAME chain.
def _arm7():
    """A 7-DoF arm that is NOT the Panda: its link lengths / offsets changed, another flange."""
    from stoch_gpmp_amd.robots.panda_chain import PANDA_CHAIN
    ch = [(nm, kind, tuple(rpy), list(xyz)) for nm, kind, rpy, xyz in PANDA_CHAIN]
    ch[0][3][2] = 0.36; ch[2][3][1] = -0.29; ch[3][3][0] = 0.07; ch[4][3][0] = -0.07; ch[4][3][1] = 0.41; ch[6][3][0] = 0.1
    ch[7][3][2] = 0.15                                   # a longer flange ...
    ch[8] = (ch[8][0], ch[8][1], (0.0, 0.0, 0.3), ch[8][3])          # ... turned differently
    return [(nm, kind, tuple(rpy), tuple(xyz)) for nm, kind, rpy, xyz in ch]


def _arm6():
    """A 6-DoF arm (UR-like proportions): six revolute joints and a tool flange."""
    h = 1.57079632679
    return [("j1", "revolute", (0.0, 0.0, 0.0), (0.0, 0.0, 0.1625)), ("j2", "revolute", (h, 0.0, 0.0), (0.0, 0.0, 0.0)),
            ("j3", "revolute", (0.0, 0.0, 0.0), (-0.425, 0.0, 0.0)), ("j4", "revolute", (0.0, 0.0, 0.0), (-0.3922, 0.0, 0.1333)),
            ("j5", "revolute", (h, 0.0, 0.0), (0.0, -0.0997, 0.0)), ("j6", "revolute", (-h, 0.0, 0.0), (0.0, 0.0996, 0.0)),
            ("tool", "fixed", (0.0, 0.0, 0.0), (0.0, 0.0, 0.12))]


def _arm_config(chain):
    n = sum(1 for j in chain if j[1] == "revolute")
    c = dict(SC.PANDA, n_dof=n)
    if n == 6:
        c.update(start_q=[0.1, -1.2, 1.4, -0.4, 0.8, 0.2], goal_q=[0.9, -0.7, 0.9, 0.3, 1.1, -0.4])
    return c, n


@pytest.mark.parametrize("arm,field_type", [("arm7", "rbf"), ("arm7", "sdf"), ("arm6", "rbf")])
def test_any_serial_chain_runs_the_fused_launch_through_run_time_chain_code(arm, field_type):
    """A robot the library was not built for: its chain code is generated at set-up and compiled with hiprtc, the planner
    launches `fused_step_kernel (run-time chain code)`, and costs / means follow the fp64 oracle on the restated noise (means
    re-synchronised per iteration, as in _fp32_panda_run); the stand-alone sweep (`sample_and_eval`) takes the chain's
    chunked sweep and agrees with the fused launch's costs bit for bit."""
    from oracle.native_noise import native_eps
    chain = _arm7() if arm == "arm7" else _arm6()
    c, n = _arm_config(chain)
    T, nppg, S, seed, iters = 32, 24, 32, 19, 3
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=seed))
    eps0 = torch.from_numpy(native_eps(seed, 0, range(1), nppg, T, n, "float32")).double()
    ora = SC.oracle_panda_planner(c, T, nppg, S, seed=seed, eps_init=eps0, field_type=field_type, chain=chain)
    pl = hip_panda_planner(c, T, nppg, S, F32, seed=seed, field_type=field_type, chain=chain)
    cid, secs, compiled, cached = pl._engine.fk_codegen_info()
    assert cid == 2, (cid, pl._engine.fk_codegen_error)
    print(f"\n[run-time chain code] {arm} {field_type}: hiprtc {secs:.2f} s, {compiled} compiled, {cached} from the disk cache")
    assert rel_err(pl.particle_means, ora.particle_means) < 2e-5
    pl.particle_means.copy_(ora.particle_means.to(**F32))
    scale = float(ora.particle_means.abs().max())
    within = []
    for it in range(iters):
        eps = torch.from_numpy(native_eps(seed, 2 + it, range(nppg), S, T, n, "float32")).double()
        ora.particle_means.copy_(pl.particle_means.cpu().double())
        ora.prior.set_mean(ora.particle_means.view(nppg, -1))
        costs_o, _ = ora.step(eps=eps, obstacle_spheres=sph)
        _, _, _, _, costs, _ = pl.optimize(obstacle_spheres=sph.to(**F32))
        assert pl._engine.last_cost_kernel() == "fused_step_kernel (run-time chain code)", pl._engine.last_cost_kernel()
        assert float((pl.state_samples.cpu().double() - ora.state_samples).abs().max()) < 2e-5 * float(ora.state_samples.abs().max())
        c32 = costs.cpu().double()
        assert float(((c32 - costs_o).abs() / costs_o.abs()).max()) < 2e-4
        d = (pl.particle_means.cpu().double() - ora.particle_means).abs().amax(dim=(1, 2)) / scale
        within.append(float((d < 1e-3).double().mean()))
    assert np.mean(within) >= 0.98, within
    # the stand-alone sweep of the same samples: the chain's chunked sweep, same costs as the fused launch computed
    fused_costs = pl._costs.clone()
    isw = pl._engine.is_weights(pl._means_prev.contiguous(), pl.temperature)
    again = pl._engine.cost_eval(pl.state_samples, batch_offset=0, spheres=sph.to(**F32).reshape(-1, 4).contiguous(),
                                 is_weights=isw, rows_per_particle=S).reshape(nppg, S)
    assert pl._engine.last_cost_kernel() == "cost_sweep_chunked_kernel (run-time chain code)", pl._engine.last_cost_kernel()
    assert torch.equal(again, fused_costs)
    # ... and against the generic sweep (run-time constants through LDS): same numbers to fp32 rounding
    pl._engine.set_option("no_chain_codegen", 1)
    generic = pl._engine.cost_eval(pl.state_samples, batch_offset=0, spheres=sph.to(**F32).reshape(-1, 4).contiguous(),
                                   is_weights=isw, rows_per_particle=S).reshape(nppg, S)
    assert "run-time chain code" not in pl._engine.last_cost_kernel()
    assert rel_err(generic, fused_costs) < 2e-5
    # S and T off the fused launch's grid of 8 rows x 16 waypoints: the chain's masked instantiation, against the two-launch path
    a2 = hip_panda_planner(c, 24, 5, 12, F32, seed=seed, field_type=field_type, chain=chain)
    b2 = hip_panda_planner(c, 24, 5, 12, F32, seed=seed, field_type=field_type, chain=chain)
    b2._engine.set_option("no_fused_step", 1)
    for it in range(2):
        a2.optimize(obstacle_spheres=sph.to(**F32))
        b2.optimize(obstacle_spheres=sph.to(**F32))
        assert a2._engine.last_cost_kernel() == "fused_step_kernel (run-time chain code)", a2._engine.last_cost_kernel()
        assert torch.equal(a2.state_samples, b2.state_samples) and rel_err(a2._costs, b2._costs) < 2e-5
        assert float((a2.particle_means - b2.particle_means).abs().max()) <= 1e-6 * float(b2.particle_means.abs().max())
    # ... and the chain's small-step launch (one workgroup per item; tests/conftest.py keeps it off elsewhere): the same samples
    # bit for bit, the costs to fp32 rounding (like the two-launch path above: for run-time chain code every launch is a
    # compilation of its own, and hiprtc need not pair the cost terms' multiplies and adds alike in each; the built-in chain's
    # launches ARE bit-identical: test_small_step_launch_equals_the_one_wave_per_item_launch_bitwise)
    a3 = hip_panda_planner(c, 24, 5, 12, F32, seed=seed, field_type=field_type, chain=chain)
    a3._engine.set_option("no_small_step", 0)
    for it in range(2):
        a3.optimize(obstacle_spheres=sph.to(**F32))
        assert a3._engine.last_cost_kernel() == "fused_step_small_kernel (run-time chain code)", a3._engine.last_cost_kernel()
    assert torch.equal(a3.state_samples, a2.state_samples) and rel_err(a3._costs, a2._costs) < 2e-6
    assert float((a3.particle_means - a2.particle_means).abs().max()) <= 1e-6 * float(a2.particle_means.abs().max())
    print(f"[run-time chain code] small-step launch against the one-wave launch: {int((a3._costs != a2._costs).sum())} of {a3._costs.numel()} costs differ")


def test_chain_code_of_another_robot_is_refused():
    """sgpmp_set_fk_codegen checks the code against the chain of sgpmp_set_fk (forward kinematics at random joint vectors,
    link / pair tables): code generated for ANOTHER chain must not be accepted -- the chain then stays on the generic sweep."""
    from stoch_gpmp_amd.engine import Engine, _chain_struct_source
    from stoch_gpmp_amd import _lib as L
    import ctypes
    eng = Engine(7, 16, 4, 8, tensor_args=F32)
    eng.set_fk(_arm7(), codegen=False)
    assert eng.fk_codegen_info()[0] == 0
    other = [(nm, kind, rpy, (xyz[0], xyz[1], xyz[2] + (0.01 if i == 2 else 0.0))) for i, (nm, kind, rpy, xyz) in enumerate(_arm7())]
    rc = eng.lib.sgpmp_set_fk_codegen(eng._ctx, _chain_struct_source(other).encode())
    assert rc == L.EINVAL and "does not" in L.last_error(), (rc, L.last_error())
    assert eng.fk_codegen_info()[0] == 0
    rc = eng.lib.sgpmp_set_fk_codegen(eng._ctx, b"struct Nonsense {};" + b" " * 40)
    assert rc == L.EINVAL
    rc = eng.lib.sgpmp_set_fk_codegen(eng._ctx, _chain_struct_source(_arm7()).encode())
    assert rc == L.OK and eng.fk_codegen_info()[0] == 2
    eng.set_fk(_arm6() + [("pad", "revolute", (0., 0., 0.), (0., 0., 0.1))], codegen=False)      # a new chain resets it
    assert eng.fk_codegen_info()[0] == 0


def test_run_time_chain_code_rate_against_the_built_in_panda():
    """Done-criterion of the round-3 verdict: a non-Panda 7-DoF chain on the fused launch within 10 % of the Panda's rate
    (same kernel source around another chain's constants), measured on one box in one process at config 3's size."""
    import time
    sph = torch.as_tensor(SC.panda_spheres(num=5)).to(**F32)
    rates = {}
    for name, chain in (("panda", None), ("arm7", _arm7())):
        pl = hip_panda_planner(SC.PANDA, 64, 1024, 128, F32, seed=1, chain=chain)
        pl.optimize(opt_iters=150, obstacle_spheres=sph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl.optimize(opt_iters=200, obstacle_spheres=sph)
        torch.cuda.synchronize()
        rates[name] = 200 / (time.perf_counter() - t0)
        assert pl._engine.last_cost_kernel().startswith("fused_step_kernel"), pl._engine.last_cost_kernel()
        del pl
    print(f"\n[run-time chain code] iterations/s at 1024 x 128 x 64: {rates}")
    assert rates["arm7"] > 0.9 * rates["panda"], rates


# --------------------------------------------------------------------------- checkpoint / resume
@pytest.mark.parametrize("kind", ["panda", "planar"])
def test_state_dict_resume_continues_bit_for_bit(kind, golden):
    """5 iterations, state_dict(), a NEW planner, load_state_dict(), 5 more == 10 straight, bit for bit -- and the same
    state resumed as 8 sequential world_size-8 shards (the noise is keyed on the global particle index, so a state saved at
    world 1 feeds any sharding).  Reference state: particle_means + the generator (planner.py:215,243,270)."""
    if kind == "panda":
        sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
        obs = {"obstacle_spheres": sph}
        mk = lambda **kw: hip_panda_planner(SC.PANDA, 32, 64, 32, F32, seed=9, **kw)          # noqa: E731
    else:
        om = planar_map(golden, F32)
        obs = {}
        goals = [[9., 6., 0., 0.], [9., -3., 0., 0.]]
        mk = lambda **kw: hip_planar_planner(SC.PLANAR, 64, goals, 32, 64, om, F32, seed=9, **kw)   # noqa: E731
    straight = mk()
    for k in (3, 2, 1, 4):                               # (calls of several iterations and single ones)
        straight.optimize(opt_iters=k, **obs)
    a = mk()
    a.optimize(opt_iters=3, **obs)
    a.optimize(opt_iters=2, **obs)
    sd = a.state_dict()
    stats5 = a.global_stats()
    import copy
    import io
    buf = io.BytesIO()
    torch.save(sd, buf)                                  # (a state is plain tensors and numbers: it survives torch.save)
    sd = torch.load(io.BytesIO(buf.getvalue()), weights_only=False)
    b = mk()                                             # a fresh planner of the same problem
    b.optimize(opt_iters=1, **obs)                       # ... that has already moved on
    b.load_state_dict(copy.deepcopy(sd))
    assert torch.equal(b.particle_means, a.particle_means) and b._draw == a._draw
    assert b.global_stats() == stats5
    b.optimize(opt_iters=1, **obs)
    b.optimize(opt_iters=4, **obs)
    assert torch.equal(b.particle_means, straight.particle_means)
    assert torch.equal(b._costs, straight._costs) and torch.equal(b.state_samples, straight.state_samples)
    # the same state, resumed shard by shard
    for r in range(8):
        sh = mk(rank=r, world_size=8)
        sh.load_state_dict(sd)
        sh.optimize(opt_iters=1, **obs)
        sh.optimize(opt_iters=4, **obs)
        assert torch.equal(sh.particle_means, straight.particle_means[sh.p0:sh.p1]), r
        assert torch.equal(sh._costs, straight._costs[sh.p0:sh.p1]), r
        # a shard's state does not cover another shard
        if r == 3:
            part = sh.state_dict()
            with pytest.raises(ValueError):
                mk(rank=4, world_size=8).load_state_dict(part)
            mk(rank=3, world_size=8).load_state_dict(part)
    with pytest.raises(ValueError):                      # another problem
        (hip_panda_planner(SC.PANDA, 32, 32, 32, F32, seed=9) if kind == "panda" else
         hip_planar_planner(SC.PLANAR, 64, goals, 16, 64, om, F32, seed=9)).load_state_dict(sd)


def test_optimize_returns_clones_of_the_pre_update_means():
    """planner.py:252-253: the reference returns CLONES of the (pre-update) means; so does optimize() -- a result kept
    across calls does not change under the caller.  clone_outputs=False hands out views of the persistent buffer."""
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    a = hip_panda_planner(SC.PANDA, 16, 8, 8, F32, seed=4)
    before = a.particle_means.clone()
    sp1, cp1 = a.optimize(opt_iters=1, obstacle_spheres=sph)[:2]
    assert torch.equal(sp1, before[..., :7]) and torch.equal(cp1, before[..., 7:])
    keep = sp1.clone()
    mid = a.particle_means.clone()
    sp2 = a.optimize(opt_iters=1, obstacle_spheres=sph)[0]
    assert torch.equal(sp1, keep) and torch.equal(sp2, mid[..., :7]) and sp1.data_ptr() != sp2.data_ptr()
    b = hip_panda_planner(SC.PANDA, 16, 8, 8, F32, seed=4, clone_outputs=False)
    v1 = b.optimize(opt_iters=1, obstacle_spheres=sph)[0]
    v2 = b.optimize(opt_iters=1, obstacle_spheres=sph)[0]
    assert v1.data_ptr() == v2.data_ptr() and torch.equal(v2, mid[..., :7])


def test_planar_seg_launch_with_two_waves_zeroes_all_statistics_shards(golden):
    """fused_planar_seg_kernel runs 64 * (T / 8) threads: at T = 16 that is 128, fewer than the 256 statistics words its
    first workgroup zeroes -- with more than 32 particles, shards 32..63 accumulated across kept-means steps (round-3
    advisor finding).  The statistics of several consecutive steps against the tile launch's."""
    om = planar_map(golden, F32)
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.]]
    a = hip_planar_planner(SC.PLANAR, 16, goals, 48, 64, om, F32, seed=5)
    b = hip_planar_planner(SC.PLANAR, 16, goals, 48, 64, om, F32, seed=5)
    b._engine.set_option("no_planar_seg", 1)
    for it in range(4):
        a.optimize(opt_iters=1)
        b.optimize(opt_iters=1)
        assert a._engine.last_cost_kernel() == "fused_planar_seg_kernel" and b._engine.last_cost_kernel() == "fused_planar_kernel"
        sa, sb = a.global_stats(), b.global_stats()
        raw = a._stats[a._stats_slot ^ 1].sum(0).cpu()
        assert float(raw[2]) == 96.0, (it, raw)                       # every particle counted exactly once
        assert abs(sa[0] / sb[0] - 1) < 2e-3 and abs(sa[1] / sb[1] - 1) < 2e-3, (it, sa, sb)   # (two kernels: costs agree to fp32 rounding of the samples)


# --------------------------------------------------------------------------- API surface
def test_api_surface_and_errors(golden):
    om = planar_map(golden, F32)
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.]]
    pl = hip_planar_planner(SC.PLANAR, 16, goals, 3, 8, om, F32, seed=0)
    assert pl.particle_means.shape == (6, 16, 4) and pl.state_samples.shape == (6, 8, 16, 4)
    assert pl.Sigma_inv.shape == (64, 64)
    vel, pos, vmean, pmean, costs = pl.sample_and_eval()
    assert vel.shape == (6, 8, 16, 2) and pos.shape == (6, 8, 16, 2) and costs.shape == (6, 8)
    grad = pl._update_distribution(costs, pl.state_samples)
    assert grad.shape == (6, 16, 4) and pl._weights.shape == (6, 8, 1, 1)
    assert abs(float(pl._weights.sum()) - 6.0) < 1e-4
    pos2, vel2 = pl.sample_trajectories(5)
    assert pos2.shape == (6, 5, 16, 2)
    with pytest.raises(AssertionError):
        hip_planar_planner(SC.PLANAR, 16, torch.zeros(4), 3, 8, om, F32)      # goals must be 2-D
    with pytest.raises(RuntimeError):                                          # no CPU path
        hip_planar_planner(SC.PLANAR, 16, goals, 3, 8, om,
                           {"device": torch.device("cpu"), "dtype": torch.float32})

    class ForeignCost:                                   # any object with .eval is a legal cost
        def eval(self, trajs, **obs):
            return (trajs[..., :2] ** 2).sum((-1, -2)).reshape(-1)
    from stoch_gpmp_amd.planner import StochGPMP
    ta = F32
    p2 = StochGPMP(num_particles_per_goal=2, num_samples=4, traj_len=8, opt_iters=1, dt=0.1, n_dof=2,
                   start_state=torch.zeros(4, **ta), multi_goal_states=torch.ones(1, 4, **ta),
                   cost=ForeignCost(), sigma_start_init=0.1, sigma_start_sample=0.1, sigma_goal_init=0.1,
                   sigma_goal_sample=0.1, sigma_gp_init=1., sigma_gp_sample=1., seed=0, tensor_args=ta)
    before = p2.particle_means.clone()
    out = p2.optimize(opt_iters=2)
    assert out[4].shape == (2, 4) and not torch.equal(before, p2.particle_means)


# --------------------------------------------------------------------------- fused sampler + sweep launch
@pytest.mark.parametrize("nppg,G,S,T,field_type,n_sph,fused", [
    (3, 1, 8, 16, "rbf", 5, True),            # smallest: one item per particle, one chunk
    (3, 2, 24, 48, "sdf", 9, True),           # two goals, three items per particle (not a power of two), three chunks
    (5, 1, 16, 32, "occupancy", 1, True),
    (2, 1, 8, 80, "rbf", 64, True),           # five chunks, the sphere-staging limit of the fused launch
    (2, 1, 8, 32, "sdf", 64, True),
    (2, 1, 8, 32, "rbf", 65, False),          # one sphere too many -> sampler + two-trajectory sweep
    (2, 1, 12, 32, "rbf", 5, True),           # S not a multiple of 8: the launch's masked instantiation (round 4)
    (2, 1, 8, 24, "rbf", 5, True),            # T not a multiple of 16: likewise
    (2, 2, 12, 40, "sdf", 5, True),           # both, two goals
    (2, 1, 8, 25, "rbf", 5, False),           # T odd (rows of 14 T floats are not 16-byte multiples) -> two launches
])
def test_fused_step_corners_match_the_two_launch_path(nppg, G, S, T, field_type, n_sph, fused):
    """sgpmp_step's fused launch (fused_step.inc) against the same step as sampler + sweep: identical
    noise keys, so the samples agree to the last bit or two (the two samplers order one fused
    multiply-add differently at tiny sizes) and the costs to fp32 rounding; the dispatcher's choice
    is asserted for every corner, including the fall-backs."""
    c, n = SC.PANDA, 7
    goals = None if G == 1 else [c["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n][:G]
    sph = torch.as_tensor(SC.panda_spheres(num=n_sph, seed=3)).to(**F32)
    a = hip_panda_planner(c, T, nppg, S, F32, field_type=field_type, seed=17, goals=goals)
    b = hip_panda_planner(c, T, nppg, S, F32, field_type=field_type, seed=17, goals=goals)
    b._engine.set_option("no_fused_step", 1)
    for it in range(3):
        a.optimize(obstacle_spheres=sph)
        b.optimize(obstacle_spheres=sph)
        assert (a._engine.last_cost_kernel() == "fused_step_kernel") == fused, a._engine.last_cost_kernel()
        assert b._engine.last_cost_kernel() != "fused_step_kernel"
        scale = float(b.state_samples.abs().max())
        assert float((a.state_samples - b.state_samples).abs().max()) <= 4e-7 * scale
        assert rel_err(a._costs, b._costs) < 2e-5
        assert torch.equal(a._costs.argmin(1), b._costs.argmin(1))
        assert float((a.particle_means - b.particle_means).abs().max()) <= 1e-6 * scale
        b.particle_means.copy_(a.particle_means)


def test_fused_step_random_shapes_match_the_two_launch_path():
    """Twelve seeded random shapes inside the fused launch's domain (S multiple of 8, T multiple of 16, 1-3 goals,
    1-64 spheres, every field type, with and without the clamp): fused launch vs sampler + sweep, as in the
    corner test above."""
    rng = np.random.default_rng(2024)
    c, n = SC.PANDA, 7
    all_goals = [c["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n,
                 [0.8, -0.2, 0.1, -1.0, -0.4, 1.1, 0.9] + [0.] * n]
    for trial in range(12):
        nppg, G = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        S, T = 8 * int(rng.integers(1, 6)), 16 * int(rng.integers(1, 7))
        ft = ["rbf", "sdf", "occupancy"][int(rng.integers(0, 3))]
        n_sph = int(rng.integers(1, 65))
        goals = None if G == 1 else all_goals[:G]
        sph = torch.as_tensor(SC.panda_spheres(num=n_sph, seed=trial)).to(**F32)
        a = hip_panda_planner(c, T, nppg, S, F32, field_type=ft, seed=100 + trial, goals=goals)
        b = hip_panda_planner(c, T, nppg, S, F32, field_type=ft, seed=100 + trial, goals=goals)
        b._engine.set_option("no_fused_step", 1)
        tag = f"trial {trial}: nppg={nppg} G={G} S={S} T={T} {ft} spheres={n_sph}"
        for it in range(2):
            a.optimize(obstacle_spheres=sph)
            b.optimize(obstacle_spheres=sph)
            assert a._engine.last_cost_kernel() == "fused_step_kernel", tag
            assert b._engine.last_cost_kernel() != "fused_step_kernel", tag
            scale = float(b.state_samples.abs().max())
            assert float((a.state_samples - b.state_samples).abs().max()) <= 4e-7 * scale, tag
            assert rel_err(a._costs, b._costs) < 2e-5, tag
            assert float((a.particle_means - b.particle_means).abs().max()) <= 1e-6 * scale or \
                not torch.equal(a._costs.argmin(1), b._costs.argmin(1)), tag      # (a near-tie may flip the arg-min)
            b.particle_means.copy_(a.particle_means)


@pytest.mark.parametrize("nppg,G,S,T,fused", [
    (2, 2, 8, 16, True), (3, 2, 24, 48, True), (5, 1, 16, 128, True), (1, 4, 8, 32, True),
    (2, 2, 12, 32, False),                    # S not a multiple of 8
    (2, 2, 8, 24, False),                     # T not a multiple of 16
])
def test_planar_fused_corners_match_the_two_launch_path(golden, nppg, G, S, T, fused):
    """The planar fused launch (fused_planar.inc) at its dispatch corners against sampler + generic sweep."""
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]][:G]
    om = planar_map(golden, F32)
    a = hip_planar_planner(SC.PLANAR, T, goals, nppg, S, om, F32, seed=31)
    b = hip_planar_planner(SC.PLANAR, T, goals, nppg, S, om, F32, seed=31)
    b._engine.set_option("no_fused_step", 1)
    for it in range(3):
        a.optimize()
        b.optimize()
        assert a._engine.last_cost_kernel().startswith("fused_planar") == fused, a._engine.last_cost_kernel()
        assert b._engine.last_cost_kernel() == "cost_sweep_kernel<f32, no FK>"
        scale = float(b.state_samples.abs().max())
        assert float((a.state_samples - b.state_samples).abs().max()) <= 4e-7 * scale
        assert rel_err(a._costs, b._costs) < 2e-5
        assert torch.equal(a._costs.argmin(1), b._costs.argmin(1))
        b.particle_means.copy_(a.particle_means)


@pytest.mark.parametrize("nppg,G,S,T,n", [(3, 2, 64, 64, 2), (2, 4, 64, 128, 2), (1, 3, 128, 32, 2), (5, 1, 64, 256, 2),
                                            (2, 2, 64, 16, 2), (2, 2, 192, 16, 2), (3, 1, 64, 96, 3), (2, 2, 64, 128, 3),
                                            (2, 2, 32, 64, 2), (2, 1, 64, 144, 2)])
def test_planar_launch_with_a_lane_per_sample_matches_the_tile_launch(golden, nppg, G, S, T, n):
    """fused_planar_seg.inc: lane = sample, wave = segment of 8 (16) waypoints, everything indexed by the waypoint in
    scalar registers, three barriers and a few LDS words per lane -- against fused_planar_kernel (8 samples per wave
    through an LDS tile; `no_planar_seg`) and, every iteration, against the sampler + generic sweep as two launches: same
    noise keys, samples to rounding, costs to fp32 rounding, same arg-min, same means after the update.  The last two
    shapes (S = 32; T = 144 = 18 segments) do not fit and run the tile launch."""
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]][:G]
    c = SC.PLANAR
    if n == 3:
        goals = [g[:2] + [1.0] + [0., 0., 0.] for g in goals]
        c = dict(SC.PLANAR, n_dof=3, start=[-9., -9., 0.5, 0., 0., 0.])
    om = planar_map(golden, F32)
    mk = lambda: hip_planar_planner(c, T, goals, nppg, S, om, F32, seed=59)   # noqa: E731
    a, b, two = mk(), mk(), mk()
    b._engine.set_option("no_planar_seg", 1)
    two._engine.set_option("no_fused_step", 1)
    fits = S % 64 == 0 and T % 8 == 0 and T // (8 if T <= 128 else 16) <= 16
    for it in range(3):
        a.optimize()
        b.optimize()
        two.optimize()
        assert a._engine.last_cost_kernel() == ("fused_planar_seg_kernel" if fits else "fused_planar_kernel")
        assert b._engine.last_cost_kernel() == "fused_planar_kernel"
        assert two._engine.last_cost_kernel() == "cost_sweep_kernel<f32, no FK>"
        scale = float(two.state_samples.abs().max())
        for ref in (b, two):
            assert float((a.state_samples - ref.state_samples).abs().max()) <= 1e-6 * scale, it
            assert rel_err(a._costs, ref._costs) < 2e-5
            assert torch.equal(a._costs.argmin(1), ref._costs.argmin(1))
            assert float((a.particle_means - ref.particle_means).abs().max()) <= 2e-5 * float(ref.particle_means.abs().max())
            ref.particle_means.copy_(a.particle_means)


@pytest.mark.parametrize("kind", ["panda", "panda_two_goals_sdf", "panda_ee_goal", "planar", "planar_tile"])
def test_pipelined_optimize_equals_single_steps_bitwise(golden, kind):
    """optimize(opt_iters=K) runs its iterations as two particle-half chains on the context's own streams
    (sgpmp_pipeline_begin / _end); a twin that takes the same iterations one optimize(opt_iters=1) at a time -- one
    chain, every step on the caller's stream -- must end with the same bits in every buffer, through several calls,
    a single-iteration call in between, moved obstacles and an edit of the means between two calls."""
    if kind.startswith("planar"):
        goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
        om = planar_map(golden, F32)
        tile = kind == "planar_tile"
        # (fused_planar_seg_kernel splits once a half brings a 16-wave workgroup per CU: 2 x 256 particles of 64 samples;
        # fused_planar_kernel, 8 samples per wave, from 2 x 128)

        def mk(**kw):
            pl = hip_planar_planner(SC.PLANAR, 64, goals, 64 if tile else 128, 64, om, F32, seed=41, **kw)
            pl._engine.set_option("no_planar_seg", 1 if tile else 0)
            return pl
        obs1 = obs2 = {}
        name = "fused_planar_kernel" if tile else "fused_planar_seg_kernel"
    else:
        c, n = SC.PANDA, 7
        two = kind == "panda_two_goals_sdf"
        g = [c["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n] if two else None

        def mk(**kw):
            pl = hip_panda_planner(c, 32, 64 if two else 129, 128, F32, seed=41, goals=g,
                                   field_type="sdf" if two else "rbf", **kw)
            if kind == "panda_ee_goal":          # CostGoal (end-effector term): evaluated by each half's update kernel
                from stoch_gpmp_amd.costs.cost_functions import CostGoal
                from stoch_gpmp_amd.costs.fields import EESE3DistanceField
                from oracle.fk import fk_all_links
                H = fk_all_links(torch.tensor([[0.3, -0.5, 0.2, -1.9, 0.1, 1.6, 0.4]], dtype=torch.float64))[0, -1].clone()
                pl.cost.cost_list.append(CostGoal(n, 32, field=EESE3DistanceField(H, w_pos=1., w_rot=0.5, tensor_args=F32),
                                                  sigma_goal=1e-2, tensor_args=F32))
            return pl
        obs1 = {"obstacle_spheres": torch.as_tensor(SC.panda_spheres(num=5, seed=3)).to(**F32)}
        obs2 = {"obstacle_spheres": torch.as_tensor(SC.panda_spheres(num=9, seed=4)).to(**F32)}
        name = "fused_step_kernel"
    # a: the product's default -- ONE sgpmp_optimize call per optimize() (the K-loop, the flags, the two-chain bracket on the C
    # side, round 6); b: rounds 1-5's host loop, one sgpmp_step call per iteration from Python, single chain
    a, b = mk(), mk(pipeline_steps=False, c_loop=False)
    assert a.c_loop and not b.c_loop

    def same():
        torch.cuda.synchronize()
        for x, y in ((a.particle_means, b.particle_means), (a.state_samples, b.state_samples), (a._costs, b._costs),
                     (a._weights_buf, b._weights_buf), (a._grad, b._grad), (a._means_prev, b._means_prev)):
            assert torch.equal(x, y)
        sa, sb = a.global_stats(), b.global_stats()
        assert abs(sa[0] - sb[0]) <= 1e-12 * abs(sb[0]) and abs(sa[1] - sb[1]) <= 1e-12 * abs(sb[1])

    def both(k, obs):
        ra = a.optimize(opt_iters=k, **obs)
        for _ in range(k):
            rb = b.optimize(opt_iters=1, **obs)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y)

    both(5, obs1)
    assert a._engine.last_cost_kernel().startswith(name)
    assert a._engine.pipeline_split_steps() == 5 and b._engine.pipeline_split_steps() == 0
    same()
    both(1, obs1)                                        # a single iteration: the ordinary step
    assert a._engine.pipeline_split_steps() == 5
    same()
    both(4, obs2)                                        # other obstacles
    same()
    for pl in (a, b):                                    # the caller edits the means between two calls
        pl.particle_means.mul_(0.999)
    both(3, obs1)
    assert a._engine.pipeline_split_steps() == 12
    same()
    # the switch: same planner, chains off
    a._engine.set_option("no_step_pipeline", 1)
    both(3, obs1)
    assert a._engine.pipeline_split_steps() == 12
    same()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_end_effector_goal_term_inside_the_update_kernel_bitwise(dtype):
    """CostGoal (the reference's Panda example has it, panda_environment.py:90-96): inside sgpmp_step update_kernel evaluates the
    term for its particle's rows itself, in front of its softmax -- ee_goal_kernel's function, its additions in its order --
    instead of a launch of ee_goal_kernel between the fused launch and the update (option no_ee_fold: rounds 1-4).  One launch
    less per iteration; costs, weights, means, gradient bit-identical, as single steps and as two chains, fp32 (fused launch)
    and fp64 (fused_step_f64_kernel)."""
    from stoch_gpmp_amd.costs.cost_functions import CostGoal
    from stoch_gpmp_amd.costs.fields import EESE3DistanceField
    from oracle.fk import fk_all_links
    ta = F32 if dtype == "f32" else F64
    H = fk_all_links(torch.tensor([[0.3, -0.5, 0.2, -1.9, 0.1, 1.6, 0.4]], dtype=torch.float64))[0, -1].clone()
    sph = torch.as_tensor(SC.panda_spheres(num=5, seed=3)).to(**ta)

    def mk(nppg, S, T):
        pl = hip_panda_planner(SC.PANDA, T, nppg, S, ta, seed=43)
        pl.cost.cost_list.append(CostGoal(7, T, field=EESE3DistanceField(H, w_pos=1., w_rot=0.5, tensor_args=ta), sigma_goal=1e-2,
                                          tensor_args=ta))
        return pl
    for nppg, S, T, calls in ((5, 32, 64, (1, 3, 1)), (129, 128, 32, (4, 1))):
        a, b = mk(nppg, S, T), mk(nppg, S, T)
        b._engine.set_option("no_ee_fold", 1)
        for k in calls:
            ra, rb = a.optimize(opt_iters=k, obstacle_spheres=sph), b.optimize(opt_iters=k, obstacle_spheres=sph)
            for i, (x, y) in enumerate(zip(ra, rb)):
                assert torch.equal(x, y), (k, i)
            assert torch.equal(a._costs, b._costs) and torch.equal(a._weights_buf, b._weights_buf)
            assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a._grad, b._grad)
        # (fp64 steps are one launch + the update too since round 6 -- fused_step_f64_kernel -- so the fold saves a launch there as well)
        assert a._engine.last_step_launches() == 2 and b._engine.last_step_launches() == 3


@pytest.mark.parametrize("nppg,G,S,T", [(2, 2, 8, 16), (3, 2, 24, 48), (64, 2, 64, 64)])
def test_three_dof_point_mass_fused_matches_the_two_launch_path(golden, nppg, G, S, T):
    """The no-FK fused launch for n = 3 (fused_planar_kernel<3>: two segments per chain in its scan, pair broadcasts)
    against sampler + generic sweep, and -- at the size that splits -- as two chains against single calls."""
    c3 = dict(SC.PLANAR, n_dof=3, start=[-9., -9., 0.5, 0., 0., 0.])
    goals = [[9., 6., 1., 0., 0., 0.], [9., -3., -1., 0., 0., 0.]][:G]
    om = planar_map(golden, F32)
    a = hip_planar_planner(c3, T, goals, nppg, S, om, F32, seed=37)
    b = hip_planar_planner(c3, T, goals, nppg, S, om, F32, seed=37)
    b._engine.set_option("no_fused_step", 1)
    for it in range(3):
        a.optimize()
        b.optimize()
        assert a._engine.last_cost_kernel().startswith("fused_planar")
        assert b._engine.last_cost_kernel() == "cost_sweep_kernel<f32, no FK>"
        scale = float(b.state_samples.abs().max())
        assert float((a.state_samples - b.state_samples).abs().max()) <= 4e-7 * scale
        assert rel_err(a._costs, b._costs) < 2e-5
        assert torch.equal(a._costs.argmin(1), b._costs.argmin(1))
        b.particle_means.copy_(a.particle_means)
    if nppg * G * S >= 16384:
        twin = hip_planar_planner(c3, T, goals, nppg, S, om, F32, seed=37, pipeline_steps=False)
        for _ in range(3):
            twin.optimize()
        a.optimize(opt_iters=4)
        for _ in range(4):
            twin.optimize()
        assert a._engine.pipeline_split_steps() == 4
        assert torch.equal(a.particle_means, twin.particle_means) and torch.equal(a._costs, twin._costs)


@pytest.mark.parametrize("ta,kernel", [(F32, "fused_step_kernel"), (F64, "fused_step_f64_kernel")])
def test_prepared_is_weights_follow_every_edit_of_the_means(ta, kernel):
    """A step has its update kernel prepare the next step's importance-sampling weights, and the next
    sgpmp_step skips K5 when the caller vouches (SGPMP_STEP_MEANS_KEPT) that the means are untouched (fused
    fp32 steps: the fused launch zeroes the statistics; other steps: the sampler does).  The
    planner vouches only while the tensor's version counter stands still; a twin that NEVER vouches (it bumps
    the counter before every step, so K5 always runs) must stay bit-identical through plain steps, in-place
    edits of the means, and a `sample_and_eval` + `_update_distribution` detour."""
    T, nppg, S = 32, 6, 16
    sph = torch.as_tensor(SC.panda_spheres()).to(**ta)
    a = hip_panda_planner(SC.PANDA, T, nppg, S, ta, seed=23)
    b = hip_panda_planner(SC.PANDA, T, nppg, S, ta, seed=23)

    def both(fn):
        fn(a)
        b.particle_means.add_(0)                         # version bump: b never claims "means kept"
        fn(b)
        assert a._engine.last_cost_kernel() == b._engine.last_cost_kernel()
        assert a._engine.last_cost_kernel().startswith(kernel.split("<")[0]), a._engine.last_cost_kernel()
        assert torch.equal(a._costs, b._costs) and torch.equal(a.particle_means, b.particle_means)
        sa, sb = a.global_stats(), b.global_stats()
        assert abs(sa[0] / sb[0] - 1) < 1e-12 and abs(sa[1] / sb[1] - 1) < 1e-12

    step = lambda pl: pl.optimize(obstacle_spheres=sph)          # noqa: E731
    for _ in range(3):
        both(step)
    for pl in (a, b):
        pl.particle_means.mul_(1.0005)                   # an outside edit: the prepared weights are stale now
    both(step)
    both(step)
    for pl in (a, b):                                    # K4 on its own moves the means without a version bump
        _, _, _, _, costs = pl.sample_and_eval(obstacle_spheres=sph)
        pl._update_distribution(costs, pl.state_samples)
    both(step)
    a.temperature = b.temperature = 2.0                  # the weights carry the temperature
    both(step)
    both(step)


# --------------------------------------------------------------------------- live observations / edits
def test_moving_obstacles_are_seen_on_every_call():
    """optimize(obstacle_spheres=...) must use THIS call's spheres (reference fields.py:63-76 reads the
    observation on every eval): fresh tensors per call -- whose id() CPython recycles --, CPU tensors,
    and one device tensor edited in place, all against a planner fed explicit per-call copies."""
    import gc
    T, nppg, S = 16, 4, 8
    base = torch.as_tensor(SC.panda_spheres()).float()
    a = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=3)            # fresh CPU tensors (freed at once)
    b = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=3)            # one device tensor, edited in place
    c = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=3)            # fresh device tensors
    frozen = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=3)       # never told that the spheres moved
    live = base.to(DEV).clone()
    first = base.to(DEV).clone()
    differs = 0
    for i in range(6):
        shift = torch.tensor([0.03 * i, -0.02 * i, 0.01 * i, 0.0])
        a.optimize(obstacle_spheres=torch.tensor((base + shift).tolist()))
        gc.collect()
        live.copy_((base + shift).to(DEV))
        b.optimize(obstacle_spheres=live)
        c.optimize(obstacle_spheres=(base + shift).to(DEV))
        frozen.optimize(obstacle_spheres=first)
        assert torch.equal(a._costs, b._costs) and torch.equal(a._costs, c._costs), f"call {i}"
        assert torch.equal(a.particle_means, b.particle_means) and torch.equal(a.particle_means, c.particle_means)
        differs += int(not torch.equal(a._costs, frozen._costs))
    assert differs >= 4          # the moved spheres do change the costs (the check above is not vacuous)


def test_update_target_reaches_a_live_planner():
    """EESE3DistanceField.update_target (fields.py:140-141): the reference reads target_H on every eval,
    so a planner that already compiled the cost must follow the new target on its next step."""
    from stoch_gpmp_amd.costs.cost_functions import CostGoal
    from stoch_gpmp_amd.costs.fields import EESE3DistanceField
    from oracle.fk import fk_all_links
    T, nppg, S, n = 16, 3, 8, 7
    H1 = fk_all_links(torch.tensor([[0.3, -0.5, 0.2, -1.9, 0.1, 1.6, 0.4]], dtype=torch.float64))[0, -1].clone()
    H2 = fk_all_links(torch.tensor([[-0.6, 0.1, 0.5, -1.2, -0.3, 2.2, 0.0]], dtype=torch.float64))[0, -1].clone()
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)

    def build(H):
        pl = hip_panda_planner(SC.PANDA, T, nppg, S, F32, seed=6)
        field = EESE3DistanceField(H, w_pos=1., w_rot=0.5, tensor_args=F32)
        pl.cost.cost_list.append(CostGoal(n, T, field=field, sigma_goal=1e-4, tensor_args=F32))
        pl.step_size = 0.0                                   # keep the means: both planners see the same draws
        return pl, field
    moved, field = build(H1)
    fresh, _ = build(H2)
    moved.optimize(obstacle_spheres=sph)
    fresh.optimize(obstacle_spheres=sph)
    assert not torch.equal(moved._costs, fresh._costs)       # different targets, different costs
    field.update_target(H2)
    moved.optimize(obstacle_spheres=sph)
    fresh.optimize(obstacle_spheres=sph)
    assert torch.equal(moved.state_samples, fresh.state_samples)
    assert torch.equal(moved._costs, fresh._costs)


def test_arbitrary_fk_callable_gives_the_native_results():
    """cost_functions.py:39,51-52: FK is ANY callable q[B*T,n] -> [B*T,L,4,4].  A plain Python function
    (here: a closure around the chain's FK, so the expected numbers are known) takes the frames-in
    path; costs must match the in-sweep FK path, in a composite and through a planner."""
    from stoch_gpmp_amd.costs.cost_functions import CostComposite
    from stoch_gpmp_amd.robots.panda import DifferentiableFrankaPanda
    c, n = SC.PANDA, 7
    T, nppg, S = 12, 3, 6
    for ta, tol in ((F64, 1e-10), (F32, 2e-4)):
        sph = torch.as_tensor(SC.panda_spheres()).to(**ta)
        native = hip_panda_cost(c, T, nppg, S, ta, field_type='sdf')
        robot = DifferentiableFrankaPanda(gripper=False, device=DEV)
        calls = []

        def my_fk(q):
            calls.append(q.shape)
            return robot.compute_forward_kinematics_all_links(q)
        foreign = CostComposite(n, T, hip_panda_cost(c, T, nppg, S, ta, field_type='sdf').cost_list,
                                FK=my_fk, tensor_args=ta)
        assert foreign.foreign_fk and not native.foreign_fk
        g = torch.Generator().manual_seed(2)
        trajs = torch.cat([torch.rand(nppg, S, T, n, generator=g) * 2 - 1,
                           torch.randn(nppg, S, T, n, generator=g) * 0.1], dim=-1).to(**ta)
        # drop the GP / goal terms for the comparison in fp32 (they would swamp the fields)
        native.cost_list = native.cost_list[2:]
        foreign.cost_list = foreign.cost_list[2:]
        a = native.eval(trajs, obstacle_spheres=sph)
        b = foreign.eval(trajs, obstacle_spheres=sph)
        assert calls and calls[-1] == (nppg * S * T, n)
        assert rel_err(b, a) < tol
    # through the planner: the composite is then user code to StochGPMP (slow path), and it runs
    pl = hip_panda_planner(c, T, nppg, S, F32, seed=1)
    pl_f = hip_panda_planner(c, T, nppg, S, F32, seed=1)
    robot = DifferentiableFrankaPanda(gripper=False, device=DEV)
    pl_f.cost = CostComposite(n, T, pl_f.cost.cost_list, FK=lambda q: robot.compute_forward_kinematics_all_links(q),
                              tensor_args=F32)
    pl_f._draw = 0                                       # replay the same draws as `pl` (test-only)
    pl_f.reset()
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    o1 = pl.optimize(obstacle_spheres=sph)
    o2 = pl_f.optimize(obstacle_spheres=sph)
    assert torch.equal(pl.state_samples, pl_f.state_samples)
    assert rel_err(o2[4], o1[4]) < 1e-4


# --------------------------------------------------------------------------- full-size properties
def _full_panda(P, S, T, ta, **kw):
    return hip_panda_planner(SC.PANDA, T, P, S, ta, seed=0, **kw)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_config2_shards_equal_unsharded_bitwise(golden, world):
    """BASELINE config 2 (planar, 4 goals x 64 particles x 64 samples x 128 waypoints, fp32) through the lane-per-sample
    launch: shards addressed by global particle index -- even (2, 8) and ragged with boundaries inside a goal (3) --
    reproduce the unsharded run bit for bit, single calls and a multi-iteration call; the kernel choice is a function
    of (S, T, n) alone, so a shard launches what the whole problem launches."""
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
    om = planar_map(golden, F32)
    mk = lambda **kw: hip_planar_planner(SC.PLANAR, 128, goals, 64, 64, om, F32, seed=3, **kw)   # noqa: E731
    full = mk()
    ref0 = full.particle_means.clone()
    for _ in range(2):
        full.optimize()
    full.optimize(opt_iters=3)
    assert full._engine.last_cost_kernel() == "fused_planar_seg_kernel"
    shards = []
    for r in range(world):
        h = mk(rank=r, world_size=world)
        assert torch.equal(h.particle_means, ref0[h.p0:h.p1])
        for _ in range(2):
            h.optimize()
        h.optimize(opt_iters=3)
        assert h._engine.last_cost_kernel() == "fused_planar_seg_kernel"
        shards.append(h)
    assert sum(h.num_particles_local for h in shards) == 256
    assert torch.equal(torch.cat([h.particle_means for h in shards]), full.particle_means)
    assert torch.equal(torch.cat([h._costs for h in shards]), full._costs)
    assert torch.equal(torch.cat([h.state_samples for h in shards]), full.state_samples)


def test_full_size_sharded_equals_unsharded_bitwise():
    """BASELINE config 3 shape (Panda, 1024 x 128 x 64, fp32): two half shards addressed by global
    particle index reproduce the single-GPU run bit for bit (means, costs) -- the property that
    makes the 8-GPU run equal to the 1-GPU run."""
    P, S, T = 1024, 128, 64
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    full = _full_panda(P, S, T, F32)
    ref_means0 = full.particle_means.clone()
    for _ in range(2):
        full.optimize(obstacle_spheres=sph)
    halves = []
    for r in range(2):
        h = _full_panda(P, S, T, F32, rank=r, world_size=2)
        assert torch.equal(h.particle_means, ref_means0[h.p0:h.p1])
        for _ in range(2):
            h.optimize(obstacle_spheres=sph)
        halves.append(h)
    assert torch.equal(torch.cat([h.particle_means for h in halves]), full.particle_means)
    assert torch.equal(torch.cat([h._costs for h in halves]), full._costs)
    # determinism: a second identical run is bit-identical
    again = _full_panda(P, S, T, F32)
    for _ in range(2):
        again.optimize(obstacle_spheres=sph)
    assert torch.equal(again.particle_means, full.particle_means)
    # sanity of the update at full size: weights are a distribution, means moved, all finite
    w = full._weights.reshape(P, S)
    assert torch.allclose(w.sum(1), torch.ones(P, device=DEV), atol=1e-5)
    assert torch.isfinite(full.particle_means).all() and torch.isfinite(full._costs).all()
    assert not torch.equal(full.particle_means, ref_means0)


def test_config4_eight_shards_equal_unsharded_bitwise():
    """BASELINE config 4 on one GPU: Panda 8192 x 128 x 64 fp32 (3.8 GB of samples) run unsharded, then
    as the eight `rank = r, world_size = 8` shards of 1024 particles run one after the other
    (particle_offset up to 7168): means and costs must be bit-identical -- the 8-GPU run IS the 1-GPU run."""
    P, S, T = 8192, 128, 64
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    full = _full_panda(P, S, T, F32)
    means0 = full.particle_means.clone()
    for _ in range(2):
        full.optimize(obstacle_spheres=sph)
    full_means, full_costs = full.particle_means.clone(), full._costs.clone()
    assert full._engine.last_cost_kernel() == "fused_step_kernel"
    stats_full = full._stats[full._stats_slot ^ 1].sum(0).cpu()
    del full
    torch.cuda.empty_cache()
    stats_sum = torch.zeros(4, dtype=torch.float64)
    for r in range(8):
        h = _full_panda(P, S, T, F32, rank=r, world_size=8)
        assert (h.p0, h.p1) == (1024 * r, 1024 * (r + 1))
        assert torch.equal(h.particle_means, means0[h.p0:h.p1])
        for _ in range(2):
            h.optimize(obstacle_spheres=sph)
        assert torch.equal(h.particle_means, full_means[h.p0:h.p1]), f"shard {r}: means differ"
        assert torch.equal(h._costs, full_costs[h.p0:h.p1]), f"shard {r}: costs differ"
        stats_sum += h._stats[h._stats_slot ^ 1].sum(0).cpu()
        del h
    # what the RCCL all-reduce would sum: shard statistics add up to the unsharded run's
    assert float(stats_sum[2]) == P and abs(float(stats_sum[0] / stats_full[0]) - 1) < 1e-12
    assert abs(float(stats_sum[1] / stats_full[1]) - 1) < 1e-12


def test_config5_eight_shards_equal_unsharded_bitwise():
    """BASELINE config 5 -- the WHOLE problem in one context (round-5 verdict, missing #4): Panda 4 goals x 1024 particles x 256
    samples x 128 waypoints, fp64 prior + fp32 cost path: 1 879 048 192 sample elements (a hair under 2^31), 7.5 GB of samples,
    byte offsets past 2^32.  One single-iteration call (storing), then one optimize(opt_iters=2) (a store-free step whose rows
    update_kernel regenerates + the storing last step, as two particle-half chains) unsharded; then the eight `rank = r,
    world_size = 8` shards of 512 particles one after the other (shard boundaries inside goals: p // nppg through the global
    offset): means, costs, weights and every shard's slice of the 7.5 GB sample tensor bit-identical, statistics add up."""
    c, n = SC.PANDA, 7
    nppg, S, T = 1024, 256, 128
    P = 4 * nppg
    goals = [c["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n,
             [0.9, -0.2, 0.4, -1.1, -0.3, 1.9, 0.8] + [0.] * n, [-0.8, 0.1, 0.6, -2.4, 0.4, 2.6, -0.2] + [0.] * n]
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    mk = lambda **kw: hip_panda_planner(c, T, nppg, S, F32, seed=5, goals=goals, **kw)      # noqa: E731

    def run(pl):
        pl.optimize(obstacle_spheres=sph)
        pl.optimize(opt_iters=2, obstacle_spheres=sph)
        assert pl._engine.last_cost_kernel() == "fused_step_kernel" and pl._engine.store_free_steps() == 1
    full = mk()
    assert full.num_particles == P and full.state_samples.numel() == P * S * T * 2 * n == 1879048192
    means0 = full.particle_means.clone()
    run(full)
    assert full._engine.pipeline_split_steps() == 2
    stats_full = full._stats[full._stats_slot ^ 1].sum(0).cpu()
    assert bool(torch.isfinite(full._costs).all()) and bool(torch.isfinite(full.particle_means).all())
    # the last rows of the tensor really were written by this call (element offsets near 2^31, byte offsets near 2^33)
    assert bool((full.state_samples[-1, -1] != 0).any()) and bool(torch.isfinite(full.state_samples[-1]).all())
    stats_sum = torch.zeros(4, dtype=torch.float64)
    for r in range(8):
        h = mk(rank=r, world_size=8)
        assert (h.p0, h.p1) == (512 * r, 512 * (r + 1))
        assert torch.equal(h.particle_means, means0[h.p0:h.p1])
        run(h)
        assert torch.equal(h.particle_means, full.particle_means[h.p0:h.p1]), f"shard {r}: means differ"
        assert torch.equal(h._costs, full._costs[h.p0:h.p1]), f"shard {r}: costs differ"
        assert torch.equal(h._weights_buf, full._weights_buf[h.p0:h.p1]) and torch.equal(h._grad, full._grad[h.p0:h.p1]), r
        assert torch.equal(h.state_samples, full.state_samples[h.p0:h.p1]), f"shard {r}: samples differ"
        stats_sum += h._stats[h._stats_slot ^ 1].sum(0).cpu()
        del h
        torch.cuda.empty_cache()
    assert float(stats_sum[2]) == P and abs(float(stats_sum[0] / stats_full[0]) - 1) < 1e-12
    assert abs(float(stats_sum[1] / stats_full[1]) - 1) < 1e-12


@pytest.mark.parametrize("field_type", ["rbf", "sdf"])
def test_config5_share_fast_sweep_equals_generic_sweep(field_type):
    """BASELINE config 5's per-GPU share at FULL size (Panda, 512 x 256 x 128, fp32, 4 goals -> this
    shard holds half of goal 0's particles): cost_sweep_dual_pf_multi_kernel against the single-trajectory
    generic-FK sweep on all 131 072 trajectories, with IS weights and the multi-goal prior."""
    c, n = SC.PANDA, 7
    S, T = 256, 128
    goals = [c["goal_q"] + [0.] * n, [-0.4, 0.5, -0.3, -2.0, 0.2, 1.5, -0.5] + [0.] * n,
             [0.9, -0.2, 0.4, -1.1, -0.3, 1.9, 0.8] + [0.] * n, [-0.8, 0.1, 0.6, -2.4, 0.4, 2.6, -0.2] + [0.] * n]
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    pl = hip_panda_planner(c, T, 1024, S, F32, field_type=field_type, seed=9, goals=goals, rank=3, world_size=8)
    assert pl.num_particles == 4096 and pl.num_particles_local == 512 and pl.p0 == 1536
    pl.optimize(obstacle_spheres=sph)
    assert pl._engine.last_cost_kernel() == "fused_step_kernel"
    step_costs = pl._costs.clone()                           # what the fused launch computed for its own samples
    samples = pl.state_samples
    w = pl._engine.is_weights(pl._means_prev, pl.temperature)     # weights of the PRE-update means, as in the step
    sphc = sph.reshape(-1, 4).contiguous()
    kw = dict(batch_offset=pl.p0 * S, spheres=sphc, is_weights=w, rows_per_particle=S)
    chunked = pl._engine.cost_eval(samples, **kw).clone()
    assert pl._engine.last_cost_kernel() == "cost_sweep_chunked_kernel"
    pl._engine.set_option("no_chunked_sweep", 1)
    fast = pl._engine.cost_eval(samples, **kw).clone()
    assert pl._engine.last_cost_kernel() == "cost_sweep_dual_pf_multi_kernel"
    pl._engine.set_option("no_dual_sweep", 1)
    pl._engine.set_option("force_generic_fk", 1)
    slow = pl._engine.cost_eval(samples, **kw)
    assert pl._engine.last_cost_kernel() == "cost_sweep_kernel<f32, generic FK>"
    rel = ((fast.double() - slow.double()).abs() / slow.double().abs().clamp_min(1.0)).max()
    assert fast.shape == (512 * S,) and bool(torch.isfinite(fast).all()) and float(rel) < 2e-5, float(rel)
    # ... and the fused launch (sampler + sweep in one kernel) and the chunked sweep agree with both on
    # every trajectory (the chunked sweep IS the fused launch's phase C: bit-identical costs)
    rel = ((step_costs.reshape(-1).double() - slow.double()).abs() / slow.double().abs().clamp_min(1.0)).max()
    assert float(rel) < 2e-5, float(rel)
    assert torch.equal(chunked, step_costs.reshape(-1))


def test_full_size_planar_fused_step_equals_separate_calls(golden):
    """BASELINE config 2 shape (planar, 256 x 64 x 128, fp32): sgpmp_step -- here the planar fused launch
    (fused_planar.inc: noise, recurrence, GP / goal / grid / IS terms in one kernel) -- against K5, K2, K3, K4
    called one by one through the reference-shaped methods (sample_and_eval + _update_distribution): same
    noise keys, so the samples agree to the last bit or two, the costs to fp32 rounding and the arg-min
    exactly; with the fused launch switched off the two routes are bit-identical."""
    goals = [[9., 6., 0., 0.], [9., -3., 0., 0.], [-3., 9., 0., 0.], [6., 9., 0., 0.]]
    om = planar_map(golden, F32)
    a = hip_planar_planner(SC.PLANAR, 128, goals, 64, 64, om, F32, seed=4)
    b = hip_planar_planner(SC.PLANAR, 128, goals, 64, 64, om, F32, seed=4)
    c = hip_planar_planner(SC.PLANAR, 128, goals, 64, 64, om, F32, seed=4)
    c._engine.set_option("no_fused_step", 1)
    assert torch.equal(a.particle_means, b.particle_means)
    for _ in range(3):
        a.optimize()
        c.optimize()
        _, _, _, _, costs = b.sample_and_eval()
        b._update_distribution(costs, b.state_samples)
        assert a._engine.last_cost_kernel().startswith("fused_planar")
        assert torch.equal(c.state_samples, b.state_samples) and torch.equal(c._costs, b._costs)
        assert torch.equal(c.particle_means, b.particle_means)
        scale = float(b.state_samples.abs().max())
        assert float((a.state_samples - b.state_samples).abs().max()) <= 4e-7 * scale
        assert rel_err(a._costs, b._costs) < 2e-5
        assert torch.equal(a._costs.argmin(1), b._costs.argmin(1))
        assert float((a.particle_means - b.particle_means).abs().max()) <= 1e-6 * scale
        b.particle_means.copy_(a.particle_means)
        c.particle_means.copy_(a.particle_means)
    # goal-directedness: every particle's last waypoint stays near its own goal (sigma_goal 1e-3)
    end = a.particle_means[:, -1, :2].reshape(4, 64, 2).cpu()
    assert float((end - torch.tensor(goals)[:, None, :2]).abs().max()) < 0.05


@pytest.mark.parametrize("field_type", ["rbf", "sdf"])
def test_full_size_fast_sweep_equals_generic_sweep(field_type):
    """BASELINE config 3 at FULL size (Panda, 1024 x 128 x 64, fp32): the two-trajectory LDS-prefetch
    sweep against the single-trajectory generic-FK sweep (SGPMP_NO_DUAL_SWEEP + SGPMP_FORCE_GENERIC_FK) on
    the very same 131 072 samples and importance-sampling weights -- every cost, not a sample of them."""
    c = SC.PANDA
    P, S, T = 1024, 128, 64
    sph = torch.as_tensor(SC.panda_spheres()).to(**F32)
    pl = hip_panda_planner(c, T, P, S, F32, field_type=field_type, seed=9)
    pl.optimize(obstacle_spheres=sph)
    assert pl._engine.last_cost_kernel() == "fused_step_kernel"
    step_costs = pl._costs.clone()
    samples = pl.state_samples
    w = pl._engine.is_weights(pl._means_prev, pl.temperature)     # weights of the PRE-update means, as in the step
    sphc = sph.reshape(-1, 4).contiguous()
    chunked = pl._engine.cost_eval(samples, spheres=sphc, is_weights=w, rows_per_particle=S).clone()
    assert pl._engine.last_cost_kernel() == "cost_sweep_chunked_kernel"
    assert torch.equal(chunked, step_costs.reshape(-1))      # the chunked sweep IS the fused launch's phase C
    pl._engine.set_option("no_chunked_sweep", 1)
    fast = pl._engine.cost_eval(samples, spheres=sphc, is_weights=w, rows_per_particle=S).clone()
    assert pl._engine.last_cost_kernel() == "cost_sweep_dual_pf_kernel"
    pl._engine.set_option("no_dual_sweep", 1)
    pl._engine.set_option("force_generic_fk", 1)
    slow = pl._engine.cost_eval(samples, spheres=sphc, is_weights=w, rows_per_particle=S)
    assert pl._engine.last_cost_kernel() == "cost_sweep_kernel<f32, generic FK>"
    assert fast.shape == (P * S,) and bool(torch.isfinite(fast).all())
    rel = ((fast.double() - slow.double()).abs() / slow.double().abs().clamp_min(1.0)).max()
    assert float(rel) < 2e-5, float(rel)
    # the fused launch of the planner's own iteration: same samples, same costs
    rel = ((step_costs.reshape(-1).double() - slow.double()).abs() / slow.double().abs().clamp_min(1.0)).max()
    assert float(rel) < 2e-5, float(rel)
    # a second evaluation reproduces the fast kernel's costs bit for bit
    pl._engine.set_option("no_dual_sweep", 0)
    pl._engine.set_option("force_generic_fk", 0)
    again = pl._engine.cost_eval(samples, spheres=sphc, is_weights=w, rows_per_particle=S)
    assert torch.equal(again, fast)


# --------------------------------------------------------------------------- example scripts
def test_example_scripts_run_and_make_progress():
    """examples/*.py are this package's versions of the reference's two example scripts; a short run
    must lower the mean cost (planar) and bring the end effector towards the target (Panda)."""
    import importlib.util
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    planar = load("planar_environment")
    pl0, c0 = planar.main(opt_iters=0, seed=3, num_samples=32, verbose=False)
    pl1, c1 = planar.main(opt_iters=60, seed=3, num_samples=32, verbose=False)
    assert c1.shape == c0.shape == (15, 32) and float(c1.mean()) < 0.5 * float(c0.mean())
    panda = load("panda_environment")
    p0, k0 = panda.main(opt_iters=0, seed=3, verbose=False)
    p1, k1 = panda.main(opt_iters=80, seed=3, verbose=False)
    assert k1.shape == (5, 32) and float(k1.min(1)[0].mean()) < float(k0.min(1)[0].mean())


# --------------------------------------------------------------------------- GPMP (Gauss-Newton)
def _hip_gpmp(g, tag, ta, delta, trust, method="cholesky", **kw):
    from stoch_gpmp_amd.planner import GPMP
    c = SC.PANDA
    T, nppg = [int(v) for v in g["dims"]]
    n = 7
    goals = torch.from_numpy(g["goals"]).to(**ta)
    G = goals.shape[0]
    cost = hip_panda_cost(c, T, nppg, 1, ta, goals=goals)
    init = torch.from_numpy(g[f"{tag}/means0"]).to(**ta).reshape(G, nppg, T, 2 * n)
    return GPMP(num_particles_per_goal=nppg, traj_len=T, opt_iters=1, dt=c["dt"], n_dof=n, step_size=0.5,
                temperature=1., start_state=torch.tensor(c["start_q"] + [0.] * n, **ta), multi_goal_states=goals,
                initial_particle_means=init, cost=cost,
                sigma_start_init=c["sigma_start_init"], sigma_start_sample=c["sigma_start_sample"],
                sigma_goal_init=c["sigma_goal_init"], sigma_goal_sample=c["sigma_goal_sample"],
                sigma_gp_init=c["sigma_gp_init"], sigma_gp_sample=c["sigma_gp_sample"], seed=0,
                solver_params=dict(delta=delta, trust_region=trust, method=method), tensor_args=ta, **kw)


@pytest.mark.parametrize("tag,delta,trust", [("lm", 5.0, False), ("tr", 1e-2, True)])
def test_gpmp_matches_reference_run_and_oracle(golden, tag, delta, trust):
    """The Gauss-Newton planner: block-tridiagonal HIP solve against (a) the reference's own run for the
    Levenberg mode with its correct 'inverse' solver (g7 fixture), (b) the dense oracle (proper solve)
    step by step in both damping modes -- d_theta, costs and means."""
    from oracle import gpmp_equiv as GP
    from oracle.fk import fk_all_links
    g = golden("g7_gpmp.npz")
    T, nppg = [int(v) for v in g["dims"]]
    goals, sph = torch.from_numpy(g["goals"]), torch.from_numpy(g["spheres"])
    pl = _hip_gpmp(g, tag, F64, delta, trust)
    ora = GP.OracleGPMP(torch.from_numpy(g[f"{tag}/means0"]),
                        GP.panda_systems_fn(SC.PANDA, T, nppg, goals, fk_all_links), 0.5, delta, trust, "inverse")
    for it in range(3):
        d_o, c_o = ora.step(obstacle_spheres=sph)
        vel, pos, costs = pl.optimize(obstacle_spheres=sph.to(**F64))
        assert rel_err(pl._d_theta, d_o) < 1e-7
        assert rel_err(costs, c_o) < 1e-9
        assert rel_err(pl.particle_means, ora.particle_means) < 1e-8
        assert torch.equal(pos, pl.particle_means[..., :7]) and torch.equal(vel, pl.particle_means[..., 7:])
        if tag == "lm":                                   # the reference itself (its 'inverse' branch)
            assert rel_err(pl.particle_means, torch.from_numpy(g[f"lm/means{it + 1}"])) < 1e-8
            assert rel_err(costs, torch.from_numpy(g[f"lm/costs{it + 1}"])) < 1e-9


@pytest.mark.parametrize("ta_name,tol", [("f64", 1e-9), ("f32", 2e-4)])
def test_gpmp_register_kernel_equals_the_lds_cholesky_kernel(golden, ta_name, tol):
    """Round 4's solve (block-Thomas recursion, the d x d matrices in registers, Gauss-Jordan by readlane) against round 3's
    (block Cholesky through LDS tiles, `gpmp_cholesky`): same step, same costs, both damping modes."""
    g = golden("g7_gpmp.npz")
    sph = torch.from_numpy(g["spheres"])
    ta = F64 if ta_name == "f64" else F32
    for tag, delta, trust in (("lm", 5.0, False), ("tr", 1e-2, True)):
        a, b = _hip_gpmp(g, tag, ta, delta, trust), _hip_gpmp(g, tag, ta, delta, trust)
        b._engine.set_option("gpmp_cholesky", 1)
        for it in range(3):
            b.particle_means.copy_(a.particle_means)
            _, _, ca = a.optimize(obstacle_spheres=sph.to(**ta))
            _, _, cb = b.optimize(obstacle_spheres=sph.to(**ta))
            assert rel_err(a._d_theta, b._d_theta) < tol, (tag, it)
            assert rel_err(ca, cb) < (1e-12 if ta_name == "f64" else 1e-5)


def test_gpmp_with_end_effector_goal_matches_oracle(golden):
    """GPMP with CostGoal (EESE3DistanceField) in the cost list: its one row on the last waypoint enters the
    block-tridiagonal normal equations like a collision row; d_theta, costs and means against the dense oracle
    whose Jacobian is autograd through FK."""
    from oracle import gpmp_equiv as GP
    from oracle import ref_equiv as R
    from oracle.fk import fk_all_links
    from stoch_gpmp_amd.costs.cost_functions import CostGoal
    from stoch_gpmp_amd.costs.fields import EESE3DistanceField
    g = golden("g7_gpmp.npz")
    T, nppg = [int(v) for v in g["dims"]]
    goals, sph = torch.from_numpy(g["goals"]), torch.from_numpy(g["spheres"])
    H_t = fk_all_links(torch.tensor([[0.3, -0.5, 0.2, -1.9, 0.1, 1.6, 0.4]], dtype=torch.float64))[0, -1].clone()
    sigma_ee = 0.05
    pl = _hip_gpmp(g, "lm", F64, 5.0, False)
    pl.cost.cost_list.append(CostGoal(7, T, field=EESE3DistanceField(H_t.to(**F64), w_pos=1.0, w_rot=0.5, tensor_args=F64),
                                      sigma_goal=sigma_ee, tensor_args=F64))
    pl.cost.touch()
    base = GP.panda_systems_fn(SC.PANDA, T, nppg, goals, fk_all_links)

    def systems(means, **obs):
        return base(means, **obs) + [R.goal_ee_linear_system(means, 7, fk_all_links,
                                                             lambda fr: R.field_ee_se3(fr, H_t, 1.0, 0.5), sigma_ee)]
    ora = GP.OracleGPMP(torch.from_numpy(g["lm/means0"]), systems, 0.5, 5.0, False, "inverse")
    for it in range(3):
        d_o, c_o = ora.step(obstacle_spheres=sph)
        _, _, costs = pl.optimize(obstacle_spheres=sph.to(**F64))
        assert rel_err(pl._d_theta, d_o) < 1e-7
        assert rel_err(costs, c_o) < 1e-9
        assert rel_err(pl.particle_means, ora.particle_means) < 1e-8


@pytest.mark.parametrize("clamp", [False, True])
def test_gpmp_with_sdf_collision_matches_oracle(golden, clamp):
    """GPMP with the signed-distance sphere field in the cost list (fields.py:79-83): its Jacobian -- the gradient of
    the arg-max (link, sphere) pair, what the reference gets from autograd (field_factor.py:35) -- enters the
    block-tridiagonal normal equations like the rbf rows; d_theta, costs and means against the dense oracle."""
    from oracle import gpmp_equiv as GP
    from oracle import ref_equiv as R
    from oracle.fk import fk_all_links
    g = golden("g7_gpmp.npz")
    T, nppg = [int(v) for v in g["dims"]]
    goals, sph = torch.from_numpy(g["goals"]), torch.from_numpy(g["spheres"])
    c = SC.PANDA
    ta = F64
    cost = hip_panda_cost(c, T, nppg, 1, ta, goals=goals.to(**ta), field_type="sdf", clamp_sdf=clamp)
    from stoch_gpmp_amd.planner import GPMP
    init = torch.from_numpy(g["lm/means0"]).to(**ta).reshape(goals.shape[0], nppg, T, 14)
    pl = GPMP(num_particles_per_goal=nppg, traj_len=T, opt_iters=1, dt=c["dt"], n_dof=7, step_size=0.5, temperature=1.,
              start_state=torch.tensor(c["start_q"] + [0.] * 7, **ta), multi_goal_states=goals.to(**ta),
              initial_particle_means=init, cost=cost,
              sigma_start_init=c["sigma_start_init"], sigma_start_sample=c["sigma_start_sample"],
              sigma_goal_init=c["sigma_goal_init"], sigma_goal_sample=c["sigma_goal_sample"],
              sigma_gp_init=c["sigma_gp_init"], sigma_gp_sample=c["sigma_gp_sample"], seed=0,
              solver_params=dict(delta=5.0, trust_region=False, method="cholesky"), tensor_args=ta)
    n = 7

    def systems(means, obstacle_spheres=None):
        start = torch.tensor(c["start_q"] + [0.] * n, dtype=means.dtype)
        return [GP.linear_system_gp(means, start, n, c["dt"], c["cost_sigma_start"], c["cost_sigma_gp"]),
                GP.linear_system_goal_prior(means, goals, nppg, n, c["sigma_goal_prior"]),
                R.collision_linear_system(means, n, fk_all_links, lambda fr: R.field_self(fr, margin=c["self_margin"]),
                                          c["sigma_self"]),
                R.collision_linear_system(means, n, fk_all_links,
                                          lambda fr: R.field_spheres(fr, obstacle_spheres, field_type="sdf",
                                                                     clamp_sdf=clamp), c["sigma_coll"])]
    ora = GP.OracleGPMP(torch.from_numpy(g["lm/means0"]), systems, 0.5, 5.0, False, "inverse")
    for it in range(3):
        d_o, c_o = ora.step(obstacle_spheres=sph)
        _, _, costs = pl.optimize(obstacle_spheres=sph.to(**ta))
        assert rel_err(pl._d_theta, d_o) < 1e-7
        assert rel_err(costs, c_o) < 1e-9
        assert rel_err(pl.particle_means, ora.particle_means) < 1e-8


def test_gpmp_fp32_and_errors(golden):
    from oracle import gpmp_equiv as GP
    from oracle.fk import fk_all_links
    from stoch_gpmp_amd.planner import GPMP
    g = golden("g7_gpmp.npz")
    T, nppg = [int(v) for v in g["dims"]]
    goals, sph = torch.from_numpy(g["goals"]), torch.from_numpy(g["spheres"])
    pl = _hip_gpmp(g, "lm", F32, 5.0, False, method="inverse")
    ora = GP.OracleGPMP(torch.from_numpy(g["lm/means0"]),
                        GP.panda_systems_fn(SC.PANDA, T, nppg, goals, fk_all_links), 0.5, 5.0, False, "inverse")
    d_o, c_o = ora.step(obstacle_spheres=sph)
    _, _, costs = pl.optimize(obstacle_spheres=sph.to(**F32))
    assert rel_err(costs, c_o) < 1e-4 and rel_err(pl.particle_means, ora.particle_means) < 1e-4
    with pytest.raises(NotImplementedError):
        _hip_gpmp(g, "lm", F64, 1.0, False, method="lu")
    with pytest.raises(TypeError):
        GPMP(num_particles_per_goal=1, traj_len=4, opt_iters=1, dt=0.1, n_dof=2, cost=None, tensor_args=F64)
