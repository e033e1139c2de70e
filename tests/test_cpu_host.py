"""CPU-side checks: the C-ABI library loads and exports every symbol include/sgpmp.h declares, the
ctypes structs match the C layout, host logic (sharding, descriptors, rasteriser) behaves, and the
product fails loudly without a GPU instead of falling back.  No kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sgpmp.h")
CPU = {"device": torch.device("cpu"), "dtype": torch.float64}


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sgpmp_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    from stoch_gpmp_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"libsgpmp.so does not export {name}"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in stoch_gpmp_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.sgpmp_abi_version() == _lib.ABI_VERSION == 6


def test_ctypes_structs_match_the_c_layout(tmp_path):
    from stoch_gpmp_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "sgpmp.h"
int main(void) {
  printf("%zu %zu %zu\n", sizeof(sgpmp_dims), sizeof(sgpmp_cost_desc), sizeof(sgpmp_joint));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(sgpmp_cost_desc, sigma), offsetof(sgpmp_cost_desc, data),
         offsetof(sgpmp_cost_desc, dim0), offsetof(sgpmp_cost_desc, p0),
         offsetof(sgpmp_cost_desc, num_interpolate), offsetof(sgpmp_cost_desc, alpha));
  printf("%zu %zu\n", offsetof(sgpmp_joint, xyz), offsetof(sgpmp_joint, revolute));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    vals = [int(v) for v in out]
    assert vals[:3] == [ctypes.sizeof(_lib.Dims), ctypes.sizeof(_lib.CostDesc), ctypes.sizeof(_lib.Joint)]
    cd = _lib.CostDesc
    assert vals[3:9] == [cd.sigma.offset, cd.data.offset, cd.dim0.offset, cd.p0.offset,
                         cd.num_interpolate.offset, cd.alpha.offset]
    assert vals[9:] == [_lib.Joint.xyz.offset, _lib.Joint.revolute.offset]


def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback: constructing anything that would compute raises on a CPU device."""
    from stoch_gpmp_amd.engine import Engine
    from stoch_gpmp_amd.planner import StochGPMP
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(2, 8, 1, 1, tensor_args=CPU)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        StochGPMP(2, 4, 8, 1, dt=0.1, n_dof=2, start_state=torch.zeros(4),
                  multi_goal_states=torch.ones(1, 4), sigma_start_init=1., sigma_start_sample=1.,
                  sigma_goal_init=1., sigma_goal_sample=1., sigma_gp_init=1., sigma_gp_sample=1.,
                  tensor_args=CPU)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            Engine(2, 8, 1, 1, tensor_args={"device": torch.device("cuda:0"), "dtype": torch.float32})


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "stoch_gpmp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "/root/reference" not in text, f


def test_shard_range_partitions_particles():
    from stoch_gpmp_amd.dist import shard_range
    for P in (1, 7, 8, 1024, 4096, 8192, 1000):
        for W in (1, 2, 3, 4, 8):
            rs = [shard_range(P, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == P
            assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(8192, 3, 8) == (3072, 4096)


def test_cost_descriptors_follow_the_reference_constructors():
    from stoch_gpmp_amd import _lib as L
    from stoch_gpmp_amd.costs.cost_functions import (CostCollision, CostComposite, CostGoal, CostGP,
                                                     CostGPTrajectory, CostGoalPrior)
    from stoch_gpmp_amd.costs.fields import (EESE3DistanceField, LinkDistanceField,
                                             LinkSelfDistanceField)
    from stoch_gpmp_amd.envs.obst_map import ObstacleMap
    from stoch_gpmp_amd.robots.panda import DifferentiableFrankaPanda
    n, T = 7, 16
    start = torch.arange(14, dtype=torch.float64)
    goals = torch.ones(3, 14, dtype=torch.float64)
    gp = CostGP(n, T, start, 0.05, dict(sigma_start=1e-4, sigma_gp=7e-4), CPU)
    d = gp.descriptors()[0]
    assert d["kind"] == L.COST_GP and d["flags"] == L.FLAG_GP_START and d["sigma"] == 7e-4
    assert d["sigma2"] == 1e-4 and d["host_data"] == list(map(float, range(14)))
    assert CostGPTrajectory(n, T, start, 0.05, dict(sigma_gp=0.1), CPU).descriptors()[0]["flags"] == 0
    gl = CostGoalPrior(n, T, multi_goal_states=goals, num_particles_per_goal=5, num_samples=8,
                       sigma_goal_prior=20., tensor_args=CPU).descriptors()[0]
    assert gl["kind"] == L.COST_GOAL_PRIOR and gl["dim0"] == 3 and gl["dim1"] == 40
    sp = CostCollision(n, T, field=LinkDistanceField(field_type='sdf', clamp_sdf=True, num_interpolate=2,
                                                     tensor_args=CPU), sigma_coll=0.01).descriptors()[0]
    assert sp["kind"] == L.COST_SPHERES and sp["flags"] == (L.FIELD_SDF | L.FLAG_SDF_CLAMP)
    assert sp["interp_lo"] == 5 and sp["interp_hi"] == 7
    np.testing.assert_allclose(sp["alpha"], [np.float32(1 / 3), np.float32(2 / 3)], rtol=1e-7)
    se = CostCollision(n, T, field=LinkSelfDistanceField(margin=0.03, tensor_args=CPU),
                       sigma_coll=0.01).descriptors()[0]
    assert se["kind"] == L.COST_SELF and se["sigma2"] == 0.03
    assert CostCollision(n, T, field=None, sigma_coll=1.).descriptors() == []
    om = ObstacleMap([20, 20], 0.1, tensor_args=CPU)
    gr = CostCollision(2, T, field=om, sigma_coll=1e-5).descriptors()[0]
    assert gr["kind"] == L.COST_GRID and (gr["dim0"], gr["dim1"]) == (200, 200)
    assert (gr["p0"], gr["p1"], gr["p2"]) == (0.1, 100.0, 100.0)
    fk = DifferentiableFrankaPanda(gripper=False, tensor_args=CPU)
    cc = CostComposite(n, T, [gp, CostCollision(n, T, field=LinkSelfDistanceField(tensor_args=CPU),
                                                sigma_coll=0.01)],
                       FK=fk.compute_forward_kinematics_all_links, tensor_args=CPU)
    assert len(cc.descriptors()) == 2 and len(cc.chain) == 10 and not cc.needs_spheres()
    # an arbitrary FK callable (cost_functions.py:39,51-52) is legal: its link-field children leave the
    # compiled program and are evaluated on the frames the callable returns
    fc = CostComposite(n, T, [gp, CostCollision(n, T, field=LinkSelfDistanceField(tensor_args=CPU),
                                                sigma_coll=0.01)], FK=lambda q: q, tensor_args=CPU)
    assert fc.foreign_fk and fc.chain is None and len(fc.descriptors()) == 1 and not cc.foreign_fk
    # edit counters: a moved target / re-built factor is visible to whoever compiled the cost
    v0 = cc.version()
    gp.set_cost_factors()
    assert cc.version() > v0
    # a CostGoal without a field contributes nothing, like the reference's (cost_functions.py:324-325)
    assert CostGoal(n, T, sigma_goal=1.).get_linear_system(None) == (None, None, None)
    H = torch.eye(4, dtype=torch.float64)
    H[:3, 3] = torch.tensor([0.4, 0.1, 0.5])
    ee = CostGoal(n, T, field=EESE3DistanceField(H, w_pos=2., w_rot=0.5, tensor_args=CPU),
                  sigma_goal=7e-3, tensor_args=CPU)
    de = ee.descriptors()[0]
    assert de["kind"] == L.COST_EE_GOAL and de["flags"] == L.FLAG_EE_SQUARE and de["sigma"] == 7e-3
    assert (de["p0"], de["p1"]) == (2., 0.5) and de["host_data"] == [float(v) for v in H.flatten()]
    assert ee.goal_factor.K == 1. / 7e-3 ** 2
    assert CostGoal(n, T, sigma_goal=1.).descriptors() == []    # cost_functions.py:306: no field -> 0


def test_rasteriser_and_synthetic_scene(golden):
    from stoch_gpmp_amd.envs.obst_map import (ObstacleCircle, ObstacleMap, ObstacleRectangle,
                                              synthetic_obstacle_map)
    om = ObstacleMap([20, 20], 0.1, tensor_args=CPU)
    assert om.map.shape == (200, 200) and (om.origin_xi, om.origin_yi) == (100, 100)
    ObstacleRectangle(1.0, -2.0, 2, 2)._add_to_map(om)
    assert om.map.sum() == 400 and om.map[80, 110] == 1 and om.map[69, 110] == 0 and om.map[70, 100] == 1
    ObstacleCircle(-3.0, 3.0, 1.0)._add_to_map(om)
    ys, xs = np.nonzero(om.map[120:141, 60:81])
    assert abs((om.map.sum() - 400) - np.pi * 100) < 15 and om.map[130, 70] == 1 and om.map[130, 59] == 0
    a = synthetic_obstacle_map(seed=3, tensor_args=CPU)
    b = synthetic_obstacle_map(seed=3, tensor_args=CPU)
    c = synthetic_obstacle_map(seed=4, tensor_args=CPU)
    assert np.array_equal(a.map, b.map) and not np.array_equal(a.map, c.map)
    assert a.map.max() == 1 and 2000 < a.map.sum() < 7000           # 15 non-overlapping obstacles
    g = golden("g2_planar_e2e.npz")
    ref = ObstacleMap.from_grid(g["grid"], float(g["cell_size"]), tensor_args=CPU)
    assert np.array_equal(ref.map, g["grid"].astype(np.float64))
    assert [ref.origin_xi, ref.origin_yi] == list(g["c_offset"])
    assert ref.map_torch.shape == (200, 200)


def test_scene_generators_reproduce_the_reference_scenes(golden):
    """generate_obstacle_map (map_generator.py:9-92) and random_init_static_sphere (envs/panda.py:42-66)
    consume the global `random` / `np.random` streams exactly like the reference: same seeds, same
    scene -- the example's 200x200 grid (g2) and the fixed+random coarse maps / sphere sets of g6."""
    import random
    from stoch_gpmp_amd.envs.map_generator import generate_obstacle_map
    from stoch_gpmp_amd.envs.obst_map import ObstacleCircle, ObstacleRectangle
    from stoch_gpmp_amd.envs.spheres import random_init_static_sphere, spawn_obstacle_spheres
    random.seed(0)
    np.random.seed(0)
    om, obs = generate_obstacle_map(map_dim=[20, 20], obst_list=[], cell_size=0.1, random_gen=True, num_obst=15,
                                    rand_limits=[[-7.5, 7.5], [-7.5, 7.5]], rand_rect_shape=[2, 2], tensor_args=CPU)
    assert len(obs) == 15 and np.array_equal(om.map, golden("g2_planar_e2e.npz")["grid"].astype(np.float64))
    g = golden("g6_scene_tooling.npz")
    for seed in (0, 7):
        random.seed(seed)
        np.random.seed(seed)
        om, obs = generate_obstacle_map(map_dim=[10, 12], obst_list=[ObstacleRectangle(0, 0, 2, 3),
                                                                     ObstacleCircle(-3, 2, 1.)],
                                        cell_size=0.25, random_gen=True, num_obst=7,
                                        rand_limits=[[-4, 4], [-5, 5]], rand_rect_shape=[1, 2],
                                        rand_circle_radius=0.75, tensor_args=CPU)
        assert len(obs) == int(g[f"map/{seed}/n_obst"])
        assert np.array_equal(om.map, g[f"map/{seed}/grid"].astype(np.float64))
        np.random.seed(seed)
        sph = spawn_obstacle_spheres(5)
        assert sph.shape == (1, 5, 4) and np.array_equal(sph[0].numpy(), g[f"spheres/{seed}"])
    np.random.seed(0)
    r, pos = random_init_static_sphere(0.1, 0.2, np.array([0.6, -0.2, 0.6]), np.array([1., 0.2, 1]), 0.01)
    assert r == g["spheres/0"][0, 3] and np.array_equal(pos, g["spheres/0"][0, :3])
    with pytest.raises(AssertionError):
        generate_obstacle_map(map_dim=[10, 10], obst_list=[ObstacleRectangle(0, 0, 1, 1)] * 3, random_gen=True,
                              num_obst=2, rand_limits=[[-1, 1], [-1, 1]], tensor_args=CPU)


def test_factor_mirrors_expose_the_reference_quantities():
    from oracle import ref_equiv as R
    from stoch_gpmp_amd.costs.factors.field_factor import FieldFactor
    from stoch_gpmp_amd.costs.factors.gp_factor import GPFactor
    from stoch_gpmp_amd.costs.factors.unary_factor import UnaryFactor
    gp = GPFactor(3, 0.7, 0.1, 5, CPU)
    assert torch.equal(gp.phi, R.phi_matrix(3, 0.1, torch.float64))
    assert torch.equal(gp.Q_inv[0], R.q_inv_matrix(3, 0.1, 0.7, torch.float64))
    assert gp.Q_inv.shape == (5, 6, 6)
    uf = UnaryFactor(6, 0.05, torch.ones(6, dtype=torch.float64), CPU)
    assert torch.equal(uf.K, R.unary_K(6, 0.05, torch.float64))
    assert torch.equal(uf.get_error(torch.zeros(2, 1, 6, dtype=torch.float64), calc_jacobian=False),
                       torch.ones(2, 1, 6, dtype=torch.float64))
    # the reference's defaults (gp_factor.py:54, unary_factor.py:22): a bare get_error(x) returns the Jacobians too
    e, H = uf.get_error(torch.zeros(2, 6, dtype=torch.float64))
    assert e.shape == (2, 6, 1) and torch.equal(H, torch.eye(6, dtype=torch.float64).expand(2, 6, 6))
    err, H1, H2 = gp.get_error(torch.zeros(4, 6, 6, dtype=torch.float64))
    assert err.shape == (4, 5, 6, 1) and torch.equal(H1[2], gp.calc_phi()) and torch.equal(H2[0], -torch.eye(6, dtype=torch.float64))
    assert torch.equal(gp.calc_Q_inv(), gp.Q_inv) and torch.equal(gp.calc_phi(), gp.phi)
    assert FieldFactor(2, 1e-5, [1, 64]).K == 1. / (1e-5 ** 2) and FieldFactor(2, 1., [1, 64]).length == 63


def test_bench_and_entry_points_exist():
    import importlib
    ge = importlib.import_module("__graft_entry__")
    assert callable(ge.build) and callable(ge.smoke)
    assert os.path.exists(os.path.join(ROOT, "bench.py"))


def test_dense_linear_systems_match_the_oracle():
    """get_linear_system of CostGP / CostGoalPrior / CostComposite (cost_functions.py:60-85,148-168,
    390-405) -- plain tensor assembly, no kernel involved -- against oracle/gpmp_equiv.py, which is
    pinned against a run of the reference's GPMP."""
    from oracle import gpmp_equiv as GP
    from stoch_gpmp_amd.costs.cost_functions import CostComposite, CostGP, CostGoalPrior
    n, T, nppg, dt = 3, 5, 2, 0.1
    g = torch.Generator().manual_seed(0)
    start = torch.randn(2 * n, generator=g, dtype=torch.float64)
    goals = torch.randn(2, 2 * n, generator=g, dtype=torch.float64)
    trajs = torch.randn(2 * nppg, T, 2 * n, generator=g, dtype=torch.float64)
    gp = CostGP(n, T, start, dt, dict(sigma_start=0.01, sigma_gp=0.3), CPU)
    gl = CostGoalPrior(n, T, multi_goal_states=goals, num_particles_per_goal=nppg, num_samples=1,
                       sigma_goal_prior=0.5, tensor_args=CPU)
    for got, want in ((gp.get_linear_system(trajs), GP.linear_system_gp(trajs, start, n, dt, 0.01, 0.3)),
                      (gl.get_linear_system(trajs), GP.linear_system_goal_prior(trajs, goals, nppg, n, 0.5))):
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    A, b, K = CostComposite(n, T, [gp, gl], tensor_args=CPU).get_linear_system(trajs)
    Ao, bo, Ko = GP.composite_linear_system(trajs, [GP.linear_system_gp(trajs, start, n, dt, 0.01, 0.3),
                                                    GP.linear_system_goal_prior(trajs, goals, nppg, n, 0.5)])
    assert torch.equal(A, Ao) and torch.equal(b, bo) and torch.equal(K, Ko)
    err, H1, H2 = gp.gp_prior.get_error(trajs, calc_jacobian=True)
    assert err.shape == (4, T - 1, 2 * n, 1) and H1.shape == H2.shape == (T - 1, 2 * n, 2 * n)
    e, H = gp.start_prior.get_error(trajs[:, [0]], calc_jacobian=True)
    assert e.shape == (4, 2 * n, 1) and torch.equal(H[0], torch.eye(2 * n, dtype=torch.float64))



def _canned_bench_record():
    """A record of the shape bench.py's main() assembles (values of a round-5 run), with the things that once broke or could
    break the driver's parse: NaN / Infinity, paragraphs of prose, an eight-rank table, fat per-configuration objects."""
    roof = {"bound": "valu", "kernel": "fused_step_kernel", "achieved": 5115.123456789, "peak": 8000.0, "unit": "GB/s",
            "frac": 0.6393904320987, "traffic": 480102400, "valu_frac": 0.7812345, "moved_frac": 0.3412345,
            "launch_ms": 0.175512345, "frac_launch": 0.669912345}
    sf = {"kernel": "fused_step_kernel", "iterations_per_s": 6086.123456, "ms_per_step": 0.16431, "valu_frac": 0.851234,
          "launch_ms": 0.1574, "counters": "profiles/r05/final_cfg3_store_free_* " * 20, "steps": 20}
    row = {"config": "config 5 share: Panda 4 goals, 512 of 4096 x256x128 fp32", "workload": "w" * 200,
           "iterations_per_s": 2929.123456, "store_free": dict(sf), "roofline": dict(roof), "roofline_detail": {"x": "y" * 500},
           "cost_kernel": "fused_step_kernel", "dtype": "f32", "kernel_ms_per_step": {"cost_sweep": 0.33}}
    return {
        "metric": "planner iterations/sec (and ms/iter) at fixed particles x samples x T", "value": 5438.4321012345,
        "unit": "iterations/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.18387654321,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "field": "rbf",
        "config": {"workload": "Panda 7-DoF 1024p (1024/GPU) x 128s x 64t, 1 goal, GP+goal+self+5 spheres rbf",
                   "particles_per_gpu": 1024, "particles_total": 1024, "samples": 128, "traj_len": 64, "state_dim": 14,
                   "parallelism": "single GPU", "noise": "philox4x32-7 + Box-Muller (in-kernel)", "prior_factor_dtype": "f64"},
        "roofline": roof, "roofline_detail": {"algorithmic_bytes_per_launch": 940572672, "note": "n" * 3000},
        "cpu_baseline": {"value": 0.0111234, "unit": "iterations/s", "cores": 16, "kind": "port",
                         "sample": "4+8 of 1024 particles at full S=128 T=64, float32, 16 of 256 host cores; affine "
                                   "extrapolation through the measured points to P=1024",
                         "measured_points": [{"particles": 4, "s_per_it": 0.45}] * 3, "leg_seconds": float("inf")},
        "speedup_vs_cpu_baseline": 489000.123, "single_iteration_calls": {"iterations_per_s": 5180.1, "ms_per_step": 0.19305},
        "mode": "storing", "store_free": sf,
        "sweep_alone": {"kernel": "cost_sweep_chunked_kernel", "launch_ms": 0.1296, "frac": 0.4541234,
                        "sampler_kernel": "sample_iso_kernel", "sampler_launch_ms": 0.1059, "sampler_frac": 0.5551,
                        "kernel_ms_per_step": {"a": 1.0}},
        "parity": {"ok": True, "means_rel_err_max_after_K": 2.3e-7, "cost_rel_err_max_while_tracking": float("nan"),
                   "iterations": 10, "tolerance_on_means": 1e-3, "departures_on_near_ties": [], "what": "p" * 900},
        "rccl": {"ranks": 8, "rank": 0, "version": 22204, "library": "librccl.so", "test_hooks_build": False},
        "shared_gpu_test_double": False,
        "per_rank_iterations_per_s": {"min": 5400.123456789, "max": 5500.123456789, "all": [5400.123456789 + i for i in range(8)]},
        "kernel_ms_per_step": {"cost_sweep": 0.1755, "update": 0.0139}, "launches_per_iteration": 2,
        "last_iteration": {"mean_cost_sum": 1.234e11, "mean_min_cost": 5.6e8},
        "other_configs": [dict(row) for _ in range(3)], "passes": "prose " * 400,
    }


def test_bench_line_is_compact_strict_json_with_the_contract_keys(tmp_path):
    """The driver parses bench.py's single stdout line whole or not at all (round 5: a 23 KB line was kept as `parsed:
    null`).  `compact_line` / `emit` must give < 4 KB of strict JSON carrying the contract keys, whatever the detailed
    record holds; the detail goes to bench_detail.json."""
    import json
    import bench

    def no_constants(name):
        raise AssertionError(f"non-strict JSON constant {name}")
    full = _canned_bench_record()
    line = bench.compact_line(full)
    assert len(line) < 4096 and "\n" not in line
    rec = json.loads(line, parse_constant=no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"] and rec["vs_baseline"] is None
    assert set(rec["config"]) >= {"workload", "particles_total", "samples", "traj_len"} and "model" not in rec["config"]
    assert set(rec["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "valu_frac",
                                    "moved_frac", "launch_ms"}
    assert set(rec["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert rec["single_iteration_calls"]["iterations_per_s"] > 0 and rec["store_free"]["iterations_per_s"] > 0
    assert rec["parity"]["ok"] is True and rec["parity"]["cost_rel_err_max_while_tracking"] is None     # NaN -> null
    assert rec["rccl"]["ranks"] == 8 and len(rec["per_rank_iterations_per_s"]["all"]) == 8
    assert rec["per_rank_iterations_per_s"]["min"] == full["per_rank_iterations_per_s"]["min"]          # (not rounded)
    assert rec["sweep_alone_frac"] == pytest.approx(0.4541, rel=1e-3)
    assert [set(r) for r in rec["other_configs"]] == [{"config", "it_per_s", "store_free_it_per_s", "frac", "bound", "kernel",
                                                       "dtype"}] * 3
    # a record that grew (forty configurations) loses optional parts, never the contract keys, and stays under the limit
    fat = dict(full, other_configs=[dict(full["other_configs"][0]) for _ in range(40)])
    line2 = bench.compact_line(fat)
    rec2 = json.loads(line2, parse_constant=no_constants)
    assert len(line2) < 4096 and "other_configs" not in rec2 and rec2["roofline"]["frac"] > 0 and rec2["cpu_baseline"]["value"] > 0
    # emit(): the detail file is strict JSON too and holds what the line leaves out; the line goes to the given descriptor
    r, w = os.pipe()
    detail = str(tmp_path / "bench_detail.json")
    out = bench.emit(full, w, detail)
    os.close(w)
    got = os.read(r, 65536).decode()
    os.close(r)
    assert got == out + "\n" and got.count("\n") == 1
    det = json.loads(open(detail).read(), parse_constant=no_constants)
    assert det["roofline_detail"]["algorithmic_bytes_per_launch"] == 940572672 and det["cpu_baseline"]["leg_seconds"] is None
    assert json.loads(out)["detail"] == "bench_detail.json"


def test_bench_multi_gpu_launch_starts_ranks_and_relays_failure():
    """`python bench.py --gpus 2` with no launcher around it starts its own ranks (torch.distributed.run)
    from a parent that never touches the GPU, and relays the outcome: in this GPU-less container both
    ranks fail at device selection, so the parent must come back (no hang) with a non-zero exit code and
    without printing a JSON line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import json
        assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert p.returncode != 0
        assert '"metric"' not in p.stdout


def test_hand_placed_loads_are_not_touched_before_their_wait():
    """hipcc treats an inline-asm load's destination as written at the statement; the audit compiles the cost
    kernels to gfx950 assembly and checks that nothing reads or copies such a register before the wait."""
    import shutil
    import subprocess
    import sys
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # 11 = fused_step_kernel x 3 field types x {on the 8 x 16 grid, masked} + cost_sweep_chunked_kernel x 3 + fused_planar_kernel x 2 (n = 2, 3)
    assert re.search(r"\b17 kernels audited, \d+ hand-placed loads, 0 offending", r.stdout), r.stdout
    assert "0 kernels with scratch" in r.stdout, r.stdout       # (no spilled vector register in any launch of the step)


def test_run_time_chain_code_compiles_for_gfx950_without_a_device():
    """csrc/chain_rtc.hip: the chain kernels of a robot the library was not built for -- the host generates
    `struct ChainCode_rt` (csrc/gen/chain_codegen.py) and the library compiles cost_device.h + fused_step.inc around it with
    hiprtc.  Compilation needs no GPU, so the build container can hold the kernel sources to "still includable by the
    run-time translation unit" (no host library headers, no host declarations under __HIPCC_RTC__): a 6-DoF and a 7-DoF
    chain, all three sphere-field types; and broken chain code comes back as SGPMP_ESTATE with the compiler's log."""
    import ctypes as C
    from stoch_gpmp_amd import _lib
    from stoch_gpmp_amd.engine import _chain_struct_source
    from stoch_gpmp_amd.robots.panda_chain import PANDA_CHAIN
    lib = _lib.load()
    h = 1.57079632679
    arm6 = [("j1", "revolute", (0, 0, 0), (0, 0, 0.1625)), ("j2", "revolute", (h, 0, 0), (0, 0, 0)),
            ("j3", "revolute", (0, 0, 0), (-0.425, 0, 0)), ("j4", "revolute", (0, 0, 0), (-0.3922, 0, 0.1333)),
            ("j5", "revolute", (h, 0, 0), (0, -0.0997, 0)), ("j6", "revolute", (-h, 0, 0), (0, 0.0996, 0)),
            ("tool", "fixed", (0, 0, 0), (0, 0, 0.12))]
    arm7 = [(nm, kind, rpy, (xyz[0] + (0.01 if i == 3 else 0.0), xyz[1], xyz[2])) for i, (nm, kind, rpy, xyz) in enumerate(PANDA_CHAIN)]
    for chain, fts in ((arm6, (0, 1, 2)), (arm7, (0,))):
        src = _chain_struct_source(chain)
        assert "struct ChainCode_rt" in src and f"N = {sum(1 for j in chain if j[1] == 'revolute')};" in src
        for ft in fts:
            n = C.c_int64(0)
            rc = lib.sgpmp_fk_codegen_compile(src.encode(), ft, C.byref(n))
            assert rc == _lib.OK and n.value > 10000, (rc, _lib.last_error()[:2000])
    rc = lib.sgpmp_fk_codegen_compile(b"struct ChainCode_rt { static constexpr int N = 7; };", 0, None)
    assert rc == _lib.ESTATE and "hiprtc" in _lib.last_error()


def test_run_time_chain_compiler_refuses_kernel_sources_it_was_not_built_from(tmp_path):
    """The library passes CostArgs / FlatProg / FusedArgs by value in layouts fixed when IT was built; chain kernels compiled
    from other sources (an edited tree, a library loaded from elsewhere through SGPMP_LIB_PATH) would read them with another
    layout.  The content hash of the kernel sources is baked into the library (csrc/gen/rtc_hash.py): a csrc directory whose
    files differ by one byte is refused (SGPMP_ESTATE), SGPMP_RTC_ALLOW_EDITED_SOURCES=1 is the development override.
    (Own process: SGPMP_CSRC_DIR is read when the compiler looks for its sources.)"""
    import shutil
    import subprocess
    import sys
    src = os.path.join(ROOT, "stoch_gpmp_amd", "csrc")
    inc = os.path.join(ROOT, "include")
    edited = tmp_path / "tree" / "stoch_gpmp_amd" / "csrc"
    shutil.copytree(src, edited, ignore=shutil.ignore_patterns("build*", "*.o"))
    shutil.copytree(inc, tmp_path / "tree" / "include")
    with open(edited / "fused_step.inc", "a") as f:
        f.write("// one more line\n")
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from stoch_gpmp_amd import _lib\n"
        "from stoch_gpmp_amd.engine import _chain_struct_source\n"
        "from stoch_gpmp_amd.robots.panda_chain import PANDA_CHAIN\n"
        "lib = _lib.load()\n"
        "chain = [(nm, kind, rpy, (xyz[0] + (0.01 if i == 3 else 0.0), xyz[1], xyz[2])) for i, (nm, kind, rpy, xyz) in enumerate(PANDA_CHAIN)]\n"
        "rc = lib.sgpmp_fk_codegen_compile(_chain_struct_source(chain).encode(), 0, None)\n"
        "print('RC', rc, _lib.last_error()[:300].replace(chr(10), ' '))\n" % ROOT)
    env = dict(os.environ, SGPMP_CSRC_DIR=str(edited))
    env.pop("SGPMP_RTC_ALLOW_EDITED_SOURCES", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "RC -4" in r.stdout and "not the ones this libsgpmp.so was built from" in r.stdout, r.stdout + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, SGPMP_RTC_ALLOW_EDITED_SOURCES="1"), capture_output=True,
                       text=True, timeout=600)
    assert "RC 0" in r.stdout, r.stdout + r.stderr[-2000:]


def test_host_bookkeeping_under_address_and_ub_sanitizers(tmp_path):
    """tests/host_asan: api.hip and comm.hip compiled host-only with -fsanitize=address,undefined over a stub HIP runtime
    (malloc-backed device memory, streams that execute at enqueue), stub launchers that touch the kernels' byte ranges and
    the shared-memory stand-in for librccl -- contexts, priors, cost programs, chains, the step as one chain / two
    particle-half chains / with profiling events / per-step mode statistics / an empty shard, the statistics ring beyond its
    eight slots, the reduced-event table beyond its eight entries, communicator re-attach, destroy (no stream, event or
    buffer may outlive its context).  Also run with the fused launch eligible and with lagging events (the fall-back stream
    waits), and once with a buffer deliberately one element short: that run must FAIL with an AddressSanitizer report."""
    import shutil
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang")
    d = os.path.join(ROOT, "tests", "host_asan")
    b = str(tmp_path / "build")
    r = subprocess.run(["make", "-C", d, f"B={b}", "-j4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    exe, fake = os.path.join(b, "host_asan"), os.path.join(b, "libfakerccl_stub.so")
    base = {k: v for k, v in os.environ.items() if not k.startswith(("SGPMP_", "STUB_", "HOST_ASAN"))}
    base["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    for extra in ({}, {"SGPMP_RCCL_LIB": fake}, {"SGPMP_RCCL_LIB": fake, "STUB_FUSED": "1", "STUB_EVENT_LAG": "1"},
                  {"STUB_FUSED": "1", "SGPMP_NO_STEP_PIPELINE": "1"}, {"STUB_FUSED": "1", "SGPMP_NO_DENSE_PARTIALS": "1"},
                  {"STUB_FUSED": "1", "STUB_TAIL": "1"}, {"SGPMP_RCCL_LIB": fake, "STUB_FUSED": "1", "STUB_TAIL": "1"},
                  {"STUB_FUSED": "1", "STUB_TAIL": "1", "STUB_PERSIST": "1"},      # several iterations per launch: sgpmp_optimize's chunking
                  {"STUB_FUSED": "1", "SGPMP_NO_EE_FOLD": "1"}):
        p = subprocess.run([exe], env=dict(base, **extra), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "HOST_ASAN_OK" in p.stdout, (extra, p.stdout[-1000:], p.stderr[-4000:])
        assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
        multi = int(p.stdout.split("MULTI_ITERATION_LAUNCHES")[1].split()[0])
        assert (multi > 0) == ("STUB_PERSIST" in extra), (extra, multi)
    p = subprocess.run([exe], env=dict(base, HOST_ASAN_INJECT="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "AddressSanitizer" in p.stderr, "the harness did not notice a short buffer"
    shutil.rmtree(b, ignore_errors=True)


def test_mirror_classes_accept_the_reference_signatures():
    """INTEGRATION.md section 1 says the classes of stoch_gpmp_amd keep the reference's names and signatures.  Checked against
    DATA dumped from the reference itself (oracle/gen_golden.py g11: inspect.signature of every public method of the classes
    on the path): every method exists, takes the reference's parameters under the same names in the same positional order
    with the same defaults, and may take more (a superset: extra keyword arguments with defaults)."""
    import importlib
    import inspect
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "g11_signatures.json")))
    problems = []
    checked = 0
    for cname, methods in sorted(ref.items()):
        modname, _, clsname = cname.rpartition(".")
        mod = importlib.import_module("stoch_gpmp_amd." + modname)
        owner = mod if clsname == "<module>" else getattr(mod, clsname, None)
        if owner is None:
            problems.append(f"{cname}: class missing")
            continue
        for mname, params in sorted(methods.items()):
            fn = getattr(owner, mname, None)
            if fn is None or not callable(fn):
                problems.append(f"{cname}.{mname}: missing")
                continue
            mine = [(n, p) for n, p in inspect.signature(fn).parameters.items() if n != "self"]
            names = [n for n, p in mine if p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL)]
            has_kw = any(p.kind == p.VAR_KEYWORD for _, p in mine)
            ref_named = [(n, d) for n, d in params if not n.startswith("*")]
            if any(n.startswith("**") for n, _ in params) and not has_kw:
                problems.append(f"{cname}.{mname}: the reference takes **kwargs, the mirror does not")
            for i, (n, d) in enumerate(ref_named):
                checked += 1
                if n not in names:
                    if not has_kw:
                        problems.append(f"{cname}.{mname}: parameter {n!r} not accepted")
                    continue
                if names.index(n) != i:
                    problems.append(f"{cname}.{mname}: parameter {n!r} is positional #{names.index(n)}, the reference's is #{i}")
                p = dict(mine)[n]
                if d is not None:
                    if p.default is inspect.Parameter.empty or repr(p.default) != d:
                        have = "<required>" if p.default is inspect.Parameter.empty else repr(p.default)
                        problems.append(f"{cname}.{mname}: default of {n!r} is {have}, the reference's is {d}")
            # (a parameter the mirror adds must not be required)
            for n, p in mine:
                if p.kind in (p.VAR_KEYWORD, p.VAR_POSITIONAL) or n in dict(ref_named):
                    continue
                if p.default is inspect.Parameter.empty:
                    problems.append(f"{cname}.{mname}: extra REQUIRED parameter {n!r}")
    assert checked > 150 and not problems, "\n".join(problems)
