"""Thin object wrapper over one `sgpmp_ctx` (include/sgpmp.h).  Plumbing only: it converts torch
tensors to raw device pointers and Python descriptors to the C structs; no arithmetic happens here.
"""
import ctypes as C
import os

import torch

from . import _lib as L


_CODEGEN = None


def _chain_struct_source(chain):
    """`struct ChainCode_rt { ... };` for `chain` from the generator that wrote the library's built-in chain code
    (csrc/gen/chain_codegen.py: FK as straight-line code with folded joint constants, merged link frames, q-dependent pairs)."""
    global _CODEGEN
    if _CODEGEN is None:
        import importlib.util
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "gen", "chain_codegen.py")
        spec = importlib.util.spec_from_file_location("sgpmp_chain_codegen", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _CODEGEN = mod
    norm = [(str(nm), str(kind), tuple(float(v) for v in rpy), tuple(float(v) for v in xyz)) for nm, kind, rpy, xyz in chain]
    return "\n".join(_CODEGEN.gen_chain("rt", norm)) + "\n"


class Engine:
    def __init__(self, n_dof, traj_len, num_particles, num_samples, num_goals=1,
                 num_particles_per_goal=None, particle_offset=0, num_particles_global=None,
                 tensor_args=None):
        self.lib = L.load()
        self.tensor_args = tensor_args
        self.device = L.require_cuda(tensor_args)
        self.dtype = tensor_args["dtype"]
        nppg = num_particles_per_goal if num_particles_per_goal is not None else max(num_particles, 1)
        self.dims = L.Dims(n_dof, traj_len, num_particles, particle_offset,
                           num_particles_global if num_particles_global is not None
                           else num_particles, num_samples, num_goals, nppg,
                           L.dtype_code(self.dtype), 0)
        self.n, self.T, self.d = n_dof, traj_len, 2 * n_dof
        self.P, self.S = num_particles, num_samples
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_create(C.byref(self.dims), C.byref(self._ctx)))
        self._keep = []          # tensors whose device memory the cost program points at
        self.n_links = None
        self._prior_key = {}     # which -> arguments of the last successful set_prior
        self._prior_modes = {}   # which -> number of per-mode factors after set_prior_blocks (absent: one shared factor)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.lib.sgpmp_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ development switches
    def set_option(self, name, value=1):
        """Flip a development switch of this context (include/sgpmp.h: sgpmp_set_option)."""
        L.check(self.lib.sgpmp_set_option(self._ctx, name.encode(), int(value)))

    def pipeline_begin(self):
        """The steps that follow, up to pipeline_end(), may run as two particle-half chains on the context's own
        streams (include/sgpmp.h: sgpmp_pipeline_begin); nothing else may touch their buffers meanwhile."""
        L.check(self.lib.sgpmp_pipeline_begin(self._ctx, L.stream_ptr()))

    def pipeline_end(self):
        L.check(self.lib.sgpmp_pipeline_end(self._ctx, L.stream_ptr()))

    def pipeline_split_steps(self):
        """Steps of this context that ran as two chains so far."""
        return int(self.lib.sgpmp_pipeline_split_steps(self._ctx))

    def last_step_launches(self):
        """Kernels the last step enqueued for its particle range (1: the whole iteration in one launch)."""
        return int(self.lib.sgpmp_last_step_launches(self._ctx))

    def dense_particles(self):
        """Particles whose last in-step update spread its weight over more than S / 4 samples (include/sgpmp.h)."""
        k = C.c_int64()
        L.check(self.lib.sgpmp_dense_particles(self._ctx, C.byref(k), None))
        return k.value

    def dense_armed_steps(self):
        """Steps launched with the fused launch's softmax partials armed so far (include/sgpmp.h: sgpmp_dense_particles)."""
        k, a = C.c_int64(), C.c_int64()
        L.check(self.lib.sgpmp_dense_particles(self._ctx, C.byref(k), C.byref(a)))
        return a.value

    def row_counts(self):
        """Per particle: the rows that carried weight in its last in-step update (include/sgpmp.h: sgpmp_row_counts_get) --
        state of a run: the next step's launch and update decide on it."""
        import numpy as np
        out = np.zeros(max(self.P, 1), dtype=np.uint32)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_row_counts_get(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out[:self.P]

    def set_row_counts(self, counts=None):
        """Restore the per-particle row counts (None: clear them, as a fresh run has them)."""
        import numpy as np
        ptr = None
        if counts is not None:
            arr = np.ascontiguousarray(np.asarray(counts, dtype=np.uint32))
            assert arr.shape == (self.P,)
            ptr = arr.ctypes.data_as(C.POINTER(C.c_uint32))
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_row_counts_set(self._ctx, ptr))

    def clear_row_counts(self):
        """reset()'s clear: zeroes on the current stream, no host synchronisation (sgpmp_row_counts_clear)."""
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_row_counts_clear(self._ctx, L.stream_ptr()))

    def store_free_steps(self):
        """Steps of this context that did not write their samples (SGPMP_STEP_NO_SAMPLES honoured)."""
        return int(self.lib.sgpmp_store_free_steps(self._ctx))

    def multi_iteration_launches(self):
        """Launches that ran several iterations of an optimize(opt_iters = K) call each (planar problems, 64 samples per particle)."""
        return int(self.lib.sgpmp_multi_iteration_launches(self._ctx))

    def last_cost_kernel(self):
        """Name of the cost-sweep kernel the dispatcher picked at the last launch."""
        return self.lib.sgpmp_last_cost_kernel(self._ctx).decode()

    # ------------------------------------------------------------------ multi-GPU (RCCL behind the C ABI)
    def comm_unique_id(self):
        """128-byte RCCL id (rank 0 calls this and hands the bytes to the other ranks)."""
        buf = C.create_string_buffer(128)
        L.check(self.lib.sgpmp_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, world_size, rank):
        """Collective: attach an RCCL communicator; sgpmp_step then all-reduces its statistics itself."""
        assert len(unique_id) == 128
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_comm_init(self._ctx, unique_id, int(world_size), int(rank)))

    def comm_info(self):
        """(world, rank, rccl_version) as the attached RCCL communicator itself reports them (ncclCommCount,
        ncclCommUserRank, ncclGetVersion); world == 0 when no communicator is attached."""
        w, r, v = C.c_int(), C.c_int(), C.c_int()
        L.check(self.lib.sgpmp_comm_info(self._ctx, C.byref(w), C.byref(r), C.byref(v)))
        return w.value, r.value, v.value

    def comm_library(self):
        """(name of the collective library this process bound -- "" before the first communicator --, 1 when this is the
        test-hooks build that honours SGPMP_RCCL_LIB); include/sgpmp.h: sgpmp_comm_library."""
        h = C.c_int(0)
        name = self.lib.sgpmp_comm_library(C.byref(h))
        return (name or b"").decode(), h.value

    def allreduce_stats(self, stats):
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_allreduce_stats(self._ctx, L.ptr(stats), L.stream_ptr()))

    def stats_wait(self, stats=None):
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_stats_wait(self._ctx, L.ptr(stats), L.stream_ptr()))

    def mode_stats(self, means, out=None):
        """Local per-goal sums of the particle means: out [G, T*d + 1, 2] fp64 (include/sgpmp.h: sgpmp_mode_stats)."""
        self._chk(means, "means")
        if out is None:
            out = torch.empty(self.dims.num_goals, self.T * self.d + 1, 2, device=self.device, dtype=torch.float64)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_mode_stats(self._ctx, L.ptr(means), L.ptr(out), L.stream_ptr()))
        return out

    def allreduce_f64(self, buf):
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_allreduce_f64(self._ctx, L.ptr(buf), buf.numel(), L.stream_ptr()))

    def set_step_mode_stats(self, buf):
        """Every step from now on leaves the per-goal mean statistics, summed over all ranks, in `buf` (None: off)."""
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_set_step_mode_stats(self._ctx, L.ptr(buf)))

    def mode_stats_wait(self):
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_mode_stats_wait(self._ctx, L.stream_ptr()))

    def allgather_means(self, local_means, world_size):
        self._chk(local_means, "means")
        out = torch.empty((local_means.shape[0] * world_size,) + tuple(local_means.shape[1:]),
                          **self.tensor_args)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_allgather_means(self._ctx, L.ptr(local_means), L.ptr(out), L.stream_ptr()))
        return out

    # ------------------------------------------------------------------ setup
    def set_prior(self, which, dt, sigma_start, sigma_gp, sigma_goal=None, Q_c_inv=None):
        qc, flat = None, None
        if Q_c_inv is not None:
            flat = [float(v) for v in torch.as_tensor(Q_c_inv).detach().cpu().double().flatten()]
            assert len(flat) == self.n * self.n
            qc = (C.c_double * len(flat))(*flat)
        # K1's output depends on these numbers only: a repeated call (reset() of a planner whose
        # sigmas did not change) keeps the factor that is already on the device
        key = (float(dt), float(sigma_start), None if sigma_gp is None else float(sigma_gp),
               None if sigma_goal is None else float(sigma_goal), None if flat is None else tuple(flat))
        if self._prior_key.get(which) == key:
            return
        self._prior_key.pop(which, None)
        self._prior_modes.pop(which, None)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_set_prior(
                self._ctx, which, float(dt), float(sigma_start),
                float(sigma_gp) if sigma_gp is not None else -1.0,
                float(sigma_goal) if sigma_goal is not None else -1.0, qc, L.stream_ptr()))
        self._prior_key[which] = key

    def set_priors(self, dt, init, sample):
        """Both priors of a planner's reset(): init / sample = (sigma_start, sigma_gp, sigma_goal or None).  The two K1
        factorisations (one wave, ~1.2 ms each) run concurrently (include/sgpmp.h: sgpmp_set_priors); priors whose
        numbers did not change keep the factor already on the device."""
        keys = {}
        for which, (ss, sgp, sgoal) in ((L.PRIOR_INIT, init), (L.PRIOR_SAMPLE, sample)):
            keys[which] = (float(dt), float(ss), float(sgp), None if sgoal is None else float(sgoal), None)
        stale = [w for w in keys if self._prior_key.get(w) != keys[w]]
        if len(stale) == 1:
            w = stale[0]
            ss, sgp, sgoal = init if w == L.PRIOR_INIT else sample
            return self.set_prior(w, dt, ss, sgp, sgoal)
        if not stale:
            return
        arr = lambda i: (C.c_double * 2)(*[float(-1.0 if v[i] is None else v[i]) for v in (init, sample)])   # noqa: E731
        for w in stale:
            self._prior_key.pop(w, None)
            self._prior_modes.pop(w, None)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_set_priors(self._ctx, float(dt), arr(0), arr(1), arr(2), L.stream_ptr()))
        self._prior_key.update(keys)

    def get_prior(self, which, n_modes=None, blocks_only=False):
        """-> (blocks [4,d,d], G [T,d,d], H [T,d,d]) as fp64 CPU tensors (inspection/tests); after
        set_prior_blocks pass n_modes: G, H are [n_modes,T,d,d] and blocks is None.  blocks_only: the four precision
        blocks of the shared closed-form prior alone (valid also after set_prior_blocks) -> (blocks, None, None)."""
        d, T = self.d, self.T
        if blocks_only:
            blocks = (C.c_double * (4 * d * d))()
            with torch.cuda.device(self.device):
                L.check(self.lib.sgpmp_get_prior(self._ctx, which, blocks, None, None))
            return torch.tensor(list(blocks), dtype=torch.float64).reshape(4, d, d), None, None
        held = self._prior_modes.get(which, 0)
        if held and (n_modes is None or n_modes < held):
            n_modes = held                               # (the library copies every mode's factor: size the host buffers for them)
        m = 1 if n_modes is None else n_modes
        blocks = (C.c_double * (4 * d * d))()
        G = (C.c_double * (m * T * d * d))()
        H = (C.c_double * (m * T * d * d))()
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_get_prior(self._ctx, which, blocks, G, H))
        to_t = lambda a, shape: torch.tensor(list(a), dtype=torch.float64).reshape(shape)
        if n_modes is None:
            return to_t(blocks, (4, d, d)), to_t(G, (T, d, d)), to_t(H, (T, d, d))
        return None, to_t(G, (m, T, d, d)), to_t(H, (m, T, d, d))

    def set_prior_blocks(self, which, D, E):
        """Per-mode block-tridiagonal precisions: D [modes,T,d,d], E [modes,T-1,d,d] (fp64 host tensors)."""
        D = torch.as_tensor(D, dtype=torch.float64).contiguous().cpu()
        E = torch.as_tensor(E, dtype=torch.float64).contiguous().cpu()
        modes = D.shape[0]
        assert D.shape == (modes, self.T, self.d, self.d) and E.shape == (modes, self.T - 1, self.d, self.d)
        self._prior_key.pop(which, None)
        self._prior_modes[which] = modes
        Dp = C.cast(D.data_ptr(), C.POINTER(C.c_double))
        Ep = C.cast(E.data_ptr(), C.POINTER(C.c_double))
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_set_prior_blocks(self._ctx, which, modes, Dp, Ep, L.stream_ptr()))

    def prior_quadform(self, which, x, means):
        """(x_r - mu_m)^T Sigma_m^-1 (x_r - mu_m), m = r % modes: x [rows, M], means [modes, M] -> fp64 [rows]."""
        self._chk(x, "x")
        self._chk(means, "means")
        rows, modes = x.shape[0], means.shape[0]
        out = torch.empty(rows, device=self.device, dtype=torch.float64)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_prior_quadform(self._ctx, which, L.ptr(x), rows, L.ptr(means), modes,
                                                  L.ptr(out), L.stream_ptr()))
        return out

    def set_costs(self, descs):
        """descs: list of dicts produced by the Cost classes' `descriptor()`."""
        arr = (L.CostDesc * max(len(descs), 1))()
        self._keep = []
        host_keep = []
        for i, dsc in enumerate(descs):
            cd = arr[i]
            cd.kind = dsc["kind"]
            cd.flags = dsc.get("flags", 0)
            cd.sigma = float(dsc["sigma"])
            cd.sigma2 = float(dsc.get("sigma2", 0.0))
            cd.dt = float(dsc.get("dt", 0.0))
            cd.dim0, cd.dim1 = int(dsc.get("dim0", 0)), int(dsc.get("dim1", 0))
            cd.p0, cd.p1, cd.p2 = (float(dsc.get(k, 0.0)) for k in ("p0", "p1", "p2"))
            cd.num_interpolate = int(dsc.get("num_interpolate", 0))
            cd.interp_lo, cd.interp_hi = int(dsc.get("interp_lo", 0)), int(dsc.get("interp_hi", 0))
            for a, v in enumerate(dsc.get("alpha", [])):
                cd.alpha[a] = float(v)
            if "host_data" in dsc:
                vals = [float(v) for v in dsc["host_data"]]
                buf = (C.c_double * len(vals))(*vals)
                host_keep.append(buf)
                cd.data = C.cast(buf, C.c_void_p)
            elif "device_tensor" in dsc:
                t = dsc["device_tensor"].to(device=self.device, dtype=self.dtype).contiguous()
                self._keep.append(t)
                cd.data = C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_set_costs(self._ctx, arr, len(descs)))

    def set_fk(self, chain, codegen=True):
        """chain: list of (name, 'revolute'|'fixed', rpy, xyz).  codegen: for a chain the library was not built with, also
        generate its straight-line code (csrc/gen/chain_codegen.py) and have the library compile its fast launches at run
        time (sgpmp_set_fk_codegen); where that is impossible the chain keeps the slower run-time-constant kernels and
        `fk_codegen_error` says why."""
        arr = (L.Joint * len(chain))()
        for i, (_, kind, rpy, xyz) in enumerate(chain):
            arr[i].rpy = (C.c_double * 3)(*[float(v) for v in rpy])
            arr[i].xyz = (C.c_double * 3)(*[float(v) for v in xyz])
            arr[i].revolute = 1 if kind == "revolute" else 0
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_set_fk(self._ctx, arr, len(chain)))
        self.n_links = len(chain) + 1
        self.fk_codegen_error = None
        if codegen and self.fk_codegen_info()[0] == 0 and self.dtype == torch.float32 and self.n <= 7 \
                and not os.environ.get("SGPMP_NO_RTC"):
            try:
                src = _chain_struct_source(chain)
                with torch.cuda.device(self.device):
                    L.check(self.lib.sgpmp_set_fk_codegen(self._ctx, src.encode()))
            except (ValueError, RuntimeError, L.SgpmpError) as e:        # the chain stays on the slower kernels
                self.fk_codegen_error = str(e)
                import warnings
                warnings.warn(f"stoch_gpmp_amd: no run-time chain code for this robot ({e}); using the generic cost sweep")

    def fk_codegen_info(self):
        """(codegen_id, seconds in hiprtc, code objects compiled, taken from the disk cache); codegen_id: 0 run-time
        constants, 1 the code built with the library (Panda), 2 compiled at run time (include/sgpmp.h)."""
        cid, cs, nc, nf = C.c_int(), C.c_double(), C.c_int(), C.c_int()
        L.check(self.lib.sgpmp_fk_codegen_info(self._ctx, C.byref(cid), C.byref(cs), C.byref(nc), C.byref(nf)))
        return cid.value, cs.value, nc.value, nf.value

    # ------------------------------------------------------------------ kernels
    def _chk(self, t, name):
        if t is None:
            return
        if not t.is_cuda or t.dtype != self.dtype or not t.is_contiguous():
            raise ValueError(f"{name}: expected a contiguous {self.dtype} tensor on {self.device}")

    def sample(self, which, seed, draw, means, n_samples, out=None, eps=None, eps_mode_offset=0,
               mode_offset=0):
        n_modes = means.shape[0]
        self._chk(means, "means")
        if out is None:
            out = torch.empty(n_modes, n_samples, self.T, self.d, **self.tensor_args)
        self._chk(out, "out")
        eps_modes = 0
        if eps is not None:
            self._chk(eps, "eps")
            assert eps.dim() == 3 and eps.shape[0] == n_samples and eps.shape[2] == self.T * self.d
            eps_modes = eps.shape[1]
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_sample(self._ctx, which, int(seed), int(draw), L.ptr(means),
                                          n_modes, mode_offset, n_samples, L.ptr(eps), eps_modes,
                                          eps_mode_offset, L.ptr(out), L.stream_ptr()))
        return out

    def noise(self, seed, draw, n_modes, n_samples, mode_offset=0):
        """The eps the kernels draw for (seed, draw, global particles mode_offset .. + n_modes, samples 0 .. n_samples) in
        torch's randn(n_samples, n_modes, T*d) layout (include/sgpmp.h: sgpmp_noise)."""
        out = torch.empty(n_samples, n_modes, self.T * self.d, **self.tensor_args)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_noise(self._ctx, int(seed), int(draw), int(n_modes), int(mode_offset), int(n_samples),
                                         L.ptr(out), L.stream_ptr()))
        return out

    def cost_eval(self, trajs, batch_offset=0, spheres=None, is_weights=None, rows_per_particle=1,
                  out=None, out64=None):
        self._chk(trajs, "trajs")
        B = trajs.numel() // (self.T * self.d)
        if out is None and out64 is None:
            out = torch.empty(B, **self.tensor_args)
        n_sph = 0
        if spheres is not None:
            spheres = spheres.reshape(-1, 4)
            self._chk(spheres, "obstacle_spheres")
            n_sph = spheres.shape[0]
        self._chk(is_weights, "is_weights")
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_cost_eval(self._ctx, L.ptr(trajs), B, batch_offset,
                                             L.ptr(spheres), n_sph, L.ptr(is_weights),
                                             rows_per_particle, L.ptr(out), L.ptr(out64),
                                             L.stream_ptr()))
        return out if out is not None else out64

    def is_weights(self, means, temperature, out=None):
        self._chk(means, "means")
        P = means.shape[0]
        if out is None:
            out = torch.empty(P, self.T + 1, self.d, **self.tensor_args)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_is_weights(self._ctx, L.ptr(means), P, float(temperature),
                                              L.ptr(out), L.stream_ptr()))
        return out

    def update(self, costs, samples, means, temperature, step_size, weights=None, grad=None,
               means_prev=None, stats=None):
        self._chk(samples, "samples")
        self._chk(means, "means")
        if costs.dtype == torch.float64:
            cd = L.SGPMP_F64
        else:
            cd = L.dtype_code(costs.dtype)
        assert costs.is_cuda and costs.is_contiguous()
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_update(self._ctx, L.ptr(costs), cd, L.ptr(samples), L.ptr(means),
                                          float(temperature), float(step_size), L.ptr(weights),
                                          L.ptr(grad), L.ptr(means_prev), L.ptr(stats),
                                          L.stream_ptr()))

    def step(self, seed, draw, means, samples, temperature, step_size, costs=None, weights=None,
             grad=None, means_prev=None, spheres=None, eps=None, eps_mode_offset=0, stats=None, flags=0):
        n_sph = 0
        if spheres is not None:
            n_sph = spheres.shape[0]
        eps_modes = 0 if eps is None else eps.shape[1]
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_step(self._ctx, int(seed), int(draw), L.ptr(eps), eps_modes,
                                        eps_mode_offset, L.ptr(means), L.ptr(samples), L.ptr(costs),
                                        L.ptr(weights), L.ptr(grad), L.ptr(means_prev),
                                        L.ptr(spheres), n_sph,
                                        float(temperature), float(step_size), L.ptr(stats), int(flags),
                                        L.stream_ptr()))

    def prepare_step(self, seed, means, samples, temperature, step_size, costs=None, weights=None,
                     grad=None, means_prev=None, spheres=None, stats=None):
        """Pre-bind every argument of sgpmp_step except the draw counter and the stream (the buffers
        are persistent, so their device pointers do not change): the returned callable costs one
        ctypes call per iteration, which matters for the small, launch-bound configurations."""
        fn, ctx = self.lib.sgpmp_step, self._ctx
        n_sph = 0 if spheres is None else spheres.shape[0]
        fixed = (L.ptr(means), L.ptr(samples), L.ptr(costs), L.ptr(weights), L.ptr(grad),
                 L.ptr(means_prev), L.ptr(spheres), n_sph, float(temperature), float(step_size),
                 L.ptr(stats))
        seed = int(seed)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

        def call(draw, flags=0, means_prev=None):
            """means_prev: another destination for this step's pre-update means (a raw pointer) than the pre-bound one."""
            if torch.cuda.current_device() != dev_index:
                torch.cuda.set_device(dev_index)
            args = fixed if means_prev is None else fixed[:5] + (means_prev,) + fixed[6:]
            L.check(fn(ctx, seed, draw, None, 0, 0, *args, flags, L.stream_ptr(dev_index)))
        return call

    def prepare_optimize(self, seed, means, samples, temperature, step_size, costs=None, weights=None, grad=None,
                         means_prev=None, spheres=None, stats_pair=None):
        """Pre-bind sgpmp_optimize (the whole K-loop of optimize() behind the C ABI: ONE ctypes call per optimize()) except
        for the iteration count, the draw counter, the statistics slot, the flags and the last step's means_prev tensor."""
        fn, ctx = self.lib.sgpmp_optimize, self._ctx
        n_sph = 0 if spheres is None else spheres.shape[0]
        head = (L.ptr(means), L.ptr(samples), L.ptr(costs), L.ptr(weights), L.ptr(grad))
        scratch = L.ptr(means_prev)
        tail = (L.ptr(spheres), n_sph, float(temperature), float(step_size), L.ptr(stats_pair))
        seed = int(seed)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

        def call(opt_iters, draw0, first_slot, flags, means_prev_last=None):
            if torch.cuda.current_device() != dev_index:
                torch.cuda.set_device(dev_index)
            L.check(fn(ctx, opt_iters, seed, draw0, *head, scratch, scratch if means_prev_last is None else means_prev_last,
                       *tail, first_slot, flags, L.stream_ptr(dev_index)))
        return call

    def fk(self, q):
        self._chk(q, "q")
        B = q.shape[0]
        out = torch.empty(B, self.n_links, 4, 4, **self.tensor_args)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_fk(self._ctx, L.ptr(q), B, L.ptr(out), L.stream_ptr()))
        return out

    def grid_lookup(self, term, xy):
        self._chk(xy, "X")
        B = xy.numel() // 2
        out = torch.empty(B, **self.tensor_args)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_grid_lookup(self._ctx, term, L.ptr(xy), B, L.ptr(out),
                                               L.stream_ptr()))
        return out

    def field_eval(self, term, frames, spheres=None):
        self._chk(frames, "link_tensor")
        n_links = frames.shape[-3]
        B = frames.numel() // (n_links * 16)
        out = torch.empty(B, **self.tensor_args)
        n_sph = 0
        if spheres is not None:
            spheres = spheres.reshape(-1, 4)
            self._chk(spheres, "obstacle_spheres")
            n_sph = spheres.shape[0]
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_field_eval(self._ctx, term, L.ptr(frames), B, n_links,
                                              L.ptr(spheres), n_sph, L.ptr(out), L.stream_ptr()))
        return out

    def link_distances(self, frames, spheres=None, mode=0, buffer=0.0):
        """fields.py:40-61,100-112 on frames [..,L,4,4]: mode 0 -> distances [B,L,O] (spheres) / [B,L,L] (self),
        mode 1 -> 1.0 where any distance < buffer, mode 2 -> summed distances (include/sgpmp.h: sgpmp_link_distances)."""
        self._chk(frames, "link_tensor")
        n_links = frames.shape[-3]
        B = frames.numel() // (n_links * 16)
        n_sph = 0
        if spheres is not None:
            spheres = spheres.reshape(-1, 4)
            self._chk(spheres, "obstacle_spheres")
            n_sph = spheres.shape[0]
        shape = (B, n_links, n_sph if spheres is not None else n_links) if mode == 0 else (B,)
        out = torch.empty(shape, **self.tensor_args)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_link_distances(self._ctx, L.ptr(frames), B, n_links, L.ptr(spheres), n_sph, int(mode),
                                                  float(buffer), L.ptr(out), L.stream_ptr()))
        return out

    def field_grad(self, term, q, spheres=None):
        """Value [B] and gradient [B,n] of link-field term `term` at joint configurations q [B,n]
        (analytic FK Jacobians in `field_grad_kernel`)."""
        self._chk(q, "q")
        assert q.shape[-1] == self.n
        B = q.numel() // self.n
        value = torch.empty(B, **self.tensor_args)
        grad = torch.empty(B, self.n, **self.tensor_args)
        n_sph = 0
        if spheres is not None:
            spheres = spheres.reshape(-1, 4)
            self._chk(spheres, "obstacle_spheres")
            n_sph = spheres.shape[0]
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_field_grad(self._ctx, term, L.ptr(q), B, L.ptr(spheres), n_sph,
                                              L.ptr(value), L.ptr(grad), L.stream_ptr()))
        return value, grad

    # ------------------------------------------------------------------ GPMP (Gauss-Newton planner)
    def gpmp_linearize(self, means, spheres=None, diag_sum=None):
        """Fields + Jacobians of the cost list at the particle means; optionally the local sum of the
        field part of diag(A^T K A) into `diag_sum` [T*d] (fp64) for the trust-region damping."""
        self._chk(means, "means")
        n_sph = 0
        if spheres is not None:
            spheres = spheres.reshape(-1, 4)
            self._chk(spheres, "obstacle_spheres")
            n_sph = spheres.shape[0]
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_gpmp_linearize(self._ctx, L.ptr(means), L.ptr(spheres), n_sph,
                                                  L.ptr(diag_sum), L.stream_ptr()))

    def gpmp_solve(self, means, delta, step_size, diag_sum=None, d_theta=None, costs=None):
        """Block-tridiagonal Gauss-Newton solve + in-place update of `means`; -> (d_theta, costs)."""
        self._chk(means, "means")
        if d_theta is None:
            d_theta = torch.empty_like(means)
        if costs is None:
            costs = torch.empty(means.shape[0], **self.tensor_args)
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_gpmp_solve(self._ctx, L.ptr(means), L.ptr(diag_sum), float(delta),
                                              float(step_size), L.ptr(d_theta), L.ptr(costs), L.stream_ptr()))
        return d_theta, costs

    # ------------------------------------------------------------------ profiling
    def profile_enable(self, on=True):
        L.check(self.lib.sgpmp_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self):
        ms = (C.c_double * 4)()
        n = C.c_int64()
        with torch.cuda.device(self.device):
            L.check(self.lib.sgpmp_profile_read(self._ctx, ms, C.byref(n)))
        return dict(zip(("is_weights", "sample", "cost_sweep", "update"), list(ms))), n.value
