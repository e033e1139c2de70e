"""ctypes binding of libsgpmp.so (the C ABI in include/sgpmp.h).

There is deliberately no fallback: if the HIP library is missing or fails to load, importing a
symbol from here raises, and every product entry point depends on it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGPMP_LIB_PATH: development override (A/B of two builds on one GPU box, tools/ab.sh); still a HIP build
LIB_PATH = os.environ.get("SGPMP_LIB_PATH") or os.path.join(_HERE, "libsgpmp.so")

SGPMP_F32, SGPMP_F64 = 0, 1
PRIOR_INIT, PRIOR_SAMPLE = 0, 1
COST_GP, COST_GOAL_PRIOR, COST_GRID, COST_SPHERES, COST_SELF, COST_EE_GOAL = 1, 2, 3, 4, 5, 6
FIELD_RBF, FIELD_SDF, FIELD_OCCUPANCY = 0, 1, 2
FLAG_GP_START, FLAG_SDF_CLAMP, FLAG_EE_SQUARE = 1, 16, 32
MAX_TERMS, MAX_JOINTS, MAX_DOF, MAX_INTERP = 8, 16, 8, 8
STAT_SHARDS = 64
OK, EINVAL, ENOTPD, EHIP, ESTATE = 0, -1, -2, -3, -4
STEP_MEANS_KEPT = 1
STEP_NO_SAMPLES = 2
OPT_PIPELINE = 4                     # include/sgpmp.h SGPMP_OPT_* (flags of sgpmp_optimize)
OPT_STORE_FREE = 8
ABI_VERSION = 6                      # include/sgpmp.h SGPMP_ABI_VERSION


class Dims(C.Structure):
    _fields_ = [("n_dof", C.c_int32), ("traj_len", C.c_int32), ("num_particles", C.c_int32),
                ("particle_offset", C.c_int32), ("num_particles_global", C.c_int32),
                ("num_samples", C.c_int32), ("num_goals", C.c_int32),
                ("num_particles_per_goal", C.c_int32), ("dtype", C.c_int32),
                ("reserved", C.c_int32)]


class CostDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32), ("sigma", C.c_double),
                ("sigma2", C.c_double), ("dt", C.c_double), ("data", C.c_void_p),
                ("dim0", C.c_int32), ("dim1", C.c_int32), ("p0", C.c_double), ("p1", C.c_double),
                ("p2", C.c_double), ("num_interpolate", C.c_int32), ("interp_lo", C.c_int32),
                ("interp_hi", C.c_int32), ("reserved", C.c_int32),
                ("alpha", C.c_double * MAX_INTERP)]


class Joint(C.Structure):
    _fields_ = [("rpy", C.c_double * 3), ("xyz", C.c_double * 3), ("revolute", C.c_int32),
                ("reserved", C.c_int32)]


# name -> (restype, argtypes); every symbol include/sgpmp.h declares
_P, _I, _I64, _U64, _D = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
SIGNATURES = {
    "sgpmp_abi_version": (_I, []),
    "sgpmp_philox_rounds": (_I, []),
    "sgpmp_last_error": (C.c_char_p, []),
    "sgpmp_create": (_I, [C.POINTER(Dims), C.POINTER(_P)]),
    "sgpmp_destroy": (None, [_P]),
    "sgpmp_set_option": (_I, [_P, C.c_char_p, C.c_longlong]),
    "sgpmp_comm_unique_id": (_I, [C.c_char_p]),
    "sgpmp_comm_init": (_I, [_P, C.c_char_p, _I, _I]),
    "sgpmp_comm_destroy": (_I, [_P]),
    "sgpmp_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "sgpmp_comm_library": (C.c_char_p, [C.POINTER(_I)]),
    "sgpmp_allreduce_stats": (_I, [_P, _P, _P]),
    "sgpmp_stats_wait": (_I, [_P, _P, _P]),
    "sgpmp_allgather_means": (_I, [_P, _P, _P, _P]),
    "sgpmp_mode_stats": (_I, [_P, _P, _P, _P]),
    "sgpmp_allreduce_f64": (_I, [_P, _P, _I64, _P]),
    "sgpmp_set_step_mode_stats": (_I, [_P, _P]),
    "sgpmp_mode_stats_wait": (_I, [_P, _P]),
    "sgpmp_last_cost_kernel": (C.c_char_p, [_P]),
    "sgpmp_pipeline_begin": (_I, [_P, _P]),
    "sgpmp_pipeline_end": (_I, [_P, _P]),
    "sgpmp_pipeline_split_steps": (C.c_longlong, [_P]),
    "sgpmp_last_step_launches": (_I, [_P]),
    "sgpmp_set_prior": (_I, [_P, _I, _D, _D, _D, _D, C.POINTER(_D), _P]),
    "sgpmp_set_priors": (_I, [_P, _D, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), _P]),
    "sgpmp_get_prior": (_I, [_P, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)]),
    "sgpmp_set_prior_blocks": (_I, [_P, _I, _I, C.POINTER(_D), C.POINTER(_D), _P]),
    "sgpmp_prior_quadform": (_I, [_P, _I, _P, _I64, _P, _I, _P, _P]),
    "sgpmp_set_costs": (_I, [_P, C.POINTER(CostDesc), _I]),
    "sgpmp_set_fk": (_I, [_P, C.POINTER(Joint), _I]),
    "sgpmp_set_fk_codegen": (_I, [_P, C.c_char_p]),
    "sgpmp_fk_codegen_compile": (_I, [C.c_char_p, _I, C.POINTER(_I64)]),
    "sgpmp_fk_codegen_info": (_I, [_P, C.POINTER(_I), C.POINTER(_D), C.POINTER(_I), C.POINTER(_I)]),
    "sgpmp_sample": (_I, [_P, _I, _U64, _U64, _P, _I, _I, _I, _P, _I, _I, _P, _P]),
    "sgpmp_noise": (_I, [_P, _U64, _U64, _I, _I, _I, _P, _P]),
    "sgpmp_cost_eval": (_I, [_P, _P, _I64, _I64, _P, _I, _P, _I, _P, _P, _P]),
    "sgpmp_is_weights": (_I, [_P, _P, _I, _D, _P, _P]),
    "sgpmp_update": (_I, [_P, _P, _I, _P, _P, _D, _D, _P, _P, _P, _P, _P]),
    "sgpmp_dense_particles": (_I, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "sgpmp_row_counts_get": (_I, [_P, C.POINTER(C.c_uint32)]),
    "sgpmp_row_counts_set": (_I, [_P, C.POINTER(C.c_uint32)]),
    "sgpmp_row_counts_clear": (_I, [_P, _P]),
    "sgpmp_store_free_steps": (C.c_longlong, [_P]),
    "sgpmp_multi_iteration_launches": (C.c_longlong, [_P]),
    "sgpmp_optimize": (_I, [_P, _I, _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _P, _I, _D, _D, _P, _I, _I, _P]),
    "sgpmp_step": (_I, [_P, _U64, _U64, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _D, _D, _P, _I, _P]),
    "sgpmp_fk": (_I, [_P, _P, _I64, _P, _P]),
    "sgpmp_grid_lookup": (_I, [_P, _I, _P, _I64, _P, _P]),
    "sgpmp_field_eval": (_I, [_P, _I, _P, _I64, _I, _P, _I, _P, _P]),
    "sgpmp_link_distances": (_I, [_P, _P, _I64, _I, _P, _I, _I, _D, _P, _P]),
    "sgpmp_field_grad": (_I, [_P, _I, _P, _I64, _P, _I, _P, _P, _P]),
    "sgpmp_gpmp_linearize": (_I, [_P, _P, _P, _I, _P, _P]),
    "sgpmp_gpmp_solve": (_I, [_P, _P, _P, C.c_double, C.c_double, _P, _P, _P]),
    "sgpmp_event_create": (_I, [C.POINTER(_P)]),
    "sgpmp_event_record": (_I, [_P, _P]),
    "sgpmp_event_elapsed_ms": (_I, [_P, _P, C.POINTER(C.c_float)]),
    "sgpmp_event_destroy": (_I, [_P]),
    "sgpmp_profile_enable": (_I, [_P, _I]),
    "sgpmp_profile_read": (_I, [_P, C.POINTER(_D), C.POINTER(_I64)]),
}

_lib = None


class SgpmpError(RuntimeError):
    pass


def load():
    """Load libsgpmp.so (once). Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C stoch_gpmp_amd/csrc). "
            "stoch_gpmp_amd has no CPU/torch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.sgpmp_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.sgpmp_abi_version()}, this binding needs {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def last_error():
    return load().sgpmp_last_error().decode()


def check(rc):
    """Map a status code to the exception type the reference raises for the same condition."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ENOTPD:
        # torch's MultivariateNormal raises ValueError (constraint PositiveDefinite) in the reference
        raise ValueError(msg)
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ESTATE:
        raise RuntimeError(msg)
    raise SgpmpError(msg)


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return SGPMP_F32
    if torch_dtype == torch.float64:
        return SGPMP_F64
    raise ValueError(f"stoch_gpmp_amd supports float32/float64 tensors, got {torch_dtype}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device_index=None):
    """The current HIP stream of `device_index` (default: the current device) as a raw handle.
    (torch._C._cuda_getCurrentRawStream where this torch has it: 0.3 us instead of the 7 us that building a
    torch.cuda.Stream object costs -- a third of the host time of a step at the launch-bound sizes.)"""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device() if device_index is None else device_index))
    return C.c_void_p(torch.cuda.current_stream(device_index).cuda_stream)


def require_cuda(tensor_args):
    """The product has no CPU path: fail loudly when asked to run anywhere but the GPU."""
    import torch
    dev = torch.device(tensor_args["device"])
    if dev.type != "cuda":
        raise RuntimeError(
            "stoch_gpmp_amd runs on the MI355X only (tensor_args['device'] must be a cuda/HIP "
            f"device, got {dev}); there is no CPU fallback")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible; stoch_gpmp_amd has no CPU fallback")
    return dev
