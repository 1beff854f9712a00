"""ctypes binding of librome_mi355.so -- one Python signature per symbol of include/rome_mi355.h.

There is no CPU fallback: if the shared library is missing, or no HIP device is present when a
Context is created, this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("ROME_MI355_LIB") or os.path.join(HERE, "librome_mi355.so")  # override: kernel A/B experiments

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_POSDEF, ERR_UNSUPPORTED_N, ERR_ALLOC = -1, -2, -3, -4, -5, -6
SOLVER_CLOSED_FORM, SOLVER_NEWTON, SOLVER_NELDER_MEAD, SOLVER_GAUSS_NEWTON = 0, 1, 2, 3
NOISE_STANDARD_NORMALS, NOISE_MEASUREMENTS = 0, 1
LAYOUT_SOA, LAYOUT_AOS, LAYOUT_AOS_POINTS = 0, 1, 2
MAX_PARTICLES = 4096            # ROME_MAX_PARTICLES (the register-resident kernels: MAX_PARTICLES_REGISTER)
MAX_PARTICLES_REGISTER = 512
MAX_PARTICLES_KDE, MAX_PARTICLES_PRODUCT, MAX_PARTICLES_PRODUCT_POSE3, MAX_PARTICLES_GIBBS = 512, 512, 256, 256   # per-stage limits (header)
FACTOR_PRIORPOSE2, FACTOR_POSE2POSE2, FACTOR_POSE2POINT2BR, FACTOR_PRIORPOINT2, FACTOR_POSE3POSE3, FACTOR_PRIORPOSE3 = range(6)


class RomeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librome_mi355: %s (code %d)" % (msg, code))
        self.code = code


class Opts(C.Structure):
    _fields_ = [("n_particles", C.c_int32), ("solver", C.c_int32), ("max_iters", C.c_int32),
                ("inflate_cycles", C.c_int32), ("tol", C.c_double), ("inflation", C.c_double),
                ("seed", C.c_uint64), ("stream_offset", C.c_uint64), ("layout", C.c_int32),
                ("presampled", C.c_int32), ("spread_nh", C.c_double), ("nullhypo", C.c_double)]


class ConvDev(C.Structure):
    _fields_ = [("n_conv", C.c_int32), ("dir_all", C.c_int32),
                ("factor", C.c_void_p), ("dir", C.c_void_p), ("fixed_var", C.c_void_p), ("target_var", C.c_void_p),
                ("mu", C.c_void_p), ("L", C.c_void_p), ("bel_fixed", C.c_void_p), ("bel_target", C.c_void_p),
                ("noise", C.c_void_p), ("out", C.c_void_p), ("status", C.c_void_p),
                ("n_mirror", C.c_int32), ("mirror_row", C.c_int32 * 4), ("reserved", C.c_int32), ("mirror_out", C.c_void_p),
                ("alt_var", C.c_void_p), ("hypo_w", C.c_void_p), ("nullhypo", C.c_void_p), ("rows4", C.c_void_p),
                ("mirror_map", C.c_void_p)]


_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)
_CTX = C.c_void_p
_PO = C.POINTER(Opts)
_PT = C.POINTER(ConvDev)

# symbol -> (restype, argtypes): EXACTLY the declarations of include/rome_mi355.h
SIGNATURES = {
    "rome_version": (C.c_int, []),
    "rome_strerror": (C.c_char_p, [C.c_int]),
    "rome_last_hip_error": (C.c_int, [_CTX]),
    "rome_last_hip_error_string": (C.c_char_p, [_CTX]),
    "rome_opts_default": (None, [_PO, C.c_int32]),
    "rome_ctx_create": (C.c_int, [C.POINTER(_CTX), C.c_int]),
    "rome_ctx_destroy": (None, [_CTX]),
    "rome_ctx_set_stream": (C.c_int, [_CTX, C.c_void_p]),
    "rome_ctx_use_own_stream": (C.c_int, [_CTX]),
    "rome_ctx_synchronize": (C.c_int, [_CTX]),
    "rome_device_count": (C.c_int, []),
    "rome_points_to_coords": (C.c_int, [_CTX, C.c_int32, C.c_int32, _PD, _PD]),
    "rome_coords_to_points": (C.c_int, [_CTX, C.c_int32, C.c_int32, _PD, _PD]),
    "rome_cholesky_lower": (C.c_int, [C.c_int32, C.c_int32, _PD, _PD]),
    "rome_residual_pose2pose2": (C.c_int, [_CTX, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_residual_priorpose2": (C.c_int, [_CTX, C.c_int32, _PD, _PD, _PD]),
    "rome_residual_pose2point2br": (C.c_int, [_CTX, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_residual_pose2point2br_pt": (C.c_int, [_CTX, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_residual_pose3pose3": (C.c_int, [_CTX, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_residual_pose3pose3_pt": (C.c_int, [_CTX, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_residual_priorpose3": (C.c_int, [_CTX, C.c_int32, _PD, _PD, _PD]),
    "rome_conv_pose2pose2": (C.c_int, [_CTX, _PO, C.c_int32, _PI, _PD, _PD, _PD, _PD, _PD, _PI]),
    "rome_conv_pose2point2br": (C.c_int, [_CTX, _PO, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD, _PD, _PI]),
    "rome_conv_pose2point2br_mh": (C.c_int, [_CTX, _PO, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PI]),
    "rome_conv_pose2pose2_mh": (C.c_int, [_CTX, _PO, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PI]),
    "rome_conv_pose3pose3": (C.c_int, [_CTX, _PO, C.c_int32, _PI, _PD, _PD, _PD, _PD, _PD, _PI]),
    "rome_sample_priorpose2": (C.c_int, [_CTX, _PO, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_sample_priorpose3": (C.c_int, [_CTX, _PO, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_sample_priorpoint2": (C.c_int, [_CTX, _PO, C.c_int32, _PD, _PD, _PD, _PD]),
    "rome_conv_pose2pose2_dev": (C.c_int, [_CTX, _PO, _PT]),
    "rome_conv_pose2point2br_dev": (C.c_int, [_CTX, _PO, _PT]),
    "rome_conv_pose3pose3_dev": (C.c_int, [_CTX, _PO, _PT]),
    "rome_sweep_pose2_dev": (C.c_int, [_CTX, _PO, _PT, _PT, _PT, C.POINTER(C.c_uint64)]),
    "rome_sample_priorpose2_dev": (C.c_int, [_CTX, _PO, _PT]),
    "rome_sample_priorpose3_dev": (C.c_int, [_CTX, _PO, _PT]),
    "rome_sample_priorpoint2_dev": (C.c_int, [_CTX, _PO, _PT]),
    "rome_linearize": (C.c_int, [_CTX, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD, _PD, _PD, _PD]),
    "rome_linearize_dev": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rome_belief_stats_dev": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rome_belief_stats": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, _PD, _PD, _PD]),
    "rome_kde_bandwidth_dev": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_void_p]),
    "rome_kde_bandwidth": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, _PD, C.c_uint32, C.c_double, C.c_double, _PD]),
    "rome_kde_max_dev": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "rome_kde_max": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, _PD, _PD, C.c_int32, _PD]),
    "rome_product_bw_dev": (C.c_int, [_CTX, _PO, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rome_product_dev": (C.c_int, [_CTX, _PO, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rome_clique_proposals": (C.c_int, [_CTX, _PO, C.c_void_p]),
    "rome_clique_upsolve": (C.c_int, [_CTX, _PO, C.c_void_p]),
    "rome_store_create": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "rome_store_wrap": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rome_store_destroy": (None, [C.c_void_p]),
    "rome_store_upload": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _PD]),
    "rome_store_download": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _PD]),
    "rome_store_ptr": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), _PI]),
    "rome_upsolve_plan_create": (C.c_int, [_CTX, C.c_void_p, _PO, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rome_upsolve_plan_run": (C.c_int, [C.c_void_p, _PO, C.c_void_p, C.c_int64]),
    "rome_upsolve_plan_destroy": (None, [C.c_void_p]),
    "rome_blockop_plan_create": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32, _PI, _PI, _PI, _PI, C.POINTER(C.c_void_p)]),
    "rome_blockop_plan_create_ex": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32, _PI, _PI, _PI, _PI, _PD, C.POINTER(C.c_void_p)]),
    "rome_blockop_plan_run": (C.c_int, [C.c_void_p]),
    "rome_blockop_plan_destroy": (None, [C.c_void_p]),
    "rome_scatter_plan_create": (C.c_int, [_CTX, C.c_void_p, C.c_int32, _PI, _PI, _PI, C.c_int64, C.POINTER(C.c_void_p)]),
    "rome_scatter_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rome_scatter_plan_destroy": (None, [C.c_void_p]),
    "rome_product_gibbs_dev": (C.c_int, [_CTX, _PO, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]),
    "rome_dev_alloc": (C.c_int, [_CTX, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rome_dev_free": (C.c_int, [_CTX, C.c_void_p]),
    "rome_dev_upload": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rome_dev_download": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_uint64]),
}

_lib = None


def load():
    """dlopen librome_mi355.so and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch (when installed) bundles its own HIP runtime; it has to be the one this process maps
    # FIRST so that librome_mi355's libamdhip64 dependency resolves to the same runtime and device
    # pointers / streams can be shared.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(SO):
        raise RuntimeError(
            "librome_mi355.so is not built (%s). Run `python __graft_entry__.py` or "
            "`python rome.jl_amd/_build.py`; there is no CPU fallback." % SO)
    lib = C.CDLL(SO)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, ctx=None):
    if code == OK:
        return
    lib = load()
    msg = lib.rome_strerror(code).decode()
    if code == ERR_HIP and ctx is not None:
        msg += ": " + lib.rome_last_hip_error_string(ctx).decode()
    raise RomeError(code, msg)


def default_opts(solver=SOLVER_NEWTON, **kw):
    o = Opts()
    load().rome_opts_default(C.byref(o), solver)
    for k, v in kw.items():
        if v is not None:
            setattr(o, k, v)
    return o


class Context:
    """Owns a rome_ctx (device id + HIP stream + staging buffers)."""

    def __init__(self, device=0):
        self._lib = load()
        h = _CTX()
        check(self._lib.rome_ctx_create(C.byref(h), int(device)))
        self.handle = h
        self.device = int(device)

    def set_stream(self, hip_stream_ptr):
        check(self._lib.rome_ctx_set_stream(self.handle, C.c_void_p(hip_stream_ptr or 0)), self.handle)

    def use_own_stream(self):
        check(self._lib.rome_ctx_use_own_stream(self.handle), self.handle)

    def synchronize(self):
        check(self._lib.rome_ctx_synchronize(self.handle), self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rome_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
