"""Host-side mirror of the C-ABI in include/swf_solver.h (ctypes over libswf_hip.so).

Plumbing only: numpy arrays <-> C structs.  All arithmetic of a solve runs in the HIP
kernels; if the library or a GPU is missing every call raises — there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np

from .flat import FlatWindowC, OptionsC, SummaryC, default_options

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libswf_hip.so")
_pd = C.POINTER(C.c_double)
_lib = None


class SwfError(RuntimeError):
    pass


class TimingC(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("eval_ms", C.c_double), ("eliminate_ms", C.c_double),
                ("reduced_ms", C.c_double), ("other_ms", C.c_double), ("jacobian_bytes", C.c_int64),
                ("n_linearizations", C.c_int32), ("reserved", C.c_int32)]


EXPORTED = [
    "swf_version", "swf_device_count", "swf_set_device", "swf_last_error",
    "swf_batch_create", "swf_batch_destroy", "swf_batch_upload_state", "swf_batch_reset_state",
    "swf_batch_solve", "swf_batch_sync", "swf_batch_download_state", "swf_batch_summaries",
    "swf_batch_export_reduced", "swf_batch_export_vectors", "swf_batch_dims",
    "swf_batch_enable_timing", "swf_batch_timing",
    "swf_problem_create", "swf_problem_destroy", "swf_add_parameter_block", "swf_has_parameter_block",
    "swf_remove_parameter_block", "swf_set_parameter_block_constant", "swf_set_parameter_block_variable",
    "swf_is_parameter_block_constant", "swf_parameter_block_size", "swf_num_parameter_blocks",
    "swf_num_residual_blocks", "swf_add_projection", "swf_add_imu", "swf_add_rtk_carrier_phase",
    "swf_add_rtk_pseudorange", "swf_add_doppler", "swf_add_scalar_prior", "swf_add_linear_prior",
    "swf_remove_factor", "swf_factor_set_enabled", "swf_set_constants", "swf_set_ordering",
    "swf_set_export_tail", "swf_problem_solve", "swf_get_reduced",
]


def lib():
    """Load libswf_hip.so (must have been built by build.py / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwfError("libswf_hip.so not built: run `python -m rtk_visual_inertial_navigation_amd.build` "
                           "(no CPU fallback exists)")
        _lib = C.CDLL(LIB_PATH)
        _lib.swf_last_error.restype = C.c_char_p
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise SwfError("%s failed (%d): %s" % (what, rc, lib().swf_last_error().decode()))


def device_count():
    n = C.c_int32()
    rc = lib().swf_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def set_device(d):
    _chk(lib().swf_set_device(C.c_int32(d)), "swf_set_device")


class BatchSolver:
    """A batch of flat windows resident on the current HIP device (swf_batch_*)."""

    def __init__(self, windows, stream=None):
        self.windows = list(windows)
        self._structs = [w.c_struct() for w in self.windows]
        arr = (C.POINTER(FlatWindowC) * len(self._structs))(*[C.pointer(s) for s in self._structs])
        self._h = C.c_void_p()
        _chk(lib().swf_batch_create(arr, C.c_int32(len(self._structs)), C.c_void_p(stream or 0), C.byref(self._h)),
             "swf_batch_create")
        self.n = len(self.windows)

    def close(self):
        if self._h:
            lib().swf_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_state(self):
        _chk(lib().swf_batch_upload_state(self._h), "swf_batch_upload_state")

    def reset_state(self):
        _chk(lib().swf_batch_reset_state(self._h), "swf_batch_reset_state")

    def solve_async(self, opt=None):
        self._opt = opt if opt is not None else default_options()
        _chk(lib().swf_batch_solve(self._h, C.byref(self._opt)), "swf_batch_solve")

    def sync(self):
        _chk(lib().swf_batch_sync(self._h), "swf_batch_sync")

    def solve(self, opt=None, download=True):
        self.solve_async(opt)
        self.sync()
        if download:
            self.download_state()
        return self.summaries()

    def download_state(self):
        _chk(lib().swf_batch_download_state(self._h), "swf_batch_download_state")

    def summaries(self):
        out = (SummaryC * self.n)()
        _chk(lib().swf_batch_summaries(self._h, out), "swf_batch_summaries")
        return list(out)

    def dims(self, w=0):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _chk(lib().swf_batch_dims(self._h, C.c_int32(w), C.byref(a), C.byref(b), C.byref(c)), "swf_batch_dims")
        return dict(n_loc=a.value, n_e=b.value, n_red=c.value)

    def export_reduced(self, w=0):
        n = self.dims(w)["n_red"]
        S, rhs, L = np.zeros((n, n)), np.zeros(n), np.zeros((n, n))
        _chk(lib().swf_batch_export_reduced(self._h, C.c_int32(w), S.ctypes.data_as(_pd), rhs.ctypes.data_as(_pd),
                                            L.ctypes.data_as(_pd)), "swf_batch_export_reduced")
        return S, rhs, L

    def export_vectors(self, w=0):
        n = self.dims(w)["n_loc"]
        g, d, y = np.zeros(n), np.zeros(n), np.zeros(n)
        _chk(lib().swf_batch_export_vectors(self._h, C.c_int32(w), g.ctypes.data_as(_pd), d.ctypes.data_as(_pd),
                                            y.ctypes.data_as(_pd)), "swf_batch_export_vectors")
        return g, d, y

    def enable_timing(self, on=True):
        _chk(lib().swf_batch_enable_timing(self._h, C.c_int32(1 if on else 0)), "swf_batch_enable_timing")

    def timing(self):
        t = TimingC()
        _chk(lib().swf_batch_timing(self._h, C.byref(t)), "swf_batch_timing")
        return {k: getattr(t, k) for k, _ in TimingC._fields_}
