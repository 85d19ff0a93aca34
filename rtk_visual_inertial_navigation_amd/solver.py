"""Host-side mirror of the C-ABI in include/swf_solver.h (ctypes over libswf_hip.so).

Plumbing only: numpy arrays <-> C structs.  All arithmetic of a solve runs in the HIP
kernels; if the library or a GPU is missing every call raises — there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np

from .flat import FlatWindowC, OptionsC, SummaryC, default_options

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWF_LIB", os.path.join(_HERE, "libswf_hip.so"))   # SWF_LIB: A/B builds for tuning
_pd = C.POINTER(C.c_double)
_lib = None


class SwfError(RuntimeError):
    pass


K_NAMES = ["total", "eval_ps", "eval_imu", "frame_sums", "eval_prior", "lm_schur", "clique_elim", "lm_elim",
           "assemble", "chol_solve", "post_chol", "post_dogleg", "dogleg", "cand_eval", "decide", "unused"]


class TimingC(C.Structure):
    _fields_ = [("ms", C.c_double * 16), ("calls", C.c_int32 * 16), ("jacobian_bytes", C.c_int64),
                ("proj_bytes", C.c_int64), ("chol_flops", C.c_int64), ("lm_schur_flops", C.c_int64), ("n_obs", C.c_int64),
                ("n_linearizations", C.c_int32), ("reserved", C.c_int32),
                ("lm_schur_flops_sym", C.c_int64), ("lm_schur_mfma", C.c_int64)]


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
    "swf_set_export_tail", "swf_problem_solve", "swf_get_reduced", "swf_problem_marginalize",
    "swf_batch_marginalize", "swf_batch_get_prior",
    "swf_add_spp_pseudorange", "swf_add_spp_carrier_phase", "swf_add_fixed_integer",
    "swf_preintegrate_batch", "swf_triangulate_batch",
    "swf_batch_tail_covariance", "swf_batch_get_tail_covariance", "swf_problem_tail_covariance",
    "swf_composite_create", "swf_composite_evaluate", "swf_composite_hidden", "swf_composite_destroy", "swf_add_imu_gnss",
    "swf_eval_inverse_depth_batch", "swf_add_projection_inverse_depth",
    "swf_factor_is_enabled", "swf_get_residual_blocks", "swf_get_residual_blocks_for_parameter_block",
    "swf_get_parameter_blocks", "swf_get_parameter_blocks_for_residual_block", "swf_batch_export_jacobian",
    "swf_batch_marginal_priors", "swf_composite_assemble",
    "swf_composite_set_mid_links", "swf_composite_add_mid_prior", "swf_set_imu_gnss_mid_link", "swf_composite_set_root",
    "swf_batch_create_on", "swf_batch_create_sharded", "swf_solve_batches", "swf_batch_device", "swf_default_options", "swf_shard_partition",
    "swf_prior_reset_linearization_point",
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
    """A batch of flat windows resident on the current HIP device (swf_batch_*), or on `device` (swf_batch_create_on)."""

    def __init__(self, windows, stream=None, device=None, _handle=None):
        self.windows = list(windows)
        self.n = len(self.windows)
        if _handle is not None:                      # adopted from swf_batch_create_sharded
            self._structs = None
            self._h = _handle
            return
        self._structs = [w.c_struct() for w in self.windows]
        arr = (C.POINTER(FlatWindowC) * len(self._structs))(*[C.pointer(s) for s in self._structs])
        self._h = C.c_void_p()
        if device is None:
            _chk(lib().swf_batch_create(arr, C.c_int32(len(self._structs)), C.c_void_p(stream or 0), C.byref(self._h)), "swf_batch_create")
        else:
            _chk(lib().swf_batch_create_on(C.c_int32(device), arr, C.c_int32(len(self._structs)), C.c_void_p(stream or 0), C.byref(self._h)),
                 "swf_batch_create_on")

    def device(self):
        d = C.c_int32(-1)
        _chk(lib().swf_batch_device(self._h, C.byref(d)), "swf_batch_device")
        return d.value

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

    def export_jacobian(self, w=0):
        """(r [n_res], J [n_res][n_loc]) of window w's last linearisation as the device holds it (swf_batch_export_jacobian)."""
        nr, nl = C.c_int32(), C.c_int32()
        _chk(lib().swf_batch_export_jacobian(self._h, C.c_int32(w), None, None, C.byref(nr), C.byref(nl)), "swf_batch_export_jacobian")
        r, J = np.zeros(nr.value), np.zeros((nr.value, nl.value))
        _chk(lib().swf_batch_export_jacobian(self._h, C.c_int32(w), r.ctypes.data_as(_pd), J.ctypes.data_as(_pd), C.byref(nr), C.byref(nl)),
             "swf_batch_export_jacobian")
        return r, J

    PRIOR_EIGEN, PRIOR_CHOLESKY = 0, 1

    def marginalize(self, eps=1e-8, form=0):
        """UpdateSchur + setmarginalizeinfo(Sqrt=true) for every window, on the device, from the last
        ASSEMBLE_ELIMINATE_ONLY solve (R/swf/swf_gnss.cpp:25-61, R/factor/marginalization_factor.cpp:449-488)."""
        _chk(lib().swf_batch_marginalize(self._h, C.c_double(eps), C.c_int32(form)), "swf_batch_marginalize")

    def get_prior(self, w=0):
        n, rank = C.c_int32(0), C.c_int32(0)
        _chk(lib().swf_batch_get_prior(self._h, C.c_int32(w), None, None, None, None, None, C.byref(n), C.byref(rank)), "swf_batch_get_prior")
        k = n.value
        A, J, b, r0, eig = np.zeros((k, k)), np.zeros((k, k)), np.zeros(k), np.zeros(k), np.zeros(k)
        _chk(lib().swf_batch_get_prior(self._h, C.c_int32(w), A.ctypes.data_as(_pd), b.ctypes.data_as(_pd), J.ctypes.data_as(_pd),
                                       r0.ctypes.data_as(_pd), eig.ctypes.data_as(_pd), C.byref(n), C.byref(rank)), "swf_batch_get_prior")
        return dict(A=A, b=b, J=J, r0=r0, eig=eig, n=k, rank=rank.value)

    def tail_covariance(self, w=None):
        """UpdateSchurHessianOnly + LambdaSearch's covariance (R/swf/swf_gnss.cpp:65-94, swf_lambda.cpp:94-99): information A and
        covariance Qy = A^-1 of the parameter_head states of every window, from the factor of the last linear solve.
        w = None: list over windows; else one window's dict(A, Qy, n)."""
        _chk(lib().swf_batch_tail_covariance(self._h), "swf_batch_tail_covariance")

        def one(i):
            n = C.c_int32(0)
            _chk(lib().swf_batch_get_tail_covariance(self._h, C.c_int32(i), None, None, C.byref(n)), "swf_batch_get_tail_covariance")
            k = n.value
            A, Q = np.zeros((k, k)), np.zeros((k, k))
            _chk(lib().swf_batch_get_tail_covariance(self._h, C.c_int32(i), A.ctypes.data_as(_pd), Q.ctypes.data_as(_pd), C.byref(n)),
                 "swf_batch_get_tail_covariance")
            return dict(A=A, Qy=Q, n=k)
        return [one(i) for i in range(self.n)] if w is None else one(w)

    def enable_timing(self, mask=1):
        """mask: bit k brackets kernel K_NAMES[k] with a HIP event pair per launch (bit 0 = whole solve);
        True = everything."""
        if mask is True:
            mask = 0xffff
        _chk(lib().swf_batch_enable_timing(self._h, C.c_int32(int(mask))), "swf_batch_enable_timing")

    def timing(self):
        t = TimingC()
        _chk(lib().swf_batch_timing(self._h, C.byref(t)), "swf_batch_timing")
        d = dict(jacobian_bytes=t.jacobian_bytes, proj_bytes=t.proj_bytes, chol_flops=t.chol_flops,
                 lm_schur_flops=t.lm_schur_flops, lm_schur_flops_sym=t.lm_schur_flops_sym, lm_schur_mfma=t.lm_schur_mfma, n_obs=t.n_obs,
                 n_linearizations=t.n_linearizations, total_ms=t.ms[0])
        d["kernels"] = {K_NAMES[k]: dict(ms=t.ms[k], calls=t.calls[k]) for k in range(16) if t.calls[k]}
        return d


class ShardedBatchSolver:
    """Windows sharded over the GPUs of one node from ONE process (swf_batch_create_sharded + swf_solve_batches): contiguous, near-equal
    blocks per device of `device_mask` (0 = every visible device), no data-path collective, one host thread driving every device."""

    def __init__(self, windows, device_mask=0):
        self.windows = list(windows)
        self._structs = [w.c_struct() for w in self.windows]
        n = len(self._structs)
        arr = (C.POINTER(FlatWindowC) * n)(*[C.pointer(s) for s in self._structs])
        cap = max(1, device_count())
        hs = (C.c_void_p * cap)(); first = (C.c_int32 * cap)(); count = (C.c_int32 * cap)(); nb = C.c_int32(0)
        _chk(lib().swf_batch_create_sharded(arr, C.c_int32(n), C.c_uint32(device_mask), hs, first, count, C.byref(nb)), "swf_batch_create_sharded")
        self.parts = [(first[k], count[k]) for k in range(nb.value)]
        self.batches = [BatchSolver(self.windows[first[k]:first[k] + count[k]], _handle=C.c_void_p(hs[k])) for k in range(nb.value)]

    def solve(self, opt=None, download=True):
        self._opt = opt if opt is not None else default_options()
        hs = (C.c_void_p * len(self.batches))(*[b._h for b in self.batches])
        _chk(lib().swf_solve_batches(hs, C.c_int32(len(self.batches)), C.byref(self._opt)), "swf_solve_batches")
        out = []
        for b in self.batches:
            if download:
                b.download_state()
            out += b.summaries()
        return out

    def close(self):
        for b in self.batches:
            b.close()
        self.batches = []


class Problem:
    """ceres::Problem-shaped host mirror (swf_problem_*).  Parameter blocks are numpy float64
    arrays owned by the caller and identified by address, as in Ceres; Solve() reads them and
    writes the result back in place.  Method names follow the ceres::Problem members the
    reference uses."""

    def __init__(self):
        self._h = C.c_void_p()
        _chk(lib().swf_problem_create(C.byref(self._h)), "swf_problem_create")
        self._keep = {}          # address -> array (keeps caller arrays alive)

    def close(self):
        if self._h:
            lib().swf_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _p(self, arr):
        assert isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags["C_CONTIGUOUS"]
        self._keep[arr.ctypes.data] = arr
        return arr.ctypes.data_as(_pd)

    @staticmethod
    def _d(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data_as(_pd)

    # --- ceres::Problem members
    def AddParameterBlock(self, arr, size, pose_manifold=False):
        _chk(lib().swf_add_parameter_block(self._h, self._p(arr), C.c_int32(size), C.c_int32(1 if pose_manifold else 0)), "AddParameterBlock")

    def HasParameterBlock(self, arr):
        return bool(lib().swf_has_parameter_block(self._h, arr.ctypes.data_as(_pd)))

    def RemoveParameterBlock(self, arr):
        _chk(lib().swf_remove_parameter_block(self._h, arr.ctypes.data_as(_pd)), "RemoveParameterBlock")

    def SetParameterBlockConstant(self, arr):
        _chk(lib().swf_set_parameter_block_constant(self._h, arr.ctypes.data_as(_pd)), "SetParameterBlockConstant")

    def SetParameterBlockVariable(self, arr):
        _chk(lib().swf_set_parameter_block_variable(self._h, arr.ctypes.data_as(_pd)), "SetParameterBlockVariable")

    def IsParameterBlockConstant(self, arr):
        return bool(lib().swf_is_parameter_block_constant(self._h, arr.ctypes.data_as(_pd)))

    def ParameterBlockSize(self, arr):
        return lib().swf_parameter_block_size(self._h, arr.ctypes.data_as(_pd))

    def NumParameterBlocks(self):
        return lib().swf_num_parameter_blocks(self._h)

    def NumResidualBlocks(self):
        return lib().swf_num_residual_blocks(self._h)

    def RemoveResidualBlock(self, fid):
        _chk(lib().swf_remove_factor(self._h, C.c_int32(fid)), "RemoveResidualBlock")

    def SetResidualBlockUsed(self, fid, on):          # ResidualBlock::is_use of the modified Ceres
        _chk(lib().swf_factor_set_enabled(self._h, C.c_int32(fid), C.c_int32(1 if on else 0)), "is_use")

    def IsResidualBlockUsed(self, fid):
        rc = lib().swf_factor_is_enabled(self._h, C.c_int32(fid))
        if rc < 0:
            raise SwfError("is_use: unknown residual block %d" % fid)
        return bool(rc)

    # --- the query surface (GlobalMarge, R/swf/swf_image.cpp:350-367)
    def _ids(self, fn, what, *head):
        n = C.c_int32(0)
        _chk(fn(self._h, *head, None, C.c_int32(0), C.byref(n)), what)
        buf = (C.c_int32 * max(1, n.value))()
        _chk(fn(self._h, *head, buf, C.c_int32(n.value), C.byref(n)), what)
        return [int(buf[i]) for i in range(n.value)]

    def _blocks(self, fn, what, *head):
        n = C.c_int32(0)
        _chk(fn(self._h, *head, None, C.c_int32(0), C.byref(n)), what)
        buf = (_pd * max(1, n.value))()
        _chk(fn(self._h, *head, buf, C.c_int32(n.value), C.byref(n)), what)
        return [self._keep[C.cast(buf[i], C.c_void_p).value] for i in range(n.value)]

    def GetResidualBlocks(self):
        return self._ids(lib().swf_get_residual_blocks, "GetResidualBlocks")

    def GetResidualBlocksForParameterBlock(self, arr):
        return self._ids(lib().swf_get_residual_blocks_for_parameter_block, "GetResidualBlocksForParameterBlock", arr.ctypes.data_as(_pd))

    def GetParameterBlocks(self):
        return self._blocks(lib().swf_get_parameter_blocks, "GetParameterBlocks")

    def GetParameterBlocksForResidualBlock(self, fid):
        return self._blocks(lib().swf_get_parameter_blocks_for_residual_block, "GetParameterBlocksForResidualBlock", C.c_int32(fid))

    # --- typed AddResidualBlock()s
    def _fid(self, rc, what):
        if rc < 0:
            raise SwfError("%s failed (%d): %s" % (what, rc, lib().swf_last_error().decode()))
        return rc

    def AddProjection(self, pose, ex, point, uv, sqrt_info=1000.0 / 1.5, cauchy_a=1.0):
        a, pa = self._d(uv)
        return self._fid(lib().swf_add_projection(self._h, self._p(pose), self._p(ex), self._p(point), pa,
                                                  C.c_double(sqrt_info), C.c_double(cauchy_a)), "AddProjection")

    def AddImu(self, pose_i, sb_i, pose_j, sb_j, pre):
        a, pa = self._d(pre)
        return self._fid(lib().swf_add_imu(self._h, self._p(pose_i), self._p(sb_i), self._p(pose_j), self._p(sb_j), pa), "AddImu")

    def AddRtkCarrierPhase(self, pose, amb, clk, dat):
        a, pa = self._d(dat)
        return self._fid(lib().swf_add_rtk_carrier_phase(self._h, self._p(pose), self._p(amb), self._p(clk), pa), "AddRtkCarrierPhase")

    def AddRtkPseudorange(self, pose, clk, dat):
        a, pa = self._d(dat)
        return self._fid(lib().swf_add_rtk_pseudorange(self._h, self._p(pose), self._p(clk), pa), "AddRtkPseudorange")

    def AddDoppler(self, sb, drift, pose, dat):
        a, pa = self._d(dat)
        return self._fid(lib().swf_add_doppler(self._h, self._p(sb), self._p(drift), self._p(pose), pa), "AddDoppler")

    def AddSppPseudorange(self, pose, clk, dat):
        a, pa = self._d(dat)
        return self._fid(lib().swf_add_spp_pseudorange(self._h, self._p(pose), self._p(clk), pa), "AddSppPseudorange")

    def AddSppCarrierPhase(self, pose, clk, amb, dat):
        a, pa = self._d(dat)
        return self._fid(lib().swf_add_spp_carrier_phase(self._h, self._p(pose), self._p(clk), self._p(amb), pa), "AddSppCarrierPhase")

    def AddFixedInteger(self, n_a, n_b, N21, istd):
        return self._fid(lib().swf_add_fixed_integer(self._h, self._p(n_a), self._p(n_b), C.c_double(N21), C.c_double(istd)), "AddFixedInteger")

    def AddProjectionInverseDepth(self, kind, pose_i, pose_j, ex, ex2, inv_depth, pts_i, pts_j, sqrt_info, loss_a):
        pi = (C.c_double * 3)(*[float(v) for v in pts_i]); pj = (C.c_double * 3)(*[float(v) for v in pts_j])
        pp = lambda a: self._p(a) if a is not None else None
        return self._fid(lib().swf_add_projection_inverse_depth(self._h, C.c_int32(kind), pp(pose_i), pp(pose_j), pp(ex), pp(ex2), self._p(inv_depth), pi, pj,
                                                                 C.c_double(sqrt_info), C.c_double(loss_a)), "AddProjectionInverseDepth")

    def AddImuGnss(self, pose_i, sb_i, pose_j, sb_j, ambiguities, hidden_pose, hidden_sb, pose_lin, sb_lin, Hpp, HpN, rhs_p, HNN, rhsN, pre):
        """IMUGNSSFactor: hidden_pose [M][7] / hidden_sb [M][9] are caller-owned numpy arrays the solve updates in place."""
        N, M = len(ambiguities), int(np.asarray(hidden_pose).reshape(-1, 7).shape[0])
        assert hidden_pose.dtype == np.float64 and hidden_pose.flags["C_CONTIGUOUS"] and hidden_sb.dtype == np.float64 and hidden_sb.flags["C_CONTIGUOUS"]
        self._keep[hidden_pose.ctypes.data] = hidden_pose; self._keep[hidden_sb.ctypes.data] = hidden_sb
        keys = (_pd * max(1, N))(*[self._p(a) for a in ambiguities])
        arrs = [self._d(x) for x in (pose_lin, sb_lin, Hpp, HpN if N else np.zeros(1), rhs_p, HNN if N else np.zeros(1), rhsN if N else np.zeros(1), pre)]
        return self._fid(lib().swf_add_imu_gnss(self._h, self._p(pose_i), self._p(sb_i), self._p(pose_j), self._p(sb_j), keys, C.c_int32(N), C.c_int32(M),
                                                 hidden_pose.ctypes.data_as(_pd), hidden_sb.ctypes.data_as(_pd), *[a[1] for a in arrs]), "AddImuGnss")

    def SetImuGnssMidLink(self, fid, k, H12):
        """IMUGNSSBase::AddMidMargInfo's product: link k (1..M-1) of composite factor `fid` carries the cross block H12 (15x15); k = 0 clears."""
        a, pa = self._d(H12 if k else np.zeros(225))
        _chk(lib().swf_set_imu_gnss_mid_link(self._h, C.c_int32(fid), C.c_int32(k), pa), "SetImuGnssMidLink")

    def AddScalarPrior(self, scalar, w):
        return self._fid(lib().swf_add_scalar_prior(self._h, self._p(scalar), C.c_double(w)), "AddScalarPrior")

    def AddLinearPrior(self, blocks, J, r0, x0):
        keys = (_pd * len(blocks))(*[self._p(b) for b in blocks])
        aJ, pJ = self._d(J); ar, pr = self._d(r0); ax, px = self._d(x0)
        return self._fid(lib().swf_add_linear_prior(self._h, keys, C.c_int32(len(blocks)), pJ, pr, px), "AddLinearPrior")

    # --- window constants, ordering, parameter_head
    def SetConstants(self, pbg, gw, base):
        a, pa = self._d(pbg); b, pb = self._d(gw); c, pc = self._d(base)
        _chk(lib().swf_set_constants(self._h, pa, pb, pc), "SetConstants")

    def SetOrdering(self, blocks, groups):
        """blocks = [] asks for the automatic ordering (a null linear_solver_ordering)."""
        keys = (_pd * max(1, len(blocks)))(*[b.ctypes.data_as(_pd) for b in blocks])
        g = np.ascontiguousarray(groups, dtype=np.int32)
        _chk(lib().swf_set_ordering(self._h, keys, g.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(len(blocks))), "SetOrdering")

    def SetParameterHead(self, blocks):
        keys = (_pd * max(len(blocks), 1))(*[b.ctypes.data_as(_pd) for b in blocks])
        _chk(lib().swf_set_export_tail(self._h, keys, C.c_int32(len(blocks))), "SetParameterHead")

    # --- ceres::Solve
    def Solve(self, opt=None):
        opt = opt if opt is not None else default_options()
        sm = SummaryC()
        _chk(lib().swf_problem_solve(self._h, C.byref(opt), C.byref(sm)), "Solve")
        return sm

    def Marginalize(self, eps=1e-8, form=0):
        """UpdateSchur + setmarginalizeinfo(Sqrt=true) over the parameter_head blocks, after Solve with
        step_mode = ASSEMBLE_ELIMINATE_ONLY (R/swf/swf_image.cpp:404-418).  Returns dict(J, r0, A, b, n, rank)."""
        J, r0, A, b, n, rk = _pd(), _pd(), _pd(), _pd(), C.c_int32(), C.c_int32()
        _chk(lib().swf_problem_marginalize(self._h, C.c_double(eps), C.c_int32(form), C.byref(J), C.byref(r0), C.byref(A), C.byref(b),
                                           C.byref(n), C.byref(rk)), "Marginalize")
        n = n.value
        return dict(J=np.ctypeslib.as_array(J, (n, n)).copy(), r0=np.ctypeslib.as_array(r0, (n,)).copy(),
                    A=np.ctypeslib.as_array(A, (n, n)).copy(), b=np.ctypeslib.as_array(b, (n,)).copy(), n=n, rank=rk.value)

    def TailCovariance(self):
        """UpdateSchurHessianOnly (R/swf/swf_gnss.cpp:65-94) + Qy = A^-1 (R/swf/swf_lambda.cpp:94-99) after Solve."""
        A, Q, n = _pd(), _pd(), C.c_int32()
        _chk(lib().swf_problem_tail_covariance(self._h, C.byref(A), C.byref(Q), C.byref(n)), "TailCovariance")
        n = n.value
        return dict(A=np.ctypeslib.as_array(A, (n, n)).copy(), Qy=np.ctypeslib.as_array(Q, (n, n)).copy(), n=n)

    def GetReduced(self):
        S, r, L, n = _pd(), _pd(), _pd(), C.c_int32()
        _chk(lib().swf_get_reduced(self._h, C.byref(S), C.byref(r), C.byref(L), C.byref(n)), "GetReduced")
        n = n.value
        return (np.ctypeslib.as_array(S, (n, n)).copy(), np.ctypeslib.as_array(r, (n,)).copy(),
                np.ctypeslib.as_array(L, (n, n)).copy())


class CompositeBatch:
    """n composite IMU-GNSS factors (IMUGNSSBase, R/factor/gnss_imu_factor.cpp) evaluated together on the device.
    `factors`: list of dicts with pose [M][7], sb [M][9], pose_lin, sb_lin, Hpp [M][15][15], HpN [M][15][N], rhs_p [M][15],
    HNN [N][N], rhsN [N], pre [M+1][PRE_DOUBLES]; pbg, gw shared."""

    def __init__(self, factors, pbg, gw):
        cat = lambda k: np.ascontiguousarray(np.concatenate([np.asarray(f[k], np.float64).ravel() for f in factors]) if factors else np.zeros(0))
        self.M = np.array([np.asarray(f["pose"]).reshape(-1, 7).shape[0] for f in factors], np.int32)
        self.N = np.array([np.asarray(f["rhsN"]).size for f in factors], np.int32)
        self.n = len(factors)
        arrs = [cat(k) for k in ("pose", "sb", "pose_lin", "sb_lin", "Hpp", "HpN", "rhs_p", "HNN", "rhsN", "pre")]
        arrs = [a if a.size else np.zeros(1) for a in arrs]
        pb, g = np.ascontiguousarray(pbg, np.float64), np.ascontiguousarray(gw, np.float64)
        self._h = C.c_void_p()
        pi = C.POINTER(C.c_int32)
        _chk(lib().swf_composite_create(C.c_int32(self.n), self.M.ctypes.data_as(pi), self.N.ctypes.data_as(pi),
                                        *[a.ctypes.data_as(_pd) for a in arrs], pb.ctypes.data_as(_pd), g.ctypes.data_as(_pd), None,
                                        C.byref(self._h)), "swf_composite_create")
        self.G = 30 + self.N
        self.g_off = np.concatenate([[0], np.cumsum(self.G)]); self.g2_off = np.concatenate([[0], np.cumsum(self.G.astype(np.int64) ** 2)])

    def evaluate(self, outer, Nv, want_jac):
        """outer [n][32] (pose_i sb_i pose_j sb_j), Nv: list of per-factor ambiguity vectors.  Returns per-factor lists
        (residual, J, H, rhs, status) — J is None for a cost-only call."""
        o = np.ascontiguousarray(np.asarray(outer, np.float64).reshape(self.n, 32))
        nv = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64).ravel() for v in Nv]) if int(self.N.sum()) else np.zeros(1))
        res = np.zeros(int(self.g_off[-1])); jac = np.zeros(int(self.g2_off[-1])); Hd = np.zeros_like(jac); rd = np.zeros_like(res)
        st = np.zeros(self.n, np.int32)
        _chk(lib().swf_composite_evaluate(self._h, o.ctypes.data_as(_pd), nv.ctypes.data_as(_pd), C.c_int32(1 if want_jac else 0),
                                          res.ctypes.data_as(_pd), jac.ctypes.data_as(_pd), Hd.ctypes.data_as(_pd), rd.ctypes.data_as(_pd),
                                          st.ctypes.data_as(C.POINTER(C.c_int32))), "swf_composite_evaluate")
        out = []
        for f in range(self.n):
            G = int(self.G[f]); a, b = int(self.g_off[f]), int(self.g2_off[f])
            out.append(dict(r=res[a:a + G].copy(), J=jac[b:b + G * G].reshape(G, G).copy() if want_jac else None,
                            H=Hd[b:b + G * G].reshape(G, G).copy(), rhs=rd[a:a + G].copy(), status=int(st[f])))
        return out

    ROOT_PIVOTED_CHOLESKY, ROOT_EIGEN = 0, 1

    def set_root(self, form):
        """ROOT_EIGEN: the reference's eigen square root (rows in ascending eigenvalue order); default: pivoted Cholesky rows."""
        _chk(lib().swf_composite_set_root(self._h, C.c_int32(form)), "swf_composite_set_root")

    def set_mid_links(self, mid, H12):
        """IMUGNSSBase::AddMidMargInfo's product per factor: mid[f] = link 1..M-1 carrying the cross block H12[f] (15x15), 0 = none."""
        m = np.ascontiguousarray(mid, np.int32); h = np.ascontiguousarray(np.asarray(H12, np.float64).reshape(self.n, 225))
        _chk(lib().swf_composite_set_mid_links(self._h, m.ctypes.data_as(C.POINTER(C.c_int32)), h.ctypes.data_as(_pd)), "swf_composite_set_mid_links")

    def hidden(self):
        m = int(self.M.sum())
        pose, sb = np.zeros((m, 7)), np.zeros((m, 9))
        _chk(lib().swf_composite_hidden(self._h, pose.ctypes.data_as(_pd), sb.ctypes.data_as(_pd)), "swf_composite_hidden")
        e = np.concatenate([[0], np.cumsum(self.M)])
        return [(pose[e[f]:e[f + 1]], sb[e[f]:e[f + 1]]) for f in range(self.n)]

    def close(self):
        if self._h:
            lib().swf_composite_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def marginal_priors(windows, eps=1e-8, form=0, timing=None, allow_failed=False):
    """swf_batch_marginal_priors: the linear prior over each window's parameter_head tail with everything else eliminated
    (GnssPreprocess's per-epoch marginalize, R/swf/swf_gnss.cpp:504-532), for all windows in one batch on the device.
    Returns a list of dict(n, rank, A, b, J, r0).  A window whose marginal could not be formed comes back with rank -1 and zeros:
    that raises here unless allow_failed (such a prior must not be filed into a composite factor)."""
    structs = [w.c_struct() for w in windows]
    arr = (C.POINTER(FlatWindowC) * len(structs))(*[C.pointer(s) for s in structs])
    n = len(structs)
    dims = np.zeros(n, np.int32)
    pi = C.POINTER(C.c_int32)
    _chk(lib().swf_batch_marginal_priors(arr, C.c_int32(n), C.c_double(eps), C.c_int32(form), dims.ctypes.data_as(pi), None, None, None, None, None, None),
         "swf_batch_marginal_priors")
    n2, n1 = int((dims.astype(np.int64) ** 2).sum()), int(dims.sum())
    A, J, b, r0, ranks = np.zeros(max(n2, 1)), np.zeros(max(n2, 1)), np.zeros(max(n1, 1)), np.zeros(max(n1, 1)), np.zeros(n, np.int32)
    import time as _time
    t0 = _time.perf_counter()
    _chk(lib().swf_batch_marginal_priors(arr, C.c_int32(n), C.c_double(eps), C.c_int32(form), dims.ctypes.data_as(pi), ranks.ctypes.data_as(pi),
                                         A.ctypes.data_as(_pd), b.ctypes.data_as(_pd), J.ctypes.data_as(_pd), r0.ctypes.data_as(_pd), None), "swf_batch_marginal_priors")
    if timing is not None:
        timing["c_abi_call_s"] = _time.perf_counter() - t0          # the C-ABI call alone (structure upload, solve, consumer, download), without this wrapper's marshalling
    if not allow_failed and (ranks < 0).any():
        raise SwfError("swf_batch_marginal_priors: no marginal for window(s) %s (rank -1: failed elimination)" % np.nonzero(ranks < 0)[0].tolist())
    out, o2, o1 = [], 0, 0
    for i in range(n):
        d = int(dims[i])
        out.append(dict(n=d, rank=int(ranks[i]), A=A[o2:o2 + d * d].reshape(d, d).copy(), b=b[o1:o1 + d].copy(),
                        J=J[o2:o2 + d * d].reshape(d, d).copy(), r0=r0[o1:o1 + d].copy()))
        o2 += d * d; o1 += d
    return out


def composite_assemble(epochs):
    """swf_composite_assemble: IMUGNSSBase::AddMargInfo (R/factor/gnss_imu_factor.cpp:245-352) for a chain of epochs.
    epochs: list of dict(kept = [(size, key)] in prior order — key = the numpy array of a scalar block, anything for poses /
    speed-biases —, A, b).  Returns dict(N, keys (the scalar blocks in first-seen order), Hpp [M][15][15], HpN [M][15][N],
    rhs_p [M][15], HNN [N][N], rhsN [N])."""
    M = len(epochs)
    n_kept = np.array([len(e["kept"]) for e in epochs], np.int32)
    sizes = np.array([s for e in epochs for (s, _) in e["kept"]], np.int32)
    keep_alive, ptrs = {}, []
    for e in epochs:
        for (s, k) in e["kept"]:
            if s == 1:
                assert isinstance(k, np.ndarray) and k.dtype == np.float64
                keep_alive[k.ctypes.data] = k; ptrs.append(k.ctypes.data_as(_pd))
            else:
                ptrs.append(_pd())
    keys = (_pd * max(1, len(ptrs)))(*ptrs)
    A = np.ascontiguousarray(np.concatenate([np.asarray(e["A"], np.float64).ravel() for e in epochs]))
    b = np.ascontiguousarray(np.concatenate([np.asarray(e["b"], np.float64).ravel() for e in epochs]))
    pi = C.POINTER(C.c_int32)
    N = C.c_int32(0)
    _chk(lib().swf_composite_assemble(C.c_int32(M), n_kept.ctypes.data_as(pi), sizes.ctypes.data_as(pi), keys, A.ctypes.data_as(_pd), b.ctypes.data_as(_pd),
                                      C.c_int32(0), None, C.byref(N), None, None, None, None, None), "swf_composite_assemble")
    n = N.value
    Hpp, HpN, rhs_p, HNN, rhsN = np.zeros((M, 15, 15)), np.zeros((M, 15, max(n, 1))), np.zeros((M, 15)), np.zeros((max(n, 1), max(n, 1))), np.zeros(max(n, 1))
    HpN_buf = np.zeros(M * 15 * max(n, 1))
    nk = (_pd * max(1, n))()
    _chk(lib().swf_composite_assemble(C.c_int32(M), n_kept.ctypes.data_as(pi), sizes.ctypes.data_as(pi), keys, A.ctypes.data_as(_pd), b.ctypes.data_as(_pd),
                                      C.c_int32(n), nk, C.byref(N), Hpp.ctypes.data_as(_pd), HpN_buf.ctypes.data_as(_pd), rhs_p.ctypes.data_as(_pd),
                                      HNN.ctypes.data_as(_pd), rhsN.ctypes.data_as(_pd)), "swf_composite_assemble")
    HpN = HpN_buf[:M * 15 * n].reshape(M, 15, n) if n else np.zeros((M, 15, 0))
    return dict(N=n, keys=[keep_alive[C.cast(nk[i], C.c_void_p).value] for i in range(n)], Hpp=Hpp, HpN=HpN, rhs_p=rhs_p,
                HNN=HNN[:n, :n].copy(), rhsN=rhsN[:n].copy())


def prior_reset_linearization_point(blocks, sizes, J, A, r0, b, x0):
    """swf_prior_reset_linearization_point: MarginalizationInfo::ResetLinearizationPoint (R/factor/marginalization_factor.cpp:232-258).
    blocks = the kept blocks' current values (numpy arrays, kept order), sizes = their global sizes; J / A dim x dim or None.
    Returns the shifted (r0, b, x0) as new arrays (None where the pair was not given)."""
    sizes = np.ascontiguousarray(sizes, np.int32)
    arrs = [np.ascontiguousarray(np.asarray(x, np.float64).ravel()) for x in blocks]
    ptrs = (_pd * max(1, len(arrs)))(*[a.ctypes.data_as(_pd) for a in arrs])
    dim = int(sum(6 if sz == 7 else sz for sz in sizes))
    Jc = None if J is None else np.ascontiguousarray(J, np.float64); Ac = None if A is None else np.ascontiguousarray(A, np.float64)
    r = None if r0 is None else np.array(r0, np.float64); bb = None if b is None else np.array(b, np.float64)
    x = np.array(x0, np.float64)
    pn = lambda a: a.ctypes.data_as(_pd) if a is not None else None
    _chk(lib().swf_prior_reset_linearization_point(C.c_int32(len(arrs)), sizes.ctypes.data_as(C.POINTER(C.c_int32)), ptrs, C.c_int32(dim),
                                                   pn(Jc), pn(Ac), pn(r), pn(bb), pn(x)), "swf_prior_reset_linearization_point")
    return r, bb, x


def composite_add_mid_prior(fac, k, kept, A, b):
    """swf_composite_add_mid_prior: IMUGNSSBase::AddMidMargInfo (R/factor/gnss_imu_factor.cpp:121-240).  `fac` = what composite_assemble
    returned (N, keys, Hpp, HpN, rhs_p, HNN, rhsN); kept = [(size, epoch or key)] of the marginalised stretch's prior (A, b): poses (7) and
    speed-biases (9) carry the epoch index k-1 or k, scalars (1) their numpy block.  Returns the updated dict plus mid = k and H12."""
    M = fac["Hpp"].shape[0]; N0 = int(fac["N"])
    n_new = sum(1 for (sz, _) in kept if sz == 1)
    cap = N0 + n_new
    sizes = np.array([sz for (sz, _) in kept], np.int32)
    epochs = np.array([int(e) if sz != 1 else -1 for (sz, e) in kept], np.int32)
    keep_alive = {kk.ctypes.data: kk for kk in fac["keys"]}
    ptrs = []
    for (sz, kk) in kept:
        if sz == 1:
            assert isinstance(kk, np.ndarray) and kk.dtype == np.float64
            keep_alive[kk.ctypes.data] = kk; ptrs.append(kk.ctypes.data_as(_pd))
        else:
            ptrs.append(_pd())
    keys = (_pd * max(1, len(ptrs)))(*ptrs)
    nk = (_pd * max(1, cap))(*[kk.ctypes.data_as(_pd) for kk in fac["keys"]])
    Hpp = np.ascontiguousarray(fac["Hpp"], np.float64).copy(); rhs_p = np.ascontiguousarray(fac["rhs_p"], np.float64).copy()
    HpN = np.zeros((M, 15, max(cap, 1))); HpN[:, :, :N0] = fac["HpN"]
    HNN = np.zeros((max(cap, 1), max(cap, 1))); HNN[:N0, :N0] = fac["HNN"]
    rhsN = np.zeros(max(cap, 1)); rhsN[:N0] = fac["rhsN"]
    H12 = np.zeros((15, 15))
    A_ = np.ascontiguousarray(A, np.float64); b_ = np.ascontiguousarray(b, np.float64)
    N = C.c_int32(N0)
    pi = C.POINTER(C.c_int32)
    _chk(lib().swf_composite_add_mid_prior(C.c_int32(M), C.c_int32(k), C.c_int32(len(kept)), sizes.ctypes.data_as(pi), epochs.ctypes.data_as(pi), keys,
                                           A_.ctypes.data_as(_pd), b_.ctypes.data_as(_pd), C.c_int32(cap), nk, C.byref(N), Hpp.ctypes.data_as(_pd),
                                           HpN.ctypes.data_as(_pd), rhs_p.ctypes.data_as(_pd), HNN.ctypes.data_as(_pd), rhsN.ctypes.data_as(_pd),
                                           H12.ctypes.data_as(_pd)), "swf_composite_add_mid_prior")
    n = N.value
    return dict(N=n, keys=[keep_alive[C.cast(nk[i], C.c_void_p).value] for i in range(n)], Hpp=Hpp, HpN=HpN[:, :, :n].copy(), rhs_p=rhs_p,
                HNN=HNN[:n, :n].copy(), rhsN=rhsN[:n].copy(), mid=k, H12=H12)


def eval_inverse_depth_batch(kind, idx, poses, lam, pts, sqrt_info, pbg):
    """Inverse-depth projection factors (R/factor/projection_factor.cpp:77-329) for a batch on the device.
    kind [n] (0 TwoFrameOneCam, 1 TwoFrameTwoCam, 2 OneFrameTwoCam), idx [n][5] = pose_i, pose_j, ex, ex2, lambda rows,
    poses [P][7], lam [L], pts [n][6].  Returns r [n][2], J [n][50] (pose_i | pose_j | ex | ex2 | lambda)."""
    k = np.ascontiguousarray(kind, np.int32); ix = np.ascontiguousarray(idx, np.int32).reshape(-1, 5)
    ps = np.ascontiguousarray(poses, np.float64).reshape(-1, 7); lm = np.ascontiguousarray(lam, np.float64).ravel()
    pt = np.ascontiguousarray(pts, np.float64).reshape(-1, 6); pb = np.ascontiguousarray(pbg, np.float64)
    n = k.size
    r, J = np.zeros((n, 2)), np.zeros((n, 50))
    pi = C.POINTER(C.c_int32)
    _chk(lib().swf_eval_inverse_depth_batch(k.ctypes.data_as(pi), ix.ctypes.data_as(pi), C.c_int32(n), ps.ctypes.data_as(_pd), C.c_int32(ps.shape[0]),
                                            lm.ctypes.data_as(_pd), C.c_int32(lm.size), pt.ctypes.data_as(_pd), C.c_double(sqrt_info), pb.ctypes.data_as(_pd),
                                            r.ctypes.data_as(_pd), J.ctypes.data_as(_pd), C.c_int32(0), None), "eval_inverse_depth_batch")
    return r, J


def preintegrate_batch(samples, biases, noise):
    """IntegrationBase for a batch of keyframe intervals on the device (R/factor/integration_base.cpp:5-142).
    samples: list of [n_i][7] arrays (dt, acc, gyr; the first row seeds acc_0 / gyr_0); biases [n][6] (ba, bg);
    noise = (ACC_N, GYR_N, ACC_W, GYR_W).  Returns the [n][PRE_DOUBLES] records AddImu / imu_pre take."""
    n = len(samples)
    first = np.zeros(n + 1, np.int32)
    for i, s_ in enumerate(samples):
        first[i + 1] = first[i] + np.asarray(s_).reshape(-1, 7).shape[0]
    flat = (np.ascontiguousarray(np.concatenate([np.asarray(s_, np.float64).reshape(-1, 7) for s_ in samples]))
            if n and first[n] else np.zeros((1, 7)))
    b = np.ascontiguousarray(np.asarray(biases, np.float64).reshape(n, 6))
    nz = (C.c_double * 4)(*[float(v) for v in noise])
    out = np.zeros((n, 293))
    _chk(lib().swf_preintegrate_batch(flat.ctypes.data_as(_pd), first.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(n),
                                      b.ctypes.data_as(_pd), nz, out.ctypes.data_as(_pd), C.c_int32(0), None), "preintegrate_batch")
    return out


def triangulate_batch(Ps, Rs, tic, ric, pbg, start_frame, pt0, pt1, init_depth=5.0):
    """FeatureManager::triangulate (two-view branch, R/feature/feature_manager.cpp:285-316) for a batch of features on
    the device.  Ps [F][3], Rs [F][3][3]; start_frame [n]; pt0 / pt1 [n][2] normalised coordinates in frames i, i + 1.
    Returns (depth [n], pts_world [n][3])."""
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (Ps, Rs, tic, ric, pbg, pt0, pt1)]
    st = np.ascontiguousarray(start_frame, dtype=np.int32)
    n = st.size
    depth, world = np.zeros(n), np.zeros((n, 3))
    _chk(lib().swf_triangulate_batch(a[0].ctypes.data_as(_pd), a[1].ctypes.data_as(_pd), C.c_int32(a[0].size // 3),
                                     a[2].ctypes.data_as(_pd), a[3].ctypes.data_as(_pd), a[4].ctypes.data_as(_pd),
                                     st.ctypes.data_as(C.POINTER(C.c_int32)), a[5].ctypes.data_as(_pd), a[6].ctypes.data_as(_pd),
                                     C.c_int32(n), C.c_double(init_depth), depth.ctypes.data_as(_pd), world.ctypes.data_as(_pd),
                                     C.c_int32(0), None), "triangulate_batch")
    return depth, world


def problem_from_window(w):
    """Build a Problem from a FlatWindow the way the estimator would (AddParameterBlock /
    AddResidualBlock / SetParameterBlockConstant / ordering), returning (problem, blocks) where
    blocks[global_id] is the numpy parameter block."""
    P = Problem()
    a = w.a
    pose = [a["pose"].reshape(-1, 7)[i].copy() for i in range(w.n_pose)]
    sb = [a["sb"].reshape(-1, 9)[i].copy() for i in range(w.n_sb)]
    lm = [a["lm"].reshape(-1, 3)[i].copy() for i in range(w.n_lm)]
    sc = [a["sc"][i:i + 1].copy() for i in range(w.n_sc)]
    blocks = pose + sb + lm + sc
    P.SetConstants(w.pbg, w.gw, w.base)
    for b in pose:
        P.AddParameterBlock(b, 7, True)
    for p_, e_, l_, uv in zip(*a["proj_idx"].reshape(-1, 3).T, a["proj_uv"].reshape(-1, 2)):
        P.AddProjection(pose[p_], pose[e_], lm[l_], uv, w.proj_sqrt_info, w.proj_loss_a)
    for ix, pre in zip(a["imu_idx"].reshape(-1, 4), a["imu_pre"].reshape(-1, 293)):
        P.AddImu(pose[ix[0]], sb[ix[1]], pose[ix[2]], sb[ix[3]], pre)
    for ix, d in zip(a["cp_idx"].reshape(-1, 3), a["cp_dat"].reshape(-1, 9)):
        P.AddRtkCarrierPhase(pose[ix[0]], sc[ix[1]], sc[ix[2]], d)
    for ix, d in zip(a["pr_idx"].reshape(-1, 2), a["pr_dat"].reshape(-1, 7)):
        P.AddRtkPseudorange(pose[ix[0]], sc[ix[1]], d)
    for ix, d in zip(a["dop_idx"].reshape(-1, 3), a["dop_dat"].reshape(-1, 8)):
        P.AddDoppler(sb[ix[0]], sc[ix[1]], pose[ix[2]], d)
    for i, wv in zip(a["sp_idx"], a["sp_w"]):
        P.AddScalarPrior(sc[i], wv)
    for ix, d in zip(a["spr_idx"].reshape(-1, 2), a["spr_dat"].reshape(-1, 5)):
        P.AddSppPseudorange(pose[ix[0]], sc[ix[1]], d)
    for ix, d in zip(a["scp_idx"].reshape(-1, 3), a["scp_dat"].reshape(-1, 6)):
        P.AddSppCarrierPhase(pose[ix[0]], sc[ix[1]], sc[ix[2]], d)
    for ix, d in zip(a["fix_idx"].reshape(-1, 2), a["fix_dat"].reshape(-1, 2)):
        P.AddFixedInteger(sc[ix[0]], sc[ix[1]], d[0], d[1])
    for kd, ix, pt in zip(a["idp_kind"], a["idp_idx"].reshape(-1, 5), a["idp_pts"].reshape(-1, 6)):
        P.AddProjectionInverseDepth(int(kd), pose[ix[0]] if kd != 2 else None, pose[ix[1]] if kd != 2 else None, pose[ix[2]], pose[ix[3]] if kd != 0 else None,
                                    sc[ix[4]], pt[:3], pt[3:], w.proj_sqrt_info, w.proj_loss_a)
    hidden = []
    io = e0 = pn = nn = no = 0
    for k in range(a["comp_M"].size):
        M, N = int(a["comp_M"][k]), int(a["comp_N"][k])
        ix = a["comp_idx"][io:io + 4 + N]
        hp = a["comp_pose"].reshape(-1, 7)[e0:e0 + M].copy(); hs = a["comp_sb"].reshape(-1, 9)[e0:e0 + M].copy()
        hidden.append((hp, hs))
        fid = P.AddImuGnss(pose[ix[0]], sb[ix[1]], pose[ix[2]], sb[ix[3]], [sc[i] for i in ix[4:]], hp, hs,
                     a["comp_pose_lin"].reshape(-1, 7)[e0:e0 + M], a["comp_sb_lin"].reshape(-1, 9)[e0:e0 + M], a["comp_Hpp"].reshape(-1, 225)[e0:e0 + M],
                     a["comp_HpN"][pn:pn + 15 * M * N], a["comp_rhs_p"].reshape(-1, 15)[e0:e0 + M], a["comp_HNN"][nn:nn + N * N], a["comp_rhsN"][no:no + N],
                     a["comp_pre"].reshape(-1, 293)[e0 + k:e0 + k + M + 1])
        if a["comp_mid"].size and a["comp_mid"][k]:
            P.SetImuGnssMidLink(fid, int(a["comp_mid"][k]), a["comp_H12"].reshape(-1, 225)[k])
        io += 4 + N; e0 += M; pn += 15 * M * N; nn += N * N; no += N
    P.hidden = hidden
    bo = jo = ro = xo = 0
    for nb, dim in zip(a["prior_nblk"], a["prior_dim"]):
        ids = a["prior_blk"][bo:bo + nb]
        gs = sum(blocks[i].size for i in ids)
        P.AddLinearPrior([blocks[i] for i in ids], a["prior_J"].ravel()[jo:jo + dim * dim],
                         a["prior_r0"][ro:ro + dim], a["prior_x0"][xo:xo + gs])
        bo += nb; jo += dim * dim; ro += dim; xo += gs
    for i, c in enumerate(a["is_const"]):
        if c:
            P.SetParameterBlockConstant(blocks[i])
    P.SetOrdering([blocks[i] for i in a["order_block"]], a["order_group"])
    n_tail = w.n_tail
    P.SetParameterHead([blocks[i] for i in a["order_block"][len(a["order_block"]) - n_tail:]] if n_tail else [])
    return P, blocks
