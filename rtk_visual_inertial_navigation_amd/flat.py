"""ctypes mirror of include/swf_types.h — the flat window format that crosses the C-ABI.

Host-side plumbing only: numpy arrays in, C structs out.  The layouts follow the
reference's parameter blocks (R/swf/swf.cpp:142-162) and factor constructor arguments
(R/factor/*.h); see include/swf_types.h for the per-field citations.
"""
import ctypes as C
import numpy as np

PRE_DOUBLES = 293
PRE = dict(DP=0, DQ=3, DV=7, LBA=10, LBG=13, DP_DBA=16, DP_DBG=25, DQ_DBG=34, DV_DBA=43,
           DV_DBG=52, SUMDT=61, GYRI=62, GYRJ=65, SQRTINFO=68)
CP_DOUBLES, PR_DOUBLES, DOP_DOUBLES = 9, 7, 8
MAX_TRACE = 64

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)


class FlatWindowC(C.Structure):
    _fields_ = [
        ("n_pose", C.c_int32), ("pose", _pd),
        ("n_sb", C.c_int32), ("sb", _pd),
        ("n_lm", C.c_int32), ("lm", _pd),
        ("n_sc", C.c_int32), ("sc", _pd),
        ("is_const", _pu8),
        ("n_order", C.c_int32), ("order_block", _pi), ("order_group", _pi), ("n_tail", C.c_int32),
        ("n_proj", C.c_int32), ("proj_idx", _pi), ("proj_uv", _pd),
        ("proj_sqrt_info", C.c_double), ("proj_loss_a", C.c_double),
        ("n_imu", C.c_int32), ("imu_idx", _pi), ("imu_pre", _pd),
        ("n_cp", C.c_int32), ("cp_idx", _pi), ("cp_dat", _pd),
        ("n_pr", C.c_int32), ("pr_idx", _pi), ("pr_dat", _pd),
        ("n_dop", C.c_int32), ("dop_idx", _pi), ("dop_dat", _pd),
        ("n_sp", C.c_int32), ("sp_idx", _pi), ("sp_w", _pd),
        ("n_spr", C.c_int32), ("spr_idx", _pi), ("spr_dat", _pd),
        ("n_scp", C.c_int32), ("scp_idx", _pi), ("scp_dat", _pd),
        ("n_fix", C.c_int32), ("fix_idx", _pi), ("fix_dat", _pd),
        ("n_idp", C.c_int32), ("idp_kind", _pi), ("idp_idx", _pi), ("idp_pts", _pd),
        ("n_comp", C.c_int32), ("comp_M", _pi), ("comp_N", _pi), ("comp_idx", _pi), ("comp_pose", _pd), ("comp_sb", _pd),
        ("comp_pose_lin", _pd), ("comp_sb_lin", _pd), ("comp_Hpp", _pd), ("comp_HpN", _pd), ("comp_rhs_p", _pd),
        ("comp_HNN", _pd), ("comp_rhsN", _pd), ("comp_pre", _pd), ("comp_mid", _pi), ("comp_H12", _pd),
        ("n_prior", C.c_int32), ("prior_nblk", _pi), ("prior_dim", _pi), ("prior_blk", _pi),
        ("prior_J", _pd), ("prior_r0", _pd), ("prior_x0", _pd),
        ("pbg", C.c_double * 3), ("gw", C.c_double * 3), ("base", C.c_double * 3),
    ]


class OptionsC(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("step_mode", C.c_int32),
        ("num_threads", C.c_int32), ("trust_region_strategy", C.c_int32),
        ("jacobi_scaling", C.c_int32), ("composite_root", C.c_int32),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("min_mu", C.c_double), ("max_mu", C.c_double), ("mu_increase_factor", C.c_double),
        ("min_diagonal", C.c_double), ("max_diagonal", C.c_double),
    ]


def default_options(max_num_iterations=8, step_mode=0, num_threads=1, strategy=0, jacobi_scaling=0, composite_root=0):
    """Solver::Options the reference sets (R/swf/swf.cpp:25-30) + Ceres 2.x defaults."""
    o = OptionsC()
    o.max_num_iterations = max_num_iterations
    o.step_mode = step_mode
    o.num_threads = num_threads
    o.trust_region_strategy = strategy          # 0 DOGLEG (R/swf/swf.cpp:26), 1 LEVENBERG_MARQUARDT (ceres default)
    o.jacobi_scaling = jacobi_scaling           # 0 what the reference's window solves set; 1 ceres default (LM only)
    o.composite_root = composite_root    # 0 pivoted Cholesky root, 1 the reference's eigen root (composite IMU-GNSS factors inside a solve)
    o.initial_trust_region_radius = 1e4
    o.max_trust_region_radius = 1e16
    o.min_trust_region_radius = 1e-32
    o.min_relative_decrease = 1e-3
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.min_mu, o.max_mu, o.mu_increase_factor = 1e-8, 1.0, 10.0
    o.min_diagonal, o.max_diagonal = 1e-6, 1e32
    return o


class IterationC(C.Structure):
    _fields_ = [
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double), ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double),
        ("step_is_successful", C.c_int32), ("step_is_valid", C.c_int32),
    ]


class SummaryC(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("minimizer_time_in_seconds", C.c_double),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("num_iterations", C.c_int32), ("termination", C.c_int32),
        ("reduced_dim", C.c_int32), ("tail_dim", C.c_int32),
        ("trace", IterationC * MAX_TRACE),
    ]

    def rows(self):
        out = []
        for i in range(self.num_iterations + 1):
            t = self.trace[i]
            out.append({k: getattr(t, k) for k, _ in IterationC._fields_})
        return out


TERMINATION = {0: "RUNNING", 1: "CONVERGED_GRADIENT", 2: "CONVERGED_PARAMETER",
               3: "CONVERGED_FUNCTION", 4: "NO_CONVERGENCE", 5: "RADIUS_TOO_SMALL",
               6: "LINEAR_SOLVER_FAILURE", 7: "ASSEMBLED_ONLY"}

_F64 = ("pose", "sb", "lm", "sc", "proj_uv", "imu_pre", "cp_dat", "pr_dat", "dop_dat", "sp_w",
        "spr_dat", "scp_dat", "fix_dat", "idp_pts", "comp_pose", "comp_sb", "comp_pose_lin", "comp_sb_lin", "comp_Hpp", "comp_HpN", "comp_rhs_p",
        "comp_HNN", "comp_rhsN", "comp_pre", "comp_H12", "prior_J", "prior_r0", "prior_x0")
_I32 = ("order_block", "order_group", "proj_idx", "imu_idx", "cp_idx", "pr_idx", "dop_idx",
        "sp_idx", "spr_idx", "scp_idx", "fix_idx", "idp_kind", "idp_idx", "comp_M", "comp_N", "comp_idx", "comp_mid", "prior_nblk", "prior_dim", "prior_blk")


class FlatWindow:
    """A window as a dict of contiguous numpy arrays + scalars, with a ctypes view.

    Mutable state (pose/sb/lm/sc) is owned here; a solve writes the result back into
    these arrays, exactly as ceres writes into the caller's parameter blocks.
    """

    def __init__(self, **kw):
        self.a = {}
        for k in _F64:
            self.a[k] = np.ascontiguousarray(kw.get(k, np.zeros(0)), dtype=np.float64)
        for k in _I32:
            self.a[k] = np.ascontiguousarray(kw.get(k, np.zeros(0, np.int32)), dtype=np.int32)
        self.a["is_const"] = np.ascontiguousarray(kw["is_const"], dtype=np.uint8)
        self.n_tail = int(kw.get("n_tail", 0))
        self.proj_sqrt_info = float(kw.get("proj_sqrt_info", 1000.0 / 1.5))
        self.proj_loss_a = float(kw.get("proj_loss_a", 1.0))
        self.pbg = np.asarray(kw.get("pbg", np.zeros(3)), dtype=np.float64)
        self.gw = np.asarray(kw.get("gw", np.array([0, 0, 9.8])), dtype=np.float64)
        self.base = np.asarray(kw.get("base", np.zeros(3)), dtype=np.float64)
        self.meta = kw.get("meta", {})

    # sizes --------------------------------------------------------------
    @property
    def n_pose(self): return self.a["pose"].size // 7
    @property
    def n_sb(self): return self.a["sb"].size // 9
    @property
    def n_lm(self): return self.a["lm"].size // 3
    @property
    def n_sc(self): return self.a["sc"].size
    @property
    def n_blocks(self): return self.n_pose + self.n_sb + self.n_lm + self.n_sc

    def bid_pose(self, i): return i
    def bid_sb(self, i): return self.n_pose + i
    def bid_lm(self, i): return self.n_pose + self.n_sb + i
    def bid_sc(self, i): return self.n_pose + self.n_sb + self.n_lm + i

    def block_sizes(self):
        g = np.concatenate([np.full(self.n_pose, 7), np.full(self.n_sb, 9),
                            np.full(self.n_lm, 3), np.full(self.n_sc, 1)]).astype(np.int32)
        l = np.where(g == 7, 6, g).astype(np.int32)
        return g, l

    def state(self):
        return {k: self.a[k].copy() for k in ("pose", "sb", "lm", "sc")}

    def set_state(self, st):
        for k in ("pose", "sb", "lm", "sc"):
            self.a[k][...] = np.asarray(st[k], dtype=np.float64).reshape(self.a[k].shape)

    def copy(self):
        kw = {k: v.copy() for k, v in self.a.items()}
        return FlatWindow(n_tail=self.n_tail, proj_sqrt_info=self.proj_sqrt_info,
                          proj_loss_a=self.proj_loss_a, pbg=self.pbg.copy(), gw=self.gw.copy(),
                          base=self.base.copy(), meta=dict(self.meta), **kw)

    def c_struct(self):
        """Build the C struct (pointers alias self.a arrays; keep self alive while in use)."""
        s = FlatWindowC()
        a = self.a
        s.n_pose, s.n_sb, s.n_lm, s.n_sc = self.n_pose, self.n_sb, self.n_lm, self.n_sc
        for k in _F64:
            setattr(s, k, a[k].ctypes.data_as(_pd))
        for k in _I32:
            setattr(s, k, a[k].ctypes.data_as(_pi))
        if a["comp_mid"].size == 0 or a["comp_H12"].size == 0:      # optional: no composite factor has a middle-marginalisation link
            s.comp_mid = None; s.comp_H12 = None
        s.is_const = a["is_const"].ctypes.data_as(_pu8)
        s.n_order = a["order_block"].size
        s.n_tail = self.n_tail
        s.n_proj = a["proj_idx"].size // 3
        s.n_imu = a["imu_idx"].size // 4
        s.n_cp = a["cp_idx"].size // 3
        s.n_pr = a["pr_idx"].size // 2
        s.n_dop = a["dop_idx"].size // 3
        s.n_sp = a["sp_idx"].size
        s.n_spr = a["spr_idx"].size // 2
        s.n_scp = a["scp_idx"].size // 3
        s.n_fix = a["fix_idx"].size // 2
        s.n_comp = a["comp_M"].size
        s.n_idp = a["idp_kind"].size
        s.n_prior = a["prior_nblk"].size
        s.proj_sqrt_info, s.proj_loss_a = self.proj_sqrt_info, self.proj_loss_a
        for i in range(3):
            s.pbg[i], s.gw[i], s.base[i] = self.pbg[i], self.gw[i], self.base[i]
        return s

    def counts(self):
        a = self.a
        return dict(n_pose=self.n_pose, n_sb=self.n_sb, n_lm=self.n_lm, n_sc=self.n_sc,
                    n_proj=a["proj_idx"].size // 3, n_imu=a["imu_idx"].size // 4,
                    n_cp=a["cp_idx"].size // 3, n_pr=a["pr_idx"].size // 2,
                    n_dop=a["dop_idx"].size // 3, n_sp=a["sp_idx"].size,
                    n_spr=a["spr_idx"].size // 2, n_scp=a["scp_idx"].size // 3, n_fix=a["fix_idx"].size // 2, n_comp=a["comp_M"].size, n_idp=a["idp_kind"].size,
                    n_prior=a["prior_nblk"].size,
                    prior_dim=[int(x) for x in a["prior_dim"]])
