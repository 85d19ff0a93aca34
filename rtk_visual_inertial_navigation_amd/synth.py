"""Deterministic synthetic sliding windows for the BASELINE.json configs.

Host-side input producer: trajectory, landmarks, IMU samples + mid-point pre-integration
(R/factor/integration_base.cpp:30-142), RTK carrier-phase / pseudorange records
(R/swf/swf_core.cpp:98-132), the gauge prior of InitializeSqrtInfo
(R/swf/swf_core.cpp:499-513) and the elimination order of MyOrdering
(R/swf/swf_gnss.cpp:629-783, see ordering.py).  Nothing here runs inside a solve.
"""
import numpy as np

from .flat import FlatWindow, PRE, PRE_DOUBLES, CP_DOUBLES, PR_DOUBLES, DOP_DOUBLES
from .ordering import my_ordering

# yaml/rtk_visual_inertial_config.yaml
ACC_N, GYR_N, ACC_W, GYR_W = 0.05, 0.005, 0.0005, 0.00005          # :24-27
G_NORM = 9.8                                                         # :28
PBG = np.array([-0.0051302024, 0.0091942546, 0.308739733])           # :92-96
ANCHOR = np.array([-2323932.39454, 5387298.51324, 2493096.51920])    # :119-123
BODY_T_CAM0 = np.array([                                             # :64-72
    [-1.1283524065062611e-02, 9.0570010831436121e-03, 9.9989532092917277e-01, 1.3224454035460147e-02],
    [-9.9992100257025784e-01, -5.6404389398068133e-03, -1.1232723065088990e-02, 5.7114724738452263e-02],
    [5.5381137189322582e-03, -9.9994307646982916e-01, 9.1199296318514866e-03, -1.5241815653778757e-02],
    [0., 0., 0., 1.]])
FOCAL_LENGTH, FEATUREWEIGHTINVERSE = 1000.0, 1.5                     # R/parameter/parameters.h:15-17
LAM_L1 = 0.190293672798364871256993069437                            # R/gnss/src/common_function.cpp:4-8
CLIGHT, OMGE = 299792458.0, 7.2921151467E-5                          # R/gnss/include/common_function.h:21,41
AZELMIN = 25.0 / 180 * np.pi                                         # :22

CONFIGS = {
    1: dict(K=10, F=100, S=0, prior="gauge"),
    2: dict(K=10, F=100, S=0, prior="gauge"),
    3: dict(K=20, F=300, S=10, prior="gauge"),
    4: dict(K=20, F=300, S=10, prior="gauge"),
    5: dict(K=40, F=1000, S=20, prior="dense"),
}
BASE_SEED = 0xC0FFEE


# ------------------------------------------------------------------ small helpers
def q_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def q_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_q(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def enu_to_ecef_rotation(ecef):
    x, y, z = ecef
    lon = np.arctan2(y, x)
    lat = np.arctan2(z, np.hypot(x, y))     # geocentric is enough for a synthetic frame
    sl, cl, sp, cp = np.sin(lon), np.cos(lon), np.sin(lat), np.cos(lat)
    # columns = E, N, U expressed in ECEF
    return np.array([[-sl, -sp * cl, cp * cl], [cl, -sp * sl, cp * sl], [0, cp, sp]])


# ------------------------------------------------------------------ pre-integration
def preintegrate(samples, ba, bg, acc_n=ACC_N, gyr_n=GYR_N, acc_w=ACC_W, gyr_w=GYR_W):
    """Mid-point IMU pre-integration, IntegrationBase (R/factor/integration_base.cpp:5-142).

    samples: [n][7] = dt, acc(3), gyr(3); sample 0 seeds acc_0/gyr_0.  Returns the
    SWF_PRE_DOUBLES record of include/swf_types.h.
    """
    dp, dv = np.zeros(3), np.zeros(3)
    dq = np.array([0., 0., 0., 1.])
    jac, cov = np.eye(15), np.zeros((15, 15))
    noise = np.concatenate([np.full(3, acc_n ** 2), np.full(3, gyr_n ** 2), np.full(3, acc_n ** 2),
                            np.full(3, gyr_n ** 2), np.full(3, acc_w ** 2), np.full(3, gyr_w ** 2)])
    acc0, gyr0 = samples[0, 1:4].copy(), samples[0, 4:7].copy()
    gyri, gyrj = gyr0.copy(), gyr0.copy()
    sum_dt = 0.0
    I3 = np.eye(3)
    for s in range(1, samples.shape[0]):
        dt = samples[s, 0]
        acc1, gyr1 = samples[s, 1:4], samples[s, 4:7]
        gyrj = gyr1.copy()
        a0, a1 = acc0 - ba, acc1 - ba
        w = 0.5 * (gyr0 + gyr1) - bg
        R0 = q_to_R(dq)
        rq = q_mul(dq, np.array([w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0]))
        # the reference uses the UN-normalised result_delta_q both for q*v and toRotationMatrix()
        R1u = q_to_R(rq)
        un_acc = 0.5 * (R0 @ a0 + _qrot(rq, a1))
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        Rw, Ra0, Ra1 = skew(w), skew(a0), skew(a1)
        ImRw = I3 - Rw * dt
        A, B = R0 @ Ra0, R1u @ Ra1
        Cm = B @ ImRw
        F = np.zeros((15, 15)); V = np.zeros((15, 18))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * A * dt * dt + -0.25 * Cm * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (R0 + R1u) * dt * dt
        F[0:3, 12:15] = -0.25 * B * dt * dt * -dt
        F[3:6, 3:6] = ImRw
        F[3:6, 12:15] = -I3 * dt
        F[6:9, 3:6] = -0.5 * A * dt + -0.5 * Cm * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (R0 + R1u) * dt
        F[6:9, 12:15] = -0.5 * B * dt * -dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V[0:3, 0:3] = 0.25 * R0 * dt * dt
        V[0:3, 3:6] = 0.25 * -B * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * R1u * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * R0 * dt
        V[6:9, 3:6] = 0.5 * -B * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * R1u * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        jac = F @ jac
        cov = F @ cov @ F.T + (V * noise) @ V.T
        dp, dv = rp, rv
        dq = rq / np.linalg.norm(rq)
        sum_dt += dt
        acc0, gyr0 = acc1.copy(), gyr1.copy()
    pre = np.zeros(PRE_DOUBLES)
    pre[PRE["DP"]:PRE["DP"] + 3] = dp
    pre[PRE["DQ"]:PRE["DQ"] + 4] = dq
    pre[PRE["DV"]:PRE["DV"] + 3] = dv
    pre[PRE["LBA"]:PRE["LBA"] + 3] = ba
    pre[PRE["LBG"]:PRE["LBG"] + 3] = bg
    pre[PRE["DP_DBA"]:PRE["DP_DBA"] + 9] = jac[0:3, 9:12].ravel()
    pre[PRE["DP_DBG"]:PRE["DP_DBG"] + 9] = jac[0:3, 12:15].ravel()
    pre[PRE["DQ_DBG"]:PRE["DQ_DBG"] + 9] = jac[3:6, 12:15].ravel()
    pre[PRE["DV_DBA"]:PRE["DV_DBA"] + 9] = jac[6:9, 9:12].ravel()
    pre[PRE["DV_DBG"]:PRE["DV_DBG"] + 9] = jac[6:9, 12:15].ravel()
    pre[PRE["SUMDT"]] = sum_dt
    pre[PRE["GYRI"]:PRE["GYRI"] + 3] = gyri
    pre[PRE["GYRJ"]:PRE["GYRJ"] + 3] = gyrj
    # sqrt_info = LLT(cov^-1).matrixL()^T (R/factor/integration_base.cpp:105-113)
    ci = np.linalg.inv(cov)
    ci = np.tril(ci) + np.tril(ci, -1).T
    pre[PRE["SQRTINFO"]:] = np.linalg.cholesky(ci).T.ravel()
    return pre


def _qrot(q, v):
    u = q[:3]
    uv = 2 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


# ------------------------------------------------------------------ trajectory
class Trajectory:
    """IMU-body trajectory in a local ENU frame: 2 m/s arc of radius 60-80 m with a sinusoidal
    height and gentle roll/pitch.  Heading = tangent + a fixed 75 deg yaw offset, i.e. the
    camera (which looks along body x, yaml body_T_cam0) looks ACROSS the track.  DEVIATION from
    "tangent-aligned" (SURVEY.md §8d): a forward-looking camera leaves the depth of 2-frame
    landmarks unobservable (they run away by kilometres in the oracle), which would turn fp64
    parity into a test of conditioning instead of arithmetic."""
    YAW_OFFSET = np.deg2rad(75.0)

    def __init__(self, rng):
        self.R = 60.0 + 20.0 * rng.random()
        self.v = 2.0
        self.h_amp = 0.3 + 0.2 * rng.random()
        self.h_w = 0.7 + 0.3 * rng.random()
        self.psi0 = 2 * np.pi * rng.random()
        self.p0 = np.array([15.0, -8.0, 1.5]) + rng.normal(0, 3.0, 3)
        self.roll_a, self.pitch_a = 0.03 * rng.random() + 0.01, 0.03 * rng.random() + 0.01
        self.roll_w, self.pitch_w = 1.1 + rng.random(), 0.9 + rng.random()

    def pos(self, t):
        th = self.v * t / self.R
        c, s = np.cos(self.psi0), np.sin(self.psi0)
        x, y = self.R * np.sin(th), self.R * (1 - np.cos(th))
        return self.p0 + np.array([c * x - s * y, s * x + c * y, self.h_amp * np.sin(self.h_w * t)])

    def vel(self, t, h=1e-5):
        return (self.pos(t + h) - self.pos(t - h)) / (2 * h)

    def acc(self, t, h=1e-4):
        return (self.pos(t + h) - 2 * self.pos(t) + self.pos(t - h)) / (h * h)

    def rot(self, t):
        yaw = self.psi0 + self.v * t / self.R + self.YAW_OFFSET
        return rot_zyx(yaw, self.pitch_a * np.sin(self.pitch_w * t), self.roll_a * np.sin(self.roll_w * t))

    def omega_body(self, t, h=1e-5):
        Rm, Rp = self.rot(t - h), self.rot(t + h)
        W = self.rot(t).T @ (Rp - Rm) / (2 * h)
        return np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) / 2


def _track_lengths(rng, F, K):
    """Track lengths in [2, K] with total exactly F*K/2 (mean K/2)."""
    target = F * K // 2
    raw = rng.integers(2, K + 1, size=F).astype(np.int64)
    ln = np.clip(np.rint(raw * (target / raw.sum())), 2, K).astype(np.int64)
    diff = int(target - ln.sum())
    guard = 0
    while diff != 0 and guard < 100 * F:
        i = int(rng.integers(0, F))
        if diff > 0 and ln[i] < K:
            ln[i] += 1; diff -= 1
        elif diff < 0 and ln[i] > 2:
            ln[i] -= 1; diff += 1
        guard += 1
    return ln


# ------------------------------------------------------------------ window generator
def make_window(config_id=3, seed=None, K=None, F=None, S=None, prior=None, perturb=True,
                dt_kf=0.2, imu_rate=400, doppler=False, head=None):
    """Build one synthetic window.  Returns a FlatWindow whose state is the (perturbed)
    initial guess; meta['truth'] holds the noise-free state."""
    cfg = dict(CONFIGS.get(config_id, CONFIGS[3]))
    if K is not None: cfg["K"] = K
    if F is not None: cfg["F"] = F
    if S is not None: cfg["S"] = S
    if prior is not None: cfg["prior"] = prior
    K, F, S = cfg["K"], cfg["F"], cfg["S"]
    if seed is None:
        seed = BASE_SEED + config_id
    rng = np.random.Generator(np.random.PCG64(seed))

    Rwgw = enu_to_ecef_rotation(ANCHOR)
    gw = Rwgw @ np.array([0, 0, G_NORM])
    traj = Trajectory(rng)
    ric = BODY_T_CAM0[:3, :3]; tic = BODY_T_CAM0[:3, 3]
    qic = R_to_q(ric)
    ric = q_to_R(qic)

    t_kf = np.arange(K) * dt_kf
    # true states: pose = (antenna position, q_world_body); sb = (antenna velocity, ba, bg)
    ba_true = rng.normal(0, 0.02, 3); bg_true = rng.normal(0, 0.002, 3)
    pose_t = np.zeros((K + 1, 7)); sb_t = np.zeros((K, 9))
    Rwb = []
    for k, t in enumerate(t_kf):
        R = Rwgw @ traj.rot(t)
        Rwb.append(R)
        p_imu = Rwgw @ traj.pos(t)
        v_imu = Rwgw @ traj.vel(t)
        w_b = traj.omega_body(t)
        pose_t[k, :3] = p_imu + R @ PBG
        pose_t[k, 3:] = R_to_q(R)
        sb_t[k, :3] = v_imu + R @ np.cross(w_b, PBG)
        sb_t[k, 3:6] = ba_true
        sb_t[k, 6:9] = bg_true
    pose_t[K, :3] = tic; pose_t[K, 3:] = qic          # camera extrinsic lives in the pose pool

    # ---- IMU samples + pre-integration between consecutive keyframes
    n_sub = int(round(dt_kf * imu_rate))
    dt_imu = dt_kf / n_sub
    imu_idx = np.zeros((K - 1, 4), np.int32); imu_pre = np.zeros((K - 1, PRE_DOUBLES))
    for k in range(K - 1):
        smp = np.zeros((n_sub + 1, 7))
        for s in range(n_sub + 1):
            t = t_kf[k] + s * dt_imu
            Rb = traj.rot(t)
            acc = Rb.T @ (traj.acc(t) + np.array([0, 0, G_NORM])) + ba_true + rng.normal(0, ACC_N, 3)
            gyr = traj.omega_body(t) + bg_true + rng.normal(0, GYR_N, 3)
            smp[s, 0] = dt_imu; smp[s, 1:4] = acc; smp[s, 4:7] = gyr
        # linearisation biases = current bias estimates (slightly off the truth)
        imu_pre[k] = preintegrate(smp, ba_true + rng.normal(0, 0.005, 3), bg_true + rng.normal(0, 0.0005, 3))
        imu_idx[k] = [k, k, k + 1, k + 1]

    # ---- landmarks + observations
    lens = _track_lengths(rng, F, K)
    lm_t = np.zeros((F, 3)); proj_idx = []; proj_uv = []
    for f in range(F):
        L = int(lens[f])
        for _ in range(200):
            start = int(rng.integers(0, K - L + 1))
            mid = start + L // 2
            depth = rng.uniform(5.0, 40.0)
            uvn = rng.uniform(-0.45, 0.45, 2) * np.array([1.0, 0.6])
            pc = np.array([uvn[0] * depth, uvn[1] * depth, depth])
            p_imu_mid = pose_t[mid, :3] - Rwb[mid] @ PBG
            X = Rwb[mid] @ (ric @ pc + tic) + p_imu_mid
            ok, obs = True, []
            for j in range(start, start + L):
                p_imu_j = Rwb[j].T @ (X - pose_t[j, :3])
                pcj = ric.T @ (p_imu_j + PBG - tic)
                if pcj[2] < 1.0 or abs(pcj[0] / pcj[2]) > 1.2 or abs(pcj[1] / pcj[2]) > 1.0:
                    ok = False; break
                obs.append((j, pcj[0] / pcj[2], pcj[1] / pcj[2]))
            if ok:
                break
        lm_t[f] = X
        for (j, u, v) in obs:
            proj_idx.append([j, K, f])
            proj_uv.append([u + rng.normal(0, 1.0 / FOCAL_LENGTH), v + rng.normal(0, 1.0 / FOCAL_LENGTH)])
    proj_idx = np.array(proj_idx, np.int32).reshape(-1, 3); proj_uv = np.array(proj_uv).reshape(-1, 2)

    # ---- scalars: [dummy blackvalue2] + S ambiguities + K clocks (+ 1 receiver clock drift with Doppler)
    use_dop = bool(doppler) and S > 0
    n_sc = 1 + S + (K if S > 0 else 0) + (1 if use_dop else 0)
    sc_t = np.zeros(n_sc)
    i_dummy = 0; i_amb0 = 1; i_clk0 = 1 + S
    cp_idx, cp_dat, pr_idx, pr_dat, dop_idx, dop_dat = [], [], [], [], [], []
    i_drift = 1 + S + K
    base = ANCHOR.copy()
    if S > 0:
        sc_t[i_amb0:i_amb0 + S] = np.rint(rng.normal(0, 20, S))
        sc_t[i_clk0:i_clk0 + K] = rng.uniform(-30, 30, K)
        if use_dop:
            sc_t[i_drift] = rng.uniform(-0.5, 0.5)
        up = Rwgw[:, 2]
        sat0, satv = [], []
        for s in range(S):
            az = rng.uniform(0, 2 * np.pi); el = rng.uniform(AZELMIN + 0.05, np.deg2rad(85))
            los = Rwgw @ np.array([np.cos(el) * np.sin(az), np.cos(el) * np.cos(az), np.sin(el)])
            # range so that |sat| = 26 560 km
            b = base @ los; c = base @ base - 26560e3 ** 2
            rho = -b + np.sqrt(b * b - c)
            p = base + rho * los
            vdir = np.cross(p, rng.normal(0, 1, 3)); vdir /= np.linalg.norm(vdir)
            sat0.append(p); satv.append(3.9e3 * vdir)
        dt_br = 0.4
        for k in range(K):
            xg = pose_t[k, :3] + base
            for s in range(S):
                ps = sat0[s] + satv[s] * t_kf[k]
                e = xg - ps; r = np.linalg.norm(e); e /= r
                rho = r + OMGE * (ps[0] * xg[1] - ps[1] * xg[0]) / CLIGHT
                el = np.arcsin(np.clip(-(e @ up), -1, 1))
                sig_cp = 0.004 * LAM_L1; sig_pr = 0.3
                sin_el = float(np.float32(np.sin(np.float32(el))))
                L1_lam = rho - LAM_L1 * sc_t[i_amb0 + s] + sc_t[i_clk0 + k] + rng.normal(0, sig_cp / sin_el)
                P1 = rho + sc_t[i_clk0 + k] + rng.normal(0, sig_pr / sin_el)
                cp_idx.append([k, i_amb0 + s, i_clk0 + k])
                cp_dat.append([ps[0], ps[1], ps[2], L1_lam, LAM_L1, el, dt_br, sig_cp ** 2, 1.0])
                pr_idx.append([k, i_clk0 + k])
                pr_dat.append([ps[0], ps[1], ps[2], P1, el, dt_br, sig_pr ** 2])
                if use_dop:
                    # SppDopplerFactor(satellite_vel, satellite_pos, ., D*lam, istd, base) on
                    # (speed_bias, clock drift para_gnss_dt[0]+12, pose): R/swf/swf_core.cpp:189-201
                    vs = satv[s]; vr = sb_t[k, :3]
                    rate = (vr - vs) @ e + OMGE / CLIGHT * (vs[1] * xg[0] + ps[1] * vr[0] - vs[0] * xg[1] - ps[0] * vr[1])
                    sig_d = 0.05
                    istd = sin_el * sin_el / sig_d
                    D1_lam = -(rate + sc_t[i_drift]) + rng.normal(0, 1.0 / istd)
                    dop_idx.append([k, i_drift, k])
                    dop_dat.append([ps[0], ps[1], ps[2], vs[0], vs[1], vs[2], D1_lam, istd])

    # ---- initial guess = truth perturbed (5 cm / 0.5 deg / 5 cm/s / 1 % depth)
    pose = pose_t.copy(); sb = sb_t.copy(); lm = lm_t.copy(); sc = sc_t.copy()
    if perturb:
        for k in range(K):
            pose[k, :3] += rng.normal(0, 0.05, 3)
            dth = rng.normal(0, np.deg2rad(0.5), 3)
            q = q_mul(pose[k, 3:], np.array([dth[0] / 2, dth[1] / 2, dth[2] / 2, 1.0]))
            pose[k, 3:] = q / np.linalg.norm(q)
            sb[k, :3] += rng.normal(0, 0.05, 3)
            sb[k, 3:6] += rng.normal(0, 0.01, 3)
            sb[k, 6:9] += rng.normal(0, 0.001, 3)
        for f in range(F):
            obs_frames = proj_idx[proj_idx[:, 2] == f][:, 0]
            j = int(obs_frames[0])
            d = lm[f] - pose_t[j, :3]
            lm[f] = pose_t[j, :3] + d * (1.0 + rng.normal(0, 0.01))
        if S > 0:
            sc[i_amb0:i_amb0 + S] += rng.normal(0, 0.3, S)
            sc[i_clk0:i_clk0 + K] += rng.normal(0, 1.0, K)
            if use_dop:
                sc[i_drift] += rng.normal(0, 0.1)

    n_pose, n_sb = K + 1, K
    n_blocks = n_pose + n_sb + F + n_sc
    is_const = np.zeros(n_blocks, np.uint8)
    is_const[K] = 1                                     # ESTIMATE_EXTRINSIC: 0 in every yaml

    def bid_pose(i): return i
    def bid_sb(i): return n_pose + i
    def bid_lm(i): return n_pose + n_sb + i
    def bid_sc(i): return n_pose + n_sb + F + i

    # ---- prior
    if cfg["prior"] == "gauge":
        # InitializeSqrtInfo (R/swf/swf_core.cpp:499-513): sqrt-info diagonals
        if S > 0:
            d = np.concatenate([np.full(3, 1e-3), np.full(3, 180 / np.pi / 5), np.full(3, 1e-3), np.full(3, 1e1), np.full(3, 1e2)])
        else:
            d = np.concatenate([np.full(3, 2e2), np.full(3, 2e2), np.full(3, 1e1), np.full(3, 1e1), np.full(3, 1e2)])
        prior_blk = [bid_pose(0), bid_sb(0)]
        Jp = np.diag(d); r0 = np.zeros(15)
        x0 = np.concatenate([pose[0], sb[0]])
    else:
        # dense prior of dimension 6+9+S+6*12.  DEVIATION: synthesised
        # as the square root of a random well-conditioned information matrix coupling the kept
        # blocks, not by marginalising a 41st frame (that path is SURVEY.md §8f rank 1).
        kept_pose = list(range(0, 13))
        prior_blk = [bid_pose(0), bid_sb(0)] + [bid_sc(i_amb0 + s) for s in range(S)] + [bid_pose(i) for i in kept_pose[1:]]
        dim = 6 + 9 + S + 6 * 12
        scale = np.concatenate([np.full(3, 30.0), np.full(3, 100.0), np.full(3, 5.0), np.full(3, 10.0), np.full(3, 100.0),
                                np.full(S, 2.0), np.tile(np.concatenate([np.full(3, 30.0), np.full(3, 100.0)]), 12)])
        M = rng.normal(0, 1.0, (2 * dim, dim)) / np.sqrt(2 * dim)
        A = (M.T @ M + 0.5 * np.eye(dim)) * np.outer(scale, scale)
        Jp = np.linalg.cholesky(A).T
        r0 = rng.normal(0, 0.3, dim)
        x0 = np.concatenate([pose[0], sb[0]] + [sc[i_amb0 + s:i_amb0 + s + 1] for s in range(S)] + [pose[i] for i in kept_pose[1:]])
    prior_dim = Jp.shape[0]

    roles = dict(dummy=bid_sc(i_dummy), landmarks=[bid_lm(f) for f in range(F)],
                 speed_bias=[bid_sb(k) for k in range(K)], poses=[bid_pose(k) for k in range(K)],
                 extrinsics=[bid_pose(K)], rtk_ambiguities=[bid_sc(i_amb0 + s) for s in range(S)],
                 clocks=[bid_sc(i_clk0 + k) for k in range(K)] if S > 0 else [],
                 pr_corrections=[bid_sc(i_drift)] if use_dop else [],   # the drift scalar takes its own later group
                 prior_kept=list(prior_blk), parameter_head=[])
    # ceres::internal::parameter_head (ordered last and exported / kept by the marginalisation consumer):
    #   "ambiguities": the RTK ambiguity states (UpdateNParameterHead, R/swf/swf_gnss.cpp:100-116)
    #   "frames":      every pose and speed-bias but the oldest frame's, plus the ambiguities — the neighbours a
    #                  GlobalMarge of frame 0 keeps (R/swf/swf_image.cpp:343-433)
    if head == "ambiguities":
        roles["parameter_head"] = [bid_sc(i_amb0 + s) for s in range(S)]
    elif head == "frames":
        roles["parameter_head"] = ([bid_pose(k) for k in range(1, K)] + [bid_sb(k) for k in range(1, K)]
                                   + [bid_sc(i_amb0 + s) for s in range(S)])
    roles["parameter_head"] = [b for b in roles["parameter_head"] if b not in set(prior_blk)]
    order_block, order_group, n_tail = my_ordering(roles, is_const)

    win = FlatWindow(
        pose=pose, sb=sb, lm=lm, sc=sc, is_const=is_const,
        order_block=order_block, order_group=order_group, n_tail=n_tail,
        proj_idx=proj_idx, proj_uv=proj_uv,
        proj_sqrt_info=FOCAL_LENGTH / FEATUREWEIGHTINVERSE, proj_loss_a=1.0,
        imu_idx=imu_idx, imu_pre=imu_pre,
        cp_idx=np.array(cp_idx, np.int32).reshape(-1, 3), cp_dat=np.array(cp_dat).reshape(-1, CP_DOUBLES),
        pr_idx=np.array(pr_idx, np.int32).reshape(-1, 2), pr_dat=np.array(pr_dat).reshape(-1, PR_DOUBLES),
        dop_idx=np.array(dop_idx, np.int32).reshape(-1, 3), dop_dat=np.array(dop_dat).reshape(-1, DOP_DOUBLES),
        sp_idx=np.array([i_dummy], np.int32), sp_w=np.array([1.0]),
        prior_nblk=np.array([len(prior_blk)], np.int32), prior_dim=np.array([prior_dim], np.int32),
        prior_blk=np.array(prior_blk, np.int32), prior_J=Jp, prior_r0=r0, prior_x0=x0,
        pbg=PBG, gw=gw, base=base,
        meta=dict(config_id=config_id, seed=int(seed), K=K, F=F, S=S, roles=roles,
                  truth=dict(pose=pose_t, sb=sb_t, lm=lm_t, sc=sc_t)))
    return win


def with_variable_extrinsic(win, head=False):
    """The same window with the camera extrinsic un-frozen, as the reference's marginalisation solves have it (GlobalMarge sets
    every parameter_head block variable, para_ex_Pose among them: R/swf/swf_image.cpp:384-389) and as ESTIMATE_EXTRINSIC = 1
    builds optimise it (R/swf/swf_image.cpp:168-177).  MyOrdering places it behind the poses (:712-718), or in the
    parameter_head tail with head = True."""
    w = win.copy()
    roles = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in w.meta["roles"].items()}
    is_const = w.a["is_const"].copy()
    for b in roles["extrinsics"]:
        is_const[b] = 0
    if head:
        roles["parameter_head"] = list(roles["parameter_head"]) + list(roles["extrinsics"])
    ob, og, nt = my_ordering(roles, is_const)
    w.a["is_const"] = np.ascontiguousarray(is_const, np.uint8)
    w.a["order_block"], w.a["order_group"], w.n_tail = ob, og, int(nt)
    w.meta["roles"] = roles
    return w


def with_spp_and_fixed(win, seed=7, n_fix=4, spp_stride=2):
    """Copy of an RTK window with rover-only and fixed-integer factors added on its existing blocks
    (a separate random stream, so the base window and the goldens made from it do not change):
      * SppPseudorangeFactor on (pose k, clock k) and SppCarrierPhaseFactor on (pose k, clock k, ambiguity s) for every
        `spp_stride`-th (epoch, satellite) pair — istd as the reference forms it for rover-only measurements, a fixed
        number per measurement (R/swf/swf_core.cpp:157-190);
      * `n_fix` FixedIntegerFactor(N21, 1/0.03) between ambiguity pairs, N21 = the rounded true difference
        (what the LAMBDA consumer injects after a successful fix, R/swf/swf_lambda.cpp:318-330)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = win.copy()
    K, S = win.meta["K"], win.meta["S"]
    if S <= 0:
        raise ValueError("with_spp_and_fixed needs a window with satellites")
    tr = win.meta["truth"]
    i_amb0, i_clk0 = 1, 1 + S
    cp_idx = win.a["cp_idx"].reshape(-1, 3); cp_dat = win.a["cp_dat"].reshape(-1, CP_DOUBLES)
    spr_idx, spr_dat, scp_idx, scp_dat = [], [], [], []
    for i in range(0, cp_idx.shape[0], spp_stride):
        k, ia, ic = (int(v) for v in cp_idx[i])
        ps = cp_dat[i, :3]
        xg = tr["pose"][k, :3] + win.base
        r = np.linalg.norm(xg - ps)
        rho = r + OMGE * (ps[0] * xg[1] - ps[1] * xg[0]) / CLIGHT
        istd_p, istd_l = float(rng.uniform(0.5, 3.0)), float(rng.uniform(20.0, 200.0))
        spr_idx.append([k, ic]); spr_dat.append([ps[0], ps[1], ps[2], rho + tr["sc"][ic] + rng.normal(0, 1.0 / istd_p), istd_p])
        scp_idx.append([k, ic, ia])
        scp_dat.append([ps[0], ps[1], ps[2], rho + tr["sc"][ic] - LAM_L1 * tr["sc"][ia] + rng.normal(0, 1.0 / istd_l), istd_l, LAM_L1])
    fix_idx, fix_dat = [], []
    for j in range(min(n_fix, S - 1)):
        a_, b_ = i_amb0 + j, i_amb0 + ((j + 1 + int(rng.integers(0, S - 1))) % S)
        if a_ == b_:
            b_ = i_amb0 + (j + 1) % S
        fix_idx.append([a_, b_]); fix_dat.append([float(np.rint(tr["sc"][b_] - tr["sc"][a_])), 1 / 0.03])
    w.a["spr_idx"] = np.array(spr_idx, np.int32).reshape(-1, 2); w.a["spr_dat"] = np.array(spr_dat, np.float64).reshape(-1, 5)
    w.a["scp_idx"] = np.array(scp_idx, np.int32).reshape(-1, 3); w.a["scp_dat"] = np.array(scp_dat, np.float64).reshape(-1, 6)
    w.a["fix_idx"] = np.array(fix_idx, np.int32).reshape(-1, 2); w.a["fix_dat"] = np.array(fix_dat, np.float64).reshape(-1, 2)
    return w


def make_batch(n_windows, config_id=4, seed0=None):
    """cfg4: n independent windows = cfg3 with seeds seed+i."""
    if seed0 is None:
        seed0 = BASE_SEED + config_id
    return [make_window(config_id=config_id, seed=seed0 + i) for i in range(n_windows)]
