"""Independent numpy second opinion on the factor residuals (rotation-matrix algebra, not the
quaternion-rotate formulas the oracle uses) + central-difference Jacobians on the manifold.
Formulas from SURVEY.md App. A, which restates R/factor/*.cpp.  TEST INFRASTRUCTURE."""
import numpy as np

CLIGHT, OMGE = 299792458.0, 7.2921151467E-5


def q2R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def pose_plus(x, d):
    q = qmul(x[3:], np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0]))
    return np.concatenate([x[:3] + d[:3], q / np.linalg.norm(q)])


def proj_residual(pose, ex, lm, uv, sqrt_info, pbg):
    Rj, ric = q2R(pose[3:]), q2R(ex[3:])
    p_imu = Rj.T @ (lm - pose[:3])
    pc = ric.T @ (p_imu + pbg - ex[:3])
    return sqrt_info * (pc[:2] / pc[2] - uv)


def proj_idepth_residual(kind, Pi, Pj, ex, ex2, inv_dep, pts_i, pts_j, sqrt_info, pbg):
    """projection_factor.cpp:77-329 residuals; kind 0 TwoFrameOneCam, 1 TwoFrameTwoCam, 2 OneFrameTwoCam."""
    e2 = ex if kind == 0 else ex2
    lever = np.zeros(3) if kind == 2 else pbg
    p_imu_i = q2R(ex[3:]) @ (pts_i / inv_dep) + ex[:3] - lever
    p_imu_j = p_imu_i if kind == 2 else q2R(Pj[3:]).T @ (q2R(Pi[3:]) @ p_imu_i + Pi[:3] - Pj[:3])
    pc = q2R(e2[3:]).T @ (p_imu_j + lever - e2[:3])
    return sqrt_info * (pc[:2] / pc[2] - pts_j[:2])


def imu_residual(pi, sbi, pj, sbj, pre, pbg, gw):
    dp, dq, dv = pre[0:3], pre[3:7], pre[7:10]
    lba, lbg = pre[10:13], pre[13:16]
    dp_dba, dp_dbg = pre[16:25].reshape(3, 3), pre[25:34].reshape(3, 3)
    dq_dbg = pre[34:43].reshape(3, 3)
    dv_dba, dv_dbg = pre[43:52].reshape(3, 3), pre[52:61].reshape(3, 3)
    T = pre[61]; gyri, gyrj = pre[62:65], pre[65:68]
    SI = pre[68:].reshape(15, 15)
    Pi, Qi, Vi, Bai, Bgi = pi[:3], pi[3:], sbi[:3], sbi[3:6], sbi[6:9]
    Pj, Qj, Vj, Baj, Bgj = pj[:3], pj[3:], sbj[:3], sbj[3:6], sbj[6:9]
    dba, dbg = Bai - lba, Bgi - lbg
    th = dq_dbg @ dbg
    cq = qmul(dq, np.array([th[0] / 2, th[1] / 2, th[2] / 2, 1.0]))
    cv = dv + dv_dba @ dba + dv_dbg @ dbg
    cp = dp + dp_dba @ dba + dp_dbg @ dbg
    Ri, Rj = q2R(Qi), q2R(Qj)
    wi, wj = gyri - Bgi, gyrj - Bgj
    r = np.zeros(15)
    r[0:3] = Ri.T @ (0.5 * gw * T * T + (Pj - Pi) - Rj @ pbg - Vi * T) - cp + pbg + np.cross(wi, pbg) * T
    # 2 vec(cq^-1 (Qi^-1 Qj)); Eigen inverse = conj / |q|^2
    cqi = qconj(cq) / (cq @ cq)
    r[3:6] = 2 * qmul(cqi, qmul(qconj(Qi) / (Qi @ Qi), Qj))[:3]
    r[6:9] = Ri.T @ (gw * T + Vj - Rj @ np.cross(wj, pbg) - Vi) - cv + np.cross(wi, pbg)
    r[9:12] = Baj - Bai
    r[12:15] = Bgj - Bgi
    return SI @ r


def corr_sin(el):
    return float(np.float32(np.sin(np.float64(np.float32(el)))))


def varerr2(el, dt, mv):
    b = CLIGHT * 5e-12 * dt
    s = corr_sin(el)
    return mv / s / s + b * b


def gnss_range(xg, xs):
    return np.linalg.norm(xg - xs) + OMGE * (xs[0] * xg[1] - xs[1] * xg[0]) / CLIGHT


def cp_residual(pose, amb, clk, dat, base):
    w = 1 / np.sqrt(varerr2(dat[5], dat[6], dat[7])) if dat[8] != 0 else 1.0
    return w * (gnss_range(pose[:3] + base, dat[:3]) - amb * dat[4] - dat[3] + clk)


def pr_residual(pose, clk, dat, base):
    w = 1 / np.sqrt(varerr2(dat[4], dat[5], dat[6]))
    return w * (gnss_range(pose[:3] + base, dat[:3]) - dat[3] + clk)


def spr_residual(pose, clk, dat, base):
    """SppPseudorangeFactor, gnss_factor.cpp:9-39; dat = sat[3] P1 istd."""
    return dat[4] * (gnss_range(pose[:3] + base, dat[:3]) + clk - dat[3])


def scp_residual(pose, clk, amb, dat, base):
    """SppCarrierPhaseFactor, gnss_factor.cpp:45-80; dat = sat[3] L1_lam istd lam."""
    return dat[4] * (gnss_range(pose[:3] + base, dat[:3]) + clk - amb * dat[5] - dat[3])


def fix_residual(na, nb, dat):
    """FixedIntegerFactor, gnss_factor.cpp:85-96; dat = N21 istd."""
    return dat[1] * ((nb - na) - dat[0])


def dop_residual(sb, drift, pose, dat, base):
    xg = pose[:3] + base; rs, vs = dat[:3], dat[3:6]
    e = (xg - rs) / np.linalg.norm(xg - rs)
    rate = (sb[:3] - vs) @ e + OMGE / CLIGHT * (vs[1] * xg[0] + rs[1] * sb[0] - vs[0] * xg[1] - rs[0] * sb[1])
    return dat[7] * (rate + drift + dat[6])


def fd_jac(fun, blocks, which, h):
    """Central-difference Jacobian of fun(*blocks) w.r.t. blocks[which] on the manifold."""
    x = blocks[which]
    ls = 6 if x.size == 7 else x.size
    r0 = np.atleast_1d(fun(*blocks))
    J = np.zeros((r0.size, ls))
    for j in range(ls):
        d = np.zeros(ls); d[j] = h
        bp = list(blocks); bm = list(blocks)
        bp[which] = pose_plus(x, d) if x.size == 7 else x + d
        bm[which] = pose_plus(x, -d) if x.size == 7 else x - d
        J[:, j] = (np.atleast_1d(fun(*bp)) - np.atleast_1d(fun(*bm))) / (2 * h)
    return J


def triangulate_two_view(Ps, Rs, tic, ric, pbg, f0, pt0, pt1, init_depth=5.0):
    """FeatureManager::triangulate two-view branch + triangulatePoint (feature_manager.cpp:148-161, 285-316) with LAPACK's SVD."""
    def pose(f):
        t = Ps[f] + Rs[f] @ tic; R = Rs[f] @ ric
        return np.hstack([R.T, (-R.T @ t)[:, None]])
    P0, P1 = pose(f0), pose(f0 + 1)
    D = np.stack([pt0[0] * P0[2] - P0[0], pt0[1] * P0[2] - P0[1], pt1[0] * P1[2] - P1[0], pt1[1] * P1[2] - P1[1]])
    v = np.linalg.svd(D)[2][-1]
    X = v[:3] / v[3]
    depth = (P0[:, :3] @ X + P0[:, 3])[2]
    if not depth > 0:
        depth = init_depth
    return depth, Rs[f0] @ (ric @ (np.array([pt0[0], pt0[1], 1.0]) * depth) + tic - pbg) + Ps[f0]
