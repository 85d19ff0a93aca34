"""BASELINE cfg5 with its marginalisation prior OBTAINED, not synthesised ("a dense prior ... obtained by actually
marginalising a 41st frame and its landmarks") — TEST / PROFILING INFRASTRUCTURE (needs a GPU: the marginalisation runs on the
device, swf_batch_marginalize).

  full = synth.make_window(5, K = K + 1, prior = "gauge")           one more frame than the target window
  marginalisation_window(full)   what GlobalMarge solves (R/swf/swf_image.cpp:343-433): only the factors that touch frame 0 —
                                 its pose, speed-bias, receiver clock and the landmarks first seen in it — with their
                                 neighbours (the poses that see those landmarks, pose 1 / speed-bias 1 through the IMU factor,
                                 the ambiguities through the carrier phases) as parameter_head, everything else constant
  slid_window(full, prior)       frames 1..K with the new prior in place of the gauge prior
"""
import numpy as np

from rtk_visual_inertial_navigation_amd import synth
from rtk_visual_inertial_navigation_amd.flat import FlatWindow, PRE_DOUBLES


def _own(w):
    """A window that owns its arrays (FlatWindow keeps views of contiguous slices: solving one in place would move its siblings)."""
    return w.copy()


def _layout(w):
    K = w.meta["K"]; S = w.meta["S"]
    n_pose, n_sb, F = K + 1, K, w.n_lm
    return dict(K=K, S=S, F=F, n_pose=n_pose, n_sb=n_sb, bid_sb=lambda i: n_pose + i, bid_lm=lambda i: n_pose + n_sb + i,
                bid_sc=lambda i: n_pose + n_sb + F + i, i_amb0=1, i_clk0=1 + S)


def marginalisation_window(full):
    """(window, kept) — kept = global block ids of the full window in prior (parameter_head) order."""
    L = _layout(full); a = full.a
    K, S = L["K"], L["S"]
    pi = a["proj_idx"].reshape(-1, 3)
    first = np.full(L["F"], K, np.int64)
    np.minimum.at(first, pi[:, 2], pi[:, 0])
    marg_lm = np.nonzero(first == 0)[0]
    keep_obs = np.isin(pi[:, 2], marg_lm)
    frames = sorted(set(int(p) for p in pi[keep_obs, 0]) - {0} | {1})
    head = [f for f in frames] + [L["bid_sb"](1)] + [L["bid_sc"](L["i_amb0"] + s) for s in range(S)]
    marg = [0, L["bid_sb"](0)]                                                  # ordered after group 0, before the tail
    g0 = [L["bid_lm"](int(l)) for l in marg_lm] + ([L["bid_sc"](L["i_clk0"])] if S else [])
    is_const = np.ones(full.n_blocks, np.uint8)
    for b in head + marg + g0:
        is_const[b] = 0
    order_block = g0 + marg + head
    order_group = [0] * len(g0) + list(range(1, 1 + len(marg) + len(head)))
    cp, pr = a["cp_idx"].reshape(-1, 3), a["pr_idx"].reshape(-1, 2)
    kc, kp = cp[:, 0] == 0, pr[:, 0] == 0
    w = FlatWindow(pose=a["pose"], sb=a["sb"], lm=a["lm"], sc=a["sc"], is_const=is_const,
                   order_block=np.array(order_block, np.int32), order_group=np.array(order_group, np.int32), n_tail=len(head),
                   proj_idx=pi[keep_obs], proj_uv=a["proj_uv"].reshape(-1, 2)[keep_obs], proj_sqrt_info=full.proj_sqrt_info, proj_loss_a=full.proj_loss_a,
                   imu_idx=a["imu_idx"].reshape(-1, 4)[:1], imu_pre=a["imu_pre"].reshape(-1, PRE_DOUBLES)[:1],
                   cp_idx=cp[kc], cp_dat=a["cp_dat"].reshape(-1, 9)[kc], pr_idx=pr[kp], pr_dat=a["pr_dat"].reshape(-1, 7)[kp],
                   prior_nblk=a["prior_nblk"], prior_dim=a["prior_dim"], prior_blk=a["prior_blk"], prior_J=a["prior_J"], prior_r0=a["prior_r0"],
                   prior_x0=a["prior_x0"], pbg=full.pbg, gw=full.gw, base=full.base, meta=dict(marg_lm=marg_lm, head=head))
    return _own(w), head


def slid_window(full, head, J, r0):
    """Frames 1..K of `full` with the linear prior (J, r0) over `head` (block ids of `full`), linearised at full's current state."""
    L = _layout(full); a = full.a
    K, S = L["K"], L["S"]
    pi, uv = a["proj_idx"].reshape(-1, 3), a["proj_uv"].reshape(-1, 2)
    first = np.full(L["F"], K, np.int64)
    np.minimum.at(first, pi[:, 2], pi[:, 0])
    lm_keep = np.nonzero(first > 0)[0]
    remap = -np.ones(L["F"], np.int64); remap[lm_keep] = np.arange(lm_keep.size)
    ok = remap[pi[:, 2]] >= 0
    pi2 = pi[ok].copy(); pi2[:, 0] -= 1; pi2[:, 1] = K - 1; pi2[:, 2] = remap[pi2[:, 2]]
    pose = a["pose"].reshape(-1, 7)[1:]                                          # frames 1..K-1 ... and the extrinsic (last)
    sb = a["sb"].reshape(-1, 9)[1:]
    sc_old = a["sc"]
    # scalar pool: dummy | S ambiguities | clocks 1..K-1
    keep_sc = [0] + [L["i_amb0"] + s for s in range(S)] + [L["i_clk0"] + k for k in range(1, K)]
    sc = sc_old[keep_sc]
    sc_map = {old: new for new, old in enumerate(keep_sc)}
    K2, F2 = K - 1, lm_keep.size
    n_pose, n_sb = K2 + 1, K2
    new_id = {}
    for k in range(1, K + 1):
        new_id[k] = k - 1                                                        # poses (incl. the extrinsic at index K)
    for k in range(1, K):
        new_id[L["bid_sb"](k)] = n_pose + k - 1
    for old, new in sc_map.items():
        new_id[L["bid_sc"](old)] = n_pose + n_sb + F2 + new
    cp, pr = a["cp_idx"].reshape(-1, 3).copy(), a["pr_idx"].reshape(-1, 2).copy()
    kc, kp = cp[:, 0] > 0, pr[:, 0] > 0
    cp, pr = cp[kc], pr[kp]
    cp[:, 0] -= 1; cp[:, 1] = [sc_map[int(i)] for i in cp[:, 1]]; cp[:, 2] = [sc_map[int(i)] for i in cp[:, 2]]
    pr[:, 0] -= 1; pr[:, 1] = [sc_map[int(i)] for i in pr[:, 1]]
    imu = a["imu_idx"].reshape(-1, 4)[1:] - 1
    roles = dict(dummy=n_pose + n_sb + F2, landmarks=[n_pose + n_sb + f for f in range(F2)], speed_bias=[n_pose + k for k in range(K2)],
                 poses=list(range(K2)), extrinsics=[K2], rtk_ambiguities=[n_pose + n_sb + F2 + 1 + s for s in range(S)],
                 clocks=[n_pose + n_sb + F2 + 1 + S + k for k in range(K2)] if S else [], pr_corrections=[],
                 prior_kept=[new_id[b] for b in head], parameter_head=[])
    is_const = np.zeros(n_pose + n_sb + F2 + sc.size, np.uint8); is_const[K2] = 1
    from rtk_visual_inertial_navigation_amd.ordering import my_ordering
    ob_, og_, nt = my_ordering(roles, is_const)
    blocks_full = ([a["pose"].reshape(-1, 7)[i] for i in range(K + 1)] + [a["sb"].reshape(-1, 9)[i] for i in range(K)]
                   + [a["lm"].reshape(-1, 3)[i] for i in range(L["F"])] + [a["sc"][i:i + 1] for i in range(a["sc"].size)])
    x0 = np.concatenate([blocks_full[b] for b in head])
    dim = J.shape[0]
    return _own(FlatWindow(pose=pose, sb=sb, lm=a["lm"].reshape(-1, 3)[lm_keep], sc=sc, is_const=is_const, order_block=ob_, order_group=og_, n_tail=nt,
                      proj_idx=pi2, proj_uv=uv[ok], proj_sqrt_info=full.proj_sqrt_info, proj_loss_a=full.proj_loss_a,
                      imu_idx=imu, imu_pre=a["imu_pre"].reshape(-1, PRE_DOUBLES)[1:], cp_idx=cp, cp_dat=a["cp_dat"].reshape(-1, 9)[kc],
                      pr_idx=pr, pr_dat=a["pr_dat"].reshape(-1, 7)[kp], sp_idx=np.array([0], np.int32), sp_w=a["sp_w"],
                      prior_nblk=np.array([len(head)], np.int32), prior_dim=np.array([dim], np.int32),
                      prior_blk=np.array([new_id[b] for b in head], np.int32), prior_J=J, prior_r0=r0, prior_x0=x0,
                      pbg=full.pbg, gw=full.gw, base=full.base, meta=dict(K=K2, F=F2, S=S, roles=roles, prior_dim=dim)))


def make_full(args):
    """(K, F, S, seed) -> the (K + 1)-frame window with the gauge prior (picklable: for process pools, before HIP is touched)."""
    K, F, S, seed = args
    return synth.make_window(5, K=K + 1, F=F, S=S, prior="gauge", seed=seed)


def make_cfg5_with_marginalised_prior(solver, K=40, F=1000, S=20, seed=None, full=None):
    """The cfg5 stress window whose prior comes out of swf_batch_marginalize.  Returns (window, info)."""
    from rtk_visual_inertial_navigation_amd.flat import default_options
    if full is None:
        full = make_full((K, F, S, seed))
    wm, head = marginalisation_window(full)
    bs = solver.BatchSolver([wm.copy()])
    sm = bs.solve(default_options(step_mode=1), download=False)[0]
    # small tails: the reference's eigen square root.  Large ones (the full-size stress window keeps every pose that shares a landmark
    # with the marginalised frame: 6 x 39 + 9 + 20 = 263): the Cholesky square root — the same quadratic, no 263-dimensional Jacobi
    # iteration — unless the marginal is singular on the kept states, which only the eigen form (k_marg_rescue) handles
    form = solver.BatchSolver.PRIOR_EIGEN if sm.tail_dim <= 140 else solver.BatchSolver.PRIOR_CHOLESKY
    bs.marginalize(1e-8, form)
    g = bs.get_prior(0)
    if g["rank"] < 0 and form == solver.BatchSolver.PRIOR_CHOLESKY and sm.tail_dim <= 384:
        form = solver.BatchSolver.PRIOR_EIGEN
        bs.marginalize(1e-8, form)
        g = bs.get_prior(0)
    bs.close()
    assert g["rank"] >= 0, "marginalisation failed"
    w = slid_window(full, head, g["J"], g["r0"])
    return w, dict(prior_dim=g["n"], rank=g["rank"], form="eigen" if form == solver.BatchSolver.PRIOR_EIGEN else "cholesky", n_marg_landmarks=int(wm.meta["marg_lm"].size), full=full, termination=sm.termination)
