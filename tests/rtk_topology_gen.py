"""RTK windows in the REFERENCE'S OWN topology — TEST INFRASTRUCTURE.

The reference never puts raw GNSS factors into the window problem: every GNSS epoch is a state of its own between two visual
frames; its raw factors are pre-eliminated to a linear prior over {pose, ambiguities} (GnssPreprocess, R/swf/swf_gnss.cpp:
504-532), and the epochs between two visual frames are hidden inside one composite IMU-GNSS factor (R/factor/gnss_imu_factor.cpp).
This module builds, from one synthetic trajectory:

  explicit_window   every epoch an explicit frame: visual frames carry projection factors, GNSS epochs carry raw carrier-phase /
                    pseudorange factors with a receiver clock each, IMU factors link consecutive states.  The ground truth of
                    what the composite construction must reproduce.
  epoch_windows     one tiny window per GNSS epoch: its raw factors, clock in elimination group 0, {pose, ambiguities} as the
                    parameter_head tail, ambiguity values zeroed (PhaseBiasSaveAndReset) — the inputs of swf_batch_marginal_priors.
  assemble_np       numpy restatement of IMUGNSSBase::AddMargInfo's bookkeeping (R/factor/gnss_imu_factor.cpp:245-352).
  composite_window  the window the estimator actually optimises: visual frames only, one composite factor per gap.
"""
import numpy as np

from rtk_visual_inertial_navigation_amd import synth
from rtk_visual_inertial_navigation_amd.flat import FlatWindow, PRE_DOUBLES


def _own(w):
    """A window that owns its arrays (FlatWindow keeps views of contiguous slices: solving one in place would move its siblings)."""
    return w.copy()
from rtk_visual_inertial_navigation_amd.ordering import my_ordering


def explicit_window(K_vis=4, M=2, F=24, S=6, seed=7):
    """(window, vis, hidden): T = K_vis + (K_vis - 1) M frames; vis[k] / hidden = frame indices."""
    T = K_vis + (K_vis - 1) * M
    w = synth.make_window(3, K=T, F=F, S=S, seed=seed)
    a = w.a
    vis = [k * (M + 1) for k in range(K_vis)]
    hidden = [t for t in range(T) if t not in vis]
    pi, uv = a["proj_idx"].reshape(-1, 3), a["proj_uv"].reshape(-1, 2)
    keep = np.isin(pi[:, 0], vis)
    pi, uv = pi[keep], uv[keep]
    cnt = np.bincount(pi[:, 2], minlength=w.n_lm)
    lm_keep = np.nonzero(cnt >= 2)[0]
    remap = -np.ones(w.n_lm, np.int64); remap[lm_keep] = np.arange(lm_keep.size)
    ok = remap[pi[:, 2]] >= 0
    pi, uv = pi[ok].copy(), uv[ok]
    pi[:, 2] = remap[pi[:, 2]]
    lm = a["lm"].reshape(-1, 3)[lm_keep]
    cp, cpd = a["cp_idx"].reshape(-1, 3), a["cp_dat"].reshape(-1, 9)
    pr, prd = a["pr_idx"].reshape(-1, 2), a["pr_dat"].reshape(-1, 7)
    kc, kp = np.isin(cp[:, 0], hidden), np.isin(pr[:, 0], hidden)
    n_pose, n_sb, F2, n_sc = T + 1, T, lm_keep.size, w.n_sc
    bid_sc = lambda i: n_pose + n_sb + F2 + i
    is_const = np.zeros(n_pose + n_sb + F2 + n_sc, np.uint8)
    is_const[T] = 1                                             # the extrinsic
    for k in vis:
        is_const[bid_sc(1 + S + k)] = 1                         # clocks of visual frames: no GNSS factors there
    roles = dict(dummy=bid_sc(0), landmarks=[n_pose + n_sb + f for f in range(F2)], speed_bias=[n_pose + k for k in range(T)],
                 poses=list(range(T)), extrinsics=[T], rtk_ambiguities=[bid_sc(1 + s) for s in range(S)],
                 clocks=[bid_sc(1 + S + k) for k in hidden], pr_corrections=[], prior_kept=[0, n_pose], parameter_head=[])
    ob_, og_, nt = my_ordering(roles, is_const)
    wx = FlatWindow(pose=a["pose"], sb=a["sb"], lm=lm, sc=a["sc"], is_const=is_const, order_block=ob_, order_group=og_, n_tail=nt,
                    proj_idx=pi, proj_uv=uv, proj_sqrt_info=w.proj_sqrt_info, proj_loss_a=w.proj_loss_a,
                    imu_idx=a["imu_idx"], imu_pre=a["imu_pre"], cp_idx=cp[kc], cp_dat=cpd[kc], pr_idx=pr[kp], pr_dat=prd[kp],
                    sp_idx=a["sp_idx"], sp_w=a["sp_w"], prior_nblk=a["prior_nblk"], prior_dim=a["prior_dim"],
                    prior_blk=np.array([0, n_pose], np.int32), prior_J=a["prior_J"], prior_r0=a["prior_r0"], prior_x0=a["prior_x0"],
                    pbg=w.pbg, gw=w.gw, base=w.base, meta=dict(K_vis=K_vis, M=M, S=S, T=T, vis=vis, hidden=hidden))
    return _own(wx), vis, hidden


def epoch_windows(wx):
    """One window per hidden GNSS epoch: blocks = [pose_h | S ambiguities (values zeroed), clock_h]; the clock is eliminated
    (group 0), the tail keeps pose_h and the ambiguities.  Returns (windows, kept) with kept[e] = [(7, None)] + [(1, s)] * S."""
    a, m = wx.a, wx.meta
    S, out, kept = m["S"], [], []
    pose = a["pose"].reshape(-1, 7)
    cp, cpd = a["cp_idx"].reshape(-1, 3), a["cp_dat"].reshape(-1, 9)
    pr, prd = a["pr_idx"].reshape(-1, 2), a["pr_dat"].reshape(-1, 7)
    for h in m["hidden"]:
        ic, ip = cp[:, 0] == h, pr[:, 0] == h
        ce, pe = cp[ic].copy(), pr[ip].copy()
        clk = int(ce[0, 2])
        # scalar pool of the epoch: S ambiguities then the clock
        ce[:, 0] = 0; ce[:, 1] = ce[:, 1] - 1; ce[:, 2] = S
        pe[:, 0] = 0; pe[:, 1] = S
        sc = np.concatenate([np.zeros(S), [a["sc"][clk]]])
        n_blocks = 1 + S + 1
        order_block = np.array([1 + S] + [0] + [1 + s for s in range(S)], np.int32)          # clock | pose, ambiguities
        order_group = np.array([0] + list(range(1, 2 + S)), np.int32)
        out.append(FlatWindow(pose=pose[h:h + 1].copy(), sb=np.zeros(0), lm=np.zeros(0), sc=sc, is_const=np.zeros(n_blocks, np.uint8),
                              order_block=order_block, order_group=order_group, n_tail=1 + S,
                              cp_idx=ce, cp_dat=cpd[ic], pr_idx=pe, pr_dat=prd[ip], pbg=wx.pbg, gw=wx.gw, base=wx.base,
                              meta=dict(frame=h)))
        kept.append([(7, None)] + [(1, s) for s in range(S)])
    return out, kept


def assemble_np(M, kept, priors):
    """IMUGNSSBase::AddMargInfo for a chain of M epochs in numpy: kept[e] = [(size, scalar id or None)], priors[e] = dict(A, b).
    Returns dict(ids (scalar ids in first-seen order), Hpp, HpN, rhs_p, HNN, rhsN)."""
    ids = []
    for e in range(M):
        for (s, k) in kept[e]:
            if s == 1 and k not in ids:
                ids.append(k)
    N = len(ids)
    Hpp, HpN, rhs_p, HNN, rhsN = np.zeros((M, 15, 15)), np.zeros((M, 15, N)), np.zeros((M, 15)), np.zeros((N, N)), np.zeros(N)
    for e in range(M):
        A, b = priors[e]["A"], priors[e]["b"]
        sel15, selN, o = [], [], 0
        for (s, k) in kept[e]:
            l = 6 if s == 7 else s
            if s == 7: sel15 += [(o + i, i) for i in range(6)]
            elif s == 9: sel15 += [(o + i, 6 + i) for i in range(9)]
            else: selN.append((o, ids.index(k)))
            o += l
        for (r, i) in sel15:
            rhs_p[e, i] += b[r]
            for (c, j) in sel15: Hpp[e, i, j] += A[r, c]
            for (c, j) in selN: HpN[e, i, j] += A[r, c]
        for (r, i) in selN:
            rhsN[i] += b[r]
            for (c, j) in selN: HNN[i, j] += A[r, c]
    return dict(ids=ids, Hpp=Hpp, HpN=HpN, rhs_p=rhs_p, HNN=HNN, rhsN=rhsN)


def plug_into_explicit(wx, wc, clocks_from):
    """The composite window's solution written into the explicit window's state (visual frames, hidden epochs, landmarks,
    ambiguities); the receiver clocks, which the composite topology eliminated, come from `clocks_from`."""
    m = wx.meta
    w2 = clocks_from.copy()
    P, B = w2.a["pose"].reshape(-1, 7), w2.a["sb"].reshape(-1, 9)
    pc, bc = wc.a["pose"].reshape(-1, 7), wc.a["sb"].reshape(-1, 9)
    for k, v in enumerate(m["vis"]):
        P[v] = pc[k]; B[v] = bc[k]
    hp, hs = wc.a["comp_pose"].reshape(-1, 7), wc.a["comp_sb"].reshape(-1, 9)
    for i, h in enumerate(m["hidden"]):
        P[h] = hp[i]; B[h] = hs[i]
    w2.a["lm"][...] = wc.a["lm"]
    w2.a["sc"][1:1 + m["S"]] = wc.a["sc"][1:1 + m["S"]]
    return w2


def explicit_cost(solver, wx, wc):
    """Cost of the EXPLICIT problem (no composite factor, no square root of a remainder in it) at the composite window's solution wc, the
    receiver clocks — which the composite topology eliminated — at their optimum for these states: every other block held constant, a
    few iterations on the clocks alone (the problem is linear in them).  The one yardstick two composite solutions can be compared by
    when they differ along a weakly determined direction.  Runs on the device (raw-factor kernels)."""
    from rtk_visual_inertial_navigation_amd.flat import default_options
    w0_ = plug_into_explicit(wx, wc, wx)
    a_ = {k: v.copy() for k, v in w0_.a.items()}
    ic = np.ones_like(a_["is_const"])
    first_sc = w0_.bid_sc(0)
    for b_id, g_id in zip(a_["order_block"], a_["order_group"]):          # the receiver clocks: the scalar blocks of elimination group 0 (bar the dummy anchor)
        if g_id == 0 and b_id > first_sc:
            ic[b_id] = 0
    a_["is_const"] = ic
    keep = ic[a_["order_block"]] == 0
    a_["order_block"], a_["order_group"] = a_["order_block"][keep], a_["order_group"][keep]
    w_ = FlatWindow(n_tail=0, proj_sqrt_info=w0_.proj_sqrt_info, proj_loss_a=w0_.proj_loss_a, pbg=w0_.pbg, gw=w0_.gw, base=w0_.base, meta=dict(w0_.meta), **a_)
    b_ = solver.BatchSolver([w_]); c_ = b_.solve(default_options(max_num_iterations=4), download=False)[0].final_cost; b_.close()
    return c_



def composite_window(wx, chains, reference_ordering=True):
    """The window over the visual frames only: projection factors, the gauge prior, the dummy, and one composite factor per gap
    built from chains[g] = assemble output (Hpp, HpN, rhs_p, HNN, rhsN over that gap's M epochs, ambiguity ids 0..S-1).
    reference_ordering: SWFOptimization::MyOrdering as it is (R/swf/swf_gnss.cpp:629-783: every other eligible speed-bias block
    in elimination group 0 — each composite factor then touches exactly one group-0 block); False: every frame block in a group of
    its own (rounds 1-4, when the engine refused a composite factor on a group-0 block: 90 more reduced dimensions at cfg3 size)."""
    a, m = wx.a, wx.meta
    K, M, S, vis = m["K_vis"], m["M"], m["S"], m["vis"]
    T = m["T"]
    pose_all, sb_all = a["pose"].reshape(-1, 7), a["sb"].reshape(-1, 9)
    pose = np.vstack([pose_all[vis], pose_all[T:T + 1]])                       # visual frames + the extrinsic
    sb = sb_all[vis]
    vmap = {v: k for k, v in enumerate(vis)}
    pi = a["proj_idx"].reshape(-1, 3).copy()
    pi[:, 0] = [vmap[int(p)] for p in pi[:, 0]]; pi[:, 1] = K
    sc = np.concatenate([[a["sc"][0]], a["sc"][1:1 + S]])                      # dummy + ambiguities
    F = wx.n_lm
    n_pose = K + 1
    bid_pose = lambda i: i; bid_sb = lambda i: n_pose + i; bid_lm = lambda i: n_pose + K + i; bid_sc = lambda i: n_pose + K + F + i
    is_const = np.zeros(n_pose + K + F + 1 + S, np.uint8); is_const[K] = 1
    if reference_ordering:
        roles = dict(dummy=bid_sc(0), landmarks=[bid_lm(f) for f in range(F)], speed_bias=[bid_sb(k) for k in range(K)],
                     poses=[bid_pose(k) for k in range(K)], extrinsics=[bid_pose(K)], rtk_ambiguities=[bid_sc(1 + s) for s in range(S)],
                     clocks=[], pr_corrections=[], prior_kept=[bid_pose(0), bid_sb(0)], parameter_head=[])
        ob_, og_, _ = my_ordering(roles, is_const)
        order_block, order_group = list(ob_), list(og_)
    else:
        order_block = [bid_sc(0)] + [bid_lm(f) for f in range(F)]; order_group = [0] * (1 + F)
        g = 1
        for k in range(K):
            for b in (bid_pose(k), bid_sb(k)):
                order_block.append(b); order_group.append(g); g += 1
        for s in range(S):
            order_block.append(bid_sc(1 + s)); order_group.append(g); g += 1
    comp = dict(M=[], N=[], idx=[], pose=[], sb=[], Hpp=[], HpN=[], rhs_p=[], HNN=[], rhsN=[], pre=[])
    pre_all = a["imu_pre"].reshape(-1, PRE_DOUBLES)
    for gi in range(K - 1):
        h0 = vis[gi] + 1
        ch = chains[gi]
        assert list(ch["ids"]) == list(range(S)), "every gap sees every satellite in this generator"
        comp["M"].append(M); comp["N"].append(S); comp["idx"].append([gi, gi, gi + 1, gi + 1] + [1 + s for s in range(S)])
        comp["pose"].append(pose_all[h0:h0 + M]); comp["sb"].append(sb_all[h0:h0 + M])
        for k_ in ("Hpp", "HpN", "rhs_p", "HNN", "rhsN"):
            comp[k_].append(ch[k_])
        comp["pre"].append(pre_all[vis[gi]:vis[gi] + M + 1])
    cat = lambda key: np.concatenate([np.asarray(x, np.float64).ravel() for x in comp[key]])
    return _own(FlatWindow(pose=pose, sb=sb, lm=a["lm"], sc=sc, is_const=is_const,
                      order_block=np.array(order_block, np.int32), order_group=np.array(order_group, np.int32), n_tail=0,
                      proj_idx=pi, proj_uv=a["proj_uv"], proj_sqrt_info=wx.proj_sqrt_info, proj_loss_a=wx.proj_loss_a,
                      sp_idx=np.array([0], np.int32), sp_w=a["sp_w"],
                      prior_nblk=np.array([2], np.int32), prior_dim=a["prior_dim"], prior_blk=np.array([bid_pose(0), bid_sb(0)], np.int32),
                      prior_J=a["prior_J"], prior_r0=a["prior_r0"], prior_x0=a["prior_x0"],
                      comp_M=np.array(comp["M"], np.int32), comp_N=np.array(comp["N"], np.int32), comp_idx=np.concatenate([np.array(i, np.int32) for i in comp["idx"]]),
                      comp_pose=cat("pose"), comp_sb=cat("sb"), comp_pose_lin=cat("pose"), comp_sb_lin=cat("sb"),
                      comp_Hpp=cat("Hpp"), comp_HpN=cat("HpN"), comp_rhs_p=cat("rhs_p"), comp_HNN=cat("HNN"), comp_rhsN=cat("rhsN"), comp_pre=cat("pre"),
                      pbg=wx.pbg, gw=wx.gw, base=wx.base, meta=dict(K=K, M=M, N=S, F=F)))


def _explicit_job(args):
    kw, seed = args
    return explicit_window(seed=seed, **kw)[0]


def explicit_windows(n, seed0=900, pool=True, **kw):
    """n explicit windows (seeds seed0 ..), generated in a process pool when asked to — call it BEFORE the process touches HIP."""
    jobs = [(kw, seed0 + i) for i in range(n)]
    if pool and n > 2:
        import multiprocessing as mp, os
        with mp.get_context("fork").Pool(min(n, max(1, (os.cpu_count() or 2) - 2), 48)) as p:
            return p.map(_explicit_job, jobs)
    return [_explicit_job(j) for j in jobs]


def composite_batch(solver, wxs, timing=None, reference_ordering=True):
    """The device-side construction of the reference's topology for a list of explicit windows (what bench.py's rtk_topology /
    composite legs and tools/prof/gpu_comp_prof.py time): (1) every GNSS epoch of every window as ONE batch through
    swf_batch_marginal_priors (GnssPreprocess, R/swf/swf_gnss.cpp:504-532), (2) AddMargInfo's bookkeeping on the host
    (swf_composite_assemble, R/factor/gnss_imu_factor.cpp:245-352), (3) the composite windows.  Needs a GPU (the product has no CPU path)."""
    import time
    ek = [epoch_windows(wx) for wx in wxs]
    ews = [e for (es, _) in ek for e in es]
    tm = {}
    t0 = time.perf_counter()
    pri = solver.marginal_priors(ews, 1e-8, solver.BatchSolver.PRIOR_EIGEN, timing=tm)
    t_wrap = time.perf_counter() - t0
    t0 = time.perf_counter()
    wins, o = [], 0
    for wx, (es, kept) in zip(wxs, ek):
        K_vis, M = wx.meta["K_vis"], wx.meta["M"]
        chains = []
        for g in range(K_vis - 1):
            blocks = {}
            eps = []
            for e in range(g * M, (g + 1) * M):
                eps.append(dict(kept=[(sz, blocks.setdefault(k, np.zeros(1)) if sz == 1 else None) for (sz, k) in kept[e]], A=pri[o + e]["A"], b=pri[o + e]["b"]))
            c = solver.composite_assemble(eps)
            inv = {id(v): k for k, v in blocks.items()}
            c["ids"] = [inv[id(b)] for b in c["keys"]]
            chains.append(c)
        o += len(es)
        wins.append(composite_window(wx, chains, reference_ordering=reference_ordering if isinstance(reference_ordering, bool) else reference_ordering[len(wins)]))
    if timing is not None:
        timing.update(gnss_epochs=len(ews), epoch_priors_s=tm.get("c_abi_call_s"), epoch_priors_with_python_marshalling_s=t_wrap, host_assemble_s=time.perf_counter() - t0)
    return wins


def warm_start_probe(solver, wins, opt, opt_long, seed=5, sigma_p=0.05, sigma_v=0.05, sigma_q=2e-3):
    """Why the cold windows above do not converge inside the yaml's 8 iterations, and what the estimator actually sees: the generator
    perturbs EVERY state of the window (cost 2e7 -> 9e2 in the first step) and the composite factors re-linearise their hidden epochs by
    back-substitution, a block-coordinate scheme whose tail is linear (the reference's too: the oracle walks the same path); the
    estimator optimises a window whose states are the previous frame's converged solution plus ONE new frame predicted from the IMU.
    The probe: converge the windows (opt_long), then move only the newest frame's pose / velocity by an IMU-prediction-sized error
    (5 cm, 5 cm/s, 2 mrad) and solve again with the 8-iteration budget.  Returns (converged within the budget, mean iterations used,
    mean iterations of the cold solve to its own termination)."""
    ws = [w.copy() for w in wins]
    bs = solver.BatchSolver(ws)
    cold = bs.solve(opt_long)
    bs.close()
    rng = np.random.default_rng(seed)
    for w in ws:
        K = w.meta["K"]
        pose = w.a["pose"].reshape(-1, 7); sb = w.a["sb"].reshape(-1, 9)
        pose[K - 1, :3] += rng.normal(0, sigma_p, 3)
        dq = np.concatenate([rng.normal(0, sigma_q, 3) / 2, [1.0]]); dq /= np.linalg.norm(dq)
        pose[K - 1, 3:] = synth.q_mul(pose[K - 1, 3:], dq)
        sb[K - 1, :3] += rng.normal(0, sigma_v, 3)
    bs = solver.BatchSolver(ws)
    warm = bs.solve(opt)
    bs.close()
    return (int(sum(s.termination in (1, 2, 3) for s in warm)), float(np.mean([s.num_iterations for s in warm])),
            float(np.mean([s.num_iterations for s in cold])))
