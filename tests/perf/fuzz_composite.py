"""Randomised GPU-vs-oracle parity sweep over the reference's own topology (composite IMU-GNSS factors; not part of the test suite,
run on a GPU box):   python tests/perf/fuzz_composite.py [n_cases] [seed]
Random shapes — visual frames, hidden GNSS epochs per gap, features, ambiguities (both instantiations of k_comp_elim, cliques of
every size class incl. k_clique_tall / k_clique_big), MyOrdering as it is or every frame block in a group of its own.  Per window:
the linearisation and the reduced system against the oracle's, L L^T = S, the yaml's 8 iterations (same accept / reject sequence,
first cost to rounding), both solvers to their own termination (device cost not above the oracle's), and the heterogeneous batch
== the single windows, bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import rtk_topology_gen as rt
import composite_parity as cp
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
rel = lambda a, b: np.abs(a - b).max() / max(1e-300, np.abs(b).max())

shapes, wxs, ords = [], [], []
for t in range(N):
    K = int(rng.integers(3, 13)); M = int(rng.integers(1, 6)); S = int(rng.integers(4, 41)); F = int(rng.integers(max(12, 2 * K), 160))
    if K + (K - 1) * M > 60: M = max(1, (60 - K) // (K - 1))
    kw = dict(K_vis=K, M=M, F=F, S=S, seed=int(rng.integers(1, 10 ** 6)))
    shapes.append(kw); wxs.append(rt.explicit_window(**kw)[0]); ords.append(bool(rng.random() < 0.7))
wins = rt.composite_batch(solver, wxs, reference_ordering=ords)
bad = 0
singles, offs = [], []
worst = dict(first=0.0, diffs=0.0, states=0.0, states2=0.0, lo=0.0, hi=0.0)
for t, (kw, w, ro_) in enumerate(zip(shapes, wins, ords)):
    msg, info = [], ""
    try:
        so, eo = ob.solve(w.copy(), default_options(step_mode=1))
        bs = solver.BatchSolver([w.copy()]); bs.solve(default_options(step_mode=1))
        Sg, rg, Lg = bs.export_reduced(0); g, dg, y = bs.export_vectors(0); n_red = bs.dims(0)["n_red"]
        bs.close()
        e_lin = (rel(g, eo["grad"]), rel(Sg, eo["S"]), rel(rg, eo["rhs"]))
        if n_red != eo["n_red"]: msg.append("n_red %d vs %d" % (n_red, eo["n_red"]))
        # (S to 1e-10 — seen: 4e-14; the gradient and the reduced right-hand side to 1e-9: a noise direction k the oracle's eigen square root keeps
        # and the pivoted root drops, see below, puts v_k (v_k^T rhs) into J^T r — seen: 1.2e-10 / 3e-10 of the vector in one window of 600, with
        # 34 ambiguities and a first-cost offset of 3.8e-7; 1e-12 typically)
        elif e_lin[0] > 1e-9 or e_lin[1] > 1e-10 or e_lin[2] > 1e-9: msg.append("linearisation grad %.1e S %.1e rhs %.1e" % e_lin)
        if rel(Lg @ Lg.T, Sg) > 1e-12: msg.append("LLt %.1e" % rel(Lg @ Lg.T, Sg))
        dp, fin, dc0 = {}, "", 0.0
        for iters in (8, 40):
            # two-sided parity statements (tests/composite_parity.py): (A) against the oracle's noise-free restatement, (B) against the literal
            # reference, whose cost may exceed the device's by no more than the noise terms it counted as kept — for BOTH square roots
            orc = cp.oracle_solves(w, iters)
            (sn, wn), (sl, wl), noise, count = orc
            for root in (0, 1):
                wd = w.copy()
                bs = solver.BatchSolver([wd]); sd = bs.solve(default_options(max_num_iterations=iters, composite_root=root))[0]; bs.close()
                rep = {}
                # (40 iterations: the literal oracle's noise can flip a late accept / reject decision; the noise-free one's may not)
                v = cp.check(sd, wd, orc, decisions_vs_literal=(iters == 8), report=rep, n_dec=9)
                if any(m.startswith("A: end states") for m in v):
                    # end states apart along a weakly determined direction (few satellites: the global position hangs on the gauge prior, 1e-3):
                    # the yardstick is the EXPLICIT problem's cost at both end states — no composite factor, no square root in it — two-sidedly
                    ce_d, ce_n = rt.explicit_cost(solver, wxs[t], wd), rt.explicit_cost(solver, wxs[t], wn)
                    if abs(ce_d - ce_n) <= 1e-6 * ce_n + 1e-6:
                        v = [m for m in v if not m.startswith("A: end states")]
                        if root == 0 and iters == 40: fin += "; explicit-problem cost at the end states: device %.9e, noise-free oracle %.9e" % (ce_d, ce_n)
                for m in v: msg.append("%s root, %d iterations: %s" % ("eigen" if root else "pivoted", iters, m))
                worst["first"] = max(worst["first"], rep["first"]); worst["diffs"] = max(worst["diffs"], rep["diffs"]); worst["states"] = max(worst["states"], rep["states"])
                worst["states2"] = max(worst["states2"], rep["states2"]); worst["lo"] = min(worst["lo"], rep["lo"]); worst["hi"] = max(worst["hi"], rep["hi"])
                if root == 0:
                    rd = sd.rows()
                    dc0 = (rd[0]["cost"] - sl.rows()[0]["cost"]) / sl.rows()[0]["cost"]
                    if iters == 8: offs.append(dc0); singles.append((wd, [r["cost"] for r in rd]))
                    dp[iters] = rep["states"]
                    if iters == 40: fin += " end: termination %d / %d (noise-free) / %d (literal) after %d / %d / %d iterations; literal oracle kept %d noise terms, %.3e" % (
                        sd.termination, sn.termination, sl.termination, sd.num_iterations, sn.num_iterations, sl.num_iterations, count, noise)
                fin += " | %s%d: first %.1e diffs %.1e states %.1e/%.1e B[%.1e, %.1e]" % ("e" if root else "p", iters, rep["first"], rep["diffs"], rep["states"], rep["states2"], rep["lo"], rep["hi"])
        info = "n_red %d lin %.0e/%.0e/%.0e first cost rel %+.1e poses apart %.1e (8 its) %.1e (40) %s" % (n_red, *e_lin, dc0, dp[8], dp[40], fin)
    except Exception as e:
        msg.append("exception " + repr(e)[:200])
        if len(singles) <= t: singles.append(None)
    print(t, kw, "MyOrdering" if ro_ else "own-groups", info, "OK" if not msg else "FAIL " + "; ".join(msg), flush=True)
    bad += bool(msg)
batch = [w.copy() for w in wins]
bs = solver.BatchSolver(batch); sms = bs.solve(default_options(max_num_iterations=8)); bs.close()
for i, (sg, wb, sm) in enumerate(zip(singles, batch, sms)):
    if sg is None: continue
    same = [r["cost"] for r in sm.rows()] == sg[1] and all(np.array_equal(sg[0].a[k], wb.a[k]) for k in ("pose", "sb", "lm", "sc", "comp_pose", "comp_sb"))
    if not same: print("batch != single for case", i); bad += 1
offs = np.array(offs)
print("first-cost offsets (device - oracle) / oracle: max |.| %.1e, %d of %d above 1e-9 in size, %d of those positive" % (np.abs(offs).max(), (np.abs(offs) > 1e-9).sum(), offs.size, (offs > 1e-9).sum()))
print("worst seen: first cost %.2e, cost differences %.2e of the decrease, end states %.2e (poses) / %.2e (speed-bias, scalars) against the noise-free oracle; literal-oracle bound margins lo %.2e hi %.2e"
      % (worst["first"], worst["diffs"], worst["states"], worst["states2"], worst["lo"], worst["hi"]))
print("fuzz_composite: %d cases, %d failures" % (N, bad))
sys.exit(1 if bad else 0)
