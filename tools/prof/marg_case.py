import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
rng = np.random.default_rng(3)
# regenerate the sweep's random stream up to the failing cases
import cfg5_marg_gen as mg
for t in range(26):
    vi = rng.random() < 0.3
    K = int(rng.integers(3, 13)); F = int(rng.integers(max(4, K), 60)); S = 0 if vi else int(rng.integers(4, 11))
    kind = rng.choice(["ambiguities", "frames", "globalmarge"]) if S else rng.choice(["frames", "globalmarge"])
    seed = int(rng.integers(1, 10 ** 6))
    if t not in (14, 22, 25): continue
    if kind == "globalmarge":
        full = synth.make_window(3 if S else 2, K=K + 1, F=F, S=S, prior="gauge", seed=seed); w, _ = mg.marginalisation_window(full)
    else:
        w = synth.make_window(3 if S else 2, K=K, F=F, S=S, seed=seed, head=str(kind))
    so, eo = ob.solve(w.copy(), default_options(step_mode=1))
    bs = solver.BatchSolver([w.copy()]); sg = bs.solve(default_options(step_mode=1))[0]
    n = sg.tail_dim; bs.marginalize(1e-8, 0); g = bs.get_prior(0); bs.close()
    o = ob.marginalize(eo["S"], eo["rhs"], n)
    S_ = eo["S"].astype(np.longdouble); m = S_.shape[0] - n
    ev = np.linalg.eigvalsh(eo["S"][:m, :m])
    # extended-precision reference through a scaled solve
    d = np.sqrt(np.diag(S_)[:m]); Ss = S_[:m, :m] / np.outer(d, d)
    X = np.linalg.solve(Ss.astype(np.float64), (S_[:m, m:] / d[:, None]).astype(np.float64)).astype(np.longdouble)
    for _ in range(3):
        R = (S_[:m, m:] / d[:, None]) - Ss @ X
        X = X + np.linalg.solve(Ss.astype(np.float64), R.astype(np.float64)).astype(np.longdouble)
    Aref = (S_[m:, m:] - (S_[m:, :m] / d[None, :]) @ X).astype(np.float64)
    sc = np.abs(Aref).max()
    print(t, kind, "n", n, "m", m, "eig(S_mm) min %.2e max %.2e" % (ev[0], ev[-1]), "count < 1e-8:", int((ev < 1e-8).sum()),
          "| device vs ref %.1e, oracle vs ref %.1e, device vs oracle %.1e" % (np.abs(g["A"] - Aref).max() / sc, np.abs(o["A"] - Aref).max() / sc, np.abs(g["A"] - o["A"]).max() / sc))
