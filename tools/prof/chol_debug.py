import os, sys, glob
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
from golden.make_golden import load_case
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    w, gold = load_case(f)
    bs = solver.BatchSolver([w.copy()])
    bs.solve(default_options(step_mode=1))
    S, rhs, L = bs.export_reduced(0)
    n = S.shape[0]
    Lr = np.linalg.cholesky(S)
    E = np.abs(L - Lr) / (np.abs(Lr).max())
    T = (n + 15) // 16
    print(os.path.basename(f), "n", n, "max rel err", E.max(), "LLt", np.abs(L @ L.T - S).max() / np.abs(S).max())
    for I in range(T):
        print("  ", " ".join("%8.1e" % E[16 * I:16 * I + 16, 16 * J:16 * J + 16].max() for J in range(I + 1)))
    bs.close()
