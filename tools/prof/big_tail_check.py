import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
import time
for K in (39, 41):
    wx = synth.make_window(3, K=K, F=40, S=6, seed=35, head="frames")
    bs = solver.BatchSolver([wx.copy()])
    sm = bs.solve(default_options(step_mode=1), download=False)[0]
    print("K", K, "tail", sm.tail_dim, "n_red", sm.reduced_dim, "termination", sm.termination)
    bs.marginalize(1e-8, solver.BatchSolver.PRIOR_CHOLESKY); c = bs.get_prior(0)
    t0 = time.perf_counter(); bs.marginalize(1e-8, solver.BatchSolver.PRIOR_EIGEN); g = bs.get_prior(0); t1 = time.perf_counter()
    A = g["A"]
    print("  eigen %.2f ms rank %d/%d  J^T J - A %.2e  J^T r0 - b %.2e  A==chol A %s  eig vs numpy %.2e" % (1e3 * (t1 - t0), g["rank"], g["n"],
          np.abs(g["J"].T @ g["J"] - A).max() / np.abs(A).max(), np.abs(g["J"].T @ g["r0"] - g["b"]).max() / np.abs(g["b"]).max(), np.array_equal(A, c["A"]),
          np.abs(np.sort(g["eig"]) - np.linalg.eigvalsh(A)).max() / np.abs(A).max()))
    bs.close()
