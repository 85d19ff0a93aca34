# eigen square root of the cfg5 marginalisation prior (263 dimensions) on the device: wall time and residuals,
# block Jacobi over many workgroups (k_marg_bj)
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options
import cfg5_marg_gen as cg
full = cg.make_full((40, 1000, 20, None))
wm, head = cg.marginalisation_window(full)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bs = solver.BatchSolver([wm.copy() for _ in range(nb)])
sm = bs.solve(default_options(step_mode=1), download=False)[0]
print("tail", sm.tail_dim, "n_red", sm.reduced_dim, "windows", nb)
for form, name in ((solver.BatchSolver.PRIOR_EIGEN, "eigen"), (solver.BatchSolver.PRIOR_CHOLESKY, "cholesky")):
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        bs.marginalize(1e-8, form)
        g = bs.get_prior(0)          # synchronises
        ts.append(time.perf_counter() - t0)
    A, J, b, r0 = g["A"], g["J"], g["b"], g["r0"]
    print("%-9s ms per call (incl. get_prior download): %s | rank %d | |J^T J - A| / |A| %.2e | |J^T r0 - b| / |b| %.2e" %
          (name, " ".join("%.2f" % (1e3 * t) for t in ts), g["rank"], np.abs(J.T @ J - A).max() / np.abs(A).max(), np.abs(J.T @ r0 - b).max() / np.abs(b).max()))
    if form == solver.BatchSolver.PRIOR_EIGEN:
        lam = np.linalg.eigvalsh(A)
        print("          eigenvalues vs numpy eigvalsh: max rel dev %.2e (of the largest)" % (np.abs(np.sort(g["eig"]) - lam).max() / lam.max()))
bs.close()
if os.environ.get("MG_STAMPS"):
    import ctypes as C
    out = (C.c_ulonglong * 64)()
    solver.lib().swf_debug_chol_stamps(out)
    t = list(out)
    print("d_pivoted_chol (core clock ticks, wave 0): candidate ranks", t[40], "| pool rows", t[41], "| pivot steps", t[42], "| copy-out", t[43])
    print("k_marg_bj workgroup 1 of sweep 1, block step 0 (core clock ticks): load", t[51] - t[50], "| inner steps", t[52] - t[51], "| store", t[53] - t[52], "| inner steps by part: reads", t[54], "sums", t[55], "angle", t[56], "update", t[57], "barrier", t[58])
