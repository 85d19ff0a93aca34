"""k_marg_rescue at cfg5 size (tail 263 of a 278-dimension reduced system... and the full 440-dimension marginalisation window): wall time of
swf_batch_marginalize with the rescue path forced, against the regular path's result."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options
import cfg5_marg_gen as cg
full = cg.make_full((40, 1000, 20, None))
wm, head = cg.marginalisation_window(full)
res = {}
for force in (False, True):
    if force: os.environ["SWF_FORCE_MARG_RESCUE"] = "1"
    else: os.environ.pop("SWF_FORCE_MARG_RESCUE", None)
    bs = solver.BatchSolver([wm.copy()])
    sm = bs.solve(default_options(step_mode=1), download=False)[0]
    ts = []
    for it in range(4):
        t0 = time.perf_counter(); bs.marginalize(1e-8, solver.BatchSolver.PRIOR_EIGEN); g = bs.get_prior(0); ts.append(time.perf_counter() - t0)
    res[force] = g
    print("rescue forced" if force else "regular path ", "tail", sm.tail_dim, "n_red", sm.reduced_dim, "ms per call (eigen form, incl. download):", " ".join("%.2f" % (1e3 * t) for t in ts), "rank", g["rank"])
    bs.close()
a, b = res[False], res[True]
print("A: |rescue - regular| / |A| = %.2e   b: %.2e   J^T J vs A (rescue): %.2e" % (np.abs(a["A"] - b["A"]).max() / np.abs(a["A"]).max(), np.abs(a["b"] - b["b"]).max() / np.abs(a["b"]).max(),
      np.abs(b["J"].T @ b["J"] - b["A"]).max() / np.abs(b["A"]).max()))
