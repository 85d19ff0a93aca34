import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cfg5_marg_gen as mg
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
full = mg.make_full((40, 1000, 20, synth.BASE_SEED + 5))
wm, head = mg.marginalisation_window(full)
bs = solver.BatchSolver([wm.copy()])
sm = bs.solve(default_options(step_mode=1), download=False)[0]
for form, name in ((1, "cholesky"), (0, "eigen")):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); bs.marginalize(1e-8, form); g = bs.get_prior(0); ts.append(time.perf_counter() - t0)
    print(name, "tail", sm.tail_dim, "rank", g["rank"], "ms", [round(1e3 * t, 2) for t in ts], flush=True)
bs.close()
