"""Times a cfg3-size composite window with the reference's eigen root of the remainders (swf_options::composite_root = 1) and prints the
end state's distance from the pivoted-root solve.  python tools/prof/eigroot_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rtk_topology_gen as rt
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options
wxs = rt.explicit_windows(1, seed0=900, pool=False, K_vis=20, M=4, F=300, S=10)
w = rt.composite_batch(solver, wxs)[0]
out = {}
for root in (0, 1):
    c = w.copy(); bs = solver.BatchSolver([c]); opt = default_options(composite_root=root)
    for _ in range(2): bs.reset_state(); bs.solve_async(opt); bs.sync()
    t0 = time.perf_counter()
    for _ in range(5): bs.reset_state(); bs.solve_async(opt); bs.sync()
    dt = (time.perf_counter() - t0) / 5
    bs.reset_state(); sm = bs.solve(opt)[0]; bs.close()
    out[root] = (c, sm)
    print("root", root, "%.1f us per iteration" % (1e6 * dt / sm.num_iterations), "final cost %.9e" % sm.final_cost)
print("poses apart", np.abs(out[0][0].a["pose"] - out[1][0].a["pose"]).max())
