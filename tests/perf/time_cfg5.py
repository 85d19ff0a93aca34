import time, numpy as np
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
w = synth.make_window(5)
bs = solver.BatchSolver([w]); opt = default_options(); ts = []
for _ in range(8):
    bs.reset_state(); bs.sync(); t0 = time.perf_counter(); bs.solve_async(opt); bs.sync(); ts.append(time.perf_counter() - t0)
print("cfg5 solve ms min/median", 1e3 * min(ts), 1e3 * float(np.median(ts)), "iterations", bs.summaries()[0].num_iterations)
