"""BASELINE cfg5 stress window, solve only (workload for rocprofv3; no oracle, no event brackets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
bs = solver.BatchSolver([synth.make_window(5)])
for _ in range(3):
    bs.reset_state(); bs.solve_async(default_options()); bs.sync()
print("cfg5 done", bs.summaries()[0].num_iterations)
