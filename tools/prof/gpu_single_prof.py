"""Single-window latency workload for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
w = synth.make_window(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
bs = solver.BatchSolver([w])
bs.enable_timing(1)
opt = default_options()
ts = []
for _ in range(30):
    bs.reset_state(); bs.solve_async(opt); bs.sync(); ts.append(bs.timing()["total_ms"])
print("solve ms min/median", min(ts), sorted(ts)[len(ts)//2], "iterations", bs.summaries()[0].num_iterations)
