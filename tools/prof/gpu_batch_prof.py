"""Batch-only workload for rocprofv3 (no single-window runs, no event brackets):
   rocprofv3 --kernel-trace --stats -- python tools/prof/gpu_batch_prof.py [windows] [solves]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ws = bench.make_windows(4, [synth.BASE_SEED + 4 + i for i in range(B)])
bs = solver.BatchSolver(ws)
opt = default_options()
for _ in range(n):
    bs.reset_state(); bs.solve_async(opt); bs.sync()
print("done", B, n, [s.num_iterations for s in bs.summaries()[:4]])
