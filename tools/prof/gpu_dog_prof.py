import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ws = bench.make_windows(4, [synth.BASE_SEED + 4 + i for i in range(B)])
bs = solver.BatchSolver(ws)
for _ in range(3):
    bs.reset_state(); bs.solve(default_options(max_num_iterations=1), download=False)
out = (C.c_ulonglong * 16)()
solver.lib().swf_debug_dog_stamps(out)
s = list(out)
names = ["header", "cost+aux loads", "gmax", "scalars loop", "reduce", "bookkeeping", "step loop", "Plus", "reduce2+store"]
print("windows", B, " | ".join("%s %d" % (n, v) for n, v in zip(names, s)), "| total", sum(s[:9]))
