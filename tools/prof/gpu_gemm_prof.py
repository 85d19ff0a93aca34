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
    bs.reset_state(); bs.solve(default_options(step_mode=1), download=False)
out = (C.c_ulonglong * 16)()
solver.lib().swf_debug_gemm_stamps(out)
s = list(out)
print("windows", B, "| producer wave 0: land + sums", s[0], "inverse", s[1], "wait for the buffer", s[2], "Z + write", s[3], "total", s[6],
      "| consumer wave 0: wait for chunks", s[10], "mfma", s[9], "S-direct tail", s[11], "(core clock cycles, block 0)")
