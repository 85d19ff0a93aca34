"""Phase stamps of k_comp_elim<24> for the first composite factor of one cfg3-size reference-topology window (a -DSWF_PROFILE_CHOL build):
   SWF_EXTRA_FLAGS=-DSWF_PROFILE_CHOL python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)"; python tools/prof/gpu_comp_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtk_topology_gen as rt
S_ = int(sys.argv[1]) if len(sys.argv) > 1 else 10
wxs = rt.explicit_windows(1, K_vis=20, M=4, F=300, S=S_, pool=False)
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options
wins = rt.composite_batch(solver, wxs)
bs = solver.BatchSolver(wins)
for _ in range(3):
    bs.reset_state(); bs.solve(default_options(max_num_iterations=3), download=False)
out = (C.c_ulonglong * 64)()
solver.lib().swf_debug_chol_stamps(out)
s = [int(x) for x in out]
t0 = s[32]
print("k_comp_elim, factor 0, %d ambiguities (cycles from kernel entry):" % S_)
print("  link loops start:", [s[33 + k] - t0 for k in range(5)])
print("  epoch 0: J^T J accumulated", s[40] - s[34], "| + GNSS prior", s[41] - s[40], "| inverse", s[42] - s[41], "| T products + Schur update", s[43] - s[42])
print("  links done", s[44] - t0, "| remainder staged", s[45] - s[44], "| square root", s[46] - s[45], "| end", s[47] - s[46], "| total", s[47] - t0)
