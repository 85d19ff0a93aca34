import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
w = synth.make_window(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
bs = solver.BatchSolver([w])
for _ in range(3):
    bs.reset_state(); bs.solve(default_options(step_mode=1), download=False)
out = (C.c_ulonglong * 64)()
solver.lib().swf_debug_chol_stamps(out)
s = list(out)
print("n_red", bs.dims(0)["n_red"])
print("factor total ticks", s[1] - s[0], "backward", s[2] - s[1], "(s_memtime ticks; 100 MHz const clock => x10 ns)")
print("v1: mfma+init / rr: publish+sync", s[8], "| v1: transpose / rr: diag factor+inverse", s[9], "| v1: diag / rr: panel", s[10], "| v1: trsm / rr: trailing", s[11])
print("rr2: start->A0 (load)", s[3] - s[0], "| factor loop", s[1] - s[3], "| export wait", s[4] - s[1], "| backward", s[2] - s[4])
print("rr2 sums: pivot factor+inverse", s[9], "| wait B (trailing of others)", s[8], "| B->C (panel)", s[10], "| C->A (lookahead diag)", s[11])
print("rr3 tile wave 2 (ticks from kernel start): tile list", s[16]-s[0], "| loads issued + diag tiles", s[17]-s[0], "| transposed", s[18]-s[0], "| A_0", s[19]-s[0], "| loop end", s[20]-s[0], "| exported", s[21]-s[0])
print("rr3 arrival at B_j of the profiled step, per wave, ticks after the pivot wave STARTED tile j (wave 0 = its own end; 0 = idle wave):", [int(x) - int(s[48]) if x else 0 for x in s[32:48]])
print("rr3 tile wave 2, summed over the steps: top-of-step -> at B", s[28], "| B wait", s[26], "| panel phase", s[27], "| C wait", s[22], "| mask read", s[23], "| diagonal terms", s[24], "| trailing", s[25])
print("rr3 pivot wave past B_j, ticks after kernel start:", [int(x) - int(s[0]) if x else 0 for x in s[49:64]])
print("rr3 step lengths B_j -> B_j+1:", [int(s[50 + j]) - int(s[49 + j]) for j in range(14) if s[50 + j] and s[49 + j]])
