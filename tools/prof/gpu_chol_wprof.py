"""k_chol_rr3 on one window: per-wave time stamps of the chosen steps (needs a -DSWF_PROFILE_CHOLW build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
w = synth.make_window(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
steps = [int(a) for a in sys.argv[2:]] or [0, 2, 5, 8, 11]
bs = solver.BatchSolver([w])
for _ in range(2):
    bs.reset_state(); bs.solve(default_options(step_mode=1), download=False)
print("n_red", bs.dims(0)["n_red"], "(ticks of s_memtime; relative to the pivot wave's start of the step's pivot)")
names0 = ["pivot start", "pivot end", "past B", "past C"]
namest = ["past B", "panel done", "past C", "mask read", "diag terms", "trailing", "at B+1", "past B+1"]
for st in steps:
    solver.lib().swf_debug_chol_wstep(st)
    bs.reset_state(); bs.solve(default_options(step_mode=1, max_num_iterations=1), download=False)
    out = (C.c_ulonglong * 128)()
    solver.lib().swf_debug_chol_wstamps(out)
    s = [int(x) for x in out]
    t0 = s[0]
    if st < 0:
        t0 = min(x for x in s if x)
        print("== load phase (ticks after the first stamp): per tile wave: roles known, loads issued, transposed (at A_0), past A_0")
        for wv in range(16):
            r = s[wv * 8: wv * 8 + 4]
            if any(r): print("  wave %2d:" % wv, [x - t0 if x else 0 for x in r])
        continue
    print("== step", st)
    print("  wave 0 :", ", ".join("%s %d" % (names0[k], s[k] - t0) for k in range(4) if s[k]))
    print("  wave 1 : inverse end", s[8 + 1] - t0 if s[9] else 0)
    try:
        cs = (C.c_ulonglong * 32)()
        solver.lib().swf_debug_chol_cstamps(cs)
        cs = [int(x) for x in cs]
        print("  pivot wave, column starts  :", [c - t0 for c in cs[:16]])
        print("  inverse wave, column starts:", [c - t0 for c in cs[16:32]], "(the last = rows written)")
    except AttributeError:
        pass
    try:
        ps = (C.c_ulonglong * 16)()
        solver.lib().swf_debug_chol_pstamps(ps)
        ps = [int(x) for x in ps]
        print("  owner of the next pivot tile: past B, A operand ready, panel entered, X done, diagonal tile published, panel published, at C, past C:", [x - t0 if x else 0 for x in ps[:8]])
    except AttributeError:
        pass
    for wv in range(2, 16):
        r = s[wv * 8: wv * 8 + 8]
        if not any(r): continue
        print("  wave %2d:" % wv, ", ".join("%s %d" % (namest[k], r[k] - t0) for k in range(8) if r[k]))
