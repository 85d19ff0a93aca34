import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
from rtk_visual_inertial_navigation_amd.ordering import my_ordering
base = synth.make_window(3, K=6, F=20, S=5, seed=311)
w = base.copy()
pi, uv = w.a["proj_idx"].reshape(-1, 3), w.a["proj_uv"].reshape(-1, 2)
keep, seen = [], {}
for q, (p_, e_, l_) in enumerate(pi):
    c = seen.get(int(l_), 0); seen[int(l_)] = c + 1
    if c == 0 or (c == 1 and l_ % 2 == 0): keep.append(q)
w.a["proj_idx"] = pi[keep].ravel().copy(); w.a["proj_uv"] = uv[keep].ravel().copy()
for strat in (0, 1):
    wo, wg = w.copy(), w.copy()
    so, eo = ob.solve(wo, default_options(strategy=strat), export=True)
    bs = solver.BatchSolver([wg]); sg = bs.solve(default_options(strategy=strat))[0]; bs.close()
    print("strategy", strat, "cond(S) %.2e" % np.linalg.cond(eo["S"]))
    for it, (a, b) in enumerate(zip(so.rows(), sg.rows())):
        print("  it %d cost %.12e %.12e rel %.1e  radius %.3e %.3e ok %d %d step %.6e %.6e" % (it, a["cost"], b["cost"], abs(a["cost"] - b["cost"]) / a["cost"], a["trust_region_radius"], b["trust_region_radius"], a["step_is_successful"], b["step_is_successful"], a["step_norm"], b["step_norm"]))
