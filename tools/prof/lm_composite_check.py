import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import composite_gen as cg
import idepth_gen as ig
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
rng = np.random.default_rng(5)
wins = {"composite": cg.make_window(rng, 5, 3, 6, F=30), "composite_mid": cg.make_window(rng, 4, 5, 6, F=0, mid=True),
        "idepth": ig.convert_short_tracks(synth.make_window(2, K=8, F=40, S=0, seed=3), max_track=8),
        "var_ex": synth.with_variable_extrinsic(synth.make_window(3, K=6, F=30, S=5, seed=8))}
for name, w in wins.items():
    for jac in (0, 1):
        opt = default_options(max_num_iterations=12, strategy=1, jacobi_scaling=jac)
        opt.initial_trust_region_radius = 30.0
        wo, wg = w.copy(), w.copy()
        so, _ = ob.solve(wo, opt, export=False)
        bs = solver.BatchSolver([wg]); sg = bs.solve(opt)[0]; bs.close()
        ro, rg = so.rows(), sg.rows()
        same = [r["step_is_successful"] for r in ro] == [r["step_is_successful"] for r in rg]
        rel = max(abs(a["cost"] - b["cost"]) / abs(a["cost"]) for a, b in zip(ro, rg))
        print(name, "jacobi", jac, "term", so.termination, sg.termination, "iters", len(ro) - 1, len(rg) - 1, "accept seq equal", same, "rejected", sum(1 for r in ro if not r["step_is_successful"]), "max rel cost diff %.1e" % rel,
              "pose diff %.1e" % np.abs(wo.a["pose"] - wg.a["pose"]).max())
