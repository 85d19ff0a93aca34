import os, sys
ROOT = '/root/repo' if os.path.exists('/root/repo/tests') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
for kw, head in ((dict(config_id=3, K=10, F=28, S=5, seed=390436, doppler=True), None), (dict(config_id=3, K=8, F=40, S=12, seed=3984, head="ambiguities"), None)):
    w = synth.make_window(**kw)
    for hd in (False, True):
        wv = synth.with_variable_extrinsic(w, head=hd)
        wo, wg = wv.copy(), wv.copy()
        so, eo = ob.solve(wo, default_options(), export=True)
        bs = solver.BatchSolver([wg]); sg = bs.solve(default_options())[0]; bs.close()
        d = np.abs(wg.a["pose"] - wo.a["pose"]).reshape(-1, 7)
        cS = np.linalg.cond(eo["S"])
        ro, rg = so.rows(), sg.rows()
        for it_, (a_, b_) in enumerate(zip(ro, rg)):
            print("   it %d cost %.10e %.10e  radius %.3e %.3e ok %d %d step %.4e %.4e" % (it_, a_["cost"], b_["cost"], a_["trust_region_radius"], b_["trust_region_radius"], a_["step_is_successful"], b_["step_is_successful"], a_["step_norm"], b_["step_norm"]))
        print(kw["seed"], "head", hd, "cond(S) %.2e" % cS, "max pose diff %.2e" % d.max(), "extrinsic diff %.2e" % d[-1].max(), "cost o/g %.12e %.12e" % (so.final_cost, sg.final_cost),
              "its", so.num_iterations, sg.num_iterations, "last step norms", so.rows()[-1]["step_norm"], sg.rows()[-1]["step_norm"])
