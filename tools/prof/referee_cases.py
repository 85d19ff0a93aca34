"""First-step error against the extended-precision referee (tests/referee.py) for the degenerate windows of the parity test and cfg2/cfg3:
device, oracle, eps * cond(S)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T, referee
from rtk_visual_inertial_navigation_amd import synth
from rtk_visual_inertial_navigation_amd.flat import default_options
base = synth.make_window(3, K=6, F=20, S=5, seed=311)
roles = base.meta["roles"]
cases = {}
w = base.copy(); ic = w.a["is_const"].copy()
w.a["proj_idx"] = np.zeros((0, 3), np.int32).ravel(); w.a["proj_uv"] = np.zeros(0)
for b in roles["landmarks"]: ic[b] = 1
cases["no_visual"] = T._reorder(w, ic)
w = base.copy(); ic = w.a["is_const"].copy()
w.a["imu_idx"] = np.zeros(0, np.int32); w.a["imu_pre"] = np.zeros(0)
for b in roles["speed_bias"]: ic[b] = 1
cases["no_imu"] = T._reorder(w, ic)
cases["two_frames"] = synth.make_window(3, K=2, F=6, S=4, seed=5)
cases["base"] = base
cases["cfg2"] = synth.make_window(2)
cases["cfg3"] = synth.make_window(3)
for name, w in cases.items():
    ed, eo, cond, nr = referee.first_step_errors(w, T.ob, T.gpu_solve, default_options)
    print("%-10s device %.2e  oracle %.2e  eps*cond %.2e  device / (eps cond) %.2f" % (name, ed, eo, 1.1e-16 * cond, ed / (1.1e-16 * cond)))
