import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rtk_topology_gen as rt
wxs = rt.explicit_windows(1, K_vis=20, M=4, F=300, S=10, pool=False)
from rtk_visual_inertial_navigation_amd import solver
w = rt.composite_batch(solver, wxs)[0]
a = w.a
nc = a["comp_M"].size
M, N = int(a["comp_M"][0]), int(a["comp_N"][0])
fs = []
pose = a["comp_pose"].reshape(-1, 7); sb = a["comp_sb"].reshape(-1, 9)
Hpp = a["comp_Hpp"].reshape(-1, 15, 15); HpN = a["comp_HpN"].reshape(-1, 15, N); rp = a["comp_rhs_p"].reshape(-1, 15)
HNN = a["comp_HNN"].reshape(nc, N, N); rN = a["comp_rhsN"].reshape(nc, N); pre = a["comp_pre"].reshape(-1, 293)
for f in range(nc):
    e = slice(f * M, (f + 1) * M)
    fs.append(dict(pose=pose[e], sb=sb[e], pose_lin=pose[e], sb_lin=sb[e], Hpp=Hpp[e], HpN=HpN[e], rhs_p=rp[e], HNN=HNN[f], rhsN=rN[f], pre=pre[f * (M + 1):(f + 1) * (M + 1)]))
cb = solver.CompositeBatch(fs, w.pbg, w.gw)
P = a["pose"].reshape(-1, 7); B = a["sb"].reshape(-1, 9); sc = a["sc"]
outer = np.array([np.concatenate([P[f], B[f], P[f + 1], B[f + 1]]) for f in range(nc)])
out = cb.evaluate(outer, [sc[1:1 + N] for _ in range(nc)], True)
for f, o in enumerate(out[:6]):
    H = o["H"]; ev = np.linalg.eigvalsh(H)
    print(f, "status", o["status"], "rank of the square root", int((np.abs(o["J"]).sum(1) > 0).sum()), "of", H.shape[0], "| eig min %.2e max %.2e | diag min %.2e max %.2e" % (ev[0], ev[-1], np.diag(H).min(), np.diag(H).max()))
