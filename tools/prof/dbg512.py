import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import bench
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
B = 512
seeds = [synth.BASE_SEED + 4 + i for i in range(B)]
ws = bench.make_windows(4, seeds)
batch = [w.copy() for w in ws]
bs = solver.BatchSolver(batch)
sms = bs.solve(default_options(max_num_iterations=8))
bad = [i for i, s in enumerate(sms) if s.termination not in (1, 2, 3, 4)]
print("failing windows:", bad[:20], "count", len(bad))
for i in bad[:5]:
    s = sms[i]; print(i, "termination", s.termination, "n_red", bs.dims(i)["n_red"], "iters", s.num_iterations, [r["cost"] for r in s.rows()][:4])
bs.close()
for i in bad[:3]:
    b1 = solver.BatchSolver([ws[i].copy()]); s1 = b1.solve(default_options(max_num_iterations=8))[0]
    print("alone", i, "termination", s1.termination, [r["cost"] for r in s1.rows()][:4]); b1.close()
