import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_binding as ob
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
import cfg5_marg_gen as cg
w, info = cg.make_cfg5_with_marginalised_prior(solver)
print('prior obtained by marginalising a 41st frame on the device: dim', info['prior_dim'], 'rank', info['rank'], 'landmarks marginalised', info['n_marg_landmarks'])
bs = solver.BatchSolver([w.copy()])
print("dims", bs.dims(0))
bs.enable_timing(True)
for _ in range(3):
    bs.reset_state(); bs.solve_async(default_options()); bs.sync()
t = bs.timing()
print("gpu total ms", round(t["total_ms"], 3), {k: round(v["ms"], 3) for k, v in t["kernels"].items() if k != "total"})
wo = w.copy(); t0 = time.time(); so, _ = ob.solve(wo, default_options(), export=False); print("oracle 1 thread s", round(time.time() - t0, 3), "iters", so.num_iterations)
