"""Solve a batch of windows whose frames are linked by composite IMU-GNSS factors (rows a5 / a10 inside the loop).
   python tests/perf/bench_composite_solve.py [windows] [frames] [hidden_epochs] [ambiguities]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import composite_gen as cg
import oracle_binding as ob
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options

W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 11
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
rng = np.random.default_rng(2)
base = [cg.make_window(rng, K, M, N) for _ in range(16)]
ws = [base[i % 16].copy() for i in range(W)]
bs = solver.BatchSolver(ws); opt = default_options()
ts = []
for _ in range(5):
    bs.reset_state(); bs.sync(); t0 = time.perf_counter(); bs.solve_async(opt); bs.sync(); ts.append(time.perf_counter() - t0)
sms = bs.summaries()
its = sum(s.num_iterations for s in sms)
wo = base[0].copy()
t0 = time.perf_counter(); so, _ = ob.solve(wo, opt, export=False); t_or = time.perf_counter() - t0
print(json.dumps(dict(windows=W, frames=K, hidden_epochs_per_gap=M, ambiguities=N, composite_factors=W * (K - 1), gpu_batch_solve_ms=1e3 * min(ts),
                      iterations_total=its, gpu_iterations_per_s=its / min(ts), oracle_ms_per_window_1thread=1e3 * t_or, oracle_iterations=so.num_iterations,
                      speedup_vs_1thread=t_or * W / min(ts))))
