"""The reference's own RTK topology at cfg3 size: windows of K_vis visual frames linked only by composite
IMU-GNSS factors hiding M GNSS epochs each (tests/rtk_topology_gen.py, a generator), built on the device and solved:
   python tools/prof/gpu_comp_prof.py [n_windows=64] [K_vis=20] [M=4] [F=300] [S=10] [iters=8] [mode]
mode "solve" (default): one window alone (us per iteration, its iteration rows), the batch (ms per solve, terminations), the same
windows to their own termination (50 iterations), the oracle on the first windows (CPU row); "batch" / "single": only that workload,
a few solves, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rtk_topology_gen as rt
a = sys.argv[1:]
N = int(a[0]) if len(a) > 0 else 64
K, M, F, S = (int(a[i]) if len(a) > i else d for i, d in ((1, 20), (2, 4), (3, 300), (4, 10)))
iters = int(a[5]) if len(a) > 5 else 8
mode = a[6] if len(a) > 6 else "solve"
t0 = time.perf_counter()
wxs = rt.explicit_windows(N, K_vis=K, M=M, F=F, S=S)                # (process pool: before HIP is touched)
t_gen = time.perf_counter() - t0
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options
tm = {}
wins = rt.composite_batch(solver, wxs, timing=tm)
d0 = None
opt = default_options(max_num_iterations=iters)


def timed(ws, reps=10, warm=2):
    bs = solver.BatchSolver([w.copy() for w in ws])
    for _ in range(warm):
        bs.reset_state(); bs.solve_async(opt); bs.sync()
    lat = []
    for _ in range(reps):
        bs.reset_state(); t1 = time.perf_counter(); bs.solve_async(opt); bs.sync(); lat.append(time.perf_counter() - t1)
    sms = bs.summaries()
    return bs, sms, float(np.median(lat))


if mode in ("solve", "single"):
    bs, sms, dt = timed(wins[:1], reps=20, warm=3)
    s0 = sms[0]
    print("window: %d visual frames, %d hidden epochs per gap, %d landmarks, %d observations, %d ambiguities, n_red %d"
          % (K, M, wins[0].n_lm, wins[0].a["proj_idx"].size // 3, S, bs.dims(0)["n_red"]))
    print("single window: %.3f ms per solve, %d iterations -> %.1f us per iteration; termination %d, cost %.6e -> %.6e"
          % (1e3 * dt, s0.num_iterations, 1e6 * dt / max(1, s0.num_iterations), s0.termination, s0.initial_cost, s0.final_cost))
    for r in s0.rows():
        print("   it cost %.9e radius %.3e step %.3e accepted %d gmax %.3e" % (r["cost"], r["trust_region_radius"], r.get("step_norm", 0.0), r["step_is_successful"], r["gradient_max_norm"]))
    bs.close()
if mode in ("solve", "batch"):
    bs, sms, dt = timed(wins, reps=6 if mode == "batch" else 10)
    its = sum(s.num_iterations for s in sms)
    print("batch of %d: %.3f ms per solve, %d iterations -> %.1f k iterations/s, %.2f us per window-iteration; converged (terminations 1-3) %d of %d, failures %d"
          % (N, 1e3 * dt, its, its / dt / 1e3, 1e6 * dt / its, sum(s.termination in (1, 2, 3) for s in sms), N, sum(s.termination not in (1, 2, 3, 4) for s in sms)))
    print("   mean cost reduction %.3e" % float(np.mean([s.final_cost / s.initial_cost for s in sms])))
    if mode == "solve":
        bs.reset_state(); bs.solve_async(default_options(max_num_iterations=50)); bs.sync()
        s50 = bs.summaries()
        print("   to their own termination (<= 50 iterations): converged %d of %d, mean iterations %.1f, terminations %s"
              % (sum(s.termination in (1, 2, 3) for s in s50), N, float(np.mean([s.num_iterations for s in s50])), sorted(set(int(s.termination) for s in s50))))
    bs.close()
if mode == "solve":
    cw, wi, ci = rt.warm_start_probe(solver, wins[:min(N, 16)], opt, default_options(max_num_iterations=50))
    print("warm start (the windows converged, then only the newest frame moved by an IMU-prediction-sized error): %d of %d converge within %d iterations, mean %.1f iterations (cold: %.1f to termination)" % (cw, min(N, 16), iters, wi, ci))
    print("construction: generate %.1f s; %d GNSS epochs pre-eliminated in %.2f ms (C-ABI call), host bookkeeping %.1f ms" % (t_gen, tm["gnss_epochs"], 1e3 * tm["epoch_priors_s"], 1e3 * tm["host_assemble_s"]))
    import oracle_binding as ob
    ts, its = [], 0
    for w in wins[:4]:
        wo = w.copy(); t1 = time.perf_counter(); so, _ = ob.solve(wo, opt, export=False); ts.append(time.perf_counter() - t1); its += so.num_iterations
    print("oracle (plain-C port, %s OpenMP threads) on the first 4 windows: %.2f ms per solve, %.1f us per iteration" % (os.environ.get("OMP_NUM_THREADS", "default"), 1e3 * float(np.mean(ts)), 1e6 * sum(ts) / max(1, its)))
