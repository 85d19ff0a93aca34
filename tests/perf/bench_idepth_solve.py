"""Batch solve of VI windows whose short tracks are inverse-depth landmarks (row a2 in the loop) next to the same windows with world points.
   python tests/perf/bench_idepth_solve.py [windows] [world_points|inverse_depth_short_tracks]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import idepth_gen as ig
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options

W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = [synth.make_window(2, seed=synth.BASE_SEED + 700 + i) for i in range(8)]          # cfg2: 10 keyframes, 100 features
conv = [ig.convert_short_tracks(w, max_track=7) for w in base]
out = {}
ONLY = sys.argv[2] if len(sys.argv) > 2 else None
for name, src in (("world_points", base), ("inverse_depth_short_tracks", conv)):
    if ONLY and name != ONLY: continue
    ws = [src[i % 8].copy() for i in range(W)]
    bs = solver.BatchSolver(ws); opt = default_options(); ts = []
    for _ in range(5):
        bs.reset_state(); bs.sync(); t0 = time.perf_counter(); bs.solve_async(opt); bs.sync(); ts.append(time.perf_counter() - t0)
    its = sum(s.num_iterations for s in bs.summaries())
    out[name] = dict(batch_solve_ms=1e3 * min(ts), iterations_total=its, iterations_per_s=its / min(ts), counts=src[0].counts())
    bs.close()
print(json.dumps(dict(windows=W, **out)))
