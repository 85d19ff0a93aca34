"""Medium batches of cfg5-class windows (n_red = 440): python tests/perf/time_cfg5_batch.py [windows ...]"""
import sys, time, numpy as np
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
w = synth.make_window(5)
for W in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64]:
    bs = solver.BatchSolver([w.copy() for _ in range(W)]); opt = default_options(); ts = []
    for _ in range(5):
        bs.reset_state(); bs.sync(); t0 = time.perf_counter(); bs.solve_async(opt); bs.sync(); ts.append(time.perf_counter() - t0)
    assert all(sm.termination in (1, 2, 3, 4) for sm in bs.summaries()), "a solve failed: the timing would be meaningless"
    print("cfg5 x", W, "solve ms", round(1e3 * min(ts), 3), "per window", round(1e3 * min(ts) / W, 3), flush=True)
    bs.close()
