"""Experiment: same 512 windows as 1, 2 or 4 independent batches on separate HIP streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ws = bench.make_windows(4, [synth.BASE_SEED + 4 + i for i in range(B)])
opt = default_options()
for ns in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    solvers = [solver.BatchSolver(ws[i::ns], stream=streams[i].cuda_stream) for i in range(ns)]
    for rep in range(2):
        for s in solvers: s.reset_state(); s.solve_async(opt)
        for s in solvers: s.sync()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(5):
        for s in solvers: s.reset_state(); s.solve_async(opt)
        for s in solvers: s.sync()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    its = sum(sm.num_iterations for s in solvers for sm in s.summaries())
    print("streams", ns, "ms/solve %.2f" % (dt * 1e3), "iters/s %.0f" % (its / dt))
    for s in solvers: s.close()
