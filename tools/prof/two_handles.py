"""The same job as one batch, or as H handles on H streams of ONE device driven by one host thread through swf_solve_batches
(the in-process multi-batch entry): the latency-bound kernels of one part overlap the streaming kernels of the others."""
import sys, time, ctypes as C, os
sys.path.insert(0, '.')
import bench
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
opt = default_options(max_num_iterations=8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
allw = bench.make_windows(4, [synth.BASE_SEED + 4 + i for i in range(N)])
hip = C.CDLL("libamdhip64.so")
def new_stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0          # hipStreamNonBlocking: every handle on its own stream
    return s.value
for H in (1, 2, 4):
    parts = [solver.BatchSolver([w.copy() for w in allw[k * N // H:(k + 1) * N // H]], stream=new_stream(), device=0) for k in range(H)]
    hs = (C.c_void_p * H)(*[p._h for p in parts])
    def run():
        for p in parts: p.reset_state()
        assert solver.lib().swf_solve_batches(hs, C.c_int32(H), C.byref(opt)) == 0
    for _ in range(3): run()
    t0 = time.perf_counter(); K = 10
    for _ in range(K): run()
    dt = (time.perf_counter() - t0) / K
    its = sum(s.num_iterations for p in parts for s in p.summaries())
    print("windows %4d in %d handle(s): %.3f ms per solve  %.1f k it/s" % (N, H, dt * 1e3, its / dt / 1e3), flush=True)
    for p in parts: p.close()
