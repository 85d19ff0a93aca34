import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtk_topology_gen as rt
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import FlatWindowC
wxs = [rt.explicit_window(K_vis=10, M=4, F=100, S=10, seed=900 + i)[0] for i in range(16)]
ews = [e for wx in wxs for e in rt.epoch_windows(wx)[0]]
solver.marginal_priors(ews[:8], 1e-8, 0)
t0 = time.perf_counter(); structs = [w.c_struct() for w in ews]; t1 = time.perf_counter()
arr = (C.POINTER(FlatWindowC) * len(structs))(*[C.pointer(s) for s in structs]); n = len(structs)
dims = np.zeros(n, np.int32); pi = C.POINTER(C.c_int32); _pd = C.POINTER(C.c_double)
solver.lib().swf_batch_marginal_priors(arr, C.c_int32(n), C.c_double(1e-8), C.c_int32(0), dims.ctypes.data_as(pi), None, None, None, None, None, None)
t2 = time.perf_counter()
n2, n1 = int((dims.astype(np.int64) ** 2).sum()), int(dims.sum())
A, J, b, r0, ranks = np.zeros(n2), np.zeros(n2), np.zeros(n1), np.zeros(n1), np.zeros(n, np.int32)
os.environ["SWF_TRACE_REBUILD"] = "1"
rc = solver.lib().swf_batch_marginal_priors(arr, C.c_int32(n), C.c_double(1e-8), C.c_int32(0), dims.ctypes.data_as(pi), ranks.ctypes.data_as(pi), A.ctypes.data_as(_pd), b.ctypes.data_as(_pd), J.ctypes.data_as(_pd), r0.ctypes.data_as(_pd), None)
t3 = time.perf_counter()
print("epochs", n, "c_struct %.1f ms, sizing %.1f ms, device call %.1f ms (rc %d)" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), rc))
