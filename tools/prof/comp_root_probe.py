"""Probe (GPU box): the composite path's cost / state parity per square root, device against the oracle literal (absolute 1e-8 cut) and
against the oracle's noise-free restatement (cut max(1e-8, rel * lambda_max)).  python tools/prof/comp_root_probe.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import rtk_topology_gen as rt
from rtk_visual_inertial_navigation_amd import solver
from rtk_visual_inertial_navigation_amd.flat import default_options

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
wxs, shapes = [], []
for t in range(N):
    K = int(rng.integers(3, 13)); M = int(rng.integers(1, 6)); S = int(rng.integers(4, 41)); F = int(rng.integers(max(12, 2 * K), 160))
    if K + (K - 1) * M > 60: M = max(1, (60 - K) // (K - 1))
    kw = dict(K_vis=K, M=M, F=F, S=S, seed=int(rng.integers(1, 10 ** 6)))
    shapes.append(kw); wxs.append(rt.explicit_window(**kw)[0])
wins = rt.composite_batch(solver, wxs)
for kw, w in zip(shapes, wins):
    print(kw, flush=True)
    for iters in (8, 40):
        dev = {}
        for root in (0, 1):
            wd = w.copy()
            bs = solver.BatchSolver([wd]); sd = bs.solve(default_options(max_num_iterations=iters, composite_root=root))[0]; bs.close()
            dev[root] = (wd, sd.rows(), sd)
        for rel in (0.0, 1e-14, 1e-13, 1e-12):
            wo = w.copy()
            with ob.composite_eig_cut(rel) as cut:
                so, _ = ob.solve(wo, default_options(max_num_iterations=iters), export=False)
                noise = cut.noise()
            ro = so.rows()
            line = "  its %2d oracle cut %-6g (kept noise: %d, %.3e) term %d final %.9e:" % (iters, rel, noise[1], noise[0], so.termination, so.final_cost)
            for root in (0, 1):
                wd, rd, sd = dev[root]
                same = [r["step_is_successful"] for r in rd] == [r["step_is_successful"] for r in ro]
                nn = min(len(rd), len(ro))
                dc = max(abs(rd[k]["cost"] - ro[k]["cost"]) / ro[k]["cost"] for k in range(nn))
                # cost DIFFERENCES from the first row
                dd = max(abs((rd[k]["cost"] - rd[0]["cost"]) - (ro[k]["cost"] - ro[0]["cost"])) / max(1e-300, abs(ro[0]["cost"] - ro[k]["cost"])) for k in range(1, nn)) if nn > 1 else 0.0
                dp = max(np.abs(wd.a["pose"] - wo.a["pose"]).max(), np.abs(wd.a["comp_pose"] - wo.a["comp_pose"]).max())
                line += " | %s: seq %s c0 %+.1e seq %.1e diffs %.1e states %.1e" % ("piv" if root == 0 else "eig", "=" if same else "X", (rd[0]["cost"] - ro[0]["cost"]) / ro[0]["cost"], dc, dd, dp)
            print(line, flush=True)
