# which stage makes batch != single in tests/perf/fuzz_marginalize.py (seed 11)?  Re-creates its windows and compares S, rhs, L, the
# solution vector and the prior between repeated single solves and the batch.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cfg5_marg_gen as mg
from rtk_visual_inertial_navigation_amd import synth, solver
from rtk_visual_inertial_navigation_amd.flat import default_options
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
wins = []
for t in range(N):
    vi = rng.random() < 0.3
    K = int(rng.integers(3, 13)); F = int(rng.integers(max(4, K), 60)); S = 0 if vi else int(rng.integers(4, 11))
    kind = rng.choice(["ambiguities", "frames", "globalmarge"]) if S else rng.choice(["frames", "globalmarge"])
    if kind == "globalmarge":
        full = synth.make_window(3 if S else 2, K=K + 1, F=F, S=S, prior="gauge", seed=int(rng.integers(1, 10 ** 6)))
        w, _ = mg.marginalisation_window(full)
    else:
        w = synth.make_window(3 if S else 2, K=K, F=F, S=S, seed=int(rng.integers(1, 10 ** 6)), head=str(kind))
    wins.append(w)
def one(ws, idx):
    bs = solver.BatchSolver([w.copy() for w in ws]); sm = bs.solve(default_options(step_mode=1))
    out = []
    for i in idx:
        S, rhs, L = bs.export_reduced(i); g, d, y = bs.export_vectors(i)
        out.append(dict(S=S, rhs=rhs, L=L, y=y, n=sm[i].tail_dim, nr=sm[i].reduced_dim))
    bs.marginalize(1e-8, solver.BatchSolver.PRIOR_EIGEN)
    for k, i in enumerate(idx):
        g = bs.get_prior(i); out[k].update(A=g["A"], J=g["J"], b=g["b"], r0=g["r0"], rank=g["rank"])
    bs.close()
    return out
keep = [i for i, w in enumerate(wins)]
batch = one(wins, keep)
for i in keep:
    a = one([wins[i]], [0])[0]; a2 = one([wins[i]], [0])[0]; b = batch[i]
    diff = [k for k in ("S", "rhs", "L", "y", "A", "b", "J", "r0") if not np.array_equal(a[k], b[k])]
    rep = [k for k in ("S", "rhs", "L", "y", "A", "b", "J", "r0") if not np.array_equal(a[k], a2[k])]
    if diff or rep: print("case", i, "n_red", a["nr"], "tail", a["n"], "| single vs batch differ in", diff, "| single vs single differ in", rep, flush=True)
print("done")
