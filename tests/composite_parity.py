"""Two-sided parity statements for windows with composite IMU-GNSS factors, shared by tests/test_rtk_topology.py
and tests/perf/fuzz_composite.py.

The square root of a composite factor's (30 + N)^2 remainder is where the device and the reference differ BY DESIGN, and where the
reference's own number is implementation-defined: UpdateSchurComponent (R/factor/gnss_imu_factor.cpp:454-488) cuts eigenvalues at an
absolute 1e-8 of a matrix with entries of 1e8..1e11.  A single gap's remainder is singular (it pins neither the heading nor the absolute
position), its null eigenvalues come out as rounding noise ~eps * lambda_max, most of it ABOVE 1e-8, and every kept noise direction adds
r_k^2 = (v_k^T rhs)^2 / lambda_k = O(1) to the factor's cost — a number that depends on the eigensolver's rounding.  So the statements
are made against two oracles:

  (A) the NOISE-FREE restatement (oracle_binding.composite_eig_cut(1e-14): cut at max(1e-8, 1e-14 lambda_max)) — two-sided, tight: the
      device with either root takes the same accept / reject decisions, has the same first cost, the same cost DIFFERENCES cost_k - cost_0
      and ends at the same point;
  (B) the LITERAL reference (cut 1e-8): same decisions and first cost, and its cost lies above the device's by no more than the near-null
      terms the oracle itself counted as kept (eigenvalues above its cut and at most 1e-10 lambda_max: the null space's rounding noise
      and what a rank-revealing factorisation may drop next to it; summed over the solve), and not below, beyond (A)'s tolerance:
      the offset is a checked, bounded quantity.
"""
import numpy as np
import oracle_binding as ob
from rtk_visual_inertial_navigation_amd.flat import default_options

# Defaults = what a randomised sweep over the reference's topology supports (tests/perf/fuzz_composite.py, 80 windows x 2 roots x 2 budgets,
# profiles/r06/fuzz_composite_calibration.txt: 3..12 visual frames, 1..5 hidden epochs per gap, 4..40 ambiguities; worst seen in brackets).
# Windows with four or five satellites leave directions that only the gauge prior (1e-3) pins: their end states are that loose.
# The full-size tests pass tighter values where they hold.
TOL_FIRST = 5e-7          # first cost, relative                                                        [1.3e-7; 1e-12 typically]
TOL_DIFF = 5e-5           # cost_k - cost_0 against the noise-free oracle, relative to cost_0 - cost_k  [2.2e-5 with four satellites; 3e-7 typically]
TOL_STATE = 1e-4          # end poses against the noise-free oracle; 10 x for speed-biases and scalars  [3.9e-5 / 5.5e-4]


def oracle_solves(w, iters):
    """(noise-free summary / window, literal summary / window, noise cost the literal oracle kept, count)"""
    wn, wl = w.copy(), w.copy()
    with ob.composite_eig_cut(1e-14):
        sn, _ = ob.solve(wn, default_options(max_num_iterations=iters), export=False)
    with ob.composite_eig_cut(0.0) as cut:
        sl, _ = ob.solve(wl, default_options(max_num_iterations=iters), export=False)
        noise, count = cut.noise()
    return (sn, wn), (sl, wl), noise, count


NOISE_FACTOR = 4.0        # (B) the literal oracle's cost above the device's: at most this many times the near-null cost it kept over the solve
                          # (the two do not walk the same path once noise enters the literal oracle's gradient)   [1.9 with four satellites; <= 1 otherwise]


def check(sd, wd, orc, decisions_vs_literal=True, report=None, tol_first=TOL_FIRST, tol_diff=TOL_DIFF, tol_state=TOL_STATE, n_dec=None):
    """n_dec: compare the accept / reject decisions of the first n_dec rows only (long runs: near convergence a decision may flip on rounding)."""
    """sd / wd: the device's summary and end-state window; orc = oracle_solves(...).  Returns a list of violated statements (empty = parity)."""
    (sn, wn), (sl, wl), noise, count = orc
    rd, rn, rl = sd.rows(), sn.rows(), sl.rows()
    bad = []
    acc = lambda rows: [r["step_is_successful"] for r in rows][:n_dec]

    def same_decisions(ra, sa, rb, sb):
        # the same accept / reject decisions over the common iterations; one solver may stop an iteration before the other ONLY when both
        # stop on a convergence test (the function-tolerance test |cost change| <= 1e-6 cost is a threshold: a tie breaks on rounding)
        # (the row of the iteration a solver converges in is not a successful step: where one stops an iteration early, that row is left out)
        a, b_ = acc(ra), acc(rb)
        if len(a) == len(b_): return a == b_
        n = min(len(a), len(b_))
        short = sa if len(a) < len(b_) else sb
        return abs(len(a) - len(b_)) == 1 and a[:n - 1] == b_[:n - 1] and short.termination in (1, 2, 3)
    # (A) against the noise-free restatement
    if not same_decisions(rd, sd, rn, sn): bad.append("A: accept / reject sequence differs from the noise-free oracle's")
    c0 = rn[0]["cost"]
    e0 = abs(rd[0]["cost"] - c0) / c0
    if e0 > tol_first: bad.append("A: first cost %.3e off" % e0)
    ediff = 0.0
    for k in range(1, min(len(rd), len(rn))):
        dec = abs(c0 - rn[k]["cost"])
        if dec > 0: ediff = max(ediff, abs((rd[k]["cost"] - rd[0]["cost"]) - (rn[k]["cost"] - c0)) / dec)
    if ediff > tol_diff: bad.append("A: cost differences %.3e of the decrease off" % ediff)
    est = max(np.abs(wd.a[k] - wn.a[k]).max() if wd.a[k].size else 0.0 for k in ("pose", "comp_pose"))
    est2 = max(np.abs(wd.a[k] - wn.a[k]).max() if wd.a[k].size else 0.0 for k in ("sb", "sc", "comp_sb"))
    if len(rd) == len(rn) and (est > tol_state or est2 > 10 * tol_state): bad.append("A: end states %.3e / %.3e apart" % (est, est2))
    # (B) against the literal reference
    if decisions_vs_literal and not same_decisions(rd, sd, rl, sl): bad.append("B: accept / reject sequence differs from the literal oracle's")
    l0 = rl[0]["cost"]
    if abs(rd[0]["cost"] - l0) / l0 > tol_first + NOISE_FACTOR * noise / l0: bad.append("B: first cost %.3e off" % (abs(rd[0]["cost"] - l0) / l0))
    lo = hi = 0.0
    for k in range(min(len(rd), len(rl))):
        tol_k = tol_first * l0 + tol_diff * abs(l0 - rl[k]["cost"])
        off = rl[k]["cost"] - rd[k]["cost"]               # literal oracle above the device by the noise it keeps
        lo = min(lo, (off + tol_k)); hi = max(hi, off - NOISE_FACTOR * noise - tol_k)
    if lo < 0: bad.append("B: the literal oracle's cost %.3e BELOW the device's beyond tolerance" % -lo)
    if hi > 0: bad.append("B: the literal oracle's cost above the device's by %.3e more than %g x the near-null cost it kept (%.3e in %d directions)" % (hi, NOISE_FACTOR, noise, count))
    if report is not None:
        report.update(first=e0, diffs=ediff, states=est, states2=est2, noise=noise, count=count, lo=lo, hi=hi)
    return bad
