"""MI355X-native sliding-window Gauss-Newton solver behind a ceres-shaped C-ABI.

Hot path only: factor Jacobians -> block J^T J -> Schur -> dense reduced
solve -> dogleg loop, as hand-written HIP kernels for gfx950 in csrc/.  Everything in this
package is host-side plumbing around libswf_hip.so; there is no CPU fallback.
"""
from .flat import FlatWindow, default_options, SummaryC, OptionsC, TERMINATION  # noqa: F401
