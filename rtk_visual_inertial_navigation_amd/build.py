"""Builds libswf_hip.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libswf_hip.so")
SOURCES = ["swf_engine.hip", "swf_problem.cpp", "swf_producers.hip", "swf_gnss_epochs.cpp"]
HEADERS = ["swf_dev.h", "swf_kernels.h", "swf_lmschur.h", "swf_chol_rr4.h", "swf_kernels2.h", "swf_kernels3.h", "swf_kernels4.h",
           os.path.join("..", "..", "include", "swf_types.h"),
           os.path.join("..", "..", "include", "swf_solver.h")]


def hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + \
        os.environ.get("SWF_EXTRA_FLAGS", "").split() + ["-x", "hip", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
