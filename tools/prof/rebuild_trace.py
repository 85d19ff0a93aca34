"""SWF_TRACE_REBUILD=1 python tools/prof/rebuild_trace.py — where a structure change of the Problem surface spends its time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from rtk_visual_inertial_navigation_amd import synth
w = synth.make_window(3)
print(bench.structure_change_leg(w, 8, reps=3))
