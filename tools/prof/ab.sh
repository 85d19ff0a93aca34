# usage: ab.sh "<flags>" : build with flags into the tree on the box, run bench
SWF_EXTRA_FLAGS="$1" python -m rtk_visual_inertial_navigation_amd.build > /dev/null 2>&1
python bench.py --no-single-window --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
