cd /tmp && export TMPDIR=/tmp
for n in ${SIZES:-64 32}; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_w$n -o w$n --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --windows $n --steps 20 --no-cpu-baseline --no-live-traffic --no-rtk-topology --stress-windows 0 --no-single-window > $GRAFT_REPO_ROOT/gpurun_out/bench_w$n.json 2>/dev/null
done
