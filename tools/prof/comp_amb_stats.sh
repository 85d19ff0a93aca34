cd /tmp && export TMPDIR=/tmp
for S in 24 40; do
rm -rf /tmp/cprof$S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof$S -o s -- python $GRAFT_REPO_ROOT/tools/prof/gpu_comp_prof.py 2 20 4 300 $S 8 single > /tmp/cprof_single.log 2>&1
cp "$(find /tmp/cprof$S -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/composite_single_${S}amb_kernel_stats.csv
done
