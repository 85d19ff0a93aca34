#!/bin/bash
# The composite-factor (reference-topology) workload at cfg3 size: timings, then rocprofv3 kernel stats of one window and of a batch.
#   tools/prof/comp_profile.sh <out_dir> [n_windows]
OUT=$(realpath -m "${1:-gpurun_out/comp}"); N=${2:-64}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
OMP_NUM_THREADS=4 python "$ROOT/tools/prof/gpu_comp_prof.py" $N 20 4 300 10 8 solve > "$OUT/composite_workload.txt" 2>&1
rm -rf /tmp/cprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof/single -o s -- python "$ROOT/tools/prof/gpu_comp_prof.py" 2 20 4 300 10 8 single > /tmp/cprof_single.log 2>&1
cp "$(find /tmp/cprof/single -name '*kernel_stats.csv' | head -1)" "$OUT/composite_single_kernel_stats.csv"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof/batch -o b -- python "$ROOT/tools/prof/gpu_comp_prof.py" $N 20 4 300 10 8 batch > /tmp/cprof_batch.log 2>&1
cp "$(find /tmp/cprof/batch -name '*kernel_stats.csv' | head -1)" "$OUT/composite_batch_kernel_stats.csv"
cat "$OUT/composite_workload.txt"
head -14 "$OUT/composite_single_kernel_stats.csv" | cut -c1-130
head -14 "$OUT/composite_batch_kernel_stats.csv" | cut -c1-130
