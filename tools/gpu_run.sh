# one GPU-box pass: the GPU tier, cfg5 kernel trace, bench
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest.log 2>&1; tail -8 gpurun_out/r06/gputest.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/c5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5 -o c -- python $GRAFT_REPO_ROOT/tools/prof/gpu_cfg5_run.py 1 3 > /tmp/c5.log 2>&1
cd $GRAFT_REPO_ROOT; cp "$(find /tmp/c5 -name '*kernel_stats.csv' | head -1)" gpurun_out/r06/cfg5_kernel_stats.csv; head -16 gpurun_out/r06/cfg5_kernel_stats.csv | cut -c1-130
python bench.py --no-cpu-baseline > gpurun_out/r06/bench_j.json 2> gpurun_out/r06/bench_j.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_j.json')); print(d['value'], d['ms_per_step'], d['single_window']); print(d['stress']['single_window']); print(d['rtk_topology']['single_window'], d['rtk_topology']['batch_replicated'])"
