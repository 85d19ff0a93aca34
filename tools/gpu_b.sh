mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest_b.log 2>&1; tail -15 gpurun_out/r06/gputest_b.log
bash tools/prof/timeline.sh 3 > gpurun_out/r06/timeline_b.txt 2>&1; sed -n 1,30p gpurun_out/r06/timeline_b.txt
python bench.py --no-cpu-baseline > gpurun_out/r06/bench_b.json 2> gpurun_out/r06/bench_b.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_b.json')); print(d['value'], d['ms_per_step'], d['single_window'])"
