mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest_c.log 2>&1; tail -15 gpurun_out/r06/gputest_c.log
bash tools/prof/timeline.sh 3 > gpurun_out/r06/timeline_c.txt 2>&1; sed -n 1,22p gpurun_out/r06/timeline_c.txt
python bench.py --no-cpu-baseline --no-rtk-topology --stress-windows 0 > gpurun_out/r06/bench_c.json 2> gpurun_out/r06/bench_c.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_c.json')); print(d['value'], d['ms_per_step'], d['single_window'])"
