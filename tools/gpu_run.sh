mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest.log 2>&1; tail -5 gpurun_out/r06/gputest.log
bash tools/prof/timeline.sh 3 > gpurun_out/r06/timeline_l.txt 2>&1; sed -n 5,12p gpurun_out/r06/timeline_l.txt
python bench.py --no-cpu-baseline --no-rtk-topology --stress-windows 0 > gpurun_out/r06/bench_l.json 2> gpurun_out/r06/bench_l.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_l.json')); print(d['value'], d['ms_per_step'], d['single_window'])"
