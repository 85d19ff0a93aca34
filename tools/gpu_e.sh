mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest_e.log 2>&1; tail -12 gpurun_out/r06/gputest_e.log
bash tools/prof/timeline.sh 3 > gpurun_out/r06/timeline_e.txt 2>&1; sed -n 5,13p gpurun_out/r06/timeline_e.txt
python bench.py --no-cpu-baseline --no-rtk-topology --stress-windows 0 > gpurun_out/r06/bench_e.json 2> gpurun_out/r06/bench_e.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_e.json')); print(d['value'], d['ms_per_step'], d['single_window'])"
SWF_EXTRA_FLAGS="-DSWF_PROFILE_DOG" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "dog stamps, fused:"; python tools/prof/gpu_dog_prof.py 1
