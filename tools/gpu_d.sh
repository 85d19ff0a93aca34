mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest_d.log 2>&1; tail -12 gpurun_out/r06/gputest_d.log
bash tools/prof/timeline.sh 3 > gpurun_out/r06/timeline_d.txt 2>&1; sed -n 5,13p gpurun_out/r06/timeline_d.txt
SWF_NO_STEP_FUSE=1 bash tools/prof/timeline.sh 3 > gpurun_out/r06/timeline_d_nofuse.txt 2>&1; sed -n 5,14p gpurun_out/r06/timeline_d_nofuse.txt
python bench.py --no-cpu-baseline --no-rtk-topology --stress-windows 0 > gpurun_out/r06/bench_d.json 2> gpurun_out/r06/bench_d.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_d.json')); print(d['value'], d['ms_per_step'], d['single_window'])"
SWF_EXTRA_FLAGS="-DSWF_PROFILE_DOG" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "dog stamps, fused:"; python tools/prof/gpu_dog_prof.py 1
echo "dog stamps, k_dogleg alone:"; SWF_NO_STEP_FUSE=1 python tools/prof/gpu_dog_prof.py 1
