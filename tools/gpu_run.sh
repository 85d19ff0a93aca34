mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q -x -k "composite or topology or rtk or fuzz_composite" > gpurun_out/r06/gputest_comp.log 2>&1; tail -4 gpurun_out/r06/gputest_comp.log
OMP_NUM_THREADS=4 python tools/prof/gpu_comp_prof.py 64 20 4 300 10 8 solve 2>&1 | grep -E "single window|batch of"
