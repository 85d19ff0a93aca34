mkdir -p gpurun_out/r06
python tools/prof/eigroot_time.py 2>&1 | tail -3
python -m pytest tests -m gpu -q -x -k "composite or topology or rtk" > gpurun_out/r06/gputest_eig.log 2>&1; tail -4 gpurun_out/r06/gputest_eig.log
