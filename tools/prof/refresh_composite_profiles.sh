bash tools/prof/comp_profile.sh gpurun_out/comp_final 64 > /dev/null 2>&1
OMP_NUM_THREADS=4 python tools/prof/gpu_comp_prof.py 16 20 4 300 24 8 solve > gpurun_out/comp_final/composite_workload_24amb.txt 2>&1
OMP_NUM_THREADS=4 python tools/prof/gpu_comp_prof.py 16 20 4 300 40 8 solve > gpurun_out/comp_final/composite_workload_40amb.txt 2>&1
bash tools/prof/comp_timeline.sh 10 > gpurun_out/comp_final/composite_single_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/comp_final/gputest.log
timeout 600 python bench.py > gpurun_out/comp_final/bench_default.json 2> /dev/null
