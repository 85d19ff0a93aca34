timeout 600 python -m pytest tests/test_rtk_topology.py tests/test_gpu_fuzz.py -x -q -m gpu -k "composite or topology" 2>&1 | tail -5
for S in 10 24; do
echo "== S=$S fused"; python tools/prof/gpu_comp_prof.py 2 20 4 300 $S 8 solve 2>&1 | grep "single window"
echo "== S=$S separate"; SWF_NO_COMP_FUSE=1 python tools/prof/gpu_comp_prof.py 2 20 4 300 $S 8 solve 2>&1 | grep "single window"
done
