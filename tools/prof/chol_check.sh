# quick check of the reduced solve after a kernel change: single-window solve time, kernel duration in the trace, the Cholesky-related GPU tests
cd $GRAFT_REPO_ROOT
python tools/prof/gpu_single_prof.py 3
bash tools/prof/timeline.sh 3 2>&1 | grep -m2 chol
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
