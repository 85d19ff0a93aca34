# A/B of k_lm_schur's landmark parts per workgroup at small batch sizes
export SWEEP=${SWEEP:-96,64,48,32}
for q in 1 2 4 8; do echo "== SWF_LS_QPB=$q"; SWF_LS_QPB=$q python tools/prof/batch_size_sweep.py; done
