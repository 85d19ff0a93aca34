# k_lm_schur experiments on one box: per-phase stamps, producers idle, consumers idle (rocprofv3 kernel stats of the batch-only workload)
cd $GRAFT_REPO_ROOT
run() { # flags label
  SWF_EXTRA_FLAGS="$1" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== $2"; bash tools/prof/kstats.sh ${WIN:-512} 2 2>&1 | grep "lm_schur"
}
run "-DSWF_LS_NOMFMA" "consumers idle"
run "-DSWF_LS_NOPROD" "producers idle"
SWF_EXTRA_FLAGS="-DSWF_PROFILE_GEMM" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "== stamps"; python tools/prof/gpu_gemm_prof.py 512; python tools/prof/gpu_gemm_prof.py 1
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
