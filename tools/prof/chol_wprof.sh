# per-wave stamps inside k_chol_rr3 for a few steps of one cfg3 window
cd $GRAFT_REPO_ROOT
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CHOLW" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
python tools/prof/gpu_chol_wprof.py "$@"
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
