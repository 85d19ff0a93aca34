# k_chol_rr3 on one cfg3 window: phase stamps and the length of every step (B_j -> B_j+1), from a -DSWF_PROFILE_CHOL build
cd $GRAFT_REPO_ROOT
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CHOL" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
python tools/prof/gpu_chol_prof.py ${1:-3}
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
