cd $GRAFT_REPO_ROOT
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CHOL" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
MG_STAMPS=1 python tools/prof/marg_eigen_time.py | tail -1
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
