cd $GRAFT_REPO_ROOT
for F in "-DRR3_EXP_NOMASK" ""; do
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CHOL -DSWF_PROFILE_CHOL_STEP=8 $F" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "== flags: $F"; python tools/prof/gpu_chol_prof.py 3 | tail -1 | cut -c170-330
done
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
