# k_chol_rr3: who arrives when at barrier B_j (per-wave stamps), for a few steps j
cd $GRAFT_REPO_ROOT
for J in 1 4 8 12; do
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CHOL -DSWF_PROFILE_CHOL_STEP=$J" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "== step $J"; python tools/prof/gpu_chol_prof.py 3 | tail -2
done
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
