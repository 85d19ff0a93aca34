cd $GRAFT_REPO_ROOT
for F in "-DSWF_PROFILE_CHOL" "-DSWF_PROFILE_CHOL -DSWF_CHOL_PRIO"; do
SWF_EXTRA_FLAGS="$F" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "== $F"; python tools/prof/gpu_chol_prof.py 3
done
SWF_EXTRA_FLAGS="-DSWF_CHOL_PRIO" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
bash tools/prof/kstats.sh 512 2 2>&1 | grep "chol"
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
