cd $GRAFT_REPO_ROOT
for F in "" "-DSWF_EXP_V_NOBPERM" "-DSWF_EXP_V_NOPOLL" "-DSWF_EXP_V_NOBPERM -DSWF_EXP_V_NOPOLL"; do
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CHOLW $F" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "== flags: $F"; python tools/prof/gpu_chol_wprof.py 3 11 2>&1 | grep "wave 0\|wave 11\|inverse wave"
done
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
