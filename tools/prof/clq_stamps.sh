# per-phase cycle stamps of one clique (index $1 of the launch's list; one cfg3 window: 0 = dummy, 1..10 speed-bias cliques, then receiver clocks)
cd $GRAFT_REPO_ROOT
for I in ${@:-1 12}; do
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CLQ -DSWF_PROFILE_CLQ_IDX=$I" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "== clique $I (fused grid)"; python tools/prof/gpu_clq_prof.py 1
echo "== clique $I (k_clique_elim, one wave)"; SWF_NO_LAT_FUSE=1 python tools/prof/gpu_clq_prof.py 1
done
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
