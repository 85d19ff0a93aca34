# which segment does k_post_chol<2> wait for?  Durations of the launch with segments left out (1 cliques, 2 scalar J v, 4 IMU J v, 8 prior J v)
cd $GRAFT_REPO_ROOT
for M in 0 14 13 11 7; do
SWF_EXTRA_FLAGS="-DSWF_DEBUG_POST_SKIP=$M" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "== segments skipped (mask) $M"; SWF_POST_SPLIT=1 bash tools/prof/timeline.sh 3 | grep "k_post_chol<2>" | tail -2
done
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
