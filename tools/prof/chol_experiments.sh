# the register-resident Cholesky on one box: kernel stats (batch 512, single window), per-phase stamps
cd $GRAFT_REPO_ROOT
bash tools/prof/kstats.sh 512 2 2>&1 | grep "chol\|lm_schur"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o s -- python $GRAFT_REPO_ROOT/tools/prof/gpu_single_prof.py 20 > /tmp/ks1.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks1/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:8]:
    print('1win %-58s calls %4s avg %8.1f us %5s%%'%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
cd $GRAFT_REPO_ROOT
SWF_EXTRA_FLAGS="-DSWF_PROFILE_CHOL" python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
python tools/prof/gpu_chol_prof.py 3
python -c "from rtk_visual_inertial_navigation_amd import build; build.build(force=True)" > /dev/null 2>&1
