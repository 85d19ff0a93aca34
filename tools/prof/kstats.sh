# usage: kstats.sh <windows> <solves> : rocprofv3 kernel stats of the batch-only workload, top kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python $GRAFT_REPO_ROOT/tools/prof/gpu_batch_prof.py ${1:-512} ${2:-4} > /tmp/ks.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True)[0]
tot=0
for r in list(csv.DictReader(open(f)))[:17]:
    print('%-58s calls %4s avg %8.1f us %5s%%'%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
