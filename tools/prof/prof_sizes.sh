cd /tmp && export TMPDIR=/tmp
for B in 256 64; do
rm -rf /tmp/pp$B
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$B -o b -- python $GRAFT_REPO_ROOT/tools/prof/gpu_batch_prof.py $B 4 > /tmp/pp$B.log 2>&1
python - $B <<'PY'
import csv,glob,sys
B=sys.argv[1]
f=glob.glob('/tmp/pp%s/**/*kernel_stats.csv'%B, recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print('--- windows',B)
tot=0
for r in rows[:16]:
    print('%-58s calls %4s avg %8.1f us %5s%%'%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
