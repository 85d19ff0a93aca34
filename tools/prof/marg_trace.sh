cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mg1 && rocprofv3 --kernel-trace --output-format csv -d /tmp/mg1 -o s -- python $GRAFT_REPO_ROOT/tools/prof/marg_eigen_time.py > /tmp/mg1.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/mg1/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_marg_bj' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print(len(d), 'k_marg_bj launches in 4 calls')
per=len(d)//4
one=d[per:2*per]
import os
W=int(os.environ.get('LPS','34'))
print('second call: mean duration (us) of the launches of each sweep (%d launches per sweep):'%W)
print(' '.join('%5.1f'%(sum(one[s:s+W])/max(1,len(one[s:s+W]))) for s in range(0,len(one),W)))
st=[int(r['Start_Timestamp']) for r in rows][per:2*per]; en=[int(r['End_Timestamp']) for r in rows][per:2*per]
print('span of the k_marg_bj launches of one call: %.2f ms; sum of durations %.2f ms'%((en[-1]-st[0])/1e6, sum(one)/1e3))
rows=[r for r in csv.DictReader(open(f)) if 'k_marginalize' in r['Kernel_Name']]
print('k_marginalize launches:', [(r['Kernel_Name'][:30], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows[:6]])
PY
