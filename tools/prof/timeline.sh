# per-dispatch timeline of ONE iteration of the single-window solve (both streams): kernel, start / end relative to the solve's first dispatch
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/tools/prof/gpu_single_prof.py ${1:-3} > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = max(i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_init'))
sel = rows[idx:]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel[:60]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%-46s q%-3s start %8.2f us  dur %7.2f us  end %8.2f' % (r['Kernel_Name'][:46], r.get('Queue_Id', '?'), s / 1e3, (e - s) / 1e3, e / 1e3))
PY
