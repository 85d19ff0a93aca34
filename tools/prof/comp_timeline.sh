# kernel timeline of one iteration of a cfg3-size reference-topology window:  tools/prof/comp_timeline.sh [ambiguities]
S=${1:-10}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ctl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ctl -o t -- python "$ROOT/tools/prof/gpu_comp_prof.py" 2 20 4 300 $S 8 single > /tmp/ctl.log 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/ctl/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i0 = len(rows) * 2 // 3
while 'k_chol' not in rows[i0]['Kernel_Name']: i0 += 1
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + 18]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-58s start %8.2f us  dur %7.2f us" % (r['Kernel_Name'][:58], s / 1e3, (e - s) / 1e3))
PY
