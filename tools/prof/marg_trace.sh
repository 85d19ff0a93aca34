#!/bin/bash
# kernel-level account of ONE eigen marginalisation call of the cfg5 window (263-dimension tail): rocprofv3 kernel trace of
# tools/prof/marg_eigen_time.py, the launches between the last two k_marg_rescue launches of the eigen form summed by kernel
# usage (on the GPU box): bash tools/prof/marg_trace.sh
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mgtr
rocprofv3 --kernel-trace --output-format csv -d /tmp/mgtr -o r -- python $GRAFT_REPO_ROOT/tools/prof/marg_eigen_time.py > /tmp/mgtr.log 2>&1
sed -n 2,2p /tmp/mgtr.log
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/mgtr/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_marg_pchol")]
i0 = idx[-1]
j = i0
while not rows[j]["Kernel_Name"].startswith("k_marg_rescue"): j -= 1
t0 = int(rows[j]["Start_Timestamp"])
acc = collections.OrderedDict()
k = j
while k < len(rows):
    nm = rows[k]["Kernel_Name"].split("(")[0][:40]
    if k > i0 and nm.startswith("k_marg_rescue"): break
    d = (int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3
    a = acc.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += d
    last = int(rows[k]["End_Timestamp"])
    k += 1
for nm, (c, t) in acc.items(): print("%-42s n %4d total %8.1f us  avg %6.1f" % (nm, c, t, t / c))
print("first launch to last end: %.1f us" % ((last - t0) / 1e3))
PY
