# usage: pmc_kernel.sh "<COUNTER ...>" <kernel-name-substring> [windows] : per-launch averages of the counters for the matching kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout 300 rocprofv3 --pmc $1 --output-format csv -d /tmp/pk -o p -- python $GRAFT_REPO_ROOT/tools/prof/gpu_batch_prof.py ${3:-512} 2 > /tmp/pk.log 2>&1
python - "$2" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pk/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] not in r["Kernel_Name"]: continue
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]][r["Counter_Name"]] += 1
for k, v in acc.items():
    print(k[:70], {c: round(x / max(1, n[k][c])) for c, x in v.items()})
PY
