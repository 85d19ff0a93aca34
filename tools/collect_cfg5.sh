#!/bin/bash
# BASELINE config 5 (stress: 40 keyframes / 1000 features / 20 satellites / marginalisation prior obtained on the device) under
# rocprofv3:   tools/collect_cfg5.sh <out_dir> [batch_windows = 32]
# kernel trace + the HBM (FETCH_SIZE, WRITE_SIZE: separate passes) and fp64 matrix-core counters, summarised per kernel, for the
# single window (latency regime) AND for a batch of cfg5 windows (where the reduced-solve MFMA utilisation is non-trivial).
set -u
OUT=$(realpath -m "${1:-gpurun_out/prof}")
NB=${2:-32}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/swf5 && mkdir -p /tmp/swf5
export PYTHONPATH="$ROOT"
( cd "$ROOT" && python tools/prof/gpu_cfg5_prof.py ) > "$OUT/cfg5_single_window.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/swf5/kt -o k -- python "$ROOT/tools/prof/gpu_cfg5_run.py" > /tmp/swf5/kt.log 2>&1
cp "$(find /tmp/swf5/kt -name '*kernel_stats.csv' | head -1)" "$OUT/cfg5_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/swf5/pmc_$C -o p -- python "$ROOT/tools/prof/gpu_cfg5_run.py" > /tmp/swf5/pmc_$C.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/swf5/pmc_mfma -o p -- python "$ROOT/tools/prof/gpu_cfg5_run.py" > /tmp/swf5/pmc_mfma.log 2>&1
python "$ROOT/tools/summarize_mfma_pmc.py" "$(find /tmp/swf5/pmc_mfma -name '*counter_collection.csv' | head -1)" "$OUT/cfg5_pmc_mfma.json" "tools/prof/gpu_cfg5_run.py 1 3 (ONE cfg5 window, latency regime: the chip is mostly idle, so utilisation figures are tiny by construction)"
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"note": "BASELINE cfg5 (one window: 40 keyframes, 1000 features, 20000 observations, 20 satellites, 107-dim dense prior; n_red = 440), "
               "tools/prof/gpu_cfg5_run.py = 3 solves of 8 dogleg iterations; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; unit KB; "
               "gfx950: hbm_read_bytes ~= 2 * FETCH_SIZE * 1024 for streaming reads (MI355X_MICROARCH.md).  A single window lives in L2 / Infinity Cache: "
               "these are traffic counts, not a bandwidth claim."}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/swf5/pmc_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0, 0.0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") != c: continue
            a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    res[c] = {k: {"launches": v[0], "avg_kb_per_launch": v[1] / max(1, v[0])} for k, v in acc.items()}
json.dump(res, open(out + "/cfg5_pmc_fetch_write.json", "w"), indent=1)
PY
head -14 "$OUT/cfg5_kernel_stats.csv" | cut -c1-150

# ---- the same for a batch of $NB cfg5 windows
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/swf5/ktb -o k -- python "$ROOT/tools/prof/gpu_cfg5_run.py" $NB 2 > /tmp/swf5/ktb.log 2>&1
cp "$(find /tmp/swf5/ktb -name '*kernel_stats.csv' | head -1)" "$OUT/cfg5_batch${NB}_kernel_stats.csv"
grep -E "cfg5 done|skipped" /tmp/swf5/ktb.log > "$OUT/cfg5_batch${NB}_workload.txt"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/swf5/pmcb_$C -o p -- python "$ROOT/tools/prof/gpu_cfg5_run.py" $NB 1 > /tmp/swf5/pmcb_$C.log 2>&1
done
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/swf5/pmcb_mfma -o p -- python "$ROOT/tools/prof/gpu_cfg5_run.py" $NB 1 > /tmp/swf5/pmcb_mfma.log 2>&1
python "$ROOT/tools/summarize_mfma_pmc.py" "$(find /tmp/swf5/pmcb_mfma -name '*counter_collection.csv' | head -1)" "$OUT/cfg5_batch${NB}_pmc_mfma.json" "tools/prof/gpu_cfg5_run.py $NB 1 ($NB cfg5 windows, marginalisation prior obtained on the device, one 8-iteration solve)"
python - "$OUT" "$NB" <<'PY'
import csv, glob, json, sys, collections
out, nb = sys.argv[1], sys.argv[2]
res = {"note": "BASELINE cfg5 as a batch of %s windows (40 keyframes, 1000 features, 20 satellites, marginalisation prior obtained by marginalising a 41st "
               "frame on the device; n_red ~ 440), tools/prof/gpu_cfg5_run.py %s 1 = one 8-iteration solve; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
               "separate passes; unit KB; gfx950: hbm_read_bytes ~= 2 * FETCH_SIZE * 1024 for streaming reads (MI355X_MICROARCH.md)." % (nb, nb)}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/swf5/pmcb_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0, 0.0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") != c: continue
            a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    res[c] = {k: {"launches": v[0], "avg_kb_per_launch": v[1] / max(1, v[0])} for k, v in acc.items()}
json.dump(res, open(out + "/cfg5_batch%s_pmc_fetch_write.json" % nb, "w"), indent=1)
PY
head -14 "$OUT/cfg5_batch${NB}_kernel_stats.csv" | cut -c1-150
