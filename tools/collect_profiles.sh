#!/bin/bash
# Collect the rocprofv3 evidence bench.py / DESIGN.md cite, on a GPU box:
#   tools/collect_profiles.sh <out_dir>          (e.g. gpurun_out/r01)
#  1. --kernel-trace --stats of the batch-only workload (512 windows, tools/prof/gpu_batch_prof.py)
#  2. --kernel-trace --stats of the single-window workload (tools/prof/gpu_single_prof.py)
#  3. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slot limit; never combined with tracing)
#  4. --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (matrix-core utilisation per kernel)
# Summaries (csv / json) land in <out_dir>; copy them to profiles/<round>/ to have them judged.
set -u
OUT=$(realpath -m "${1:-gpurun_out/prof}")
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/swfprof && mkdir -p /tmp/swfprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/swfprof/batch -o b -- python "$ROOT/tools/prof/gpu_batch_prof.py" 512 4 > /tmp/swfprof/batch.log 2>&1
cp "$(find /tmp/swfprof/batch -name '*kernel_stats.csv' | head -1)" "$OUT/batch512_kernel_stats.csv"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/swfprof/single -o s -- python "$ROOT/tools/prof/gpu_single_prof.py" 20 > /tmp/swfprof/single.log 2>&1
cp "$(find /tmp/swfprof/single -name '*kernel_stats.csv' | head -1)" "$OUT/single_window_kernel_stats.csv"
# 2b. the bench command itself, without the single-window / CPU legs (same kernel names would dilute the averages)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/swfprof/bench -o n -- python "$ROOT/bench.py" --no-single-window --no-cpu-baseline --no-live-traffic > "$OUT/bench_nosingle.json" 2> /tmp/swfprof/bench.log
cp "$(find /tmp/swfprof/bench -name '*kernel_stats.csv' | head -1)" "$OUT/bench_nosingle_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/swfprof/pmc_$C -o p -- python "$ROOT/tools/prof/gpu_batch_prof.py" 512 2 > /tmp/swfprof/pmc_$C.log 2>&1
done
# 4. matrix-core counters in their own pass: instructions, busy cycles, GPU-active cycles -> MFMA utilisation per kernel
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/swfprof/pmc_mfma -o p -- python "$ROOT/tools/prof/gpu_batch_prof.py" 512 2 > /tmp/swfprof/pmc_mfma.log 2>&1
python "$ROOT/tools/summarize_mfma_pmc.py" "$(find /tmp/swfprof/pmc_mfma -name '*counter_collection.csv' | head -1)" "$OUT/batch512_pmc_mfma.json"
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slot limit), workload tools/prof/gpu_batch_prof.py 512 2 "
               "(512 cfg4 windows, 8 dogleg iterations, batch only).  Counter unit = KB.  Per MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE "
               "reports 1/2 of the bytes of wide coalesced reads, so hbm_read_bytes ~= 2 * FETCH_SIZE * 1024 for streaming kernels; "
               "WRITE_SIZE is uncalibrated.  Infinity-Cache hits are counted."}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/swfprof/pmc_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0, 0.0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") != c: continue
            a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    res[c] = {k: {"launches": v[0], "avg_kb_per_launch": v[1] / max(1, v[0])} for k, v in acc.items()}
json.dump(res, open(out + "/batch512_pmc_fetch_write.json", "w"), indent=1)
print("pmc kernels:", len(res["FETCH_SIZE"]), len(res["WRITE_SIZE"]))
PY
head -16 "$OUT/batch512_kernel_stats.csv" | cut -c1-150
# 5. (round 5) the per-dispatch timeline of one window's solve, and the batch-size sweep
bash "$ROOT/tools/prof/timeline.sh" 3 > "$OUT/single_window_timeline.txt" 2>&1
cd "$ROOT" && python tools/prof/batch_size_sweep.py > "$OUT/batch_size_sweep.txt" 2>&1
