#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv holding SQ_INSTS_VALU_MFMA_F64, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE
into per-kernel matrix-core figures:  tools/summarize_mfma_pmc.py <counter_collection.csv> <out.json>

Normalisation (checked against the kernel-trace durations of the same workload):
  * GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, so wall cycles = GRBM_GUI_ACTIVE / 8 (k_chol_rr2: 3.79e6 / 8 = 473k cycles
    = 197 us at 2.4 GHz, kernel trace says 191 us);
  * SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs and comes out at exactly 64 cycles per v_mfma_f64_16x16x4_f64 (16 passes x 4),
    i.e. the datasheet issue rate; so mfma_util_pct = busy / (wall cycles * 1024 SIMDs) is utilisation against the 78.6 TFLOP/s
    datasheet peak.  The sustained ceiling measured on this part is 47.7 TFLOP/s (one MFMA per ~100 cycles per SIMD), so
    frac_of_measured_ceiling = mfma_util_pct * 78.6 / 47.7.
  * flops = 2048 per instruction (16 x 16 x 4 x 2); executed flops include zero-padded tile lanes.
"""
import csv, json, sys, collections

XCDS, SIMDS, CLK_GHZ = 8, 1024, 2.4
src, dst = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "tools/prof/gpu_batch_prof.py 512 2 (512 cfg4 windows, batch only)"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(src)):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[r["Kernel_Name"]] += 1
res = {"note": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE in its own pass (no tracing), workload "
               + workload + "; per-launch averages.  See tools/summarize_mfma_pmc.py for the "
               "normalisation.  Durations under counter collection are a few % longer than in the kernel trace.", "kernels": {}}
for k, v in acc.items():
    n = max(1, cnt[k]); ins = v.get("SQ_INSTS_VALU_MFMA_F64", 0.0) / n
    if ins <= 0: continue
    wall = v["GRBM_GUI_ACTIVE"] / n / XCDS; busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    util = 100.0 * busy / (wall * SIMDS)
    res["kernels"][k] = {"launches": n, "mfma_f64_instructions": ins, "mfma_busy_cycles": busy, "wall_cycles": wall,
                         "wall_us_at_2.4GHz": wall / CLK_GHZ / 1e3, "executed_gflop": ins * 2048 / 1e9,
                         "executed_tflops": ins * 2048 / (wall / CLK_GHZ) / 1e3, "mfma_util_pct": util,
                         "pct_of_measured_47.7TF_ceiling": util * 78.6 / 47.7}
json.dump(res, open(dst, "w"), indent=1)
for k, v in res["kernels"].items(): print(k[:44], {a: round(b, 3) for a, b in v.items()})
