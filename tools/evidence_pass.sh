#!/bin/bash
# The round's evidence pass on ONE GPU box (gpurun -- 'bash tools/evidence_pass.sh <out_dir>'): the GPU tier, the rocprofv3 sets
# (batch / single window / cfg5 / reference topology) and the default bench line.  Copy what is to be judged into profiles/<round>/.
OUT=${1:-gpurun_out/evidence}
mkdir -p "$OUT"
python -m pytest tests -m gpu -q > "$OUT/gputest.log" 2>&1; tail -3 "$OUT/gputest.log"
bash tools/collect_profiles.sh "$OUT" > "$OUT/collect.log" 2>&1; tail -4 "$OUT/collect.log"
bash tools/collect_cfg5.sh "$OUT" 128 > "$OUT/collect_cfg5.log" 2>&1; tail -3 "$OUT/collect_cfg5.log"
bash tools/prof/comp_profile.sh "$OUT" 64 > "$OUT/collect_comp.log" 2>&1; head -3 "$OUT/composite_workload.txt"
python tools/prof/eigroot_time.py > "$OUT/composite_eigen_root.txt" 2>&1; cat "$OUT/composite_eigen_root.txt"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python -c "
import json, sys; d = json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['single_window']['us_per_iteration'], d['stress']['single_window']['us_per_iteration'], d['rtk_topology']['single_window'], d['rtk_topology']['eigen_root'])"
