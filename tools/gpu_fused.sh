#!/bin/bash
# GPU box, repo root: the fused-step tests, then the default bench line under rocprofv3 (kernel trace + stats) and plain.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -m gpu -x -q > $OUT/r6_t3.log 2>&1; tail -15 $OUT/r6_t3.log
rm -rf $OUT/prof_r6
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r6 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-gat-plus --no-companion --no-train-loop \
   2> $OUT/prof_r6.log | tail -1 > $OUT/r6_bench_under_rocprofv3.json
f=$(find $OUT/prof_r6 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/r6_kernel_stats.csv
rm -rf $OUT/prof_r6
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:24]:
    print(f"{r['Name'][:110]:<110} calls={r['Calls']:>6} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:8.2f}")
PY
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --no-gat-plus > $OUT/r6_b3.json 2> $OUT/r6_b3.err
python -c "
import json;d=json.load(open('gpurun_out/r6_b3.json'));print(d['value'],d['ms_per_step'],d['windows'],d['config']['fused_step'],d['config']['fused_step_off_reason']);print(d['train_loop']['value'], d.get('value_dense_slots'))"
