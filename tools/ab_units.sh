#!/bin/bash
# GPU box, repo root: A/B of per-translation-unit builds (variants/<name>.so, built with `make FLAGS_<unit>=... LIB=variants/<name>.so`)
# on one bench workload: value + per-kernel event-pair averages of each.   bash tools/ab_units.sh <workload> <name> [<name> ...]
cd ${GRAFT_REPO_ROOT:-.}
WL=$1; shift
for V in base "$@"; do
  LIBV=""; [ "$V" != base ] && LIBV="variants/$V.so"
  GADAPT_LIB=$LIBV timeout -k 10 200 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-companion --no-train-loop --no-other-workloads --no-gat-plus \
      > gpurun_out/ab_$V.json 2> gpurun_out/ab_$V.err
  python - "$V" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open(f'gpurun_out/ab_{v}.json'))
    print(v, d['value'], d['ms_per_step'], {k: {n: x['avg_us'] for n, x in kk['variants'].items()} for k, kk in d['kernels'].items()})
except Exception as e:
    print(v, 'FAILED', e, open(f'gpurun_out/ab_{v}.err').read()[-600:])
PY
done
