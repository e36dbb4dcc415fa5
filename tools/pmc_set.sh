#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_set.sh <tag> COUNTER [COUNTER ...]   -> per-kernel mean of each counter
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 tools/profile_step.py --steps 3 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        if 'grand_' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
