#!/bin/bash
# A/B library variants on the small per-step kernels: rocprofv3 kernel trace of a short bench run, mean duration per kernel.
export TMPDIR=/tmp
for lib in build/variants/*.so; do
  out=gpurun_out/ab_small_$(basename $lib .so)
  GADAPT_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  echo "== $lib"; python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('loss_forward', 'encode_linear', 'slab_reduce', 'coeffs_', 'adam_step')):
        print(f"   {r['Name'][:40]:40s} {float(r['AverageNs'])/1e3:7.2f} us")
PY
done
