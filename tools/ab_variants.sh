#!/bin/bash
# A/B the prebuilt library variants under build/variants on the GPU box: prints per-kernel µs for each.
for lib in build/variants/*.so; do
  GADAPT_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$lib', 'ms/step', d['ms_per_step'], 'meshes/s', d['value'], {n:k[n]['avg_us'] for n in k})"
done
