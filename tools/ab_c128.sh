#!/bin/bash
# A/B the prebuilt library variants under build/variants on the hidden-128 workload (BASELINE config 4).
for lib in build/variants/*.so; do
  GADAPT_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --workload burgers2d_64x64_b32_L6_C128 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$lib', 'ms/step', d['ms_per_step'], 'meshes/s', d['value'], {n:k[n]['avg_us'] for n in k})"
done
