#!/bin/bash
# A/B the prebuilt library variants under build/variants on the GPU box over the bench workloads: ms/step and per-kernel us.
WL=${WL:-"poisson2d_64x64_b32_L4_C64 burgers2d_64x64_b32_L6_C128 euler20_128x128_b16_C64 poisson2d_64x64_b32_L4_C32"}
for rep in 1 2; do
for w in $WL; do
  for lib in build/variants/*.so; do
    GADAPT_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d.get('kernels',{})
print('$w', '$lib', 'ms/step', d['ms_per_step'], 'meshes/s', d['value'], {n:k[n]['avg_us'] for n in k})"
  done
done
done
