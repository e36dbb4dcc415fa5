#!/bin/bash
# GPU box: what separates config 5 (128-node mesh rows, 262k nodes) from the metric workload (64-node rows, 131k nodes)?
# Same 4-layer hidden-64 model on four mesh batches: {64, 128}-node rows x {131k, 262k} nodes.  -> gpurun_out/locality.log
set -e
mkdir -p gpurun_out
: > gpurun_out/locality.log
for wl in poisson2d_64x64_b32_L4_C64 poisson2d_64x64_b64_L4_C64 poisson2d_128x128_b8_L4_C64 poisson2d_128x128_b16_L4_C64; do
  timeout -k 10 300 python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-train-loop --no-gat-plus --no-companion > gpurun_out/loc_$wl.json 2> gpurun_out/loc_$wl.err
  python - "$wl" >> gpurun_out/locality.log <<'PY'
import json, sys
wl = sys.argv[1]
d = json.loads(open(f'gpurun_out/loc_{wl}.json').read().strip().splitlines()[-1])
k = d['kernels']
print(f"{wl:34s} {d['value']:10.1f} meshes/s {d['ms_per_step']:.4f} ms | " + ' '.join(f"{n} {v['avg_us']:.1f}" for n, v in k.items()) + f" | dense dominant {d['roofline']['avg_launch_us']} us")
PY
done
cat gpurun_out/locality.log
