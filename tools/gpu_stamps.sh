#!/bin/bash
# GPU box, repo root: in-kernel cycle stamps of the backward kernels (variants/stamps.so = make EXTRA=-DGADAPT_STAMPS) at the metric
# workload's tiling and at config 2's (one tile per workgroup).
cd ${GRAFT_REPO_ROOT:-.}
for N in 64 32; do
  echo "== ${N}x${N} meshes, batch 32, hidden 64"
  GADAPT_LIB=variants/stamps.so STAMP_N=$N timeout -k 10 200 python tools/stamp_bwd.py 2>&1 | tail -14
done > gpurun_out/r06_stamps_backward.txt 2>&1
cat gpurun_out/r06_stamps_backward.txt
