#!/bin/bash
# GPU box: the small-mesh tests and the rollout example at the reference's sizes (round 5; profiles/r05_rollout_small_mesh.txt).
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_smallmesh.py tests/test_gpu_training.py -x -q -k "small or 1d_global" > gpurun_out/t_small.log 2>&1; tail -5 gpurun_out/t_small.log
for a in "--mesh 11 --dim 2 --hidden 8 --steps 200" "--mesh 21 --dim 1 --hidden 8 --steps 200" "--mesh 23 --dim 2 --hidden 16 --steps 100" "--mesh 15 --dim 2 --hidden 32 --steps 100"; do echo "== $a"; timeout -k 10 100 python examples/burgers_rollout.py $a 2>&1 | grep -v amdgpu.ids | grep "hipgraph\|max |per"; done > gpurun_out/rollout_small.log 2>&1; cat gpurun_out/rollout_small.log
