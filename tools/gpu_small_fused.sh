#!/bin/bash
# GPU box: captured training step at the reference's sizes with the per-layer kernels: fused 13/14-launch iteration vs the captured
# autograd iteration (GADAPT_FUSED=0), and against the one-launch pair where the policy takes it.
cd ${GRAFT_REPO_ROOT:-.}
{
for cfg in "--mesh 11 --hidden_dim 32 --num_train 512 --batch_size 8" "--mesh 15 --hidden_dim 32 --num_train 512 --batch_size 8" "--mesh 23 --hidden_dim 16 --num_train 512 --batch_size 16" "--mesh 20 --hidden_dim 16 --num_train 512 --batch_size 16" "--mesh 11 --hidden_dim 64 --num_train 512 --batch_size 8" "--mesh 23 --hidden_dim 64 --num_train 512 --batch_size 16"; do
  for F in 0 1; do
    echo "== $cfg : captured step, per-layer kernels, fused=$F"
    GADAPT_FUSED=$F GADAPT_SMALL_MESH=0 timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 4 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r06_train_small_fused.log 2>&1
grep -E "^==|meshes/s" gpurun_out/r06_train_small_fused.log | sed -e 's/; losses.*//'
