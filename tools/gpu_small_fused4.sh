#!/bin/bash
# GPU box: captured training step at the reference's own sizes on the one-launch pair: the fused 4-launch iteration (forward + loss,
# backward, slab sums, chain rule + Adam) against the captured autograd iteration over the same pair (GADAPT_FUSED=0).
cd ${GRAFT_REPO_ROOT:-.}
{
for cfg in "--mesh 11 --hidden_dim 8 --num_train 512 --batch_size 8" "--mesh 11 --hidden_dim 8 --num_train 2048 --batch_size 64" "--mesh 23 --hidden_dim 8 --num_train 512 --batch_size 16" \
           "--mesh 11 --hidden_dim 16 --num_train 512 --batch_size 8" "--mesh 15 --hidden_dim 16 --num_train 2048 --batch_size 64"; do
  for F in 0 1; do
    echo "== $cfg : captured step, one-launch pair, fused=$F"
    GADAPT_FUSED=$F timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 4 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r06_train_small_fused4.log 2>&1
grep -E "^==|meshes/s" gpurun_out/r06_train_small_fused4.log | sed -e 's/; losses.*//'
