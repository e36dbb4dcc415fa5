#!/bin/bash
# GPU box: captured / eager training loop at the reference's sizes, per-layer kernels against the one-launch pair (round 5; profiles/r05_train_small_mesh.txt).
mkdir -p gpurun_out
{
for cfg in "--mesh 11 --hidden 8 --num_train 512 --batch_size 8" "--mesh 11 --hidden 8 --num_train 2048 --batch_size 64" "--mesh 23 --hidden 8 --num_train 512 --batch_size 16"; do
  echo "== $cfg : captured step, per-layer kernels"
  GADAPT_SMALL_MESH=0 timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 4 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== $cfg : captured step, one-launch pair"
  timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 4 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== $cfg : eager loop, per-layer kernels"
  GADAPT_SMALL_MESH=0 timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 3 --eager 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== $cfg : eager loop, one-launch pair"
  timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 3 --eager 2>&1 | grep -v amdgpu.ids | tail -1
done
} > gpurun_out/train_small.log 2>&1
grep -E "^==|meshes/s" gpurun_out/train_small.log | sed -e 's/; losses.*//'
