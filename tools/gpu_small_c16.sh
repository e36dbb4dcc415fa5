#!/bin/bash
# GPU box: the policies of the one-launch small-mesh pair at hidden 16 / 32: evaluation sweep + captured training step, forced on vs off.
mkdir -p gpurun_out
timeout -k 10 400 python tools/sweep_small_mesh.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sweep_small.log
cat gpurun_out/sweep_small.log
{
for cfg in "--mesh 11 --hidden 16 --num_train 512 --batch_size 8" "--mesh 15 --hidden 16 --num_train 512 --batch_size 8" "--mesh 15 --hidden 16 --num_train 1024 --batch_size 64" "--mesh 20 --hidden 16 --num_train 512 --batch_size 16" "--mesh 11 --hidden 32 --num_train 512 --batch_size 8" "--mesh 15 --hidden 32 --num_train 512 --batch_size 8"; do
  echo "== $cfg : captured step, per-layer kernels"
  GADAPT_SMALL_MESH=0 timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 4 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== $cfg : captured step, one-launch pair (forced)"
  GADAPT_SMALL_MESH=2 timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 4 2>&1 | grep -v amdgpu.ids | tail -1
done
} > gpurun_out/train_small_c16.log 2>&1
grep -E "^==|meshes/s" gpurun_out/train_small_c16.log | sed -e 's/; losses.*//'
