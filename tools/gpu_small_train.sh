mkdir -p gpurun_out
{
for cfg in "--mesh 11 --hidden 8 --num_train 512 --batch_size 8" "--mesh 11 --hidden 8 --num_train 2048 --batch_size 64" "--mesh 23 --hidden 8 --num_train 512 --batch_size 16" "--mesh 15 --hidden 16 --num_train 512 --batch_size 16"; do
  for sm in 1 0; do
    echo "== $cfg GADAPT_SMALL_MESH=$sm"
    GADAPT_SMALL_MESH=$sm timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 4 2>&1 | grep -v amdgpu.ids | tail -3
    echo "== $cfg GADAPT_SMALL_MESH=$sm --eager"
    GADAPT_SMALL_MESH=$sm timeout -k 10 120 python examples/train_mesh_loss.py $cfg --epochs 3 --eager 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/train_small.log 2>&1
cat gpurun_out/train_small.log
