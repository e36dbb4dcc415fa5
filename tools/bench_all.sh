set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 200 --warmup 20 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/r02_bench.json
echo "bench default done"
for WL in poisson2d_32x32_b32_L4_C64 burgers2d_64x64_b32_L6_C128 euler20_128x128_b16_C64; do
  python bench.py --workload $WL --steps 50 --warmup 10 2>>gpurun_out/bench_err.log | tail -1 > gpurun_out/r02_bench_$WL.json
  echo "bench $WL done"
done
python bench.py --dense-slots --steps 100 --warmup 10 --no-cpu-baseline 2>>gpurun_out/bench_err.log | tail -1 > gpurun_out/r02_bench_dense_slots.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r02_bench_under_rocprofv3.json 2>gpurun_out/prof_bench.log
echo "rocprof bench done"
