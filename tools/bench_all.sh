#!/bin/bash
# GPU box, repo root: the round's bench lines and the rocprofv3 kernel statistics of the same commands.
#   bash tools/bench_all.sh <tag>      (after tools/profile_all.sh <tag>, which refreshes profiles/pmc.json on the box)
# (the profiled runs time ONE workload each - no other_workloads / gat_plus legs: kernels of the same instantiation at other sizes would
# be averaged into the same csv row.)
# Order matters: bench.py under rocprofv3 runs FIRST and its kernel-stats csv is copied into profiles/ of this checkout, so the
# plain bench runs that follow quote the duration of the committed summary beside their own (roofline.profile).
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out
DEF=poisson2d_64x64_b32_L4_C64
for WL in $DEF poisson2d_32x32_b32_L4_C64 burgers2d_64x64_b32_L6_C128 euler20_128x128_b16_C64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_$WL -- python3 bench.py --workload $WL --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-gat-plus \
      2> $OUT/prof_bench_$WL.log | tail -1 > $OUT/${TAG}_bench_${WL}_under_rocprofv3.json
  f=$(find $OUT/prof_bench_$WL -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" profiles/${TAG}_bench_${WL}_rocprofv3_kernel_stats.csv; cp "$f" $OUT/${TAG}_bench_${WL}_rocprofv3_kernel_stats.csv; fi
  echo "rocprof bench $WL done"
done
python bench.py --steps 200 --warmup 20 2>$OUT/bench_err.log | tail -1 > $OUT/${TAG}_bench.json
echo "bench default done"
for WL in poisson2d_32x32_b32_L4_C64 burgers2d_64x64_b32_L6_C128 euler20_128x128_b16_C64 poisson2d_64x64_b32_L4_C64_learn_step poisson2d_64x64_b32_L4_C64_GAT_plus; do
  python bench.py --workload $WL --steps 50 --warmup 10 2>>$OUT/bench_err.log | tail -1 > $OUT/${TAG}_bench_$WL.json
  echo "bench $WL done"
done
# (the dense-slot flow is timed inside every default run now: value_dense_slots / ms_per_step_dense_slots on the same line)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_gat -- python3 bench.py --workload poisson2d_64x64_b32_L4_C64_GAT_plus --steps 20 --warmup 5 --no-cpu-baseline \
    2> $OUT/prof_bench_gat.log | tail -1 > $OUT/${TAG}_bench_GAT_plus_under_rocprofv3.json
f=$(find $OUT/prof_bench_gat -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_bench_poisson2d_64x64_b32_L4_C64_GAT_plus_rocprofv3_kernel_stats.csv
echo "all done"
