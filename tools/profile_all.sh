#!/bin/bash
# GPU box, repo root: rocprofv3 stats + PMC passes for every BASELINE bench workload -> gpurun_out/prof_<tag>_<workload>/, then
# pmc.json + traffic.json (tools/make_pmc_json.py: ONE pass set feeds both) into gpurun_out/ AND into profiles/ of this checkout,
# so a bench run later in the same call reads the fresh counters.   bash tools/profile_all.sh <tag> [workload ...]
TAG=${1:-r06}; shift
WLS=${@:-poisson2d_64x64_b32_L4_C64 poisson2d_32x32_b32_L4_C64 burgers2d_64x64_b32_L6_C128 euler20_128x128_b16_C64}
rm -f gpurun_out/pmc_${TAG}.json
for WL in $WLS; do
  bash tools/rocprof_passes.sh ${TAG}_$WL $WL > gpurun_out/prof_${TAG}_$WL.txt 2>&1
  python3 tools/make_pmc_json.py gpurun_out/prof_${TAG}_$WL $WL gpurun_out/pmc_${TAG}.json
  cp gpurun_out/prof_${TAG}_$WL/summary.txt gpurun_out/${TAG}_${WL}_rocprof_summary.txt 2>/dev/null
  f=$(find gpurun_out/prof_${TAG}_$WL/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${WL}_rocprofv3_kernel_stats.csv
  echo "profiled $WL"
done
cp gpurun_out/pmc_${TAG}.json profiles/pmc.json
cp gpurun_out/traffic.json profiles/traffic.json 2>/dev/null
