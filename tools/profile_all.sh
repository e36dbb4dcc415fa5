#!/bin/bash
# GPU box, repo root: rocprofv3 stats + PMC passes for every bench workload -> gpurun_out/prof_<tag>_<workload>/, then
# profiles/pmc.json (tools/make_pmc_json.py).  bash tools/profile_all.sh <tag> [workload ...]
TAG=${1:-r02}; shift
WLS=${@:-poisson2d_64x64_b32_L4_C64 poisson2d_32x32_b32_L4_C64 burgers2d_64x64_b32_L6_C128 euler20_128x128_b16_C64}
for WL in $WLS; do
  bash tools/rocprof_passes.sh ${TAG}_$WL $WL > gpurun_out/prof_${TAG}_$WL.txt 2>&1
  python3 tools/make_pmc_json.py gpurun_out/prof_${TAG}_$WL $WL gpurun_out/pmc_${TAG}.json
  echo "profiled $WL"
done
