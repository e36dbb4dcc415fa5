#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/rocprof_passes.sh <tag> [workload]
# Writes gpurun_out/prof_<tag>/{stats,pmc*}/...; kernel-trace/stats and each --pmc set are separate runs.
set -e
TAG=${1:-r01}; WL=${2:-poisson2d_64x64_b32_L4_C64}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/profile_step.py --workload $WL --steps 10 > $OUT/stats.log 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc$i -- python3 tools/profile_step.py --workload $WL --steps 3 > $OUT/pmc$i.log 2>&1 || echo "pmc set $i failed: $SET"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
