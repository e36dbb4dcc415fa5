#!/bin/bash
# GPU box, repo root: the round's evidence in one call - stamps, PMC passes (-> profiles/pmc.json), bench lines + rocprofv3 kernel stats.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r06}
bash tools/gpu_stamps.sh > /dev/null 2>&1; echo "stamps done"
bash tools/profile_all.sh $TAG > gpurun_out/${TAG}_profile_all.log 2>&1; echo "profile_all done"; tail -3 gpurun_out/${TAG}_profile_all.log
bash tools/bench_all.sh $TAG > gpurun_out/${TAG}_bench_all.log 2>&1; echo "bench_all done"; tail -3 gpurun_out/${TAG}_bench_all.log
python -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('profile'), d['roofline']['traffic'])"
