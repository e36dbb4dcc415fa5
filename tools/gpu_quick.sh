mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -k "wide_backward" -x -q > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
GADAPT_LIB=build/diag/lib_stamps.so timeout -k 10 200 python tools/stamp_wide_bwd.py > gpurun_out/st1.log 2>&1; tail -16 gpurun_out/st1.log
timeout -k 10 400 python bench.py --steps 30 --warmup 5 > gpurun_out/b1.json 2> gpurun_out/b1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/b1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_dense_slots"), d.get("ms_per_step_dense_slots"))
for k,v in d["kernels"].items():
    print(k, v["launches_per_step"], v["avg_us"], {n:(x["avg_us"],x["launches_per_step"]) for n,x in v["variants"].items()})
PY
