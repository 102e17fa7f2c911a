#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_front.py -q -k "operating_point or scale_3_portrait or scale_1_portrait or gamma" > gpurun_out/r6g_front.txt 2>&1; tail -5 gpurun_out/r6g_front.txt; grep "rel. L2\|masks differ" gpurun_out/r6g_front.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r6g_parity.txt 2>&1; tail -3 gpurun_out/r6g_parity.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-peaks > gpurun_out/r6g_bench.json 2> gpurun_out/r6g_bench.err; tail -2 gpurun_out/r6g_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r6g_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"]); print(d["roofline"]["kernels"])
for s,v in d["roofline"]["by_scale"].items(): print(s, v["kernel_ms"])
PY
