#!/bin/bash
# round 6, second GPU visit: reactivate kernel with LDS staging (parity + time), SGM counters, streaming-solver traffic
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "loop or reactivat or update or newton" > gpurun_out/r6b_parity.txt 2>&1; tail -3 gpurun_out/r6b_parity.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-peaks > gpurun_out/r6b_bench.json 2> gpurun_out/r6b_bench.err; tail -2 gpurun_out/r6b_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r6b_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"]); print(d["roofline"]["kernels"])
for s,v in d["roofline"]["by_scale"].items(): print(s, v["kernel_ms"])
PY
timeout 900 bash tools/kernel_pmc.sh sgm "census_main_kernel,warp_kernel,cost_packed_kernel,sgm_all_paths_kernel,sgm_sum_wta_kernel" python $ROOT/tools/sgm_bench.py > gpurun_out/r6b_sgm_pmc.log 2>&1; tail -5 gpurun_out/r6b_sgm_pmc.log
PMC_SETS=traffic timeout 900 bash tools/kernel_pmc.sh stream515k "cg_spmv,cg_update,cg_init" python $ROOT/tools/cg_streaming_cost.py 1920 1080 1 > gpurun_out/r6b_stream_pmc.log 2>&1; tail -12 gpurun_out/r6b_stream_pmc.log
PMC_SETS=traffic timeout 900 bash tools/kernel_pmc.sh stream186k "cg_spmv,cg_update,cg_init" python $ROOT/tools/cg_streaming_cost.py 2304 1296 2 > gpurun_out/r6b_stream186_pmc.log 2>&1; tail -12 gpurun_out/r6b_stream186_pmc.log
