#!/bin/bash
# round 6, first GPU visit: the new headline, SGM counters, the streaming solver above the resident limit, patch flops
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
( time timeout 900 python bench.py > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err ) 2> gpurun_out/r6a_bench.time
tail -3 gpurun_out/r6a_bench.err; cat gpurun_out/r6a_bench.time | tail -3
timeout 600 python -m pytest tests/test_gpu_front.py -q -x -k "bench_contract or newton_steps_workload" > gpurun_out/r6a_contract.txt 2>&1; tail -3 gpurun_out/r6a_contract.txt
timeout 900 bash tools/kernel_pmc.sh sgm "census_main_kernel,warp_kernel,cost_packed_kernel,sgm_all_paths_kernel,sgm_sum_wta_kernel" python tools/sgm_bench.py > gpurun_out/r6a_sgm_pmc.log 2>&1; tail -5 gpurun_out/r6a_sgm_pmc.log
(timeout 300 python tools/cg_streaming_cost.py 2304 1296 2; timeout 600 python tools/cg_streaming_cost.py 1920 1080 1) > gpurun_out/r6a_streaming_cost.txt 2>&1; tail -4 gpurun_out/r6a_streaming_cost.txt
PMC_SETS=traffic timeout 900 bash tools/kernel_pmc.sh stream515k "cg_spmv,cg_update,cg_init" python tools/cg_streaming_cost.py 1920 1080 1 > gpurun_out/r6a_stream_pmc.log 2>&1; tail -12 gpurun_out/r6a_stream_pmc.log
timeout 600 bash tools/patch_flops.sh > gpurun_out/r6a_patch_flops.log 2>&1; tail -6 gpurun_out/r6a_patch_flops.log
