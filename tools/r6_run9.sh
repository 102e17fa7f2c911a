#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for wg in 120 0; do
SMVS_CG_TRACE_WG=$wg python tools/cg_trace.py gpurun_out/r6i_cg_trace_raw_$wg.txt > gpurun_out/r6i_cg_trace_wg$wg.txt 2>&1
grep -A10 "iteration 5, the waves" gpurun_out/r6i_cg_trace_wg$wg.txt | head -12
grep "it  [3-6]:" gpurun_out/r6i_cg_trace_wg$wg.txt | head -4
done
