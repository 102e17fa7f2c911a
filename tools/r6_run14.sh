#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "scale_space or async or clone" > gpurun_out/r6m_parity.txt 2>&1; tail -2 gpurun_out/r6m_parity.txt
for mode in "--sgm" ""; do
  tagname=tl${mode#--}
  (cd /tmp && TMPDIR=/tmp SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6m_$tagname -o run -- python $ROOT/tools/optimize_timeline.py run $mode > $ROOT/gpurun_out/r6m_${tagname}_run.txt 2>&1)
  trace=$(find gpurun_out/r6m_$tagname -name "*kernel_trace.csv" | head -1)
  python tools/optimize_timeline.py report $trace > gpurun_out/r6m_optimize_timeline_${tagname}.txt 2>&1
  rm -rf gpurun_out/r6m_$tagname
  head -1 gpurun_out/r6m_optimize_timeline_${tagname}.txt; grep "blur_x\|blur_y\|gradients" gpurun_out/r6m_optimize_timeline_${tagname}.txt
done
