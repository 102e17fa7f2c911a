#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6f_pytest_gpu.txt 2>&1; tail -6 gpurun_out/r6f_pytest_gpu.txt
for split in 1 0; do
  (cd /tmp && TMPDIR=/tmp SMVS_VIS_SPLIT=$split SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6f_vis$split -o run -- python $ROOT/tools/optimize_timeline.py run > $ROOT/gpurun_out/r6f_vis${split}_run.txt 2>&1)
  trace=$(find gpurun_out/r6f_vis$split -name "*kernel_trace.csv" | head -1)
  python tools/optimize_timeline.py report $trace > gpurun_out/r6f_timeline_nosgm_split$split.txt 2>&1
  rm -rf gpurun_out/r6f_vis$split
  head -1 gpurun_out/r6f_timeline_nosgm_split$split.txt; grep "topo_visibility" gpurun_out/r6f_timeline_nosgm_split$split.txt
done
