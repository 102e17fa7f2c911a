#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "clone or scale_space or async" > gpurun_out/r6j_parity.txt 2>&1; tail -2 gpurun_out/r6j_parity.txt
timeout 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_front.py -q -x -k "visib or subview or topolog or configs0 or host_optimize_matches or sphere_960 or ncc" > gpurun_out/r6j_topo.txt 2>&1; tail -2 gpurun_out/r6j_topo.txt
for variant in "new" "old"; do
  if [ $variant = old ]; then export SMVS_VIS_ORDER=patch SMVS_BLUR_XCD=0; fi
  (cd /tmp && TMPDIR=/tmp SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6j_$variant -o run -- python $ROOT/tools/optimize_timeline.py run > $ROOT/gpurun_out/r6j_${variant}_run.txt 2>&1)
  trace=$(find gpurun_out/r6j_$variant -name "*kernel_trace.csv" | head -1)
  python tools/optimize_timeline.py report $trace > gpurun_out/r6j_timeline_nosgm_$variant.txt 2>&1
  rm -rf gpurun_out/r6j_$variant
  echo "== $variant"; head -1 gpurun_out/r6j_timeline_nosgm_$variant.txt; grep "topo_visibility\|blur_y" gpurun_out/r6j_timeline_nosgm_$variant.txt
done
