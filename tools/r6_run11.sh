#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_surface.py tests/test_gpu_parity.py -q -x > gpurun_out/r6k_parity.txt 2>&1; tail -2 gpurun_out/r6k_parity.txt
timeout 1500 python -m pytest tests/test_gpu_front.py -q -x -k "plane or config1 or sphere or topology or reconstruct or config5 or host_optimize" > gpurun_out/r6k_front.txt 2>&1; tail -2 gpurun_out/r6k_front.txt
for variant in "new" "old"; do
  if [ $variant = old ]; then export SMVS_VIS_XCD=0; fi
  (cd /tmp && TMPDIR=/tmp SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6k_$variant -o run -- python $ROOT/tools/optimize_timeline.py run > $ROOT/gpurun_out/r6k_${variant}_run.txt 2>&1)
  trace=$(find gpurun_out/r6k_$variant -name "*kernel_trace.csv" | head -1)
  python tools/optimize_timeline.py report $trace > gpurun_out/r6k_timeline_nosgm_$variant.txt 2>&1
  rm -rf gpurun_out/r6k_$variant
  echo "== $variant"; head -1 gpurun_out/r6k_timeline_nosgm_$variant.txt; grep "topo_visibility\|topo_dilate\|gradients" gpurun_out/r6k_timeline_nosgm_$variant.txt
done
PMC_SETS=traffic timeout 600 bash tools/kernel_pmc.sh vis "topo_visibility_kernel,blur_y_kernel" python $ROOT/tools/optimize_timeline.py run > gpurun_out/r6k_vis_pmc.log 2>&1; grep "matching\|HBM traffic" gpurun_out/r6k_vis_pmc.log
