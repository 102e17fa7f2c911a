#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6h_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r6h_pytest_gpu.txt
for ahead in 4 1; do
for mode in "--sgm" ""; do
  tagname=tl${mode#--}_a$ahead
  (cd /tmp && TMPDIR=/tmp SMVS_TOPO_AHEAD=$ahead SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6h_$tagname -o run -- python $ROOT/tools/optimize_timeline.py run $mode > $ROOT/gpurun_out/r6h_${tagname}_run.txt 2>&1)
  trace=$(find gpurun_out/r6h_$tagname -name "*kernel_trace.csv" | head -1)
  { python tools/optimize_timeline.py report $trace; grep -a "optimize 1\|smvs host" gpurun_out/r6h_${tagname}_run.txt | tail -14; } > gpurun_out/r6h_optimize_timeline_${tagname}.txt 2>&1
  rm -rf gpurun_out/r6h_$tagname
  echo "ahead=$ahead mode=$mode"; head -1 gpurun_out/r6h_optimize_timeline_${tagname}.txt; grep "cut_boundaries" gpurun_out/r6h_optimize_timeline_${tagname}.txt
done
done
timeout 900 python bench.py --no-cpu-baseline --no-peaks > gpurun_out/r6h_bench.json 2> gpurun_out/r6h_bench.err; tail -2 gpurun_out/r6h_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r6h_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "value_optimize", d.get("value_optimize")); print(d["roofline"]["kernels"])
print(d["secondary"]["views_per_s"]["per_gpu"])
PY
