#!/bin/bash
# round 6, third GPU visit: fixed reactivate staging, async uploads, operating-point parity, timeline
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "loop or async or scale_space or newton" > gpurun_out/r6c_parity.txt 2>&1; tail -3 gpurun_out/r6c_parity.txt
timeout 1500 python -m pytest tests/test_gpu_front.py -q -x -k "operating_point or scale_3_portrait or scale_1_portrait or gamma or full_size_optimize_with_sgm" > gpurun_out/r6c_front.txt 2>&1; tail -5 gpurun_out/r6c_front.txt
timeout 900 python bench.py --no-cpu-baseline --no-peaks > gpurun_out/r6c_bench.json 2> gpurun_out/r6c_bench.err; tail -2 gpurun_out/r6c_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r6c_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "value_optimize", d.get("value_optimize")); print(d["roofline"]["kernels"])
for s,v in d["roofline"]["by_scale"].items(): print(s, v["kernel_ms"])
print(d["secondary"]["views_per_s"]["per_gpu"])
PY
for mode in "--sgm"; do
  tagname=tl${mode#--}
  (cd /tmp && TMPDIR=/tmp SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/gpurun_out/r6c_$tagname -o run -- python $ROOT/tools/optimize_timeline.py run $mode > $ROOT/gpurun_out/r6c_${tagname}_run.txt 2>&1)
  trace=$(find gpurun_out/r6c_$tagname -name "*kernel_trace.csv" | head -1)
  { python tools/optimize_timeline.py report $trace; grep -a "optimize 1\|smvs host" gpurun_out/r6c_${tagname}_run.txt | tail -14; } > gpurun_out/r6c_optimize_timeline_${tagname}.txt 2>&1
  rm -rf gpurun_out/r6c_$tagname
  head -3 gpurun_out/r6c_optimize_timeline_${tagname}.txt
done
