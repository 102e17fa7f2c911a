#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
(python tools/upload_overlap.py; SMVS_UPLOAD_STREAM=same python tools/upload_overlap.py) > gpurun_out/r6d_upload_overlap.txt 2>&1; cat gpurun_out/r6d_upload_overlap.txt
timeout 1500 python -m pytest tests/test_gpu_front.py -q -x -k "operating_point or scale_3_portrait or scale_1_portrait or gamma" > gpurun_out/r6d_front.txt 2>&1; tail -5 gpurun_out/r6d_front.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "async" > gpurun_out/r6d_async.txt 2>&1; tail -2 gpurun_out/r6d_async.txt
for mode in "--sgm"; do
  tagname=tl${mode#--}
  (cd /tmp && TMPDIR=/tmp SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/gpurun_out/r6d_$tagname -o run -- python $ROOT/tools/optimize_timeline.py run $mode > $ROOT/gpurun_out/r6d_${tagname}_run.txt 2>&1)
  trace=$(find gpurun_out/r6d_$tagname -name "*kernel_trace.csv" | head -1)
  { python tools/optimize_timeline.py report $trace; grep -a "optimize 1\|smvs host" gpurun_out/r6d_${tagname}_run.txt | tail -14; } > gpurun_out/r6d_optimize_timeline_${tagname}.txt 2>&1
  cp $(find gpurun_out/r6d_$tagname -name "*memory_copy_trace.csv" | head -1) gpurun_out/r6d_memcopy.csv 2>/dev/null
  cp $trace gpurun_out/r6d_kernel_trace.csv
  rm -rf gpurun_out/r6d_$tagname
  head -3 gpurun_out/r6d_optimize_timeline_${tagname}.txt
done
python - <<PY
import csv
k=list(csv.DictReader(open("gpurun_out/r6d_kernel_trace.csv")))
m=list(csv.DictReader(open("gpurun_out/r6d_memcopy.csv")))
print(len(k), len(m), m[0].keys() if m else None)
# the last optimize: events after the last big gap; print the copies and the first kernels around the last 'byte_to_float'
b2f=[r for r in k if "byte_to_float" in r["Kernel_Name"]]
t0=int(b2f[-9]["Start_Timestamp"])-3_000_000
for r in m:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if s>t0 and s<t0+8_000_000 and e-s>20000: print("copy", r.get("Direction"), (s-t0)/1e3, (e-s)/1e3)
for r in k:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if s>t0 and s<t0+6_000_000: print("kern", r["Kernel_Name"][:40], (s-t0)/1e3, (e-s)/1e3)
PY
