#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_front.py -q -x -k "sgm" > gpurun_out/r6l_sgm_tests.txt 2>&1; tail -2 gpurun_out/r6l_sgm_tests.txt
for v in 1 0; do
  (cd /tmp && TMPDIR=/tmp SMVS_SGM_XCD=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r6l_sgm$v -o run -- python $ROOT/tools/sgm_bench.py > $ROOT/gpurun_out/r6l_sgm${v}_run.txt 2>&1)
  f=$(find gpurun_out/r6l_sgm$v -name "*kernel_stats.csv" | head -1)
  echo "== SMVS_SGM_XCD=$v"; python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if any(k in r["Name"] for k in ("cost_packed","warp_kernel","all_paths","sum_wta")): print(r["Name"][:50], r["Calls"], r["AverageNs"])
PY
  rm -rf gpurun_out/r6l_sgm$v
done
PMC_SETS=traffic timeout 300 bash tools/kernel_pmc.sh sgmcost "cost_packed_kernel" python $ROOT/tools/sgm_bench.py > gpurun_out/r6l_cost_pmc.log 2>&1; grep "HBM traffic" gpurun_out/r6l_cost_pmc.log
