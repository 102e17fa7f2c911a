#!/bin/bash
# Runs on the GPU box (via gpurun): everything the round's profiles/ hold.
# Usage: final_round.sh [round]
R=${1:-r6}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${R}_pytest_gpu.txt 2>&1
tail -3 gpurun_out/${R}_pytest_gpu.txt
# the counter passes first: the bench line below reads the traffic they measure
timeout 900 bash tools/collect_profiles.sh $R > gpurun_out/${R}_collect.log 2>&1
tail -3 gpurun_out/${R}_collect.log
cp gpurun_out/prof_${R}/summary/traffic_${R}.json profiles/ 2>/dev/null
# the batch logs of the whole-optimize() parity tests, device / oracle, one table
cat gpurun_out/units_*.txt > gpurun_out/${R}_parity_units.txt 2>/dev/null
timeout 900 python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python - <<PY
import json
d = json.load(open("gpurun_out/${R}_bench_default.json"))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print({k: v for k, v in d["secondary"]["views_per_s"]["per_gpu"].items()} if "views_per_s" in d.get("secondary", {}) else d.get("secondary"))
PY
timeout 300 python bench.py --workload optimize > gpurun_out/${R}_bench_optimize.json 2>> gpurun_out/${R}_bench_default.err
cat gpurun_out/${R}_bench_optimize.json | cut -c1-300
(timeout 600 python tools/fuzz_parity.py 150 0; timeout 900 python tools/fuzz_parity.py 40 1 1) > gpurun_out/${R}_fuzz_parity.txt 2>&1
grep "^cases\|^worst\|Error\|assert" gpurun_out/${R}_fuzz_parity.txt | tail -8
# the resident PCG without stamps: slope = one iteration, intercept = the prologue
(timeout 300 python tools/cg_iteration_cost.py; timeout 300 python tools/cg_iteration_cost.py 3 | tail -1) > gpurun_out/${R}_cg_iteration_cost_now.txt 2>&1
tail -2 gpurun_out/${R}_cg_iteration_cost_now.txt
# kernel timeline of one warm optimize(), without and with the SGM initialisation
ROOT=$(pwd)
for mode in "" "--sgm"; do
  tagname=tl${mode#--}
  (cd /tmp && TMPDIR=/tmp SMVS_HOST_TIMING=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_$tagname -o run -- python $ROOT/tools/optimize_timeline.py run $mode > $ROOT/gpurun_out/${R}_${tagname}_run.txt 2>&1)
  trace=$(find gpurun_out/${R}_$tagname -name "*kernel_trace.csv" | head -1)
  { python tools/optimize_timeline.py report $trace; grep -a "optimize 1\|smvs host" gpurun_out/${R}_${tagname}_run.txt | tail -14; } > gpurun_out/${R}_optimize_timeline_${tagname}.txt 2>&1
  rm -rf gpurun_out/${R}_$tagname
  head -3 gpurun_out/${R}_optimize_timeline_${tagname}.txt
done
(timeout 300 python tools/upload_overlap.py) > gpurun_out/${R}_upload_overlap.txt 2>&1; tail -2 gpurun_out/${R}_upload_overlap.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
