#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel statistics and HBM
# traffic counters of the bench command.  Counters are collected in their own
# passes (--pmc with --kernel-trace only), as MI355X_MICROARCH.md prescribes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-r6}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-peaks --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- $CMD > "$OUT/write.log" 2>&1
ls -R "$OUT" | head -30
# per-step timeline of the Newton loop (kernel durations and the gaps between them)
python $ROOT/tools/step_timeline.py $(find "$OUT/stats" -name "*kernel_trace.csv" | head -1) > "$OUT/step_timeline.txt" 2>&1
# where the resident solver spends a solve (cycle stamps; the host waits for each solve)
python $ROOT/tools/cg_trace.py "$OUT/cg_trace_raw.txt" > "$OUT/cg_trace.txt" 2>&1
python $ROOT/tools/summarise_profiles.py "$OUT" "$OUT/summary" "${1:-r6}"
# the whole pipeline of one reference view (SGM front end, bilateral upsample,
# scale space, topology kernels, Newton loops) and the depth-map cut
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pipeline" -o run -- python $ROOT/tools/pipeline_profile.py > "$OUT/pipeline.log" 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/pipeline/**/*kernel_stats.csv", recursive=True)[0])))
with open("$OUT/summary/${1:-r6}_pipeline_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
tail -5 "$OUT/pipeline.log"
