#!/bin/bash
# Per-kernel HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) and times of one whole reference view
# with SGM (tools/pipeline_profile.py) -> gpurun_out/r6_pipeline_traffic.txt.  Runs on the GPU box.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROOT=$(pwd)
mkdir -p gpurun_out/pipe_traffic
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pipe_traffic/$c -o run -- python $ROOT/tools/pipeline_profile.py > $ROOT/gpurun_out/pipe_traffic/$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pipe_traffic/stats -o run -- python $ROOT/tools/pipeline_profile.py > $ROOT/gpurun_out/pipe_traffic/stats.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections
def counter(name):
    per = collections.defaultdict(list)
    for fn in glob.glob("gpurun_out/pipe_traffic/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == name:
                per[r["Kernel_Name"].split("(")[0].replace("void ","").replace("smvs_hip::","")[:44]].append(float(r["Counter_Value"]))
    return per
f = counter("FETCH_SIZE"); w = counter("WRITE_SIZE")
st = {}
for fn in glob.glob("gpurun_out/pipe_traffic/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        st[r["Name"].split("(")[0].replace("void ","").replace("smvs_hip::","")[:44]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
rows = []
for k in sorted(set(f) | set(w)):
    rd = 2 * 1024 * sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1)
    wr = 1024 * sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    calls, us = st.get(k, (0, 0.0))
    rows.append((calls * us, k, calls, us, rd / 1e6, wr / 1e6, (rd + wr) / 1e6 / max(us, 1e-9) if us else 0))
rows.sort(reverse=True)
with open("gpurun_out/r6_pipeline_traffic.txt", "w") as out:
    out.write("# whole reference view with SGM (tools/pipeline_profile.py), per kernel: calls, avg us, HBM MB read / written per launch\n")
    out.write("# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024), TB/s = MB / us\n")
    out.write("%-46s %5s %9s %10s %10s %7s\n" % ("kernel", "calls", "avg us", "read MB", "write MB", "TB/s"))
    for tot, k, calls, us, rd, wr, rate in rows:
        out.write("%-46s %5d %9.1f %10.1f %10.1f %7.2f\n" % (k, calls, us, rd, wr, rate))
print(open("gpurun_out/r6_pipeline_traffic.txt").read())
PY
rm -rf gpurun_out/pipe_traffic/FETCH_SIZE gpurun_out/pipe_traffic/WRITE_SIZE gpurun_out/pipe_traffic/stats
