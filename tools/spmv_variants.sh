#!/bin/bash
# experiment helper: SpMV variants, time (bench HIP events) + FETCH_SIZE
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/spmv_var
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 4; do
  export SMVS_SPMV_VARIANT=$v
  python $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['kernels']['cg_spmv'])"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$v -o run -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/f$v.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob('$OUT/f$v/**/*counter_collection.csv', recursive=True)
tot=0;n=0
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'cg_spmv' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
            val=float(r['Counter_Value'])
            if val>1000: tot+=val;n+=1
print('  fetch MB/launch (x2 corrected):', 2*tot*1024/max(n,1)/1e6, 'launches', n)
PY
done
