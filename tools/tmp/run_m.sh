cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python tools/fuzz_parity.py 150 0; timeout 900 python tools/fuzz_parity.py 40 1 1) > gpurun_out/r3_fuzz_parity_m.txt 2>&1
grep -n "^cases\|^worst\|Error\|DEVICE\|it False" gpurun_out/r3_fuzz_parity_m.txt | tail -20
timeout 900 python -m pytest tests -m gpu -x -q -k "cg or loop or solver or fuzz or optimize or full_size" > gpurun_out/r3_pytest_m.txt 2>&1
tail -3 gpurun_out/r3_pytest_m.txt
timeout 300 python bench.py --no-cpu-baseline --no-peaks --no-secondary > gpurun_out/r3_bench_m.json 2> gpurun_out/r3_bench_m.err
python -c "
import json
d=json.load(open('gpurun_out/r3_bench_m.json'))
print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['roofline']['kernels'].items() if v['launches']})
"
