#!/bin/bash
# PMC counter passes over the construct-only helper (patch kernel analysis):
# one counter set per pass, --kernel-trace only.  Usage: patch_pmc.sh [round]
ROUND=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/patch_pmc_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_IFETCH SQ_WAIT_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o run -- python $ROOT/tools/construct_only.py 3 > $OUT/s$i.log 2>&1
done
python - <<PY > $OUT/${ROUND}_patch_kernel_counters.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob('$OUT/s*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'gn_patch_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
waves = m.get('SQ_WAVES', 1.0)
print("# gn_patch_kernel<4>, rocprofv3 --pmc (one counter set per pass, --kernel-trace only; tools/patch_pmc.sh) over")
print("# tools/construct_only.py: all patches of the bench workload active (1920x1080, 8 neighbours, scale 2).")
print("# Per launch (mean over the launches of a pass) and per wave of 4 patches.")
print("%-36s %14s %12s" % ("counter", "per launch", "per wave"))
for k in sorted(m):
    print("%-36s %14.4g %12.1f" % (k, m[k], m[k] / waves))
g = lambda k: m.get(k, 0.0) / waves
wc = g('SQ_WAVE_CYCLES')
if wc > 0:
    print()
    print("wave lifetime split (quad-cycle counters): issuing %.1f %%, waiting on s_waitcnt / barrier %.1f %%, issue stalls %.1f %%"
          % (100 * g('SQ_ACTIVE_INST_ANY') / wc, 100 * g('SQ_WAIT_ANY') / wc, 100 * g('SQ_WAIT_INST_ANY') / wc))
fma, mul, add, tr, mf = (g('SQ_INSTS_VALU_FMA_F64'), g('SQ_INSTS_VALU_MUL_F64'), g('SQ_INSTS_VALU_ADD_F64'),
                         g('SQ_INSTS_VALU_TRANS_F64'), g('SQ_INSTS_VALU_MFMA_MOPS_F64'))
print("FP64 per wave: %.0f FMA + %.0f MUL + %.0f ADD + %.0f transcendental VALU instructions, %.1f v_mfma_f64_4x4x4" % (fma, mul, add, tr, mf))
print("executed FP64 flops per patch = ((2 FMA + MUL + ADD) * 64 + MFMA * 512) / 4 = %.4g" % (((2 * fma + mul + add) * 64 + mf * 512) / 4))
pipe = (fma + mul + add) * 4.67 + tr * 16.6 + mf * 17.1
print("FP64 pipe cycles per wave at the measured rates (tools/peaks.py: 4.67 cycles per vector FP64 instruction,")
print("16.6 per v_rcp_f64, 17.1 per v_mfma_f64_4x4x4; vector and matrix FP64 share one pipe): %.0f" % pipe)
if wc > 0:
    print("two waves per SIMD live %.0f cycles (SQ_WAVE_CYCLES * 4) and need %.0f of the pipe: %.0f %% busy"
          % (4 * wc, 2 * pipe, 100 * 2 * pipe / (4 * wc)))
PY
cat $OUT/${ROUND}_patch_kernel_counters.txt | tail -12
