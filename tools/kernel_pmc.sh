#!/bin/bash
# SQ counter passes (one counter set per pass, --kernel-trace only -- never with
# a sys / hip trace) over any command, summarised for the kernels whose name
# contains a substring.  Usage:
#   tools/kernel_pmc.sh <tag> <kernel substring> <command ...>
# -> gpurun_out/pmc_<tag>/<tag>_counters.txt
TAG=$1; shift
KERNEL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_IFETCH SQ_WAIT_IFETCH" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o run -- "$@" > $OUT/s$i.log 2>&1
done
python - <<PY > $OUT/${TAG}_counters.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob('$OUT/s*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if '$KERNEL' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
waves = m.get('SQ_WAVES', 1.0) or 1.0
print("# kernels matching '$KERNEL', rocprofv3 --pmc (one counter set per pass, --kernel-trace only;")
print("# tools/kernel_pmc.sh) over: $*")
print("# mean over the matching launches of a pass, and per wave")
print("%-36s %14s %12s" % ("counter", "per launch", "per wave"))
for k in sorted(m):
    print("%-36s %14.4g %12.1f" % (k, m[k], m[k] / waves))
g = lambda k: m.get(k, 0.0) / waves
wc = g('SQ_WAVE_CYCLES')
if wc > 0:
    print()
    print("wave lifetime split (quad-cycle counters): issuing %.1f %%, waiting on s_waitcnt / barrier %.1f %%, "
          "issue stalls %.1f %%" % (100 * g('SQ_ACTIVE_INST_ANY') / wc, 100 * g('SQ_WAIT_ANY') / wc,
                                    100 * g('SQ_WAIT_INST_ANY') / wc))
    iv, isa, il, ivm = g('SQ_INSTS_VALU'), g('SQ_INSTS_SALU'), g('SQ_INSTS_LDS'), g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')
    print("instructions per wave: %.0f VALU (FP64: %.0f FMA + %.0f MUL + %.0f ADD + %.0f transcendental), %.0f SALU, "
          "%.0f LDS, %.0f VMEM; wave lifetime %.0f cycles = %.1f cycles per instruction"
          % (iv, g('SQ_INSTS_VALU_FMA_F64'), g('SQ_INSTS_VALU_MUL_F64'), g('SQ_INSTS_VALU_ADD_F64'),
             g('SQ_INSTS_VALU_TRANS_F64'), isa, il, ivm, 4 * wc, 4 * wc / max(iv + isa + il + ivm, 1)))
PY
rm -rf $OUT/s*/  2>/dev/null
cat $OUT/${TAG}_counters.txt
