#!/bin/bash
# SQ / TCC counter passes (one counter set per pass, --kernel-trace only -- never
# with a sys / hip trace) over any command, summarised per kernel whose name
# contains one of the given substrings.  Usage:
#   tools/kernel_pmc.sh <tag> <kernel substring[,substring...]> <command ...>
# -> gpurun_out/pmc_<tag>/<tag>_counters.txt
TAG=$1; shift
KERNELS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC_SETS=traffic: only the two HBM passes
if [ "${PMC_SETS:-all}" = "traffic" ]; then
  SETS=("FETCH_SIZE" "WRITE_SIZE")
else
  SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_WR"
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_IFETCH SQ_WAIT_IFETCH"
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE"
           "FETCH_SIZE" "WRITE_SIZE")
fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o run -- "$@" > $OUT/s$i.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT "$KERNELS" "$*" > $OUT/${TAG}_counters.txt
rm -rf $OUT/s*/  2>/dev/null
cat $OUT/${TAG}_counters.txt
