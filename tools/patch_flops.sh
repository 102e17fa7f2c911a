#!/bin/bash
# FP64 flops the patch kernel executes per active patch, per kernel form (= per
# samples-per-patch): SQ instruction counters of ONE optimize() of the bench
# scene, summed over all launches of a form and divided by the active
# patch-steps of the scales that form serves (the batch log).  Runs on the GPU
# box; -> gpurun_out/patch_flops/patch_flops_r6.json (+ .txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/patch_flops
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o run -- \
      python $ROOT/tools/patch_flops.py run $OUT/log.json > $OUT/s$i.log 2>&1
done
python $ROOT/tools/patch_flops.py report $OUT > $OUT/patch_flops_r6.txt
cat $OUT/patch_flops_r6.txt
rm -rf $OUT/s*/
