set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "scale_space or topology or sgm or config1 or 960x540 or view_queue or reconstruct" > gpurun_out/r3_pytest_l.txt 2>&1
tail -3 gpurun_out/r3_pytest_l.txt
SMVS_HOST_TIMING=1 timeout 300 python tools/view_throughput.py --sgm > gpurun_out/r3_view_tp_sgm_l.log 2>&1
grep "in flight" gpurun_out/r3_view_tp_sgm_l.log
timeout 600 bash tools/patch_pmc.sh r3 2>&1 | tail -14
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_l -o run -- python $GRAFT_REPO_ROOT/tools/pipeline_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r3_pipeline_l.log 2>&1
grep "sgm front\|cut_depth" $GRAFT_REPO_ROOT/gpurun_out/r3_pipeline_l.log
