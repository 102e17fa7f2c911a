#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_front.py -q -x -k "debug_lvl" > gpurun_out/r6n_debug.txt 2>&1; tail -15 gpurun_out/r6n_debug.txt
