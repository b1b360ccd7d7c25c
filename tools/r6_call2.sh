#!/bin/bash
# round 6 call 2: last-group dy pairing in the X3 row-reuse loop: parity tests + per-layer A/B vs the unpaired (round-4) form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_exact_mode.py tests/test_gpu_production_geometry.py tests/test_gpu_gather_head.py -x -q -m gpu > gpurun_out/r2_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r2_tests.log
LAYER_PROF_PRECISION=fp32 timeout 900 tools/ab_layers.sh unp 8 2 > gpurun_out/r2_ab_x3.txt 2>&1
tail -3 gpurun_out/r2_tests.log; cat gpurun_out/r2_ab_x3.txt
