#!/bin/bash
# round 6 call 1: tap-paired X3 consumers (YR path): parity tests, smoke, per-layer A/B vs the unpaired form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_exact_mode.py tests/test_gpu_production_geometry.py tests/test_gpu_gather_head.py -x -q -m gpu > gpurun_out/r1_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r1_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r1_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r1_smoke.log
LAYER_PROF_PRECISION=fp32 timeout 900 tools/ab_layers.sh unp 8 2 > gpurun_out/r1_ab_x3.txt 2>&1
tail -5 gpurun_out/r1_tests.log; tail -3 gpurun_out/r1_smoke.log; cat gpurun_out/r1_ab_x3.txt
