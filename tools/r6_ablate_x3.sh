#!/bin/bash
# round 6: role ablation of k_conv_ws<X3> (traced build: BOA_WS_DBG bits; results are wrong by design except d=0)
cd $GRAFT_REPO_ROOT
export LAYER_PROF_PRECISION=fp32 BOA_HIP_LIB=$PWD/body-and-organ-analysis_amd/boa_hip/libboa_hip_trace.so
for d in 0 2 4096 4 8 32 64 128 16; do
  echo "=== BOA_WS_DBG=$d"
  BOA_WS_DBG=$d timeout 120 python tools/layer_prof.py 8 2>&1 | awk '/--- pass 1/{p=1} p' | grep -E "x3 conv .*(in=128x128x128 cin=32 cout=32|in=128x128x128 cin=64 cout=32|in=64x64x64 cin=64 cout=64|in=32x32x32 cin=256 cout=128)" | head -6
done
