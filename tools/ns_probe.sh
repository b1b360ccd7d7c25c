#!/bin/bash
# k_conv_ns probe on one box (traced library: tools/build_alt.sh trace -DWS_WITH_TRACE): role timelines of an interior workgroup and
# the BOA_WS_DBG ablations (2 producers off, 4 stores off, 8 epilogue off, 32 commit off, 64 transform off, 128 halo loads off) per
# stride-2 layer.   tools/ns_probe.sh [batch]  -> gpurun_out/ns_probe.txt
B=${1:-8}
PKG=$(cd "$(dirname "$0")/../body-and-organ-analysis_amd" && pwd)
OUT=$PKG/../gpurun_out; mkdir -p $OUT
export BOA_HIP_LIB=$PKG/boa_hip/libboa_hip_trace.so
{
  echo "== trace, block ${BLK:-100}"
  BOA_WS_TRACE=${BLK:-100} timeout 300 python $PKG/../tools/layer_prof.py $B 2>&1 | awk '/--- pass 1/{f=1} f' | grep "ns-trace"
  for d in ${ABL:-0 128 32 64 2 4 8 0}; do
    echo "== BOA_WS_DBG=$d"
    BOA_WS_DBG=$d timeout 200 python $PKG/../tools/layer_prof.py $B 2>&1 | awk '/--- pass 1/{p=1} p' | grep "var=2" | awk '{print $4,$5,$6,$(NF-3),$(NF-2),$(NF-1),$NF}'
  done
} > $OUT/ns_probe.txt 2>&1
