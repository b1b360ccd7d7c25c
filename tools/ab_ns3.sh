#!/bin/bash
# three-way A/B of the stride-2 layers on one box: default library, two alternates, interleaved.  tools/ab_ns3.sh altA altB [batch] [rounds]
A=$1; Bn=$2; B=${3:-8}; R=${4:-4}
PKG=$(cd "$(dirname "$0")/../body-and-organ-analysis_amd" && pwd)
for r in $(seq $R); do
  for lib in "" $A $Bn; do
    if [ -z "$lib" ]; then unset BOA_HIP_LIB; tag=base; else export BOA_HIP_LIB=$PKG/boa_hip/libboa_hip_$lib.so; tag=$lib; fi
    python $PKG/../tools/layer_prof.py $B 2>&1 | awk '/--- pass 1/{f=1} f&&/var=2/' | awk -v t=$tag '{print t, $4, $5, $6, $(NF-3)}'
  done
done | sort | awk '{k=$1" "$2" "$3" "$4; s[k]+=$5; n[k]++} END{for(k in s) printf "%-60s %8.1f us (n=%d)\n", k, s[k]/n[k], n[k]}' | sort -k2,4 -k1,1
