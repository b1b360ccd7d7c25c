#!/bin/bash
# Energy ablation of one conv layer (default: 64 -> 32 @128^3, batch 8): for every BOA_WS_DBG stage mask the launch is repeated for a few
# seconds while rocm-smi samples socket power and shader clock; prints time per launch, mean power and clock of the busy samples.
#   tools/energy_ablate.sh [Di,Cin,Cout] [repeats]
MATCH=${1:-128,64,32}; REP=${2:-1500}
for d in ${ABL:-0 2 4 8 16 32 64 128 160 176}; do
  OUT=gpurun_out/energy_$d.txt
  BOA_WS_DBG=$d BOA_LAYER_PROF_REPEAT=$REP BOA_LAYER_PROF_MATCH=$MATCH tools/power_sample.sh $OUT timeout 300 python tools/layer_prof.py 8 2>&1 | grep "\[repeat\]" | tail -1 | sed "s/^/dbg=$d /"
  python3 tools/power_parse.py $OUT
done
