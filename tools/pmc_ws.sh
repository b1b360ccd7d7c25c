#!/bin/bash
# SQ counters of k_conv_ws on one tile batch (layer_prof), one rocprofv3 --pmc pass per counter group.
# usage: bash tools/pmc_ws.sh [BOA_WS_DBG value]   (results: gpurun_out/pmc_ws_<dbg>_<group>.csv summaries)
export TMPDIR=/tmp
DBG=${1:-0}
ROOT=$(pwd)
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  out=$ROOT/gpurun_out/pmc_ws_${DBG}_$i
  rm -rf $out
  (cd /tmp && BOA_WS_DBG=$DBG timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out -- python $ROOT/tools/layer_prof.py 8 > $out.log 2>&1)
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
f=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_conv_ws<4" not in k: continue
    key=(k[:24], r["Grid_Size"], r.get("LDS_Block_Size",""))
    acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    cnt[(key,r["Counter_Name"])]+=1
for key,d in acc.items():
    print(key, {c: round(v/cnt[(key,c)]) for c,v in d.items()})
PY
done
