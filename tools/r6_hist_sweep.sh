#!/bin/bash
# round 6: k_label_hist workgroup size / table size sweep (kernel times from rocprofv3 traces of tools/agg_time.py)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_aggregation.py tests/test_gpu_fullsize.py tests/test_gpu_agg_shard.py -x -q -m gpu 2>&1 | tail -2
for cfg in "13 256 8" "13 1024 2" "12 1024 2" "14 1024 1" "13 512 2" "12 512 4" "13 1024 1" "12 1024 1"; do set -- $cfg
  out=$ROOT/gpurun_out/hist7_$1_$2_$3; rm -rf $out
  (cd /tmp && BOA_HIST_LOG2=$1 BOA_HIST_THREADS=$2 BOA_HIST_WG=$3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $ROOT/tools/agg_time.py > $out.log 2>&1)
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python - "$f" "$1 $2 $3" <<'PY'
import csv,sys
ts=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'k_label_hist' in r['Kernel_Name']]
print("LOG2 THREADS WG/CU", sys.argv[2], "phantom us", round(sorted(ts[:6])[3]), "noise us", round(sorted(ts[6:12])[3]), "n", len(ts))
PY
done
