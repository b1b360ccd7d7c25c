#!/bin/bash
# round 6: k_label_hist batched lookup (base) vs the round-5 serial form (libboa_hip_hser.so), kernel times from rocprofv3 traces
# of tools/agg_time.py (6 calls on the structured phantom, then 6 on noise labels), table sizes 2^12 / 2^13 / 2^14
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_aggregation.py tests/test_gpu_fullsize.py -x -q -m gpu -k "hist or measure or aggreg or total" 2>&1 | tail -2
for lib in base hser; do
 for cfg in "14 4" "13 8" "12 8"; do set -- $cfg
  out=$ROOT/gpurun_out/hist6_${lib}_$1_$2; rm -rf $out
  if [ $lib = hser ]; then export BOA_HIP_LIB=$ROOT/body-and-organ-analysis_amd/boa_hip/libboa_hip_hser.so; else unset BOA_HIP_LIB; fi
  (cd /tmp && BOA_HIST_LOG2=$1 BOA_HIST_WG=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $ROOT/tools/agg_time.py > $out.log 2>&1)
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python - "$f" "$lib $1 $2" <<'PY'
import csv,sys
ts=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'k_label_hist' in r['Kernel_Name']]
print("lib LOG2 WG", sys.argv[2], "phantom us", round(sorted(ts[:6])[3]), "noise us", round(sorted(ts[6:12])[3]), "n", len(ts))
PY
 done
done
