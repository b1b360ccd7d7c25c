export TMPDIR=/tmp
ROOT=$(pwd)
for cfg in "14 4" "13 8" "12 8" "12 16"; do set -- $cfg
  out=$ROOT/gpurun_out/hist_$1_$2; rm -rf $out
  (cd /tmp && BOA_HIST_LOG2=$1 BOA_HIST_WG=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $ROOT/tools/agg_time.py > $out.log 2>&1)
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python - "$f" "$1 $2" <<'PY'
import csv,sys
ts=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'k_label_hist' in r['Kernel_Name']]
# agg_time: 6 calls on the phantom, then 6 on the noise labels
print("LOG2/WG", sys.argv[2], "phantom us", round(sorted(ts[:6])[3]), "noise us", round(sorted(ts[6:12])[3]), "n", len(ts))
PY
done
