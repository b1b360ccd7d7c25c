export TMPDIR=/tmp
ROOT=$(pwd)
for cfg in "14 4" "13 8" "12 8" "12 16"; do set -- $cfg
  out=$ROOT/gpurun_out/histb_$1_$2; rm -rf $out
  (cd /tmp && BOA_HIST_LOG2=$1 BOA_HIST_WG=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-parity --no-h2h --no-lanes --no-exact --no-c3 --no-phantom > $out.log 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "LOG2/WG $1 $2: $(grep k_label_hist $f | cut -d, -f2-4)"
done
