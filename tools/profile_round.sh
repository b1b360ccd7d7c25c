#!/bin/bash
# Round profile: rocprofv3 kernel stats of the default bench command + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs,
# as MI355X_MICROARCH.md prescribes).  Run on the GPU box from the repo root; results land in gpurun_out/prof_<tag>/ and
# the summaries to commit in gpurun_out/profiles_<tag>/ (copy them into profiles/).
TAG=${1:-r01}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT $SUM; mkdir -p $OUT $SUM
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-bca"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $CMD > $SUM/${TAG}_bench512_rocprof_run.log 2>&1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $SUM/${TAG}_bench512_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o $TAG -- python $ROOT/bench.py --size 256 --steps 1 --warmup 0 --no-cpu --no-bca > $OUT/pmc_$c.log 2>&1)
done
python $ROOT/tools/pmc_summary.py $OUT $SUM/${TAG}_pmc_fetch_write_256.json
ls -la $SUM
