#!/bin/bash
# Round profile (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (total+bca, 512^3)      -> <tag>_bench512_kernel_stats.csv
#   2. FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) of the SAME workload's
#      total+bca workload at 512^3 (one volume, 1 605 tile forwards) -> <tag>_pmc_fetch_write_512.json
#   3. matrix-core counters (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) on one 8-tile batch   -> <tag>_pmc_mfma.json
# Results land in gpurun_out/prof_<tag>/, the summaries to commit in gpurun_out/profiles_<tag>/ (copy them into profiles/).
TAG=${1:-r05}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
GIT=${GIT_REV:-unknown}
rm -rf $OUT $SUM; mkdir -p $OUT $SUM
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-parity --no-h2h --no-lanes --no-exact --no-c3 --no-phantom"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $CMD > $SUM/${TAG}_bench512_rocprof_run.log 2>&1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $SUM/${TAG}_bench512_kernel_stats.csv
PMCCMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-parity --no-h2h --no-lanes --no-exact --no-c3 --no-phantom"   # the total+bca workload the bench line reports
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 1200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o $TAG -- $PMCCMD > $OUT/pmc_$c.log 2>&1)
done
python $ROOT/tools/pmc_summary.py $OUT $SUM/${TAG}_pmc_fetch_write_512.json "$PMCCMD" "$GIT"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/pmc_mfma -o $TAG -- python $ROOT/tools/layer_prof.py 8 > $OUT/pmc_mfma.log 2>&1)
python $ROOT/tools/pmc_mfma_summary.py $OUT/pmc_mfma $SUM/${TAG}_pmc_mfma.json "$GIT"
# 4. the split-precision (fp32) mode: per-layer event timings and matrix-core counters of one 8-tile batch
LAYER_PROF_PRECISION=fp32 python $ROOT/tools/layer_prof.py 8 > $SUM/${TAG}_per_layer_fp32_mode.txt 2>&1
python $ROOT/tools/layer_prof.py 8 > $SUM/${TAG}_per_layer.txt 2>&1
(cd /tmp && LAYER_PROF_PRECISION=fp32 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/pmc_mfma_x3 -o $TAG -- python $ROOT/tools/layer_prof.py 8 > $OUT/pmc_mfma_x3.log 2>&1)
python $ROOT/tools/pmc_mfma_summary.py $OUT/pmc_mfma_x3 $SUM/${TAG}_pmc_mfma_fp32_mode.json "$GIT"
ls -la $SUM
