#!/bin/bash
# LDS counters of the conv kernels on one 8-tile batch (tools/layer_prof.py): SQ_LDS_BANK_CONFLICT (extra LDS cycles), SQ_LDS_IDX_ACTIVE (all
# LDS-array cycles), SQ_WAIT_INST_LDS, SQ_ACTIVE_INST_LDS -- one rocprofv3 --pmc pass.   tools/pmc_lds.sh [lib suffix]  -> gpurun_out/pmc_lds[_suffix].txt
export TMPDIR=/tmp
ROOT=$(pwd)
SFX=$1
[ -n "$SFX" ] && [ "$SFX" != bench ] && export BOA_HIP_LIB=$ROOT/body-and-organ-analysis_amd/boa_hip/libboa_hip_$SFX.so
CMD="python $ROOT/tools/layer_prof.py 8"
# `tools/pmc_lds.sh bench`: every kernel of one total+bca volume (the bench's step) instead of one tile batch
[ "$SFX" = bench ] && CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-parity --no-h2h --no-lanes --no-exact --no-c3 --no-phantom"
out=$ROOT/gpurun_out/pmc_lds_raw$SFX
rm -rf $out; mkdir -p $ROOT/gpurun_out
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out -- $CMD > $out.log 2>&1)
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" > $ROOT/gpurun_out/pmc_lds${SFX:+_$SFX}.txt <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    key=(k.split("(")[0][:44], r["Grid_Size"])
    acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(key,r["Counter_Name"])]+=1
print("# per dispatch (mean): kernel, grid, dispatches, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, conflict share, SQ_ACTIVE_INST_LDS, SQ_WAIT_INST_LDS")
for key,d in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"]):
    n=cnt[(key,"SQ_LDS_IDX_ACTIVE")]
    if d["SQ_LDS_IDX_ACTIVE"] < 1e6: continue
    g=lambda c: d[c]/max(cnt[(key,c)],1)
    bc,ia=g("SQ_LDS_BANK_CONFLICT"),g("SQ_LDS_IDX_ACTIVE")
    print(f"{key[0]:46s} {key[1]:>8s} x{n:<3d} {bc:14.0f} {ia:14.0f} {bc/max(ia,1):6.3f} {g('SQ_ACTIVE_INST_LDS'):14.0f} {g('SQ_WAIT_INST_LDS'):14.0f}")
PY
cat $ROOT/gpurun_out/pmc_lds${SFX:+_$SFX}.txt
