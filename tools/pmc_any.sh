#!/bin/bash
# one rocprofv3 --pmc pass with arbitrary counters over a tile batch (or `bench`): per kernel and dispatch means.
#   tools/pmc_any.sh "<counters>" [bench]   -> stdout
export TMPDIR=/tmp
ROOT=$(pwd)
CMD="python $ROOT/tools/layer_prof.py 8"
[ "$2" = bench ] && CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-parity --no-h2h --no-lanes --no-exact --no-c3 --no-phantom"
out=$ROOT/gpurun_out/pmc_any_raw; rm -rf $out; mkdir -p $ROOT/gpurun_out
(cd /tmp && timeout 900 rocprofv3 --pmc $1 --output-format csv -d $out -- $CMD > $out.log 2>&1)
f=$(find $out -name "*counter_collection.csv" | head -1)
[ -z "$f" ] && { tail -5 $out.log; exit 1; }
python - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); names=[]
for r in csv.DictReader(open(sys.argv[1])):
    key=(r["Kernel_Name"].split("(")[0][:46], r["Grid_Size"]); c=r["Counter_Name"]
    if c not in names: names.append(c)
    acc[key][c]+=float(r["Counter_Value"]); cnt[(key,c)]+=1
print("# kernel, grid, dispatches, " + ", ".join(names))
for key,d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    n=cnt[(key,names[0])]
    print(f"{key[0]:48s} {key[1]:>8s} x{n:<4d} " + " ".join(f"{d[c]/max(cnt[(key,c)],1):14.0f}" for c in names))
PY
