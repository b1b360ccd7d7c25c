export TMPDIR=/tmp
ROOT=$(pwd)
for grp in "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  out=$ROOT/gpurun_out/gh_pmc_raw; rm -rf $out
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out -- python $ROOT/tools/gh_time.py > $out.log 2>&1)
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); names=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'gather_head' not in r["Kernel_Name"]: continue
    key=r["Kernel_Name"].split("(")[0][:46]; c=r["Counter_Name"]
    if c not in names: names.append(c)
    acc[key][c]+=float(r["Counter_Value"]); cnt[(key,c)]+=1
for key,d in acc.items():
    print(key, {c: round(d[c]/cnt[(key,c)]) for c in names})
PY
done
