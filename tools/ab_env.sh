#!/bin/bash
# A/B of two ENVIRONMENTS of the same library on one box (per-layer timings of tools/layer_prof.py, second pass, interleaved):
#   tools/ab_env.sh "BOA_WS_DBG=2048" [batch] [rounds]
ENVV=$1; B=${2:-8}; ROUNDS=${3:-3}
PKG=$(cd "$(dirname "$0")/../body-and-organ-analysis_amd" && pwd)
OUT=$PKG/../gpurun_out; mkdir -p $OUT
rm -f $OUT/ab_base.log $OUT/ab_alt.log
for r in $(seq $ROUNDS); do
  python $PKG/../tools/layer_prof.py $B 2>&1 | awk '/--- pass 1/{f=1} f&&/\[layer\]/' >> $OUT/ab_base.log
  env $ENVV python $PKG/../tools/layer_prof.py $B 2>&1 | awk '/--- pass 1/{f=1} f&&/\[layer\]/' >> $OUT/ab_alt.log
done
python - <<PY
import re, statistics as st
def load(fn):
    d={}; order=[]
    for l in open(fn):
        m=re.search(r'\[layer\] (?:x3 )?(\S+)\s+N=\d+ in=(\S+) cin=(\d+) cout=(\d+) k=(\d+) s=(\d+).*? ([\d.]+) us', l)
        if not m: continue
        key=m.group(1,2,3,4,6)
        d.setdefault(key,[]).append(float(m.group(7)))
        if key not in order: order.append(key)
    return d,order
b,order=load("$OUT/ab_base.log"); a,_=load("$OUT/ab_alt.log")
tb=ta=0
print(f"# A/B default vs $ENVV, batch $B, $ROUNDS rounds")
for k in order:
    mb=st.median(b[k]); ma=st.median(a.get(k,[float('nan')]))
    n=len(b[k])//$ROUNDS
    tb+=mb*n; ta+=ma*n
    print(f"{k[0]:6s} {k[1]:12s} {k[2]:>4s}->{k[3]:<4s} s{k[4]} x{n}: base {mb:8.1f} us   alt {ma:8.1f} us   {100*(ma/mb-1):+6.1f} %")
print(f"sum over one forward: base {tb:.0f} us, alt {ta:.0f} us, {100*(ta/tb-1):+.2f} %")
PY
