#!/bin/bash
# A/B of two builds of the library on the BENCH step (one box, interleaved): tools/ab_bench.sh <alt name> [rounds]
ALT=$1; ROUNDS=${2:-2}
PKG=$(cd "$(dirname "$0")/../body-and-organ-analysis_amd" && pwd)
F="--no-exact --no-c3 --no-lanes --no-phantom --no-h2h --no-cpu --no-parity --steps 2 --warmup 1"
for r in $(seq $ROUNDS); do
  for which in base alt; do
    if [ $which = alt ]; then export BOA_HIP_LIB=$PKG/boa_hip/libboa_hip_$ALT.so; else unset BOA_HIP_LIB; fi
    python $PKG/../bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', 'vol/s %.4f  ms/step %.1f  conv TFLOP/s %.1f  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']))"
  done
done
