#!/bin/bash
# Samples socket power / shader clock (rocm-smi) every 0.5 s while a command runs: is a kernel mix at the power cap?
#   tools/power_sample.sh out.txt python bench.py --steps 3 ...
OUT=$1; shift
( while true; do /opt/rocm/bin/rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|Max" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT &
SP=$!
"$@"
kill $SP
