for d in ${ABL:-0 2 8 10 4}; do
  echo "=== BOA_WS_DBG=$d"
  BOA_WS_DBG=$d timeout 120 python tools/layer_prof.py ${NB:-8} 2>&1 | awk '/--- pass 1/{p=1} p' | grep "var=2" | awk '{print $3,$4,$5,$6,$8,$(NF-3),$(NF-2),$(NF-1),$NF}'
done
