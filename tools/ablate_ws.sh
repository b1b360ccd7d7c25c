for d in ${ABL:-0 1 2 3 4 8 16 32 64 66 9}; do
  echo "=== BOA_WS_DBG=$d"
  BOA_WS_DBG=$d timeout 120 python tools/layer_prof.py 8 2>&1 | awk '/--- pass 1/{p=1} p' | grep -E "in=128x128x128 cin=32 cout=32|in=128x128x128 cin=32 cout=64|in=128x128x128 cin=64 cout=32|in=64x64x64 cin=64 cout=64|in=64x64x64 cin=64 cout=128" | head -6
done
