mkdir -p gpurun_out/r05
for d in 0 4096 4104 4136 4168 4232 4264 6; do
  echo "== BOA_WS_DBG=$d"
  BOA_WS_DBG=$d timeout 200 python tools/layer_prof.py 8 2>&1 | awk '/--- pass 1/{f=1} f&&/\[layer\] conv/' | grep "in=128x128x128 cin=32 cout=32\|in=128x128x128 cin=64 cout=32\|in=64x64x64 cin=128 cout=64\|in=32x32x32 cin=256" | awk '{print $4,$5,$6, $(NF-3),$(NF-2)}'
done
