for d in 0 2 16 128 144 32 64 8 4 ; do
  echo "=== BOA_WS_DBG=$d"
  BOA_WS_DBG=$d timeout 120 python tools/layer_prof.py 8 2>&1 | awk '/--- pass 1/{p=1} p' | grep -E "s=222|in=8x8x8|in=4x4x4 cin=320 cout=320 k=333|in=16x16x16 cin=256 cout=256" | awk '{print $3,$4,$5,$6,$8,$(NF-3),$(NF-2),$(NF-1),$NF}'
done
