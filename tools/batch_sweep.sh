#!/bin/bash
# per-TILE time of the full-resolution conv layers at several tile batches (does the previous layer's output stay in the 256 MB
# Infinity Cache at small batches?  3.4 GB at batch 25, 134 MB per sample).   tools/batch_sweep.sh "1 2 4 8 16"
for B in ${1:-1 2 4 8 16}; do
  python tools/layer_prof.py $B 2>&1 | awk '/--- pass 1/{f=1} f&&/\[layer\]/' | grep "in=128x128x128\|in=64x64x64 cin=64 cout=64\|in=64x64x64 cin=128" | \
    awk -v B=$B '{printf "batch %2d %-8s %s %s %s  %8.1f us per tile\n", B, $2, $4, $5, $6, $(NF-3)/B}'
done
