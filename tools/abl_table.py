#!/usr/bin/env python3
"""Tabulate tools/ablate_*.sh logs: one row per layer, one column per BOA_WS_DBG value (us)."""
import collections
import sys
d = collections.OrderedDict()
cur = None
for l in open(sys.argv[1]):
    if l.startswith('==='):
        cur = l.split('=')[-1].strip()
        continue
    p = l.split()
    if len(p) < 7:
        continue
    d.setdefault(' '.join(p[1:5]), []).append((cur, p[5]))
for k, v in d.items():
    print(k, ' '.join(f"{c}:{t}" for c, t in v))
