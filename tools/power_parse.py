#!/usr/bin/env python3
"""Mean socket power / shader clock of the busy samples (power > 700 W) of a tools/power_sample.sh trace."""
import re
import sys

busy = []
for line in open(sys.argv[1]):
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", line)
    p = re.search(r"Current Socket Graphics Package Power \(W\): ([\d.]+)", line)
    if c and p and float(p.group(1)) > float(sys.argv[2] if len(sys.argv) > 2 else 700):
        busy.append((float(p.group(1)), float(c.group(1))))
busy = busy[1:-1] if len(busy) > 4 else busy      # (the first / last busy sample straddle the start / end of the load)
if busy:
    print(f"   busy samples {len(busy)}: {sum(b[0] for b in busy) / len(busy):.0f} W, {sum(b[1] for b in busy) / len(busy):.0f} MHz")
else:
    print("   no busy samples")
