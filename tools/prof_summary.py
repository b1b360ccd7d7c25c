#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) per (kernel, grid, LDS) = per layer shape."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, count(*), avg(end-start)/1e3, sum(end-start)/1e6, "
        "max(vgpr_count), max(accum_vgpr_count), max(scratch_size) "
        "from kernels group by name, grid_x, grid_y, grid_z, lds_size order by sum(end-start) desc").fetchall()
    tot = sum(r[8] for r in rows)
    print(f"total kernel time {tot:.2f} ms")
    for r in rows[:top]:
        blocks = r[1] // max(r[4], 1)
        print(f"{r[0][:30]:30s} blocks=({blocks},{r[2]},{r[3]}) wg={r[4]} lds={r[5]:6d} vgpr={r[9]}+{r[10]} scr={r[11]} "
              f"n={r[6]:4d} avg={r[7]:8.1f}us sum={r[8]:8.2f}ms ({100 * r[8] / tot:4.1f}%)")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
