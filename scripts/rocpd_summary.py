#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max (µs)
plus VGPR/SGPR/LDS of each kernel.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total µs | avg µs | min µs | max µs | % | vgpr | agpr | sgpr | lds | scratch | grid | wg |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (
            r[0][:110], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
