#!/usr/bin/env python3
"""Timeline of the LAST n kernel launches of a rocprofv3 (rocpd sqlite) kernel trace: start / end in µs relative to the first of them, the
queue (stream) each ran on -- what shows launches overlapping.  Usage: rocpd_timeline.py results.db [n=24]"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute("select name, start, end, %s from kernels order by start desc limit %d" % (q, n)).fetchall()[::-1]
t0 = rows[0][1]
print("| kernel | queue | start µs | end µs |\n|---|---|---|---|")
for name, s, e, qq in rows:
    print("| `%s` | %s | %.1f | %.1f |" % (name[:60], qq, (s - t0) / 1e3, (e - t0) / 1e3))
