#!/usr/bin/env python3
"""How much do the kernels of a traced region overlap?  usage: trace_overlap.py results.db  (between the two profile markers):
wall clock, sum of kernel durations, time with >= 1 / >= 2 / >= 3 kernels in flight, dispatches per queue."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marks = db.execute("select K.start, K.end from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid "
                   "where S.display_name like '%profile_marker_kernel%' order by K.start").fetchall()
lo, hi = marks[0][1], marks[1][0]
cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
rows = db.execute("select K.start, K.end%s from rocpd_kernel_dispatch K where K.start > %d and K.end < %d order by K.start" % (", K." + qcol if qcol else "", lo, hi)).fetchall()
ev = []
for r in rows:
    ev.append((r[0], 1)); ev.append((r[1], -1))
ev.sort()
depth, last, at = 0, lo, {}
for t, d in ev:
    at[depth] = at.get(depth, 0) + (t - last)
    last = t
    depth += d
at[0] = at.get(0, 0) + (hi - last)
wall = (hi - lo) / 1e3
print("wall %.1f us, kernels %d, sum of durations %.1f us" % (wall, len(rows), sum(r[1] - r[0] for r in rows) / 1e3))
for k in sorted(at):
    print("  %d kernels in flight: %.1f us (%.0f %%)" % (k, at[k] / 1e3, 100.0 * at[k] / 1e3 / wall))
if qcol:
    from collections import Counter
    print("  dispatches per queue:", dict(Counter(r[2] for r in rows)))
