#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database as a markdown table of kernels.
usage: rocpd_stats.py results.db "title" [--between-markers] [--steps N] > profiles/xyz.md
--between-markers: only the dispatches between the first two pats::profile_marker_kernel launches (bench.py brackets its
timed steps with them: no set-up, no warm-up, no parity leg in the table); --steps N adds a per-step column;
--list PATTERN appends the individual dispatches whose kernel name contains PATTERN, in launch order."""
import sqlite3
import sys


def main():
    args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and sys.argv[i] != "--list"]
    db = sqlite3.connect(args[0])
    title = args[1] if len(args) > 1 else args[0]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
    if steps:
        args = [a for a in args if a != str(steps)] or args
    where, note = "", ""
    if "--between-markers" in sys.argv:
        marks = db.execute("select K.start, K.end from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id = K.kernel_id "
                           "and S.guid = K.guid where S.display_name like '%profile_marker_kernel%' order by K.start").fetchall()
        if len(marks) < 2:
            raise SystemExit("rocpd_stats: fewer than two profile markers in the trace")
        where = " where K.start > %d and K.end < %d" % (marks[0][1], marks[1][0])
        note = "Only the dispatches between the two `pats::profile_marker_kernel` launches that bracket the timed steps " \
               "(%.2f ms of wall clock between them).\n\n" % ((marks[1][0] - marks[0][1]) / 1e6)
    rows = db.execute("select S.display_name, count(*), sum(K.end - K.start) / 1000.0, avg(K.end - K.start) / 1000.0 "
                      "from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid"
                      + where + " group by S.display_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    print("# %s\n" % title)
    sys.stdout.write(note)
    print("| kernel | calls | total us | avg us | % |" + (" us per step |" if steps else ""))
    print("|---|---|---|---|---|" + ("---|" if steps else ""))
    for n, c, t, a in rows[:40]:
        n = n.replace("void ", "")
        n = n.split("(")[0] if n.startswith("pats::") or "<" not in n else n[:70]
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (n[:90], c, t, a, 100.0 * t / total) + (" %.1f |" % (t / steps) if steps else ""))
    print("\nsum of kernel time: %.1f us" % total + (" = %.1f us per step" % (total / steps) if steps else ""))
    if "--list" in sys.argv:
        pat = sys.argv[sys.argv.index("--list") + 1]
        each = db.execute("select S.display_name, (K.end - K.start) / 1000.0, K.grid_size_x, K.workgroup_size_x from rocpd_kernel_dispatch K "
                          "join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid" + where + " order by K.start").fetchall()
        print("\ndispatches matching `%s`, in launch order (us, grid threads, workgroup):\n" % pat)
        for n, d, gx, wx in each:
            if pat in n:
                print("- `%s` %.1f us, grid %d, workgroup %d" % (n.replace("void ", "").split("(")[0][:60], d, gx, wx))
    foreign = [r for r in rows if not r[0].replace("void ", "").startswith("pats::")]
    print("\nkernels outside `pats::` in the table: %d%s" % (len(foreign), "" if not foreign else " (" + ", ".join(sorted(set(r[0][:50] for r in foreign))) + ")"))


if __name__ == "__main__":
    main()
