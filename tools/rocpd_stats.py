#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (top_kernels view) as a markdown table.
usage: rocpd_stats.py results.db "title" > profiles/xyz.md"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    rows = db.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    print("# %s\n" % title)
    print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
    for n, c, t, a, p in rows[:28]:
        n = n.replace("void ", "")
        n = n.split("(")[0] if n.startswith("pats::") or "<" not in n else n[:70]
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (n[:90], c, t, a, p))


if __name__ == "__main__":
    main()
