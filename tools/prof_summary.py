#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) from a rocpd .db into a small CSV/markdown
that can be committed under profiles/.   usage: prof_summary.py results.db out.md [note]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\n{note}\n\n")
        f.write("| kernel | calls | total (us) | avg (us) | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, tot, avg, pct in rows:
            f.write(f"| `{name[:100]}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
