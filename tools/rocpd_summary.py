#!/usr/bin/env python3
"""Dump the per-kernel stats of a rocprofv3 rocpd database (…_results.db) as CSV/markdown for profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["kernel,calls,total_us,avg_us,percent"]
    for name, calls, total, avg, pct in rows:
        lines.append(f"\"{name}\",{calls},{total / 1e3 if total > 1e7 else total:.3f},{avg / 1e3 if total > 1e7 else avg:.3f},{pct:.2f}")
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
