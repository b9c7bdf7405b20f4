#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result as the per-kernel --stats table (CSV)."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = ["Name,Calls,TotalDurationUs,AverageUs,Percentage"]
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        lines.append('"%s",%d,%.3f,%.3f,%.4f' % (short, calls, tot, avg, pct))
    text = "\n".join(lines) + "\n"
    if out_path:
        with open(out_path, "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
