"""Kernel timeline of one timed bench step from a rocprofv3 --kernel-trace csv: start / end / duration (us), stream, kernel.
usage: python tools/step_timeline.py <kernel_trace.csv> [step index among the k_walk<false> launches, default 4]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_walk<false>" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f  s%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][:60]))
