#!/usr/bin/env python3
"""Prints a window of a rocprofv3 kernel (+ memory copy) trace as a timeline, one line per dispatch/copy:
   python tools/trace_timeline.py <dir> [start_fraction=0.6] [count=140]"""
import csv, glob, re, sys

def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("hess::", "")
    return re.sub(r"\(.*", "", name)[:40]

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 140
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", ""))))
rows.sort()
i0 = int(len(rows) * frac)
t0 = rows[i0][0]
for s, e, q, n in rows[i0:i0 + cnt]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {q:6s} {n}")
