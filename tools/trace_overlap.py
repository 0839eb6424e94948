#!/usr/bin/env python3
"""Busy fraction and overlap of the kernels in a rocprofv3 kernel trace (several streams):
   python tools/trace_overlap.py <dir with *_kernel_trace.csv> [skip_fraction]"""
import csv
import glob
import sys


def main():
    f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    iv = []
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    iv.sort()
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    lo = t0 + (t1 - t0) * skip  # steady state: skip warm-up
    ev = []
    tot = 0
    for s, e, _ in iv:
        if e <= lo:
            continue
        s = max(s, lo)
        ev.append((s, 1)); ev.append((e, -1)); tot += e - s
    ev.sort()
    busy, depth, last = 0, 0, None
    hist = {}
    for t, d in ev:
        if last is not None and depth > 0:
            busy += t - last
        if last is not None:
            hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    span = t1 - lo
    print(f"span {span/1e6:.2f} ms, union busy {busy/span:.3f}, sum of kernel time / span {tot/span:.3f}")
    for k in sorted(hist):
        print(f"  {k} kernels in flight: {hist[k]/span:.3f}")


if __name__ == "__main__":
    main()
