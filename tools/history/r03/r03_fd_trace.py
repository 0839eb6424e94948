#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/r03/r03_fill_drain.py: for the LAST timed run of a pipeline depth, the device's busy
share and the number of distinct batches (descriptor launches delimit them per queue) in 0.5 ms windows.
   python tools/r03/r03_fd_trace.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
# split into bursts separated by > 3 ms of silence; take the last burst with >= 18 descriptor launches
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(r[1] for r in cur[-50:]) > 3_000_000:
        bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
for bi, bu in enumerate(bursts):
    nd = sum(1 for r in bu if "descriptor_kernel" in r[3])
    if nd < 18:
        continue
    t0 = bu[0][0]; t1 = max(r[1] for r in bu)
    print(f"burst {bi}: {nd} descriptor launches, {len(bu)} dispatches, {(t1 - t0) / 1e6:.2f} ms")
    W = 500_000
    nb = (t1 - t0) // W + 1
    busy = [0] * nb; kern = [0.0] * nb
    ev = []
    for s, e, q, n in bu:
        ev.append((s, 1)); ev.append((e, -1))
        # summed kernel time per window
        a = s
        while a < e:
            k = (a - t0) // W
            bnd = min(e, t0 + (k + 1) * W)
            kern[k] += bnd - a
            a = bnd
    ev.sort()
    active, last = 0, t0
    for t, dlt in ev:
        if active > 0:
            a = last
            while a < t:
                k = (a - t0) // W
                bnd = min(t, t0 + (k + 1) * W)
                busy[k] += bnd - a
                a = bnd
        active += dlt; last = t
    desc_end = sorted((e - t0) / 1e6 for s, e, q, n in bu if "descriptor_kernel" in n)
    print("   descriptor launches end at (ms):", " ".join(f"{x:.2f}" for x in desc_end))
    print("   window busy share / mean kernels in flight:")
    print("   " + " ".join(f"{busy[k] / W:.2f}/{kern[k] / W:.1f}" for k in range(nb)))
