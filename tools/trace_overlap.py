#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 --kernel-trace CSV of a pipelined bench run:

  python tools/trace_overlap.py <dir with *_kernel_trace.csv> [skip_fraction]

Over the steady part of the trace (the first `skip_fraction`, default 0.3, of the dispatches is dropped): wall time,
time with at least one kernel running, mean number of kernels in flight, and per kernel the summed duration next to
its share of the wall time -- set against a single-stream trace this shows which kernels stretch when they overlap.
Then the IDLE intervals (no kernel in flight): how they are distributed by length, and which kernel ended before / which
started after the ones that carry the idle time (round 5: where do the pipelined device's gaps come from?).
Then the hardware queues: per queue the streams it serves, its share of the wall time with a kernel running, and the gaps
between its consecutive kernels by length; and the time with exactly ONE kernel in flight by that kernel's name (is a
memory-bound kernel alone on the device while an issue-bound one waits in another queue?).
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("hess::", "")
    return re.sub(r"\(.*", "", name)


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = []
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Thread_Id", "?"))))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = 0; depth = 0; last = t0; area = 0
    hist = collections.Counter()
    for t, k in ev:
        if depth > 0:
            busy += t - last
        area += depth * (t - last)
        hist[depth] += t - last
        depth += k; last = t
    wall = t1 - t0
    print(f"dispatches {len(rows)}  wall {wall/1e6:.3f} ms  busy(>=1 kernel) {busy/1e6:.3f} ms ({busy/wall:.3f})  mean kernels in flight {area/wall:.2f}")
    print("time share by kernels in flight:", {k: round(v / wall, 3) for k, v in sorted(hist.items())})
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n, *_ in rows:
        per[n][0] += e - s; per[n][1] += 1
    print("kernel,calls,sum_ms,avg_us,sum/wall")
    for n, (tot, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print(f"{n},{c},{tot/1e6:.3f},{tot/c/1e3:.2f},{tot/wall:.3f}")
    # idle intervals: sweep the dispatches in start order, keep the running maximum of the end times
    gaps = []
    cur_end, cur = rows[0][1], rows[0]
    for r in rows[1:]:
        if r[0] > cur_end:
            gaps.append((r[0] - cur_end, cur, r))
        if r[1] > cur_end:
            cur_end, cur = r[1], r
    idle = sum(g[0] for g in gaps)
    print(f"idle intervals: {len(gaps)}, {idle/1e6:.3f} ms in all ({idle/wall:.3f} of the wall time)")
    bins = [(0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 20e3), (20e3, 50e3), (50e3, 1e12)]
    for lo, hi in bins:
        sel = [g for g in gaps if lo <= g[0] < hi]
        print(f"  {lo/1e3:5.0f} - {hi/1e3 if hi < 1e11 else float('inf'):5.0f} us: {len(sel):5d} intervals, {sum(g[0] for g in sel)/1e6:8.3f} ms")
    pair = collections.defaultdict(lambda: [0, 0, 0])
    for g, a, b in gaps:
        k = (a[2], b[2], "same queue" if a[3] == b[3] else "other queue")
        pair[k][0] += g; pair[k][1] += 1
    print("idle time by (kernel that ended before, kernel that started after, queue): total ms, intervals, mean us")
    for k, (tot, c, _) in sorted(pair.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {tot/1e6:7.3f} ms {c:5d} x {tot/c/1e3:6.1f} us   {k[0]} -> {k[1]} ({k[2]})")
    # hardware queues
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    print("queue: streams, dispatches, busy/wall, gaps between consecutive kernels of the queue (count / ms): <3us, 3-8, 8-20, 20-100, >100")
    for q, rs in sorted(byq.items()):
        rs.sort()
        qbusy = 0; cur_end = rs[0][0]
        gl = [[0, 0] for _ in range(5)]
        for a in rs:
            if a[0] > cur_end:
                g = a[0] - cur_end
                i = 0 if g < 3e3 else 1 if g < 8e3 else 2 if g < 20e3 else 3 if g < 100e3 else 4
                gl[i][0] += 1; gl[i][1] += g
            qbusy += max(0, a[1] - max(a[0], cur_end))
            cur_end = max(cur_end, a[1])
        streams = sorted({a[4] for a in rs})
        print(f"  queue {q}: streams {','.join(streams)}  {len(rs)} dispatches  busy {qbusy/wall:.3f}  " + "  ".join(f"{c}/{t/1e6:.2f}" for c, t in gl))
    # time with exactly one kernel in flight, by kernel
    ev2 = []
    for i, (s_, e_, *_rest) in enumerate(rows):
        ev2.append((s_, 1, i)); ev2.append((e_, -1, i))
    ev2.sort()
    live = set(); last = t0
    alone = collections.Counter(); pairs = collections.Counter()
    for t, k, i in ev2:
        if len(live) == 1:
            alone[rows[next(iter(live))][2]] += t - last
        elif len(live) == 2:
            a, b = sorted(rows[j][2] for j in live)
            pairs[(a, b)] += t - last
        if k > 0: live.add(i)
        else: live.discard(i)
        last = t
    print("time with exactly one kernel in flight, by kernel (share of the wall time):")
    for n, t in alone.most_common(10):
        print(f"  {t/wall:.3f}  {n}")
    print("time with exactly two kernels in flight, by pair:")
    for (a, b), t in pairs.most_common(14):
        print(f"  {t/wall:.3f}  {a} + {b}")


if __name__ == "__main__":
    main()
