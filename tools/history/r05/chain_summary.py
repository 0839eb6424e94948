#!/usr/bin/env python3
"""Summary of a HESS_CHAIN_STAMPS=1 stderr log of bench.py (pipelined part): per context the mean chain length, start delay
after submit, time from chain end to the wait's return; which contexts alternate on one hardware queue (a chain that
starts within 40 us of another context's chain end); phase offsets between the queues."""
import re, sys, json, collections
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"hess chain ctx (\S+): host submit ([\d.]+) - ([\d.]+)  device ([\d.]+) - ([\d.]+)  host wait ([\d.]+) - ([\d.]+)", l)
    if m:
        rows.append((m.group(1),) + tuple(float(m.group(i)) for i in range(2, 8)))
ctx = {c: i for i, c in enumerate(dict.fromkeys(r[0] for r in rows))}
pip = [r for r in rows if r[4] - r[3] > 1.45]          # pipelined chains are stretched; the single-stream legs are not
pip = pip[len(pip) // 4:]                               # steady part
per = collections.defaultdict(list)
for r in pip:
    per[ctx[r[0]]].append(r)
out = {}
for c, rs in sorted(per.items()):
    out[c] = (sum(r[4] - r[3] for r in rs) / len(rs), sum(r[3] - r[2] for r in rs) / len(rs), sum(r[6] - r[4] for r in rs) / len(rs))
ends = sorted((r[4], ctx[r[0]]) for r in pip)
follow = collections.Counter()
for r in pip:
    for e, c in ends:
        if c != ctx[r[0]] and 0 <= r[3] - e < 0.04:
            follow[(c, ctx[r[0]])] += 1
span = (max(r[4] for r in pip) - min(r[3] for r in pip)) / len(pip)
print(f"steps {len(pip)}  ms/step {span:.4f}  chain/start-delay/after-end by context: " + "  ".join(f"{c}:{a:.2f}/{b:.2f}/{d:.2f}" for c, (a, b, d) in out.items()))
print("  follows (ctx a's chain end -> ctx b's chain start within 40 us):", dict(follow.most_common(8)))
