#!/usr/bin/env python3
"""A/B of the single-stream time per batch (one context, device-resident input) between library builds:
   python tools/ab_latency.py lib1.so lib2.so ...   ("cur" = the in-tree library).  Each library runs in its own
child process, alternating, three rounds; prints ms per batch for batches of 1 and 8."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch, fixtures, hessgpu_amd
from hessgpu_amd import _abi
W, H = 1920, 1080
out = {}
for B in (1, 8):
    imgs = np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(B)])
    c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
    c.reserve(W, H, B)
    d = torch.from_numpy(imgs).to("cuda:0")
    for _ in range(5): c.run_device(d.data_ptr(), B, H, W)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(40): c.run_device(d.data_ptr(), B, H, W)
        best = min(best, (time.perf_counter() - t0) / 40)
    out[B] = round(best * 1e3, 4)
    c.close()
print(json.dumps(out))
''' % (ROOT, ROOT)


def main():
    libs = sys.argv[1:] or ["cur"]
    for rnd in range(3):
        for lib in libs:
            env = dict(os.environ)
            if lib == "cur":
                env.pop("HESS_LIB", None)
            else:
                env["HESS_LIB"] = os.path.join(ROOT, "tools", "_variants", lib, "libhessgpu.so") if not os.path.exists(lib) else lib
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
            print(lib, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
