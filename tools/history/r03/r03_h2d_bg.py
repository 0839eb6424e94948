#!/usr/bin/env python3
"""Is the gap between the device-resident and the host-to-host rate the transfer itself or its place in the context's
stream?  The device-resident pipeline (six contexts) with a background thread that uploads one batch of pixels per step
on a stream of its own (same bytes over the link, no dependency on any context), against the two bench figures."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixtures, hessgpu_amd
from hessgpu_amd import _abi
W, H, B, K, NCTX = 1920, 1080, 8, 200, 6
imgs = np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(B)])
d = torch.from_numpy(imgs).cuda()
pinned = torch.from_numpy(imgs).pin_memory()
ctxs = [hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096) for _ in range(NCTX)]
for c in ctxs:
    c.reserve(W, H, B)

def run(n, submit):
    infl = []
    for i in range(n):
        c = ctxs[i % NCTX]
        if len(infl) == NCTX:
            infl.pop(0).wait()
        submit(c)
        infl.append(c)
    while infl:
        infl.pop(0).wait()

def measure(name, submit, bg=False):
    stop = threading.Event()
    steps_done = [0]
    def uploader():
        st = torch.cuda.Stream()
        dst = torch.empty_like(d)
        n = 0
        with torch.cuda.stream(st):
            while not stop.is_set():
                if n <= steps_done[0] + 2:      # keep about in step with the pipeline: one upload per step
                    dst.copy_(pinned, non_blocking=True)
                    st.synchronize()
                    n += 1
                else:
                    time.sleep(0.0001)
    run(12, submit); torch.cuda.synchronize()
    th = None
    if bg:
        th = threading.Thread(target=uploader); th.start()
    t0 = time.perf_counter()
    infl = []
    for i in range(K):
        c = ctxs[i % NCTX]
        if len(infl) == NCTX:
            infl.pop(0).wait(); steps_done[0] += 1
        submit(c); infl.append(c)
    while infl:
        infl.pop(0).wait(); steps_done[0] += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if th:
        stop.set(); th.join()
    print(f"{name}: {B*K*W*H/dt/1e6:.0f} Mpix/s ({dt*1e3/K:.3f} ms/step)")

sub_dev = lambda c: c.submit_device(d.data_ptr(), B, H, W)
sub_host = lambda c: c.submit_host(ptr=pinned.data_ptr(), batch=B, height=H, width=W)
for rep in range(2):
    measure("device-resident", sub_dev)
    measure("device-resident + background upload of one batch per step", sub_dev, bg=True)
    measure("host to host (upload on the context's stream)", sub_host)
