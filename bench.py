#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Hessian + SIFT hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1: run directly)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        (N>1, one rank per GPU)

Metric (BASELINE.json): Mpixels/s end-to-end (pyramid -> descriptor) on 1920x1080 images.
Workload: BASELINE.json configs[1] -- 1920x1080 synthetic blobs, default octaves / DoG levels, top-K = 4096 -- as
batches of `--batch` distinct images per GPU and step; for N > 1 this is configs[3] (a batch of 8 x N images sharded
over N GPUs, 64 images at N = 8, RCCL gather of the feature lists to rank 0).  Per-GPU work is the same at every N
(weak scaling).  A step = one pass of the hot path over one batch per GPU, the u8 luminance pixels already resident
in HBM when the timed region starts; a step ends with the keypoints + descriptors of the batch in host memory
(hess_wait returns) and, for N > 1, the gather.  `--contexts` contexts (streams) per GPU are pipelined.

Added to the contract's JSON line (rank 0, N = 1 unless noted):
  roofline            the kernel that takes most device time per step, roofline_secondary the runner-up (descriptor
                      and Gaussian kernels): algorithmic bytes per launch / average launch duration, measured with
                      hipEvents on the context's stream in a single-stream leg of the same run; the descriptor entry
                      also carries "valu": vector instructions per launch (PMC pass kept under profiles/) / duration
                      against the issue peak 1024 SIMDs x 2.4 GHz / 2 cycles
  value_host_to_host  the same steps starting from pinned HOST pixels (hess_submit_host: one asynchronous transfer per
                      batch, pipelined over the contexts) -- SURVEY 8(d)'s definition of the metric, PCIe included
  latency_ms_single_image   one 1080p image, host pixels -> host results, one context (the drop-in RunSIFT call)
  parity_checked      image 0 of the timed run compared bit for bit with the CPU oracle on the same pixels
  cpu_baseline        the CPU oracle (a port of the reference's CUDA path; the reference has no CPU path) timed on
                      rank 0's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, TOPK = 1920, 1080, 4096
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINST = 1024 * 2.4 / 2.0     # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (configs[3]: 64 images over 8 GPUs)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic images per GPU (0 = --batch: all distinct)")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (streams) pipelined per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel hipEvents")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host and single-image legs")
    ap.add_argument("--octaves", type=int, default=-1, help="developer experiments only: limit the octave count (-no); "
                    "the headline workload uses the default (all 7 octaves of 1920x1080)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import fixtures  # tests/fixtures.py: the synthetic generator of configs[1]
    import hessgpu_amd
    from hessgpu_amd import _abi, dist as hdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HESS_BENCH_FORCE_DIST=1 runs the RCCL gather path with a single rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("HESS_BENCH_FORCE_DIST") == "1"
    json_fd = None
    if use_dist:
        import torch.distributed as tdist

        # RCCL and gloo print banners on stdout when their communicators come up; the contract is ONE JSON line on
        # rank 0's stdout, so everything but that line goes to stderr from here on
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group(backend="nccl", device_id=dev)
        # per-image counts travel over a gloo side group (host, loopback: one node), the payload over RCCL
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        try:
            hdist.enable_host_count_exchange()
        except Exception as e:  # same environment on every rank: all ranks fall back together
            if rank == 0:
                print(f"bench.py: gloo side group unavailable ({e}); counts go through RCCL", file=sys.stderr)

    B = args.batch
    nd = max(1, min(args.distinct or B, B))
    # image index = position in the global batch of configs[3]: rank r owns images r*B .. r*B+B-1
    imgs = np.stack([fixtures.synthetic_blobs(W, H, rank * B + i) for i in range(nd)])
    imgs = np.concatenate([imgs] * ((B + nd - 1) // nd))[:B]
    d_imgs = torch.from_numpy(imgs).to(dev)  # [B,H,W] u8 resident in HBM

    # Contexts used round-robin: while one batch's results travel to the host (and, for N > 1, are gathered over
    # RCCL), the next batches' kernels already run on the other contexts' streams.
    nctx = max(1, args.contexts)
    ctxs = [hessgpu_amd.HessContext(local_rank, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK,
                                    octave_num=args.octaves)
            for _ in range(nctx)]
    for c in ctxs:
        c.reserve(W, H, B)

    def finish(c):
        c.wait()
        counts = [c.count(b) for b in range(B)]
        if use_dist:
            keys, desc = hdist.device_feature_tensors(c, counts, dev)
            hdist.gather_feature_lists(counts, keys, desc, dst=0)
        return counts

    def run_steps(n, submit):
        """n steps, software-pipelined over the contexts; every step is submitted and finished inside."""
        counts = None
        inflight = []
        for i in range(n):
            c = ctxs[i % nctx]
            if len(inflight) == nctx:
                counts = finish(inflight.pop(0))
            submit(c)
            inflight.append(c)
        while inflight:
            counts = finish(inflight.pop(0))
        return counts

    def submit_resident(c):
        c.submit_device(d_imgs.data_ptr(), B, H, W)

    def fence():
        if use_dist:
            tdist.barrier()
        torch.cuda.synchronize()

    counts = run_steps(max(args.warmup, 1), submit_resident)
    fence()
    t0 = time.perf_counter()
    counts = run_steps(args.steps, submit_resident)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    last = ctxs[(args.steps - 1) % nctx]          # context that ran the last timed step: its results are still there
    timed_k0, timed_d0 = last.fetch(0)            # image 0 of the timed run (parity_checked below)
    timed_keys = [last.fetch(b)[0] for b in range(B)]

    # Legs outside the timed region (rank 0 alone reports them; every rank runs the device ones to stay in step).
    # Roofline leg: per-kernel hipEvent durations are only meaningful when kernels of different streams do not
    # overlap, so the events are recorded in a separate single-stream leg of the same run (same batch, one context).
    prof, prof_dev, roof_steps = None, None, 0
    if not args.no_profile:
        c = ctxs[0]
        roof_steps = max(3, min(args.steps, 10))
        c.profile_enable(True)
        c.profile_reset()
        for _ in range(roof_steps):
            c.run_device(d_imgs.data_ptr(), B, H, W)
        prof = c.profile()
        c.profile_enable(False)
        # The shipped descriptor launch also stores its 536 B per feature into pinned host memory (posted PCIe
        # writes); alone on the device it then waits for the link.  The same leg on a context that delivers by a
        # copy after the kernels shows the kernel's own duration.
        saved = os.environ.get("HESS_HOST_DIRECT")
        os.environ["HESS_HOST_DIRECT"] = "0"
        cd = hessgpu_amd.HessContext(local_rank, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK,
                                     octave_num=args.octaves)
        if saved is None:
            del os.environ["HESS_HOST_DIRECT"]
        else:
            os.environ["HESS_HOST_DIRECT"] = saved
        cd.reserve(W, H, B)
        cd.run_device(d_imgs.data_ptr(), B, H, W)
        cd.profile_enable(True)
        cd.profile_reset()
        for _ in range(roof_steps):
            cd.run_device(d_imgs.data_ptr(), B, H, W)
        prof_dev = cd.profile()
        cd.close()
    host = None
    if world == 1 and not use_dist and not args.no_host_leg:
        host = host_legs(ctxs, nctx, imgs, B, args, run_steps, fence, torch)

    if rank == 0:
        pixels = float(world) * B * args.steps * W * H
        value = pixels / dt / 1e6
        workload = "1920x1080 synthetic blobs (tests/fixtures.py), default octaves/DoG levels, top-K=4096 [configs[1]]"
        if args.octaves > 0:
            workload += f" -- DEVELOPER RUN limited to {args.octaves} octaves, not the headline workload"
        if world > 1:
            workload = (f"batch of {B * world} synthetic 1920x1080 images sharded over {world} GPUs, {B} per GPU per step, "
                        "top-K=4096, RCCL gather of the feature lists to rank 0 [configs[3]: 64 images at 8 GPUs]")
        out = {
            "metric": "Mpixels/s end-to-end (pyramid->descriptor), 1920x1080",
            "value": round(value, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "images_per_gpu_per_step": B,
                "pipelined_contexts_per_gpu": nctx,
                "distinct_images_per_gpu": nd,
                "features_per_image_mean": round(float(np.mean(counts)), 1),
                "sharding": (f"images over {world} rank(s), exact-size RCCL send/recv of the feature lists to rank 0"
                             if use_dist else "single GPU"),
                "input": "u8 luminance resident in HBM; results delivered to host memory",
            },
        }
        if prof is not None:
            roofs = rooflines(prof, roof_steps, timed_keys, B, prof_dev)
            ranked = sorted(roofs, key=lambda r: -r["ms_per_step"])
            if ranked:
                out["roofline"] = ranked[0]
                out["roofline"]["leg"] = f"{roof_steps} single-stream steps after the timed region (kernels do not overlap)"
            if len(ranked) > 1:
                out["roofline_secondary"] = ranked[1]
            out["kernel_ms_per_step"] = {k: round(v["ms"] / roof_steps, 4) for k, v in prof.items() if v["launches"]}
        if host is not None:
            out.update(host)
        if world == 1 and not args.no_cpu_baseline:
            out["parity_checked"] = parity_check(imgs[0], timed_k0, timed_d0)
            out["cpu_baseline"] = cpu_baseline(imgs[:min(nd, 4)])
        if json_fd is None:
            print(json.dumps(out), flush=True)
        else:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    for c in ctxs:
        c.close()
    if use_dist:
        tdist.barrier()
        tdist.destroy_process_group()


def host_legs(ctxs, nctx, imgs, B, args, run_steps, fence, torch):
    """Host-to-host throughput (pinned input, one asynchronous transfer per batch, contexts pipelined) and the
    latency of one image through one context."""
    pinned = torch.from_numpy(imgs).pin_memory()

    def submit_pinned(c):
        c.submit_host(ptr=pinned.data_ptr(), batch=B, height=H, width=W)

    run_steps(max(2, nctx), submit_pinned)
    fence()
    n = max(4, args.steps)
    t0 = time.perf_counter()
    run_steps(n, submit_pinned)
    fence()
    dth = time.perf_counter() - t0
    one = imgs[:1].copy()  # pageable memory, as a caller of RunSIFT(w, h, data, ...) would hand it over
    c = ctxs[0]
    for _ in range(3):
        c.run(one)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        c.run(one)
    lat = (time.perf_counter() - t0) / reps
    c.reserve(W, H, B)
    return {
        "value_host_to_host": round(B * n * W * H / dth / 1e6, 2),
        "host_to_host": f"{n} steps from pinned host pixels (hess_submit_host), {nctx} contexts pipelined, PCIe transfer included",
        "latency_ms_single_image": round(lat * 1e3, 4),
    }


def rooflines(prof, steps, timed_keys, images, prof_dev=None):
    """Roofline entries of the two heavy kernels from the single-stream profile leg."""
    import numpy as np

    out = []
    g = prof["gauss"]
    if g["launches"]:
        achieved = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        out.append({
            "bound": "hbm",
            "kernel": "gauss_kernel (separable Gaussian + fused det-Hessian/gradient, one pyramid level of the batch per launch)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _profile_value("gauss_traffic.json", "hbm_bytes_per_launch"),
            "avg_launch_us": round(g["ms"] * 1e3 / g["launches"], 2),
            "algorithmic_bytes_per_launch": round(g["bytes"] / g["launches"], 1),
            "launches": g["launches"], "ms_per_step": round(g["ms"] / steps, 4),
        })
        gi = _profile_value("gauss_traffic.json", "valu_insts_per_image")
        if gi:
            rate = gi * images * steps / (g["ms"] * 1e-3) / 1e9
            out[-1]["valu"] = {"bound": "valu", "achieved": round(rate, 1), "peak": VALU_PEAK_GINST, "unit": "Ginst/s",
                               "frac": round(rate / VALU_PEAK_GINST, 4), "instructions_per_image": gi,
                               "source": "SQ_INSTS_VALU of the PMC pass in profiles/gauss_traffic.json x images of this run "
                                         "(about half are packed-FP32 or division/sqrt helper instructions that issue at half rate or less)"}
    d = prof["descriptor"]
    if d["launches"]:
        # algorithmic bytes of one launch (per output feature): the rotated 5x5-cell footprint of side
        # 5 * 3 * scale (scale at the octave's resolution) of 8-byte (gradient, theta) samples read once, 16 B of
        # feature record, 24 + 512 B written
        nfeat, fbytes = 0, 0.0
        for k in timed_keys:
            s_oct = k["s"].astype(np.float64) / (2.0 ** (k["level"] // 3))
            fbytes += float(np.sum((15.0 * s_oct) ** 2 * 8.0 + 16.0 + 24.0 + 512.0))
            nfeat += len(k)
        dur = d["ms"] * 1e-3 / d["launches"]
        achieved = fbytes / dur / 1e9
        e = {
            "bound": "hbm",
            "kernel": "descriptor_kernel<true> (one wavefront per feature: rotated-grid histogram + normalisation + result stores incl. the pinned host mirror)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _profile_value("descriptor_counters.json", "hbm_bytes_per_launch"),
            "avg_launch_us": round(dur * 1e6, 2), "algorithmic_bytes_per_launch": round(fbytes, 1),
            "features_per_launch": nfeat, "launches": d["launches"], "ms_per_step": round(d["ms"] / steps, 4),
        }
        insts = _profile_value("descriptor_counters.json", "valu_insts_per_feature")
        if insts:
            rate = insts * nfeat / dur / 1e9
            e["valu"] = {"bound": "valu", "achieved": round(rate, 1), "peak": VALU_PEAK_GINST, "unit": "Ginst/s",
                         "frac": round(rate / VALU_PEAK_GINST, 4), "instructions_per_feature": insts,
                         "source": "SQ_INSTS_VALU of the PMC pass in profiles/descriptor_counters.json x features of this run"}
        dd = (prof_dev or {}).get("descriptor")
        if dd and dd["launches"]:
            ddur = dd["ms"] * 1e-3 / dd["launches"]
            e["without_host_mirror"] = {
                "avg_launch_us": round(ddur * 1e6, 2), "achieved": round(fbytes / ddur / 1e9, 1), "unit": "GB/s",
                "kernel": "descriptor_kernel<false>",
                "note": "same launch on a context that delivers results by a copy after the kernels (HESS_HOST_DIRECT=0): "
                        "the shipped launch also stores keypoints + descriptors into pinned host memory and, alone "
                        "on the device, waits for PCIe",
            }
            if insts:
                e["without_host_mirror"]["valu_frac"] = round(insts * nfeat / ddur / 1e9 / VALU_PEAK_GINST, 4)
        out.append(e)
    return out


def _profile_value(name, key):
    """A number from a committed PMC summary under profiles/, if one exists."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f).get(key)
    except Exception:
        return None


def _host_cores():
    """CPU share of this process: cgroup quota if one is set, else the affinity mask; capped at 16
    (the GPU box gives one GPU's job 16 cores; more OpenMP threads than that only oversubscribe)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def parity_check(img0, gk, gd):
    """Image 0 of the timed run against the CPU oracle on the same pixels: keypoints and descriptors bit for bit."""
    import numpy as np
    from hessgpu_amd import _abi
    from oracle_lib import OracleSession  # the checker

    o = OracleSession(threads=_host_cores(), keep_levels=False, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK)
    o.run(img0[None])
    ok, od = o.fetch(0)
    o.close()
    return bool(len(ok) == len(gk) and ok.tobytes() == gk.tobytes() and np.array_equal(od.view(np.uint32), gd.view(np.uint32)))


def cpu_baseline(sample_imgs):
    """The CPU oracle on the same workload, all host cores (OpenMP), bounded sample."""
    from hessgpu_amd import _abi
    from oracle_lib import OracleSession  # the checker, timed here as the CPU baseline only

    cores = _host_cores()
    o = OracleSession(threads=cores, keep_levels=False, truncate_method=_abi.TRUNC_TOPK,
                      feature_count_threshold=TOPK)
    o.run(sample_imgs[:1])  # warm-up (page faults, OpenMP pool)
    n, t0 = 0, time.perf_counter()
    while True:
        for i in range(len(sample_imgs)):
            o.run(sample_imgs[i:i + 1])
            n += 1
        if time.perf_counter() - t0 > 10.0 or n >= 256:
            break
    dt = time.perf_counter() - t0
    o.close()
    # single-thread figure (SURVEY 8d asks for both): two images, about 2 s
    o1 = OracleSession(threads=1, keep_levels=False, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK)
    t1 = time.perf_counter()
    n1 = 0
    for i in range(min(2, len(sample_imgs))):
        o1.run(sample_imgs[i:i + 1])
        n1 += 1
    dt1 = time.perf_counter() - t1
    o1.close()
    return {
        "value_single_thread": round(n1 * W * H / dt1 / 1e6, 2),
        "value": round(n * W * H / dt / 1e6, 2),
        "unit": "Mpix/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} runs of the bench workload (1920x1080, top-K 4096) in {dt:.1f} s, "
                  f"oracle/hess_oracle.c with {cores} OpenMP threads",
    }


if __name__ == "__main__":
    main()
