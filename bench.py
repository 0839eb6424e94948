#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Hessian + SIFT hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1: run directly)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        (N>1, one rank per GPU)

Metric (BASELINE.json): Mpixels/s end-to-end (pyramid -> descriptor) on 1920x1080 images.
Workload = BASELINE.json configs[1]: 1920x1080 synthetic blobs, default octaves / DoG levels,
top-K = 4096.  A step = one pass of the hot path over one batch of `--batch` images per GPU, the
u8 luminance pixels already resident in HBM when the timed region starts; a step ends with the
keypoints + descriptors of the batch in host memory (hess_run_device returns) and, for N > 1,
the RCCL gather of the feature lists to rank 0.  Weak scaling: per-GPU work is fixed.

Adds to the contract's JSON line:
  roofline      dominant kernel (separable Gaussian): algorithmic bytes per launch / average launch
                duration measured with hipEvents on the context's stream during the timed region
  cpu_baseline  the CPU oracle (a port of the reference's CUDA path; the reference has no CPU
                path) timed on rank 0's host cores on a bounded sample of the same workload
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic images per GPU (tiled to --batch)")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (streams) pipelined per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel hipEvents")
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
    if use_dist:
        import torch.distributed as tdist

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
    nd = max(1, min(args.distinct, B))
    imgs = np.stack([fixtures.synthetic_blobs(W, H, rank * B + i) for i in range(nd)])
    imgs = np.concatenate([imgs] * ((B + nd - 1) // nd))[:B]
    d_imgs = torch.from_numpy(imgs).to(dev)  # [B,H,W] u8 resident in HBM

    # Contexts used round-robin (three measured best on MI355X: 1 -> 10.8, 2 -> 12.4, 3 -> 12.7, 4 -> 12.1 Gpix/s):
    # while one batch's results travel to the host (and, for N > 1,
    # are gathered over RCCL), the next batch's kernels already run on the other context's stream.
    nctx = max(1, args.contexts)
    ctxs = [hessgpu_amd.HessContext(local_rank, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK)
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

    def run_steps(n):
        """n steps, software-pipelined over the contexts; every step is submitted and finished inside."""
        counts = None
        inflight = []
        for i in range(n):
            c = ctxs[i % nctx]
            if len(inflight) == nctx:
                counts = finish(inflight.pop(0))
            c.submit_device(d_imgs.data_ptr(), B, H, W)
            inflight.append(c)
        while inflight:
            counts = finish(inflight.pop(0))
        return counts

    def fence():
        if use_dist:
            tdist.barrier()
        torch.cuda.synchronize()

    counts = run_steps(max(args.warmup, 1))
    fence()
    t0 = time.perf_counter()
    counts = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    for c in ctxs:
        c.profile_enable(False)
    # Roofline leg: per-kernel hipEvent durations are only meaningful when kernels of different
    # streams do not overlap, so with more than one pipelined context the events are recorded in a
    # separate single-stream leg of the same run (same batch, same inputs, one context).
    prof, roof_steps = None, 0
    if not args.no_profile:
        c = ctxs[0]
        roof_steps = max(3, min(args.steps, 10))
        c.profile_enable(True)
        c.profile_reset()
        for _ in range(roof_steps):
            c.run_device(d_imgs.data_ptr(), B, H, W)
        prof = c.profile()
        c.profile_enable(False)

    if rank == 0:
        pixels = float(world) * B * args.steps * W * H
        value = pixels / dt / 1e6
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
                "workload": "1920x1080 synthetic blobs (tests/fixtures.py), default octaves/DoG levels, top-K=4096",
                "images_per_gpu_per_step": B,
                "pipelined_contexts_per_gpu": nctx,
                "distinct_images_per_gpu": nd,
                "features_per_image_mean": round(float(np.mean(counts)), 1),
                "sharding": f"images over {world} rank(s), RCCL gather of feature lists" if use_dist else "single GPU",
                "input": "u8 luminance resident in HBM; results delivered to host memory",
            },
        }
        if prof is not None:
            g = prof["gauss"]
            if g["launches"]:
                achieved = g["bytes"] / (g["ms"] * 1e-3) / 1e9
                out["roofline"] = {
                    "bound": "hbm",
                    "kernel": "gauss_kernel (separable Gaussian, one pyramid level of the batch per launch)",
                    "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": _traffic_from_profiles(),
                    "avg_launch_us": round(g["ms"] * 1e3 / g["launches"], 2),
                    "algorithmic_bytes_per_launch": round(g["bytes"] / g["launches"], 1),
                    "launches": g["launches"],
                }
                out["roofline"]["leg"] = f"{roof_steps} single-stream steps after the timed region (kernels do not overlap)"
            out["kernel_ms_per_step"] = {k: round(v["ms"] / roof_steps, 4) for k, v in prof.items() if v["launches"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(imgs[:nd])
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if use_dist:
        tdist.barrier()
        tdist.destroy_process_group()


def _traffic_from_profiles():
    """HBM bytes per launch of the dominant kernel from the committed PMC pass, if one exists."""
    p = os.path.join(ROOT, "profiles", "gauss_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("hbm_bytes_per_launch")
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


def cpu_baseline(sample_imgs):
    """The CPU oracle on the same workload, all host cores (OpenMP), bounded sample."""
    import numpy as np
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
