#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Hessian + SIFT hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1: runs in this process; N>1 with no WORLD_SIZE in the
                                                          environment: starts its own N ranks as child processes,
                                                          hessgpu_amd/launch.py, and relays rank 0's line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        (N>1 under a launcher, one rank per GPU)

Metric (BASELINE.json): Mpixels/s end-to-end (pyramid -> descriptor) on 1920x1080 images.
Workload: BASELINE.json configs[1] -- 1920x1080 synthetic blobs, default octaves / DoG levels, top-K = 4096 -- as
batches of `--batch` distinct images per GPU and step; for N > 1 this is configs[3] (a batch of 8 x N images sharded
over N GPUs, 64 images at N = 8, RCCL gather of the feature lists to rank 0, landed in rank 0's host memory).  Per-GPU
work is the same at every N (weak scaling).  A step = one pass of the hot path over one batch per GPU; `value` (=
`value_device_resident`) is measured as the driver contract states it -- the u8 luminance pixels already resident in HBM
when the timed region starts -- and a step ends with the keypoints + descriptors of the batch in host memory (hess_wait
returns) and, for N > 1, the gathered lists in rank 0's host memory.  `--contexts` contexts (streams) per GPU are
pipelined.

Added to the contract's JSON line (rank 0; N = 1 unless noted):
  value_host_to_host  the same steps starting from pinned HOST pixels (hess_submit_host: one asynchronous transfer per
                      batch, pipelined over the contexts) -- SURVEY 8(d)'s definition of the metric, PCIe included
  value_device_resident   = value, under its own name
  roofline            the kernel that takes most device time per step, roofline_secondary the runner-up (descriptor
                      and Gaussian kernels): algorithmic bytes per launch / average launch duration, measured with
                      hipEvents on the context's stream in a single-stream leg of the same run; each carries "valu":
                      vector instructions per launch (PMC pass kept under profiles/) / duration against the NOMINAL
                      issue peak (1024 SIMDs x 2.4 GHz / 2 cycles) and against the MEASURED sustained all-CU rate of
                      tools/micro/valu_peak.hip (profiles/valu_peak.json)
  value_steady_state  (runs of fewer than 150 steps, single rank) the same pipeline over 200 steps after the timed region:
                      the contract's fences put the pipeline's fill and drain inside the K timed steps (about 2 ms of
                      the 21 at K = 20)
  latency_ms_single_image   one 1080p image, pageable host pixels -> host results, one context (the C-ABI call under
                      the drop-in RunSIFT)
  value_siftgpu_api_1thread / value_siftgpu_api_threads   the reference's own calling pattern through libsiftgpu.so:
                      RunSIFT(w, h, data, GL_LUMINANCE, GL_UNSIGNED_BYTE) + GetFeatureVector per image, one instance
                      (speed.cpp:107-124) and one instance per host thread (MultiThreadSIFT.cpp:83-156,231-244), run
                      by apps/multithread.cpp -mem as a child process after the timed region
  configs4            BASELINE.json configs[4] (one 4096x4096 image, -maxd 4096 -topk 65536 -half) after the timed
                      region: Mpix/s with one and with three contexts, features, its descriptor launch's roofline entry
  parity_checked      image 0 of the timed run (N > 1: image 0 of every rank, from the gathered lists) compared bit for
                      bit with the CPU oracle on the same pixels
  cpu_baseline        the CPU oracle (a port of the reference's CUDA path; the reference has no CPU path) timed on
                      rank 0's host cores on a bounded sample of the same workload
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, TOPK = 1920, 1080, 4096
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINST = 1024 * 2.4 / 2.0     # nominal: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (configs[3]: 64 images over 8 GPUs)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic images per GPU (0 = --batch: all distinct)")
    ap.add_argument("--contexts", type=int, default=0, help="contexts (streams) pipelined per GPU; 0 = 6 on one GPU, 5 per rank for N > 1.  "
                    "Round 5, same call, the driver's --steps 20 --warmup 5 / 200 steady steps, Gpix/s: 3: 18.4 - 20.0 / 21.8, 4: 18.9 - 19.2 / 20.6, "
                    "5: 19.7 - 19.8 / 21.9, 6: 20.5 / 21.3 - 22.4, 7: 20.2 - 20.4 / 22.5, 8: 19.7 / 20.4 - 21.0, 9: 20.2 - 20.4 / 22.0, 11: 19.9 - 20.5 / 22.3 "
                    "(the streams share four hardware queues).  N > 1: five contexts per rank keep a node of eight ranks at 40 copier "
                    "threads / 40 streams and the node-shared result buffers near 1.2 GB for 3 % of one GPU's steady rate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel hipEvents")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host and single-image legs")
    ap.add_argument("--no-api-leg", action="store_true", help="skip the libsiftgpu.so (RunSIFT + GetFeatureVector) legs")
    ap.add_argument("--no-configs4", action="store_true", help="skip the 4096x4096 leg (BASELINE.json configs[4])")
    ap.add_argument("--no-real-images", action="store_true", help="skip the leg on the reference's own data/ images")
    ap.add_argument("--no-matcher", action="store_true", help="skip the descriptor matcher leg")
    ap.add_argument("--no-steady", action="store_true", help="skip the 200-step steady-state leg that follows a timed region of "
                    "fewer than 150 steps (counter passes: every launch of the run is then one of warmup + steps + profile leg)")
    ap.add_argument("--api-threads", type=int, default=8, help="SiftGPU instances (host threads) of the multi-instance leg")
    ap.add_argument("--gather-dest", choices=("shm", "host", "hbm"), default="shm",
                    help="N > 1: where rank 0 finds the feature lists of the global batch after the RCCL gather: shm = "
                         "also in host memory, read in place from the node-shared pinned buffers every rank's own "
                         "copier delivers into (each GPU over its own host link); host = rank 0 copies the gathered "
                         "lists from its HBM to pinned memory through its own link; hbm = in rank 0's HBM only")
    ap.add_argument("--desc-order", type=int, default=-1, help="descriptor summation order (include/hess_abi.h HESS_DESC_ORDER_*: 0 interleaved, "
                    "1 sequential = the reference's, 2 pixel raster); default: what hess_default_params chooses")
    ap.add_argument("--octaves", type=int, default=-1, help="developer experiments only: limit the octave count (-no); "
                    "the headline workload uses the default (all 7 octaves of 1920x1080)")
    ap.add_argument("--spawn", action="store_true", help="start the rank(s) as child processes even for --gpus 1 (the one rank then "
                    "takes the N > 1 code path: process group, helper-thread gather; HESS_BENCH_FORCE_DIST=1)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds after which a self-launched job is ended (exit code 124)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # No launcher around us: be the launcher.  This process makes no GPU call (it never imports torch); the ranks are
        # child processes (hessgpu_amd/launch.py: rendezvous environment, rank 0's JSON line relayed, worst exit code,
        # the whole job ended when a rank dies or the time is up) -- the reference starts its workers itself too
        # (TestWin/MultiThreadSIFT.cpp:231-244, ServerSiftGPU/ServerSiftGPU.cpp:156-194).
        from hessgpu_amd import launch
        env = dict(os.environ)
        if args.gpus == 1:
            env["HESS_BENCH_FORCE_DIST"] = "1"
        child = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--spawn"]
        sys.exit(launch.run_ranks(child, args.gpus, timeout_s=args.launch_timeout, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HESS_BENCH_SAME_GPU") == "1":   # rehearsal of the N > 1 control flow on a one-GPU box (with HESS_BENCH_BACKEND=gloo)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU: keep this rank's host threads on the CPUs next to its GPU -- before any GPU call, so that
    # the runtime's threads and pinned allocations inherit the binding (one rank alone on a node keeps all its CPUs)
    bound = None
    if world > 1 or os.environ.get("HESS_BENCH_FORCE_DIST") == "1" or os.environ.get("HESS_BENCH_BIND") == "1":
        from hessgpu_amd import numa
        bound = numa.bind_to_gpu(local_rank)

    import numpy as np
    import torch

    import fixtures  # tests/fixtures.py: the synthetic generator of configs[1]
    import hessgpu_amd
    from hessgpu_amd import _abi, dist as hdist

    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment: the launcher's rank count and --gpus must agree "
                  "(without WORLD_SIZE set, bench.py starts its own ranks)", file=sys.stderr)
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
        backend = os.environ.get("HESS_BENCH_BACKEND", "nccl")   # "gloo": rehearsal only (ranks sharing one GPU cannot form an RCCL group)
        if backend == "nccl":
            tdist.init_process_group(backend="nccl", device_id=dev)
        else:
            tdist.init_process_group(backend=backend)
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
    nctx = args.contexts if args.contexts > 0 else (6 if world == 1 else 5)
    order_kw = {"descriptor_order": args.desc_order} if args.desc_order >= 0 else {}
    def make_contexts():
        return [hessgpu_amd.HessContext(local_rank, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK,
                                        octave_num=args.octaves, **order_kw) for _ in range(nctx)]

    ctxs = make_contexts()
    desc_order = int(ctxs[0].params.descriptor_order)   # hess_default_params' choice (include/hess_abi.h, HESS_DESC_ORDER_*)
    readers = None
    gather_dest = args.gather_dest if use_dist else None
    if gather_dest == "shm":
        # every rank keeps its contexts' pinned result buffers in shared memory of the node; rank 0 maps them
        tok = [f"{os.environ.get('MASTER_PORT', '0')}_{os.getpid()}"] if rank == 0 else [None]
        shm_prefix = os.environ.get("HESS_BENCH_SHM_PREFIX", "hessbench")   # (a prefix with a '/' rehearses the fallback)
        tdist.broadcast_object_list(tok, src=0, group=hdist.count_group())
        ok = 1
        try:
            for j, c in enumerate(ctxs):
                c.share_results(f"{shm_prefix}_{tok[0]}_r{rank}_c{j}")
                c.reserve(W, H, B)
        except Exception as e:   # e.g. /dev/shm too small for this job
            print(f"bench.py: rank {rank}: node-shared result buffers unavailable ({e})", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)   # every rank takes the same decision
        if int(flag.item()) == 0:
            gather_dest = "host"   # the landing through rank 0's host link instead
            for c in ctxs:
                c.close()
            ctxs = make_contexts()
    for c in ctxs:
        c.reserve(W, H, B)
    placement = None
    if gather_dest == "shm":
        # The shared result buffers are sized by need (the first batch + 25 %), so they are allocated by the first batch of
        # every context: run it here, locally (no collective inside), and let every rank take the same decision.  Where
        # /dev/shm has no room the library puts a buffer into a file under HESS_SHARE_DIR / TMPDIR instead (mapped and
        # registered the same way); only when that fails too do all ranks fall back to the landing through rank 0's link.
        ok = 1
        try:
            for c in ctxs:
                c.run_device(d_imgs.data_ptr(), B, H, W)
        except Exception as e:
            print(f"bench.py: rank {rank}: node-shared result buffers unavailable ({e})", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            gather_dest = "host"
            for c in ctxs:
                c.close()
            ctxs = make_contexts()
            for c in ctxs:
                c.reserve(W, H, B)
        else:
            mine = [hdist.SharedResultsReader(f"{shm_prefix}_{tok[0]}_r{rank}_c{j}").placement() for j in range(nctx)]
            allp = [None] * world
            tdist.all_gather_object(allp, mine, group=hdist.count_group())
            placement = allp
            if rank == 0:
                readers = {r: [hdist.SharedResultsReader(f"{shm_prefix}_{tok[0]}_r{r}_c{j}") for j in range(nctx)]
                           for r in range(1, world)}
                try:
                    st = os.statvfs("/dev/shm")
                    free_mb = st.f_bavail * st.f_frsize / 1e6
                except OSError:
                    free_mb = float("nan")
                tot = sum(p["bytes"] for ps in allp for p in ps)
                in_files = sum(p["bytes"] for ps in allp for p in ps if not p["desc"].startswith("/dev/shm/"))
                print(f"bench.py: node-shared result buffers: {tot / 1e6:.0f} MB for {world} rank(s) x {nctx} contexts "
                      f"({tot / 1e6 / max(1, world * nctx):.1f} MB each, sized by need); /dev/shm has {free_mb:.0f} MB free now; "
                      f"{in_files / 1e6:.0f} MB of them in files instead (no room in /dev/shm); per node: {world * nctx} streams, "
                      f"{world * nctx} copier threads", file=sys.stderr)
                for r, ps in enumerate(allp):
                    where = sorted({os.path.dirname(p["desc"]) for p in ps})
                    print(f"bench.py:   rank {r}: {sum(p['bytes'] for p in ps) / 1e6:.0f} MB in {', '.join(where)}", file=sys.stderr)
    if use_dist and rank == 0 and gather_dest != "shm":
        print(f"bench.py: gather destination of every rank: {gather_dest}", file=sys.stderr)
    landing = hdist.HostLanding() if gather_dest == "host" else None
    gathered = {}
    # N > 1: a step's exchange (wait for the context, counts to rank 0, grouped send/recv of the lists) runs on a helper
    # thread, in step order; the submitting thread only waits for the exchange of the context it is about to reuse,
    # pipelined_contexts steps later.  A rank that falls behind then uses up its own pipeline depth instead of stopping
    # every rank's submissions at every step.
    worker = hdist.GatherWorker(dev) if use_dist else None

    def finish(c):
        c.wait()
        counts = [c.count(b) for b in range(B)]
        if use_dist:
            keys, desc = hdist.device_feature_tensors(c, counts, dev)
            allc, gk, gd = hdist.gather_feature_lists(counts, keys, desc, dst=0, counts_to_dst_only=True)
            if rank == 0:
                if landing is not None:   # the other ranks' lists into pinned host memory: the step ends where N = 1 ends
                    gk, gd = landing.land(gk, gd, own_rank=0)
                gathered["counts"], gathered["keys"], gathered["desc"] = allc, gk, gd
                if readers is not None:   # ... or read in place where every rank's copier has put them
                    slot = ctxs.index(c)
                    dim = c.desc_dim()
                    gathered["host"] = {r: readers[r][slot].views(int(sum(allc[r])), dim) for r in readers}
        return counts

    stamps = [] if os.environ.get("HESS_BENCH_STAMPS") == "1" else None   # completion time of every step (diagnostics, stderr)

    host_submit = [0.0, 0]   # (diagnostics) seconds the submitting thread spent inside the submit call, calls

    # (tests only) "rank:step:code": that rank leaves with that exit code before submitting that step of a run_steps call
    # of at least that many steps -- a rank that dies mid-run, for the launcher's supervision (tests/test_bench_contract_gpu.py)
    test_exit = [int(x) for x in os.environ.get("HESS_BENCH_TEST_EXIT", "-1:0:0").split(":")]

    if worker is not None:
        def posted(c):
            return worker.post(finish, c)
        finished = hdist.GatherWorker.result
    else:
        def posted(c):
            return c
        finished = finish

    def run_steps(n, submit):
        """n steps, software-pipelined over the contexts; every step is submitted and finished inside."""
        counts = None
        inflight = []
        for i in range(n):
            c = ctxs[i % nctx]
            if len(inflight) == nctx:
                counts = finished(inflight.pop(0))
                if stamps is not None:
                    stamps.append(time.perf_counter())
            if rank == test_exit[0] and i == test_exit[1]:
                os._exit(test_exit[2])
            if stamps is not None:
                t_sub = time.perf_counter()
                submit(c)
                host_submit[0] += time.perf_counter() - t_sub
                host_submit[1] += 1
            else:
                submit(c)
            inflight.append(posted(c))
        while inflight:
            counts = finished(inflight.pop(0))
            if stamps is not None:
                stamps.append(time.perf_counter())
        return counts

    def submit_resident(c):
        c.submit_device(d_imgs.data_ptr(), B, H, W)

    def fence():
        if use_dist:
            tdist.barrier()
        torch.cuda.synchronize()

    # Every context has run a batch before the timed region, whatever --warmup says: hess_reserve's dry batch (a context's
    # first batch takes twice its steady time, hess_abi.hip prime()) and one real batch each here -- set-up, like the
    # reservation itself (round 4's driver run had two of seven contexts run their FIRST batch inside the 20 timed steps).
    if not use_dist or gather_dest != "shm":   # (the shm destination has run one batch per context already, see above)
        for c in ctxs:
            c.run_device(d_imgs.data_ptr(), B, H, W)
    counts = run_steps(max(args.warmup, 1), submit_resident)
    fence()
    if stamps is not None:
        del stamps[:]
        host_submit[0], host_submit[1] = 0.0, 0
    # (the interpreter's cyclic collector stays out of the timed steps: a full collection with torch imported takes
    # milliseconds, a fifth of the driver's 20-step region when it falls inside)
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    counts = run_steps(args.steps, submit_resident)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    if stamps is not None:
        gaps = [stamps[0] - t0] + [b - a for a, b in zip(stamps, stamps[1:])]
        print("bench.py: completion gaps of the timed steps (ms):", " ".join(f"{g * 1e3:.2f}" for g in gaps), file=sys.stderr)
        print(f"bench.py: the submitting thread spent {host_submit[0] / max(host_submit[1], 1) * 1e3:.3f} ms per step inside the submit call "
              f"({host_submit[1]} calls; {dt / args.steps * 1e3:.3f} ms per step in all)", file=sys.stderr)
        stamps = None
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    last = ctxs[(args.steps - 1) % nctx]          # context that ran the last timed step: its results are still there
    timed_k0, timed_d0 = last.fetch(0)            # image 0 of the timed run (parity_checked below)
    timed_keys = [last.fetch(b)[0] for b in range(B)]
    # N > 1: image 0 of every other rank out of the gathered lists of the last timed step (rank 0 checks them below)
    gathered_first = {}
    if use_dist and rank == 0 and gathered:
        for r in range(1, world):
            n0 = gathered["counts"][r][0]
            if "host" in gathered:   # the node-shared host buffers are what rank 0 would hand on: check those
                hk, hd = gathered["host"][r]
                gathered_first[r] = (hk[:n0].copy(), hd[:n0].copy())
            else:
                gathered_first[r] = (gathered["keys"][r][:n0].cpu().numpy().copy(),
                                     gathered["desc"][r][:n0].cpu().numpy().copy())

    # The contract's region holds the pipeline's fill and drain (the fences empty it on both sides): at the driver's 20
    # steps that is about 2 ms of 21.  The steady-state rate of the same pipeline is reported beside it, from a leg of
    # its own (single rank only: no collective in an extra leg).
    steady = None
    if not use_dist and args.steps < 150 and not args.no_steady:
        run_steps(nctx, submit_resident)
        fence()
        ts = time.perf_counter()
        run_steps(200, submit_resident)
        fence()
        steady = B * 200 * W * H / (time.perf_counter() - ts) / 1e6
    # Legs outside the timed region (rank 0 alone reports them; every rank runs the device ones to stay in step).
    # Roofline leg: per-kernel hipEvent durations are only meaningful when kernels of different streams do not
    # overlap, so the events are recorded in a separate single-stream leg of the same run (same batch, one context).
    prof, prof_mirror, roof_steps = None, None, 0
    if not args.no_profile:
        c = ctxs[0]
        roof_steps = max(3, min(args.steps, 10))
        c.profile_enable(True)
        c.profile_reset()
        for _ in range(roof_steps):
            c.run_device(d_imgs.data_ptr(), B, H, W)
        prof = c.profile()
        c.profile_enable(False)
        # For comparison, the same leg with the other form of the descriptor launch: descriptor_kernel<true, ..> also
        # stores the results into pinned host memory (what batches of one or two images use by default; alone on the
        # device it waits for PCIe), descriptor_kernel<false, ..> leaves them in HBM for the copier thread's DMA copy.
        saved = os.environ.get("HESS_DELIVERY")
        os.environ["HESS_DELIVERY"] = "dma" if _mirror_is_default(B) else "mirror"   # the other form
        cd = hessgpu_amd.HessContext(local_rank, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK,
                                     octave_num=args.octaves, **order_kw)
        if saved is None:
            del os.environ["HESS_DELIVERY"]
        else:
            os.environ["HESS_DELIVERY"] = saved
        cd.reserve(W, H, B)
        cd.run_device(d_imgs.data_ptr(), B, H, W)
        cd.profile_enable(True)
        cd.profile_reset()
        for _ in range(roof_steps):
            cd.run_device(d_imgs.data_ptr(), B, H, W)
        prof_mirror = cd.profile()
        cd.close()
    host = None
    if world == 1 and not use_dist and not args.no_host_leg:
        host = host_legs(ctxs, nctx, imgs, B, args, run_steps, fence, torch)
    for c in ctxs:
        c.close()
    ctxs = []
    cfg4 = None
    if world == 1 and not use_dist and not args.no_configs4:
        cfg4 = configs4_leg(local_rank, torch)
    real = None
    if world == 1 and not use_dist and not args.no_real_images:
        real = real_images_leg(local_rank, torch)
    match = None
    if world == 1 and not use_dist and not args.no_matcher:
        match = matcher_leg(local_rank, check=not args.no_cpu_baseline)
    api = None
    if world == 1 and not use_dist and not args.no_api_leg:
        api = api_legs(imgs[0], args.api_threads)

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
                "streams_and_copier_threads_per_node": world * nctx,
                "descriptor_order": {0: "interleaved", 1: "sequential", 2: "pixel"}.get(desc_order, str(desc_order)),
                "distinct_images_per_gpu": nd,
                "features_per_image_mean": round(float(np.mean(counts)), 1),
                "sharding": (f"images over {world} rank(s), exact-size RCCL send/recv of the feature lists to rank 0"
                             + (", landed in rank 0's pinned host memory" if landing is not None else
                                " (HBM); every rank's lists also in node-shared pinned host memory, mapped by rank 0" if gather_dest == "shm" else " (HBM)")
                             if use_dist else "single GPU"),
                "input": "u8 luminance resident in HBM; results delivered to host memory",
                **({"gather_dest": gather_dest,
                    "exchange": "helper thread per rank, in step order; counts to rank 0 only; the submitting thread waits for the "
                                "exchange of the context it reuses"} if use_dist else {}),
                **({"shared_result_buffers_mb": round(sum(p["bytes"] for ps in placement for p in ps) / 1e6, 1),
                    "shared_result_buffers_in_files_mb": round(sum(p["bytes"] for ps in placement for p in ps
                                                                   if not p["desc"].startswith("/dev/shm/")) / 1e6, 1)}
                   if placement else {}),
                "result_delivery": "copier thread: DMA copy of the exact byte count, no dependency on a kernel "
                                   "(batches of 1-2 images: stores of the descriptor kernel into pinned memory)",
            },
            "value_device_resident": round(value, 2),
            "value_definition": "driver contract: pixels resident in HBM when the timed region starts, results in host "
                                "memory when a step ends; value_host_to_host is the same from pinned host pixels (SURVEY 8d)",
        }
        if steady is not None:
            out["value_steady_state"] = round(steady, 2)
            out["value_steady_state_definition"] = ("the same pipeline over 200 steps after the timed region: `value` at "
                                                    f"{args.steps} steps includes the pipeline's fill and drain between the contract's fences")
        if bound is not None:
            out["config"]["cpu_binding"] = f"{len(bound)} CPUs local to the rank's GPU"
        if prof is not None:
            roofs = rooflines(prof, roof_steps, timed_keys, B, prof_mirror, _mirror_is_default(B), desc_order)
            ranked = sorted(roofs, key=lambda r: -r["ms_per_step"])
            if ranked:
                out["roofline"] = ranked[0]
                out["roofline"]["leg"] = f"{roof_steps} single-stream steps after the timed region (kernels do not overlap)"
            if len(ranked) > 1:
                out["roofline_secondary"] = ranked[1]
            out["kernel_ms_per_step"] = {k: round(v["ms"] / roof_steps, 4) for k, v in prof.items() if v["launches"] and k != "gauss_octave0"}
            # The whole path over the TIMED (pipelined) step, SURVEY 8(d): A_px = 139.7 B per u8 input pixel (every array of
            # the reference's layout written once and read once, octaves summed as 4/3) + the per-feature bytes of the
            # descriptor entry (footprint samples, record, keypoint + descriptor out).
            dsc = [r for r in roofs if "descriptor" in r["kernel"]]
            per_step = 139.7 * B * W * H + (dsc[0]["algorithmic_bytes_per_launch"] * dsc[0]["launches_per_step"] if dsc else 0.0)
            ach = per_step / (dt / args.steps) / 1e9
            out["roofline_whole_path"] = {
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_step": round(per_step, 1), "ms_per_step": round(dt / args.steps * 1e3, 4),
                "definition": "SURVEY 8(d): 139.7 B per input pixel + per-feature bytes, over ms_per_step of the timed region -- the "
                              "reference layout's bytes (arrays this build keeps in LDS or never materialises included): a speed "
                              "normalised to that layout, NOT HBM utilisation (the launches' own bytes: roofline.achieved)"}
            # the dominant kernel of the committed kernel trace (top row of the rocprofv3 statistics of the last profiled
            # round), priced with the bytes of that profiled run: a copy, so that the line and the trace name the same kernel
            top = _profile_json("kernel_stats_top.json")
            if top:   # (not measured by this run: a verbatim copy of the committed file, marked as such -- and as stale when it
                      #  was measured on other kernel sources than this run's)
                out["roofline_by_rocprof"] = dict(top, from_committed_profile=True)
            stale = [n for n in ("gauss_traffic.json", "descriptor_counters.json", "kernel_stats_top.json") if (_profile_json(n) or {}).get("stale")]
            if stale:
                out["committed_profiles_stale"] = {"files": stale, "effect": "valu / traffic entries derived from them are left off this line",
                                                   "reason": (_profile_json(stale[0]) or {}).get("stale_reason")}
        if host is not None:
            out.update(host)
        if api is not None:
            out.update(api)
        if cfg4 is not None:
            out["configs4"] = cfg4
        if real is not None:
            out["real_images"] = real
        if match is not None:
            out["matcher"] = match
        if not args.no_cpu_baseline:
            ok = parity_check(imgs[0], timed_k0, timed_d0, desc_order)
            for r, (gk, gd) in gathered_first.items():   # N > 1: image 0 of every rank, from the gathered lists
                ref_img = fixtures.synthetic_blobs(W, H, r * B)
                ok = ok and parity_check(ref_img, np.frombuffer(gk.tobytes(), dtype=_abi.KEYPOINT_DTYPE), gd, desc_order)
            out["parity_checked"] = ok
            out["parity"] = dict(_parity_detail)
            if use_dist:
                out["parity_checked_ranks"] = 1 + len(gathered_first)
            if world == 1:
                out["cpu_baseline"] = cpu_baseline(imgs[:min(nd, 4)])
        bad = bad_fractions(out)
        if bad:   # say so on the line itself rather than print an impossible number unmarked
            out["consistency_error"] = [f"{w} = {v}: not a fraction of its peak" for w, v in bad]
            print("bench.py: inconsistent roofline entries:", out["consistency_error"], file=sys.stderr)
        if json_fd is None:
            print(json.dumps(out), flush=True)
        else:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        worker.close()
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
    n = max(100, args.steps)   # (a leg of its own, outside the contract's timed region: long enough for the pipeline's steady state)
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


def api_legs(img0, nthreads):
    """The reference's own calling pattern through libsiftgpu.so (RunSIFT(w,h,data,GL_LUMINANCE,GL_UNSIGNED_BYTE) +
    GetFeatureVector per image): one instance, and one instance per host thread on the device -- apps/multithread.cpp
    -mem as a child process (this process no longer holds a context)."""
    import re
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "hessgpu_amd", "bin", "multithread")
    if not os.path.exists(exe):
        return {"value_siftgpu_api_1thread": None, "value_siftgpu_api_threads": None, "siftgpu_api": f"{exe} not built"}
    out = {}
    with tempfile.TemporaryDirectory() as td:
        pgm = os.path.join(td, "bench.pgm")
        with open(pgm, "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (img0.shape[1], img0.shape[0]))
            f.write(img0.tobytes())
        for key, k, reps in (("value_siftgpu_api_1thread", 1, 200), ("value_siftgpu_api_threads", nthreads, 150)):
            r = subprocess.run([exe, "-i", pgm, "-mem", "-n", str(reps), "-devices", "1", "-per-device", str(k),
                                "-topk", str(TOPK)], capture_output=True, text=True, timeout=300)
            m = re.search(r"MPIX: ([0-9.]+)", r.stdout)
            out[key] = float(m.group(1)) if (m and r.returncode == 0) else None
            if out[key] is None:
                print("bench.py: multithread failed:", r.stdout[-500:], r.stderr[-500:], file=sys.stderr)
    out["siftgpu_api"] = (f"libsiftgpu.so, RunSIFT(w,h,data,GL_LUMINANCE,GL_UNSIGNED_BYTE) + GetFeatureVector per 1080p image "
                          f"(pageable pixels in, caller's arrays out), -topk {TOPK}; threads leg: {nthreads} instances, "
                          "one per host thread (MultiThreadSIFT.cpp pattern)")
    return out


def configs4_leg(local_rank, torch):
    """BASELINE.json configs[4]: one 4096x4096 synthetic image, -maxd 4096 -topk 65536 -half (64-d descriptors),
    device-resident input, results to host memory; one context synchronously and three contexts pipelined, plus the
    roofline entry of its descriptor launch (single-stream hipEvents)."""
    import numpy as np

    import fixtures
    import hessgpu_amd
    from hessgpu_amd import _abi

    S = 4096
    img = fixtures.synthetic_blobs(S, S, 0)   # the generator scales its blob count with the area
    d = torch.from_numpy(img[None]).to(torch.device("cuda", local_rank))
    ctxs = [hessgpu_amd.HessContext(local_rank, tex_max_dim=4096, half_sift=1, truncate_method=_abi.TRUNC_TOPK,
                                    feature_count_threshold=65536) for _ in range(5)]
    for c in ctxs:
        c.reserve(S, S, 1)
        c.run_device(d.data_ptr(), 1, S, S)
    n = ctxs[0].count(0)
    keys = ctxs[0].fetch(0)[0]
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctxs[0].run_device(d.data_ptr(), 1, S, S)
    lat = (time.perf_counter() - t0) / reps
    def pipelined(n, steps=40):
        inflight = []
        t0 = time.perf_counter()
        for i in range(steps):
            c = ctxs[i % n]
            if len(inflight) == n:
                inflight.pop(0).wait()
            c.submit_device(d.data_ptr(), 1, S, S)
            inflight.append(c)
        while inflight:
            inflight.pop(0).wait()
        return (time.perf_counter() - t0) / steps

    pipelined(5, 10)
    dt = pipelined(3)
    dt5 = pipelined(5)   # (same call, 2 .. 6 contexts: 11.3 / 11.1 - 12.2 / 12.8 / 13.6 / 13.5 Gpix/s, tools/r06/cfg4_ctx.sh)
    # per-kernel split of the SUBMITTED form (what the pipelined figures above run: copier delivery, the descriptors in four
    # launches over quarters of the list), one context, single stream; beside it the descriptor launch of the synchronous
    # form (hess_run_device keeps the descriptor kernel's own host stores: 28 MB through the link inside the kernel)
    c = ctxs[0]
    c.profile_enable(True)
    c.profile_reset()
    for _ in range(5):
        c.submit_device(d.data_ptr(), 1, S, S)
        c.wait()
    prof = c.profile()
    c.profile_reset()
    for _ in range(3):
        c.run_device(d.data_ptr(), 1, S, S)
    prof_sync = c.profile()
    for c in ctxs:
        c.close()
    dk = prof["descriptor"]
    dur = dk["ms"] * 1e-3 / max(1, dk["launches"])
    s_oct = keys["s"].astype(np.float64) / (2.0 ** (keys["level"] // 3))
    fbytes = float(np.sum((15.0 * s_oct) ** 2 * 8.0 + 16.0 + 24.0 + 256.0))
    # (one large image delivered by the copier thread has its descriptors in four launches over quarters of its features)
    per_image = max(1, round(dk["launches"] / 5))
    fbytes /= per_image
    n_launch = int(round(n / per_image))
    # (this leg calls the synchronous run: the in-kernel mirror whatever the result size -- unless HESS_DELIVERY says otherwise;
    #  the three-context figure above comes from asynchronous submissions: copier thread, four launches per image)
    mirror = per_image == 1 and _mirror_is_default(1, result_bytes=0)
    return {
        "workload": "4096x4096 synthetic blobs, -maxd 4096 -topk 65536 -half (64-d descriptors) [configs[4]]",
        "features": n,
        "Mpix_per_s_one_context": round(S * S / lat / 1e6, 1), "ms_per_image_one_context": round(lat * 1e3, 3),
        "Mpix_per_s_three_contexts": round(S * S / dt / 1e6, 1), "ms_per_image_three_contexts": round(dt * 1e3, 3),
        "Mpix_per_s_five_contexts": round(S * S / dt5 / 1e6, 1), "ms_per_image_five_contexts": round(dt5 * 1e3, 3),
        "kernel_ms_per_image": {k: round(v["ms"] / 5, 4) for k, v in prof.items() if v["launches"]},
        "descriptor_ms_per_image_synchronous_form": round(prof_sync["descriptor"]["ms"] / 3, 4),
        "roofline_descriptor": {
            "bound": "hbm", "kernel": _desc_kernel_name(int(ctxs[0].params.descriptor_order), mirror) + " (half descriptors)", "achieved": round(fbytes / dur / 1e9, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fbytes / dur / 1e9 / HBM_PEAK_GBS, 4),
            "avg_launch_us": round(dur * 1e6, 2), "algorithmic_bytes_per_launch": round(fbytes, 1), "features_per_launch": n_launch,
            "launches_per_image": per_image,
        },
    }


def real_images_leg(local_rank, torch):
    """The reference's own data/ images (tests/golden/data, decoded here with PIL to u8 luminance -- decode is before the
    hot path) as device-resident batches, default parameters + -topk 4096 like the headline: three contexts
    pipelined for the rate, one context with per-kernel hipEvents for the split.  Every number on the headline line is
    measured on synthetic blobs, the densest case for the gradient planes (a descriptor footprint on 92 % of the 64 x 32
    tiles of levels 1-3; these photographs: 40 - 53 %)."""
    import numpy as np
    from PIL import Image

    import hessgpu_amd
    from hessgpu_amd import _abi

    data = os.path.join(ROOT, "tests", "golden", "data")

    def lum(name):
        return np.ascontiguousarray(np.asarray(Image.open(os.path.join(data, name)).convert("L")))

    sets = (("1600.jpg x 8 (2048x1536)", ["1600.jpg"] * 8),
            ("list640.txt: 640-1..5.jpg (640x480)", ["640-%d.jpg" % i for i in range(1, 6)]),
            ("listx.txt, its 800x600 images: 800-1..4.jpg", ["800-%d.jpg" % i for i in range(1, 5)]))
    out = {}
    for label, names in sets:
        try:
            imgs = np.stack([lum(n) for n in names])
        except Exception as e:   # (no decoder on this box: say so, the leg is not part of the contract)
            out[label] = {"error": str(e)}
            continue
        B, h, w = imgs.shape
        d = torch.from_numpy(imgs).to(torch.device("cuda", local_rank))
        ctxs = [hessgpu_amd.HessContext(local_rank, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK) for _ in range(3)]
        for c in ctxs:
            c.reserve(w, h, B)
            c.run_device(d.data_ptr(), B, h, w)
        counts = [ctxs[0].count(b) for b in range(B)]
        steps, inflight = 60, []
        t0 = time.perf_counter()
        for i in range(steps):
            c = ctxs[i % 3]
            if len(inflight) == 3:
                inflight.pop(0).wait()
            c.submit_device(d.data_ptr(), B, h, w)
            inflight.append(c)
        while inflight:
            inflight.pop(0).wait()
        dt = (time.perf_counter() - t0) / steps
        c = ctxs[0]
        c.profile_enable(True)
        c.profile_reset()
        for _ in range(5):
            c.run_device(d.data_ptr(), B, h, w)
        prof = c.profile()
        for c in ctxs:
            c.close()
        out[label] = {"images_per_step": B, "width": int(w & ~3), "height": int(h), "features_per_image_mean": round(float(np.mean(counts)), 1),
                      "Mpix_per_s_three_contexts": round(B * w * h / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 4),
                      "kernel_ms_per_step": {k: round(v["ms"] / 5, 4) for k, v in prof.items() if v["launches"] and k != "gauss_octave0"}}
    out["input"] = "u8 luminance (PIL convert('L') of the reference's JPEGs) resident in HBM; -topk 4096; results to host memory"
    return out


def matcher_leg(local_rank, check=True):
    """The descriptor matcher (hess_matcher_*): device time of one unguided mutual-best match (multiply
    kernel + merge launch, hipEvents inside the library) at 4096^2 and 8192^2 random byte descriptors, against the dense
    i8 matrix-core peak; results compared bit for bit with the oracle's matcher on a size the CPU finishes in a second."""
    import numpy as np

    from hessgpu_amd.matcher import Matcher

    rng = np.random.RandomState(0)
    out = {"bound": "mfma_i8", "unit": "TOP/s", "peak": 5000.0,
           "peak_source": "MI355X_MICROARCH.md: i8 MFMA = 2 x BF16 per clock, BF16 ~2.5 PFLOP/s dense; 1 MAC = 2 OP",
           "kernel": "match_mfma_kernel<true> (v_mfma_i32_32x32x32_i8 tiles, row / column folds as packed keys) + match_finish_kernel"}
    for n in (4096, 8192):
        d1 = (rng.rand(n, 128) * 45).astype(np.uint8)
        d2 = (rng.rand(n, 128) * 45).astype(np.uint8)
        m = Matcher(local_rank, max_sift=n)
        m.set_descriptors(0, d1)
        m.set_descriptors(1, d2)
        m.match(max_match=n)
        ts = []
        for _ in range(10):
            m.match(max_match=n)
            ts.append(m.last_ms())
        m.close()
        ms = float(np.median(ts))
        out[f"{n}x{n}"] = {"device_ms": round(ms, 4), "TMAC_per_s": round(n * n * 128 / ms / 1e9, 1)}
    ms = out["8192x8192"]["device_ms"]
    out["achieved"] = round(2 * 8192 * 8192 * 128 / ms / 1e9, 1)
    out["frac"] = round(out["achieved"] / out["peak"], 4)
    if not check:
        return out
    from oracle_lib import oracle_match  # the checker

    a = rng.randint(0, 256, size=(1500, 128)).astype(np.uint8)
    b = rng.randint(0, 256, size=(3100, 128)).astype(np.uint8)
    b[7] = a[2]; b[3098] = a[2]; a[1499] = a[2]   # ties across rows and columns
    m = Matcher(local_rank, max_sift=4096)
    m.set_descriptors(0, a)
    m.set_descriptors(1, b)
    got = m.match(max_match=4096)
    m.close()
    out["parity_checked"] = bool(np.array_equal(got, oracle_match(a, b, max_match=4096)))
    out["parity"] = "1500 x 3100 full-range byte descriptors with ties, matrix-core path, == oracle_match (bit for bit)"
    return out


def _valu_entry(rate, insts, unit_key, source):
    """Vector-issue entry: the kernel's instruction rate against the nominal peak and against the all-CU rate this
    chip sustains for plain v_fma_f32 (tools/micro/valu_peak.hip -> profiles/valu_peak.json): the clock under an
    all-CU vector load is below the nominal 2.4 GHz, so the two fractions differ."""
    e = {"bound": "valu", "achieved": round(rate, 1), "peak": VALU_PEAK_GINST, "unit": "Ginst/s",
         "frac": round(rate / VALU_PEAK_GINST, 4), unit_key: insts, "source": source}
    measured = _profile_value("valu_peak.json", "sustained_fma_ginst_all_cus")
    if measured:
        e["peak_measured"] = measured
        e["frac_of_measured"] = round(rate / measured, 4)
    return e


def _mirror_is_default(batch, result_bytes=None):
    """Whether a batch of this size is delivered by the descriptor kernel's own host stores (hess_copier.hip,
    choose_delivery): HESS_DELIVERY overrides, else batches up to two images whose results (keypoints + descriptors of
    the context's batch before) stay within 16 MB."""
    pref = os.environ.get("HESS_DELIVERY")
    if pref in ("mirror", "dma", "blit"):
        return pref == "mirror"
    if result_bytes is None:
        result_bytes = batch * 6000 * (24 + 128 * 4)   # the bench workload: about 5.6 k features per image, 128-d
    return batch <= 2 and result_bytes <= (16 << 20)


def _desc_kernel_name(order, mirror):
    """The descriptor launch as a rocprofv3 trace prints it: descriptor_pixel_kernel<host mirror> for the pixel order,
    descriptor_kernel<host mirror, sequential order> for the float orders."""
    m = "true" if mirror else "false"
    return f"descriptor_pixel_kernel<{m}>" if order == 2 else f"descriptor_kernel<{m}, {'true' if order == 1 else 'false'}>"


def rooflines(prof, steps, timed_keys, images, prof_other=None, mirror=False, order=2):
    """Roofline entries of the two heavy kernels from the single-stream profile leg (prof); prof_other: the same leg
    with the other form of the descriptor launch."""
    import numpy as np

    out = []
    g = prof["gauss"]
    if g["launches"]:
        # `achieved` / `frac`: the bytes these launches actually have to MOVE (what `traffic`, the PMC figure, compares
        # with) over their time.  SURVEY 8(d)'s accounting -- every array of the reference's layout written once and read
        # once: Gaussian level 4 W + 4 R, det-H 4 W, gradient/theta 8 W per level pixel, 1 B per input pixel -- counts two
        # arrays that never leave LDS in this build (the octave's top level, level 0 of octave 0: 8 B per pixel each); that
        # figure is a speed normalised to the reference's layout, not HBM utilisation, and is reported as a side field.
        layout = g["bytes"] + g.get("bytes_in_lds", 0.0)
        on_layout = layout / (g["ms"] * 1e-3) / 1e9
        moved = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        out.append({
            "bound": "hbm",
            "kernel": "gauss_kernel / gauss_pair_kernel / gauss_first_kernel (separable Gaussian + fused det-Hessian/gradient; one pyramid level of the batch per launch; levels 0 + 1 of octave 0 share a launch, the top level of an octave shares one with level 1 of the next)",
            "achieved": round(moved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(moved / HBM_PEAK_GBS, 4),
            "traffic": _profile_value("gauss_traffic.json", "hbm_bytes_per_launch"),
            "avg_launch_us": round(g["ms"] * 1e3 / g["launches"], 2),
            "algorithmic_bytes_per_launch": round(g["bytes"] / g["launches"], 1),
            "algorithmic_bytes": "the bytes the launches move: source level read once, produced level + det-H + gradient/theta written once (levels kept in LDS are not counted)",
            "reference_layout_bytes_per_launch": round(layout / g["launches"], 1),
            "achieved_on_reference_layout": round(on_layout, 1),
            "achieved_on_reference_layout_note": "SURVEY 8(d)'s layout figure (arrays kept in LDS counted as if moved): a speed normalised to the reference's layout, not HBM utilisation",
            "launches": g["launches"], "ms_per_step": round(g["ms"] / steps, 4),
        })
        mix = _profile_value("hbm_mix.json", "one_read_four_writes_gbs")
        if mix:  # what this chip sustains for a level launch's traffic mix (4 B read, 16 B written per pixel)
            out[-1]["achievable_for_mix"] = {"peak": mix, "unit": "GB/s", "frac": round(moved / mix, 4),
                                             "source": "tools/micro/hbm_mix_layout.hip, one read to four writes into planes that do not share "
                                                       "HBM channels (profiles/r05_experiments/hbm_mix_layout.txt; round 3's 4 637 GB/s was the "
                                                       "worst case of planes exactly 512 MiB apart); of the moved bytes"}
        g0 = prof.get("gauss_octave0")
        if g0 and g0["launches"]:
            # the launches that work on octave 0 (three quarters of the stage's bytes): large enough to be bound by
            # memory bandwidth; the rest of the stage is the dependent chain of small launches of the other octaves
            m0 = g0["bytes"] / (g0["ms"] * 1e-3) / 1e9
            out[-1]["octave0_launches"] = {
                "launches": g0["launches"], "avg_launch_us": round(g0["ms"] * 1e3 / g0["launches"], 2),
                "algorithmic_bytes_per_launch": round(g0["bytes"] / g0["launches"], 1),
                "share_of_stage_bytes": round(g0["bytes"] / g["bytes"], 3), "share_of_stage_time": round(g0["ms"] / g["ms"], 3),
                "achieved": round(m0, 1), "unit": "GB/s", "frac": round(m0 / HBM_PEAK_GBS, 4),
                # (the mix rate is a rate of bytes that move)
                "frac_of_mix": round(m0 / mix, 4) if mix else None}
        gi = _profile_value("gauss_traffic.json", "valu_insts_per_image")
        if gi:
            rate = gi * images * steps / (g["ms"] * 1e-3) / 1e9
            out[-1]["valu"] = _valu_entry(rate, gi, "instructions_per_image",
                                          "SQ_INSTS_VALU of the PMC pass in profiles/gauss_traffic.json x images of this run "
                                          "(about half are packed-FP32 or division/sqrt helper instructions that issue at half rate or less)")
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
        # (a batch of four or more images delivered by the copier thread has its descriptors in two launches, each over
        # half of the images: per-launch figures are per kernel launch, as a kernel trace counts them)
        per_step = max(1, round(d["launches"] / steps))
        fbytes_step = fbytes
        fbytes /= per_step
        nfeat = int(round(nfeat / per_step))
        dur = d["ms"] * 1e-3 / d["launches"]
        achieved = fbytes / dur / 1e9
        what = ("one wavefront per feature: one raster over the footprint, fixed-point sums" if order == 2 else
                "one wavefront per feature: rotated-grid histogram")
        names = {False: _desc_kernel_name(order, False) + f" ({what} + normalisation + packed "
                        "result stores in HBM; the copier thread's DMA copy takes them to the host)",
                 True: _desc_kernel_name(order, True) + f" ({what} + normalisation + result "
                       "stores incl. the pinned host mirror)"}
        e = {
            "bound": "hbm",
            "kernel": names[mirror],
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _profile_value("descriptor_counters.json", "hbm_bytes_per_launch"),
            "avg_launch_us": round(dur * 1e6, 2), "algorithmic_bytes_per_launch": round(fbytes, 1),
            "features_per_launch": nfeat, "launches": d["launches"], "launches_per_step": per_step,
            "ms_per_step": round(d["ms"] / steps, 4),
        }
        insts = _profile_value("descriptor_counters.json", "valu_insts_per_feature")
        if insts:
            e["valu"] = _valu_entry(insts * nfeat / dur / 1e9, insts, "instructions_per_feature",
                                    "SQ_INSTS_VALU of the PMC pass in profiles/descriptor_counters.json x features of this run")
        dd = (prof_other or {}).get("descriptor")
        if dd and dd["launches"]:
            ddur = dd["ms"] * 1e-3 / dd["launches"]
            dbytes = fbytes_step / max(1, round(dd["launches"] / steps))
            e["without_host_mirror" if mirror else "with_host_mirror"] = {
                "avg_launch_us": round(ddur * 1e6, 2), "achieved": round(dbytes / ddur / 1e9, 1), "unit": "GB/s",
                "launches_per_step": max(1, round(dd["launches"] / steps)),
                "kernel": _desc_kernel_name(order, not mirror),
                "note": "same launch on a context created with HESS_DELIVERY=" + ("dma" if mirror else "mirror") +
                        ": the <true> form also stores keypoints + descriptors into pinned host memory and, "
                        "alone on the device, waits for PCIe",
            }
        out.append(e)
    return out


_sources_sha16 = []


def _profile_json(name):
    """A committed summary under profiles/, if it exists.  Summaries of kernel profiles record the sources they were
    measured on (`kernel_sources_sha16`, hessgpu_amd/build.py sources_digest()); one measured on OTHER sources than the ones
    this run executes comes back with "stale": True -- numbers derived from it are then left off the line."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
    except Exception:
        return None
    if isinstance(d, dict) and "kernel_sources_sha16" in d:
        if not _sources_sha16:
            try:
                from hessgpu_amd import build
                _sources_sha16.append(build.sources_digest())
            except Exception:
                _sources_sha16.append(None)
        if _sources_sha16[0] and d["kernel_sources_sha16"] != _sources_sha16[0]:
            d = dict(d, stale=True, stale_reason=f"measured on kernel sources {d['kernel_sources_sha16']} (commit {d.get('commit', '?')}), "
                                                 f"this run executes {_sources_sha16[0]}")
    return d


def _profile_value(name, key):
    """A number from a committed PMC summary under profiles/, if one exists and was measured on the sources this run
    executes (chip constants -- valu_peak.json, hbm_mix.json -- carry no source hash and are always used)."""
    d = _profile_json(name) or {}
    return None if d.get("stale") else d.get(key)


def bad_fractions(node, path=""):
    """Every frac / frac_of_* on the line that is not in (0, 1]: a fraction of a peak above 1 is a bookkeeping error
    (round 3 published 15.94), never a measurement."""
    bad = []
    if isinstance(node, dict):
        for k, v in node.items():
            if k in ("frac", "frac_of_measured", "frac_of_mix"):
                if v is not None and not (0.0 < v <= 1.0):
                    bad.append((path + "/" + k, v))
            else:
                bad += bad_fractions(v, path + "/" + k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            bad += bad_fractions(v, f"{path}[{i}]")
    return bad


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


# unit-norm descriptors of the bench's 1080p image: 6e-6 measured.  (include/hess_abi.h lists the measured distance per
# config; tests/test_gpu_parity.py bounds all of them, 4096^2's 1.6e-5 included, by TOL_ORDER = 3e-5; north star 1e-4.)
PARITY_TOL_VS_REFERENCE_ORDER = 1e-5
PARITY_TOL_VS_FORMULA = 1e-6           # against the reference's formula in double precision (oracle, HESS_ORACLE_DESC_EXACT = 3)
_parity_detail = {}


def parity_check(img0, gk, gd, order=0):
    """Image 0 of the timed run against the CPU oracle on the same pixels:
      same_order       keypoints and descriptors bit for bit against the oracle's restatement of the SAME descriptor
                       summation order (hess_params.descriptor_order);
      reference_order  keypoints bit for bit, descriptors within PARITY_TOL_VS_REFERENCE_ORDER of the REFERENCE's sequential
                       float order (ProgramCU.cu:1723-1774);
      formula          descriptors against the reference's formula evaluated in double precision (what every float order
                       approximates; the sequential float order is itself ~6e-6 from it at 1080p)."""
    import numpy as np
    from hessgpu_amd import _abi
    from oracle_lib import OracleSession  # the checker

    res = {}
    for name, o_order in (("same_order", order), ("reference_order", _abi.DESC_ORDER_SEQUENTIAL), ("formula", 3)):
        o = OracleSession(threads=_host_cores(), keep_levels=False, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK,
                          descriptor_order=o_order)
        o.run(img0[None])
        ok, od = o.fetch(0)
        o.close()
        keys_equal = bool(len(ok) == len(gk) and ok.tobytes() == gk.tobytes())
        bitwise = bool(keys_equal and np.array_equal(od.view(np.uint32), gd.view(np.uint32)))
        diff = float(np.abs(od.astype(np.float64) - gd.astype(np.float64)).max()) if keys_equal and od.size else (0.0 if keys_equal else float("inf"))
        res[name] = (keys_equal, bitwise, diff, od)
    seq_vs_formula = float(np.abs(res["reference_order"][3].astype(np.float64) - res["formula"][3]).max()) if res["formula"][3].size else 0.0
    tol_formula = PARITY_TOL_VS_FORMULA if order == _abi.DESC_ORDER_PIXEL else PARITY_TOL_VS_REFERENCE_ORDER
    good = (res["same_order"][1] and res["reference_order"][0] and res["reference_order"][2] <= PARITY_TOL_VS_REFERENCE_ORDER
            and res["formula"][2] <= tol_formula)
    _parity_detail.update({
        "descriptor_order": {0: "interleaved", 1: "sequential (the reference's)", 2: "pixel raster, fixed point"}.get(order, str(order)),
        "bitwise_vs_oracle_in_the_same_order": res["same_order"][1],
        "keypoints_bitwise_vs_oracle_in_the_reference_order": res["reference_order"][0],
        "descriptor_max_abs_diff_vs_reference_order": max(_parity_detail.get("descriptor_max_abs_diff_vs_reference_order", 0.0), res["reference_order"][2]),
        "tolerance_vs_reference_order": PARITY_TOL_VS_REFERENCE_ORDER,
        "descriptor_max_abs_diff_vs_reference_formula_in_double": max(_parity_detail.get("descriptor_max_abs_diff_vs_reference_formula_in_double", 0.0), res["formula"][2]),
        "tolerance_vs_reference_formula": tol_formula,
        "reference_order_vs_its_own_formula_in_double": max(_parity_detail.get("reference_order_vs_its_own_formula_in_double", 0.0), seq_vs_formula),
    })
    return bool(good)


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
