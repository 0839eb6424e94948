"""Image-level sharding over ranks and the gather of the final feature lists.

The path shards trivially: every image is independent (the reference runs one SiftGPU instance
per device, TestWin/MultiThreadSIFT.cpp:231-244, or one TCP server process per GPU,
ServerSiftGPU.cpp:156-194).  Here: one process per GPU, image i of a batch goes to the rank that
owns the contiguous block containing i, each rank runs the whole path locally, and the only
exchange step is the gather of the variable-length feature lists to one rank -- an all_gather of
the per-image counts (a few integers that are already on the host: over a gloo side group when
enable_host_count_exchange() was called, else through the data group) followed by one grouped batch of
exact-size sends of keypoints and descriptors to the destination rank (torch.distributed: backend "nccl" =
RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

KEY_BYTES = 24  # sizeof(hess_keypoint) = sizeof(SiftGPU::SiftKeypoint)


def shard_range(n_items, rank, world):
    """Contiguous block partition of range(n_items): -> (first, last_exclusive) of `rank`."""
    base, rem = divmod(n_items, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


_count_group = {}  # data group (None = default) -> gloo side group for the count exchange


def enable_host_count_exchange(group=None):
    """Create a gloo side group for the per-image counts (collective: call on every rank, once).

    The counts are on the host already (hess_count); exchanging them over gloo keeps the GPU out of the
    control step.  Measured in bench.py on MI355X with three pipelined contexts: RCCL all_gather of the
    counts + event wait = -7 % throughput (the host waits for a tiny kernel queued behind three streams of
    long ones), gloo = no measurable cost."""
    ranks = dist.get_process_group_ranks(group) if group is not None else None
    _count_group[group] = dist.new_group(ranks=ranks, backend="gloo")


def count_group(group=None):
    """The gloo side group of `group` (None when enable_host_count_exchange was not called): host-side control
    messages of the job (object broadcasts, count exchange) travel there."""
    return _count_group.get(group)


def _exchange_counts(counts, dev, world, group, dst=None):
    """all_gather of the per-image counts -> list[world][n_local] of ints.  With `dst` (a rank inside `group`) and a
    host path for the counts (gloo), only dst collects them: the other ranks get their own row and None for the rest,
    and do not wait for anybody -- a sender needs no count but its own."""
    local = torch.tensor(counts, dtype=torch.int32)
    side = _count_group.get(group)
    if dev.type != "cuda" or side is not None:
        g = side if dev.type == "cuda" else group
        if dst is not None:
            rank = dist.get_rank(group)
            rows = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
            gdst = dist.get_global_rank(group, dst) if group is not None else dst
            dist.gather(local, rows, dst=gdst, group=g)
            if rank == dst:
                return [c.tolist() for c in rows]
            return [list(counts) if r == rank else None for r in range(world)]
        allc = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(allc, local, group=g)
        return [c.tolist() for c in allc]
    # no side group: through the device, with pinned buffers and asynchronous copies (a synchronous copy
    # from pageable memory stalls every stream of the device on ROCm)
    pin_local = local.pin_memory()
    dlocal = torch.empty(len(counts), dtype=torch.int32, device=dev)
    dlocal.copy_(pin_local, non_blocking=True)
    allc = torch.empty((world, len(counts)), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allc, dlocal, group=group)
    pin_all = torch.empty((world, len(counts)), dtype=torch.int32).pin_memory()
    pin_all.copy_(allc, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    done.synchronize()
    return pin_all.tolist()


def gather_feature_lists(counts, keys_u8, desc_f32, dst=0, group=None, counts_to_dst_only=False):
    """Gather per-image feature lists to rank `dst`, every rank sending exactly its own records.

    dst       destination rank INSIDE `group` (group-local numbering; with group=None that is the global rank).
              Use dist.get_group_rank(group, global_rank) to convert a global rank.
    counts    list[int], features per local image (same number of local images on every rank)
    keys_u8   uint8 tensor [sum(counts), 24]  (hess_keypoint records, local images back to back)
    desc_f32  float32 tensor [sum(counts), dim] or None when descriptors are off
    counts_to_dst_only   only dst learns every rank's counts (a gather instead of an all_gather where the counts
              travel over the host): the senders then wait for dst alone, never for each other
    Returns on dst: (all_counts [world][n_local], keys list[world] of uint8 [n_r,24],
    desc list[world] of float32 [n_r,dim] or None); on other ranks (all_counts, None, None) -- with
    counts_to_dst_only all_counts holds the rank's own row and None elsewhere.

    The counts are exchanged first (every rank then knows every block size), after which rank r sends its
    n_r x 24 and n_r x dim blocks to dst with one grouped batch of point-to-point operations (RCCL: one
    ncclGroup of send/recv pairs over xGMI; no padding to the largest rank, nothing sent for an empty rank;
    dst's own block is not copied at all).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = keys_u8.device
    all_counts = _exchange_counts(counts, dev, world, group, dst if counts_to_dst_only else None)
    totals = [int(sum(c)) if c is not None else -1 for c in all_counts]
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    dim = desc_f32.shape[1] if desc_f32 is not None else 0
    ops = []
    if rank == dst:
        keys = [keys_u8 if r == dst else torch.empty((totals[r], KEY_BYTES), dtype=torch.uint8, device=dev)
                for r in range(world)]
        desc = None
        if desc_f32 is not None:
            desc = [desc_f32 if r == dst else torch.empty((totals[r], dim), dtype=torch.float32, device=dev)
                    for r in range(world)]
        for r in range(world):
            if r != dst and totals[r] > 0:
                ops.append(dist.P2POp(dist.irecv, keys[r], peer(r), group))
                if desc is not None:
                    ops.append(dist.P2POp(dist.irecv, desc[r], peer(r), group))
    elif totals[rank] > 0:
        ops.append(dist.P2POp(dist.isend, keys_u8.contiguous(), peer(dst), group))
        if desc_f32 is not None:
            ops.append(dist.P2POp(dist.isend, desc_f32.contiguous(), peer(dst), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if dev.type == "cuda":
            # RCCL: wait() orders the CURRENT STREAM after the transfers, the host goes on.  The send buffers are the
            # context's own result arrays, which its next batch overwrites from another stream: the caller may only
            # resubmit once the transfers are really over, so the host waits here (callers that care run this
            # function on a helper thread: GatherWorker)
            torch.cuda.current_stream(dev).synchronize()
    if rank != dst:
        return all_counts, None, None
    return all_counts, keys, desc


class GatherWorker:
    """One helper thread that runs the per-step exchange (wait for the context, count exchange, grouped send/recv)
    away from the thread that submits batches: step i's exchange then overlaps the submission of steps i+1 .. and a
    rank that falls behind holds up its own pipeline depth, not every other rank's submitting thread at every step.

    Jobs run strictly in the order they were posted, so every rank issues its collectives in the same order as long
    as every rank posts the same sequence.  post(fn, *args) -> a ticket; result(ticket) waits for that job and returns
    fn's value or re-raises what it raised (and every later ticket then fails the same way: after a failed collective
    the ranks are out of step for good).  The thread selects `device` first (the current device is per thread)."""

    def __init__(self, device=None):
        import queue
        import threading

        self._q = queue.Queue()
        self._device = device
        self._failed = None
        self._t = threading.Thread(target=self._loop, name="hess-gather", daemon=True)
        self._t.start()

    def _loop(self):
        if self._device is not None and self._device.type == "cuda":
            torch.cuda.set_device(self._device)
        while True:
            job = self._q.get()
            if job is None:
                return
            fn, args, ticket = job
            try:
                if self._failed is not None:
                    raise RuntimeError("an earlier exchange of this worker failed") from self._failed
                ticket["value"] = fn(*args)
            except BaseException as e:   # handed to whoever asks for the result
                ticket["error"] = e
                if self._failed is None:
                    self._failed = e
            ticket["done"].set()

    def post(self, fn, *args):
        import threading

        ticket = {"done": threading.Event(), "value": None, "error": None}
        self._q.put((fn, args, ticket))
        return ticket

    @staticmethod
    def result(ticket, timeout=None):
        if not ticket["done"].wait(timeout):
            raise TimeoutError("exchange still running")
        if ticket["error"] is not None:
            raise ticket["error"]
        return ticket["value"]

    def close(self):
        self._q.put(None)
        self._t.join(timeout=30.0)


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_feature_tensors(ctx, counts, device):
    """Zero-copy views of the device-resident, already packed results of the last run
    (hess_device_results): -> (keys uint8 [total,24], desc float32 [total,dim] or None)."""
    kptr, dptr, total = ctx.device_results()
    assert total == sum(counts)
    dim = ctx.desc_dim()
    if total == 0:
        return (torch.zeros((0, KEY_BYTES), dtype=torch.uint8, device=device),
                torch.zeros((0, dim), dtype=torch.float32, device=device) if dim else None)
    keys = torch.as_tensor(_DevArray(kptr, (total, KEY_BYTES), "|u1"), device=device)
    desc = torch.as_tensor(_DevArray(dptr, (total, dim), "<f4"), device=device) if (dim and dptr) else None
    return keys, desc


def host_feature_tensors(session, counts):
    """Same packing from host results (any backend): used by the gloo CPU tests."""
    ks, ds = [], []
    for b in range(len(counts)):
        k, d = session.fetch(b)
        ks.append(np.frombuffer(k.tobytes(), dtype=np.uint8).reshape(-1, KEY_BYTES))
        ds.append(d)
    keys = torch.from_numpy(np.concatenate(ks).copy()) if ks else torch.zeros((0, KEY_BYTES), dtype=torch.uint8)
    dim = session.desc_dim()
    desc = torch.from_numpy(np.concatenate(ds).copy()) if dim else None
    return keys, desc


class HostLanding:
    """Pinned host buffers on the destination rank for the gathered lists of the OTHER ranks, so that a multi-rank
    step ends where a single-rank step ends: with every feature list of the global batch in host memory (the
    destination's own block is already there, delivered by its context).

    The copy is issued on a stream of its own after the host has waited for the receives, i.e. with no stream-order
    dependency on a kernel: the runtime then uses the DMA engine instead of a blit kernel (hess_copier.hip,
    kDeliverDma, for the same reason).  Buffers grow on demand and are reused from step to step."""

    def __init__(self):
        self._keys = None
        self._desc = None
        self._stream = None
        self.keys = []   # per source rank: uint8 [n_r, 24] host views (None for the destination's own block)
        self.desc = []

    @staticmethod
    def _buffer(old, nbytes, pin):
        if old is not None and old.numel() >= nbytes:
            return old
        t = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8)
        return t.pin_memory() if pin else t

    def land(self, keys_list, desc_list, own_rank):
        """keys_list / desc_list as returned by gather_feature_lists on the destination rank."""
        on_gpu = any(k.is_cuda for k in keys_list)
        dim = desc_list[0].shape[1] if desc_list is not None else 0
        nk = sum(k.shape[0] for r, k in enumerate(keys_list) if r != own_rank)
        if nk == 0:  # nothing from other ranks (a single rank, or empty lists): nothing to wait for or to copy
            self.keys = [None if r == own_rank else k.cpu() for r, k in enumerate(keys_list)]
            self.desc = [None if (r == own_rank or desc_list is None) else desc_list[r].cpu() for r in range(len(keys_list))]
            return self.keys, self.desc
        self._keys = self._buffer(self._keys, nk * KEY_BYTES, on_gpu)
        if dim:
            self._desc = self._buffer(self._desc, nk * dim * 4, on_gpu)
        if on_gpu:
            torch.cuda.current_stream().synchronize()      # the receives are done: nothing below follows a kernel
            if self._stream is None:
                self._stream = torch.cuda.Stream()
        self.keys, self.desc = [], []
        at = 0
        ctx = torch.cuda.stream(self._stream) if on_gpu else _NullContext()
        with ctx:
            for r, k in enumerate(keys_list):
                if r == own_rank:
                    self.keys.append(None)
                    self.desc.append(None)
                    continue
                n = k.shape[0]
                hk = self._keys[at * KEY_BYTES:(at + n) * KEY_BYTES].view(n, KEY_BYTES)
                hk.copy_(k, non_blocking=on_gpu)
                self.keys.append(hk)
                if dim:
                    hd = self._desc[at * dim * 4:(at + n) * dim * 4].view(torch.float32).view(n, dim)
                    hd.copy_(desc_list[r], non_blocking=on_gpu)
                    self.desc.append(hd)
                else:
                    self.desc.append(None)
                at += n
        if on_gpu:
            self._stream.synchronize()
        return self.keys, self.desc


class SharedResultsReader:
    """Reader side of hess_share_results (include/hess_abi.h): maps the result buffers another process of the node
    keeps in POSIX shared memory and hands out zero-copy views of a batch's feature lists.

    On one node every rank's results reach host memory over its own GPU's host link (the context's copier thread);
    the rank that collects the global batch reads them in place instead of pulling them through its own link a second
    time (HostLanding: 8 x 17.7 MB per step through one ~50 GB/s link is 2.8 ms against a 1 ms step at eight ranks).
    The producer's hess_wait / run must have returned before the views are read -- the count exchange that tells the
    reader how many records there are is that ordering -- and the views are valid until the producer submits the
    context's next batch."""

    _HDR = np.dtype([("magic", "<u4"), ("gen_keys", "<u4"), ("gen_desc", "<u4"), ("pad", "<u4"),
                     ("keys_bytes", "<u8"), ("desc_bytes", "<u8"), ("keys_path", "S1024"), ("desc_path", "S1024")])

    def __init__(self, name, shm_dir="/dev/shm"):
        import mmap
        import os
        self._mmap, self._os = mmap, os
        self._name, self._dir = name, shm_dir
        fd = os.open(os.path.join(shm_dir, name + ".h"), os.O_RDONLY)
        try:
            self._hdr_map = mmap.mmap(fd, 4096, prot=mmap.PROT_READ)
        finally:
            os.close(fd)
        self._hdr = np.frombuffer(self._hdr_map, dtype=self._HDR, count=1)
        if int(self._hdr["magic"][0]) != 0x48455353:
            raise RuntimeError(f"{name}.h is not a hess result directory")
        self._maps = {"k": (0, None), "d": (0, None)}   # which -> (generation, mmap)

    def _buffer(self, which, gen):
        have, m = self._maps[which]
        if have != gen:
            # (the old mapping is dropped, not closed: views of an earlier batch may still be alive and keep it mapped;
            # the producer publishes size and path before the generation number, so a new generation has its path)
            path = bytes(self._hdr["keys_path" if which == "k" else "desc_path"][0]).split(b"\0")[0].decode()
            if not path:   # a directory written without paths: the object's name under the shm directory
                path = self._os.path.join(self._dir, f"{self._name}.{which}{gen}")
            fd = self._os.open(path, self._os.O_RDONLY)
            try:
                m = self._mmap.mmap(fd, 0, prot=self._mmap.PROT_READ)
            finally:
                self._os.close(fd)
            self._maps[which] = (gen, m)
        return m

    def placement(self):
        """-> {"keys": path, "desc": path, "bytes": total}: where the producer's current buffers live."""
        h = self._hdr
        return {"keys": bytes(h["keys_path"][0]).split(b"\0")[0].decode(), "desc": bytes(h["desc_path"][0]).split(b"\0")[0].decode(),
                "bytes": int(h["keys_bytes"][0]) + int(h["desc_bytes"][0])}

    def views(self, total, dim):
        """-> (keys uint8 [total, 24], desc float32 [total, dim] or None): read-only numpy views of the first `total`
        records of the producer's last batch."""
        if total == 0:
            return np.zeros((0, KEY_BYTES), np.uint8), (np.zeros((0, dim), np.float32) if dim else None)
        gk, gd = int(self._hdr["gen_keys"][0]), int(self._hdr["gen_desc"][0])
        keys = np.frombuffer(self._buffer("k", gk), dtype=np.uint8, count=total * KEY_BYTES).reshape(total, KEY_BYTES)
        desc = None
        if dim:
            desc = np.frombuffer(self._buffer("d", gd), dtype=np.float32, count=total * dim).reshape(total, dim)
        return keys, desc

    def close(self):
        self._hdr = None
        for which, (_, m) in self._maps.items():
            if m is not None:
                try:
                    m.close()
                except BufferError:   # a view handed out earlier is still alive: the mapping goes with it
                    pass
        self._maps = {"k": (0, None), "d": (0, None)}
        try:
            self._hdr_map.close()
        except BufferError:
            pass


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
