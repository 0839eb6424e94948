"""GPU: the asynchronous submit/wait pair with two contexts, the packed device-resident results,
and the RCCL gather path (backend "nccl", world size 1 on the one-GPU box)."""
import os
import socket

import numpy as np
import pytest

import fixtures

pytestmark = pytest.mark.gpu


def _imgs(n):
    return np.stack([fixtures.synthetic_blobs(320, 240, i) for i in range(n)])


def test_submit_wait_two_contexts_match_synchronous_run(gpu_ctx_factory):
    import torch

    a, b, ref = gpu_ctx_factory(), gpu_ctx_factory(), gpu_ctx_factory()
    i1, i2 = _imgs(3), _imgs(6)[3:]
    d1, d2 = torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()
    a.submit_device(d1.data_ptr(), 3, 240, 320)
    b.submit_device(d2.data_ptr(), 3, 240, 320)
    from hessgpu_amd.session import HessError
    with pytest.raises(HessError):  # one batch in flight per context
        a.submit_device(d1.data_ptr(), 3, 240, 320)
    a.wait()
    b.wait()
    for ctx, imgs in ((a, i1), (b, i2)):
        n = ref.run(imgs)
        assert [ctx.count(k) for k in range(3)] == n and sum(n) > 0
        for k in range(3):
            kk, dd = ctx.fetch(k)
            rk, rd = ref.fetch(k)
            assert kk.tobytes() == rk.tobytes() and dd.tobytes() == rd.tobytes()
    with pytest.raises(HessError):
        a.wait()  # nothing submitted


def test_device_results_are_packed_and_gather_over_nccl(gpu_ctx_factory):
    import torch
    import torch.distributed as dist

    from hessgpu_amd import dist as hdist

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        g = gpu_ctx_factory(truncate_method=3, feature_count_threshold=200)
        imgs = _imgs(4)
        counts = g.run(imgs)
        keys, desc = hdist.device_feature_tensors(g, counts, dev)
        assert keys.shape == (sum(counts), 24) and desc.shape == (sum(counts), 128)
        allc, gk, gd = hdist.gather_feature_lists(counts, keys, desc, dst=0)
        assert allc == [counts]
        hk = np.concatenate([np.frombuffer(g.fetch(b)[0].tobytes(), np.uint8).reshape(-1, 24) for b in range(4)])
        hd = np.concatenate([g.fetch(b)[1] for b in range(4)])
        assert np.array_equal(gk[0].cpu().numpy(), hk)
        assert np.array_equal(gd[0].cpu().numpy().view(np.uint32), hd.view(np.uint32))
        # same gather with the counts exchanged over the gloo side group (what bench.py does)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        hdist.enable_host_count_exchange()
        allc2, gk2, gd2 = hdist.gather_feature_lists(counts, keys, desc, dst=0)
        assert allc2 == [counts]
        assert np.array_equal(gk2[0].cpu().numpy(), hk)
        assert np.array_equal(gd2[0].cpu().numpy().view(np.uint32), hd.view(np.uint32))
    finally:
        hdist._count_group.clear()
        dist.destroy_process_group()


def test_one_context_per_host_thread(gpu_ctx_factory):
    """MultiThreadSIFT.cpp's usage: one instance per host thread, RunSIFT concurrently.  Contexts share
    nothing (own stream, buffers, parameters); results must equal the single-threaded ones."""
    import threading

    from oracle_lib import OracleSession

    names = fixtures.list640()[:4]
    imgs = [fixtures.load_rgb(n) for n in names]
    ref = []
    o = OracleSession(threads=8, keep_levels=False)
    for im in imgs:
        o.run(im[None])
        ref.append(tuple(a.tobytes() for a in o.fetch(0)))
    out, errs = {}, []

    def work(tid):
        try:
            g = gpu_ctx_factory()
            for rep in range(6):
                for k in range(len(imgs)):
                    i = (k + tid) % len(imgs)
                    g.run(imgs[i][None])
                    got = tuple(a.tobytes() for a in g.fetch(0))
                    if got != ref[i]:
                        errs.append((tid, rep, i))
            out[tid] = True
        except Exception as e:  # surfaces in the main thread's assert
            errs.append((tid, repr(e)))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs and len(out) == 4, errs[:3]
