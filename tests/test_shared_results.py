"""Node-shared result buffers (hess_share_results, include/hess_abi.h): the reader side on fabricated objects (CPU), the
producer side on the GPU -- one process reading its own context's buffers, and a second process reading them in place."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHM = "/dev/shm"


def _write_dir(name, gk, gd, kb, db):
    from hessgpu_amd.dist import SharedResultsReader
    hdr = np.zeros(1, dtype=SharedResultsReader._HDR)
    hdr["magic"], hdr["gen_keys"], hdr["gen_desc"], hdr["keys_bytes"], hdr["desc_bytes"] = 0x48455353, gk, gd, kb, db
    with open(os.path.join(SHM, name + ".h"), "r+b" if os.path.exists(os.path.join(SHM, name + ".h")) else "wb") as f:
        f.write(hdr.tobytes().ljust(4096, b"\0"))


@pytest.mark.skipif(not os.path.isdir(SHM), reason="no /dev/shm")
def test_reader_follows_the_directory_generations():
    from hessgpu_amd.dist import KEY_BYTES, SharedResultsReader
    name = f"hess_test_reader_{os.getpid()}"
    rng = np.random.RandomState(3)
    k1 = rng.randint(0, 256, (50, KEY_BYTES)).astype(np.uint8)
    d1 = rng.rand(50, 128).astype(np.float32)
    files = [name + ".h", name + ".k1", name + ".d1", name + ".k2", name + ".d4"]
    try:
        open(os.path.join(SHM, name + ".k1"), "wb").write(k1.tobytes().ljust(8192, b"\0"))
        open(os.path.join(SHM, name + ".d1"), "wb").write(d1.tobytes().ljust(32768, b"\0"))
        _write_dir(name, 1, 1, 8192, 32768)
        r = SharedResultsReader(name)
        keys, desc = r.views(50, 128)
        assert np.array_equal(keys, k1) and np.array_equal(desc.view(np.uint32), d1.view(np.uint32))
        keys, desc = r.views(7, 128)                       # a shorter batch in the same buffers
        assert keys.shape == (7, KEY_BYTES) and np.array_equal(desc, d1[:7])
        ek, ed = r.views(0, 128)
        assert ek.shape == (0, KEY_BYTES) and ed.shape == (0, 128)
        assert r.views(3, 0)[1] is None                    # descriptors off
        # the producer grew both buffers: new objects, new generations in the directory
        k2 = rng.randint(0, 256, (400, KEY_BYTES)).astype(np.uint8)
        d2 = rng.rand(400, 128).astype(np.float32)
        open(os.path.join(SHM, name + ".k2"), "wb").write(k2.tobytes())
        open(os.path.join(SHM, name + ".d4"), "wb").write(d2.tobytes())
        del keys, desc, ek, ed
        _write_dir(name, 2, 4, k2.nbytes, d2.nbytes)
        keys, desc = r.views(400, 128)
        assert np.array_equal(keys, k2) and np.array_equal(desc, d2)
        del keys, desc
        r.close()
        with pytest.raises(OSError):
            SharedResultsReader(name + "_missing")
    finally:
        for f in files:
            try:
                os.unlink(os.path.join(SHM, f))
            except OSError:
                pass


def _imgs(n, w=320, h=240):
    return np.stack([fixtures.synthetic_blobs(w, h, i) for i in range(n)])


def _expect(ctx, batch):
    ks = np.concatenate([np.frombuffer(ctx.fetch(b)[0].tobytes(), np.uint8).reshape(-1, 24) for b in range(batch)])
    ds = np.concatenate([ctx.fetch(b)[1] for b in range(batch)])
    return ks, ds


@pytest.mark.gpu
@pytest.mark.parametrize("delivery", ["dma", "mirror", "blit"])
def test_shared_buffers_hold_the_results_of_every_delivery_mode(gpu_ctx_factory, delivery, monkeypatch):
    from hessgpu_amd.dist import SharedResultsReader
    monkeypatch.setenv("HESS_DELIVERY", delivery)
    name = f"hess_test_{os.getpid()}_{delivery}"
    c = gpu_ctx_factory(truncate_method=3, feature_count_threshold=300)
    ref = gpu_ctx_factory(truncate_method=3, feature_count_threshold=300)
    c.share_results(name)
    imgs = _imgs(3)
    counts = c.run(imgs)
    assert counts == ref.run(imgs) and sum(counts) > 0
    gk, gd, kb, db = c.shared_results_info()
    assert gk >= 1 and gd >= 1 and kb >= sum(counts) * 24 and db >= sum(counts) * 128 * 4
    assert os.path.exists(f"{SHM}/{name}.h") and os.path.exists(f"{SHM}/{name}.k{gk}") and os.path.exists(f"{SHM}/{name}.d{gd}")
    r = SharedResultsReader(name)
    keys, desc = r.views(sum(counts), c.desc_dim())
    ek, ed = _expect(ref, 3)
    assert np.array_equal(keys, ek) and np.array_equal(desc.view(np.uint32), ed.view(np.uint32))
    # fetch through the C ABI reads the same buffers
    fk, fd = _expect(c, 3)
    assert np.array_equal(fk, ek) and np.array_equal(fd.view(np.uint32), ed.view(np.uint32))
    # a larger image: the buffers are reallocated under new generations, the old objects are gone
    big = _imgs(2, 640, 480)
    counts2 = c.run(big)
    assert counts2 == ref.run(big)
    gk2, gd2, _, _ = c.shared_results_info()
    del keys, desc
    keys, desc = r.views(sum(counts2), c.desc_dim())
    ek, ed = _expect(ref, 2)
    assert np.array_equal(keys, ek) and np.array_equal(desc.view(np.uint32), ed.view(np.uint32))
    if gk2 != gk:
        assert not os.path.exists(f"{SHM}/{name}.k{gk}")
    del keys, desc
    r.close()
    c.close()
    assert not [f for f in os.listdir(SHM) if f.startswith(name)]   # hess_destroy unlinks everything


@pytest.mark.gpu
def test_shared_buffers_are_sized_by_need_and_fall_back_to_files(gpu_ctx_factory, monkeypatch, tmp_path):
    """A batch the copier delivers: the shared buffers hold what the batches seen need (+ 25 %), not the worst case;
    and with no room in /dev/shm (forced) they are files under HESS_SHARE_DIR, found through the directory's paths."""
    from hessgpu_amd.dist import SharedResultsReader
    imgs = _imgs(4)
    ref = gpu_ctx_factory(truncate_method=3, feature_count_threshold=300)
    counts = ref.run(imgs)
    ek, ed = _expect(ref, 4)
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("HESS_SHARE_FORCE_FILE", "1")
            monkeypatch.setenv("HESS_SHARE_DIR", str(tmp_path))
        name = f"hess_test_need_{os.getpid()}_{int(forced)}"
        # (HESS_SHARE_FORCE_FILE is a test hook of the developer build)
        c = gpu_ctx_factory(dev_switches=forced, truncate_method=3, feature_count_threshold=300)
        c.share_results(name)
        assert c.run(imgs) == counts
        gk, gd, kb, db = c.shared_results_info()
        need_k, need_d = sum(counts) * 24, sum(counts) * 128 * 4
        assert need_k <= kb and need_d <= db
        if os.environ.get("HESS_DELIVERY") != "mirror":   # (the in-kernel mirror needs the worst case up front, by design)
            assert kb <= 1.3 * need_k + 8192 and db <= 1.3 * need_d + 8192   # 4 x 1200 records would be the worst case
        r = SharedResultsReader(name)
        where = r.placement()
        assert where["bytes"] == kb + db
        assert os.path.dirname(where["desc"]) == (str(tmp_path) if forced else SHM) and os.path.exists(where["desc"])
        keys, desc = r.views(sum(counts), 128)
        assert np.array_equal(keys, ek) and np.array_equal(desc.view(np.uint32), ed.view(np.uint32))
        del keys, desc
        r.close()
        c.close()
        assert not [f for f in os.listdir(SHM) if f.startswith(name)] and not list(tmp_path.iterdir())


@pytest.mark.gpu
def test_share_results_argument_errors(gpu_ctx_factory):
    from hessgpu_amd.session import HessError
    c = gpu_ctx_factory()
    with pytest.raises(HessError):
        c.share_results("a/b")
    with pytest.raises(HessError):
        c.shared_results_info()          # not shared yet
    c.share_results(f"hess_test_args_{os.getpid()}")
    with pytest.raises(HessError):
        c.share_results("again")
    assert c.run(_imgs(1)) and c.count(0) > 0


_PRODUCER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import fixtures, hessgpu_amd
from hessgpu_amd import _abi
c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=300)
c.share_results({name!r})
imgs = np.stack([fixtures.synthetic_blobs(320, 240, 10 + i) for i in range(4)])
counts = c.run(imgs)
print("COUNTS", " ".join(map(str, counts)), c.desc_dim(), flush=True)
sys.stdin.readline()          # the reader is done
c.close()
print("CLOSED", flush=True)
"""


@pytest.mark.gpu
def test_another_process_reads_the_results_in_place(gpu_ctx_factory):
    """The multi-rank shape on one GPU: a producer process runs a batch, this process maps its buffers."""
    from hessgpu_amd.dist import SharedResultsReader
    name = f"hess_test_2p_{os.getpid()}"
    p = subprocess.Popen([sys.executable, "-c", _PRODUCER.format(root=ROOT, name=name)], stdin=subprocess.PIPE,
                         stdout=subprocess.PIPE, text=True)
    try:
        line = ""
        while not line.startswith("COUNTS"):
            line = p.stdout.readline()
            assert line, "producer ended early"
        vals = list(map(int, line.split()[1:]))
        counts, dim = vals[:-1], vals[-1]
        ref = gpu_ctx_factory(truncate_method=3, feature_count_threshold=300)
        imgs = np.stack([fixtures.synthetic_blobs(320, 240, 10 + i) for i in range(4)])
        assert ref.run(imgs) == counts
        r = SharedResultsReader(name)
        keys, desc = r.views(sum(counts), dim)
        ek, ed = _expect(ref, 4)
        assert np.array_equal(keys, ek) and np.array_equal(desc.view(np.uint32), ed.view(np.uint32))
        del keys, desc
        r.close()
    finally:
        try:
            p.stdin.write("\n"); p.stdin.flush()
        except OSError:
            pass
        p.wait(timeout=120)
    assert p.returncode == 0
    assert not [f for f in os.listdir(SHM) if f.startswith(name)]
