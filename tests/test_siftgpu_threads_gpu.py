"""libsiftgpu.so under the reference's multi-threaded usage (MultiThreadSIFT.cpp:83-156: one instance per host thread,
RunSIFT concurrently): every result must be bit-identical to what a lone instance returns, whatever the other threads
are doing (other images, other sizes, a keypoint-list run, a parameter change).  (Written for a cross-instance batching
engine that was measured and not kept, profiles/r03_experiments/siftgpu_shared_engine.*; the tests are what stays.)"""
import ctypes as C
import threading

import numpy as np
import pytest

import fixtures
import siftgpu_lib

pytestmark = pytest.mark.gpu
LUM, U8 = siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE


def _solo(imgs, args=()):
    """Results of a single instance, nothing else running."""
    s = siftgpu_lib.SiftGPU(list(args))
    assert s.create_context() == 2
    out = []
    for im in imgs:
        assert s.run(im, LUM, U8) == 1
        k, d = s.features()
        out.append((k.tobytes(), d.tobytes()))
    s.close()
    return out


def _threads(images_per_thread, args=(), after=None):
    """One instance per thread, all created first, then RunSIFT concurrently."""
    n = len(images_per_thread)
    inst = [siftgpu_lib.SiftGPU(list(args)) for _ in range(n)]
    for s in inst:
        assert s.create_context() == 2
    results = [[] for _ in range(n)]
    errors = []
    start = threading.Barrier(n)

    def work(t):
        try:
            start.wait()
            for im in images_per_thread[t]:
                if inst[t].run(im, LUM, U8) != 1:
                    errors.append((t, "RunSIFT returned 0"))
                    return
                k, d = inst[t].features()
                results[t].append((k.tobytes(), d.tobytes()))
            if after:
                after(t, inst[t], results[t])
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(n)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not any(x.is_alive() for x in th), "a thread is stuck"
    for s in inst:
        s.close()
    assert not errors, errors
    return results


def test_eight_threads_get_the_lone_instance_results():
    pool = [fixtures.synthetic_blobs(640, 480, i) for i in range(12)]
    ref = _solo(pool, ["-topk", "500"])
    per_thread = [[pool[(3 * t + j) % 12] for j in range(9)] for t in range(8)]
    got = _threads(per_thread, ["-topk", "500"])
    for t in range(8):
        for j in range(9):
            assert got[t][j] == ref[(3 * t + j) % 12], (t, j)


def test_mixed_sizes_and_a_late_starter():
    small = [fixtures.synthetic_blobs(320, 240, i) for i in range(4)]
    large = [fixtures.synthetic_blobs(640, 480, 20 + i) for i in range(4)]
    ref_s, ref_l = _solo(small), _solo(large)
    per_thread = [small * 2, large * 2, small[::-1] * 2, large[::-1] * 2, small + large]
    got = _threads(per_thread)
    assert got[0] == ref_s * 2 and got[1] == ref_l * 2
    assert got[2] == ref_s[::-1] * 2 and got[3] == ref_l[::-1] * 2
    assert got[4] == ref_s + ref_l


def test_keypoint_run_and_params_change_beside_running_threads():
    """RunSIFT(num, keys, flag) on the current image, and an instance whose parameters change, while the other threads'
    instances go on."""
    imgs = [fixtures.synthetic_blobs(640, 480, 40 + i) for i in range(3)]
    ref = _solo(imgs)
    ref_half = _solo(imgs, ["-half"])
    L = siftgpu_lib.lib()
    L.siftgpu_run_keys.restype = C.c_int
    L.siftgpu_run_keys.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    extra = {}

    def after(t, s, res):
        if t == 0:   # describe the instance's own keypoints of the last image again: same descriptors
            k, d = s.features()
            assert L.siftgpu_run_keys(s.h, len(k), k.ctypes.data, 1) == 1
            k2, d2 = s.features()
            extra["keys_again"] = (k2.tobytes(), d2.tobytes(), k.tobytes(), d.tobytes())
        if t == 1:   # parameters change: new context
            s.parse(["-half"])
            out = []
            for im in imgs:
                assert s.run(im, LUM, U8) == 1
                k, d = s.features()
                out.append((k.tobytes(), d.tobytes()))
            extra["half"] = out

    got = _threads([imgs, imgs, imgs, imgs], after=after)
    for t in range(4):
        assert got[t] == ref
    # the same two calls on a lone instance
    s = siftgpu_lib.SiftGPU([])
    assert s.create_context() == 2 and s.run(imgs[-1], LUM, U8) == 1
    k, d = s.features()
    assert L.siftgpu_run_keys(s.h, len(k), k.ctypes.data, 1) == 1
    rk, rd = s.features()
    s.close()
    k2, d2, k0, d0 = extra["keys_again"]
    assert (k0, d0) == (k.tobytes(), d.tobytes()) and (k2, d2) == (rk.tobytes(), rd.tobytes())
    assert extra["half"] == ref_half
