"""Pins the oracle's elementary functions (oracle/hess_math_ref.h) against independent
implementations: numpy.float16 for the half conversions (exact), libm in float64 for
exp / atan2 / sin / cos (<= 2 ulp, the error bound CUDA documents for the functions they model)."""
import ctypes as C

import numpy as np

from oracle_lib import lib


def _ulp_err(got, ref64):
    got = got.astype(np.float64)
    ulp = np.spacing(np.abs(ref64).astype(np.float32)).astype(np.float64)
    return np.abs(got - ref64) / ulp


def test_half_to_float_all_65536_patterns():
    L = lib()
    h = np.arange(65536, dtype=np.uint16)
    ref = h.view(np.float16).astype(np.float32)
    got = np.array([L.hess_cpu_h2f(int(v)) for v in h], dtype=np.float32)
    ok = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert ok.all()


def test_float_to_half_round_to_nearest_even():
    L = lib()
    rng = np.random.RandomState(1)
    vals = np.concatenate([
        rng.randn(50000).astype(np.float32) * 0.05,           # the range of det-H responses
        rng.randn(5000).astype(np.float32) * 1e-6,            # half subnormals
        (rng.rand(5000).astype(np.float32) * 70000),          # up to overflow
        np.array([0.0, -0.0, 65504, 65519.99, 65520, 1e10, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25,
                  2.0 ** -14, 6.1e-5, np.inf, -np.inf], dtype=np.float32),
    ])
    # exact ties: halfway between consecutive halves
    h = rng.randint(0, 0x7bff, 5000).astype(np.uint16)
    lo, hi = h.view(np.float16).astype(np.float64), (h + 1).astype(np.uint16).view(np.float16).astype(np.float64)
    vals = np.concatenate([vals, ((lo + hi) / 2).astype(np.float32)])
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16).view(np.uint16)
    got = np.array([L.hess_cpu_f2h(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, ref)


def test_expf_within_2ulp():
    L = lib()
    x = np.concatenate([-np.random.RandomState(2).rand(40000) * 87.0, [0.0, -1e-9, -86.99]]).astype(np.float32)
    got = np.array([L.hess_cpu_expf(float(v)) for v in x], dtype=np.float32)
    assert _ulp_err(got, np.exp(x.astype(np.float64))).max() <= 2.0
    assert L.hess_cpu_expf(-100.0) == 0.0


def test_atan2f_within_2ulp():
    L = lib()
    rng = np.random.RandomState(3)
    y, x = rng.randn(40000).astype(np.float32), rng.randn(40000).astype(np.float32)
    got = np.array([L.hess_cpu_atan2f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert _ulp_err(got, ref).max() <= 2.0
    assert L.hess_cpu_atan2f(0.0, 0.0) == 0.0
    assert abs(L.hess_cpu_atan2f(0.0, -1.0) - np.pi) < 1e-6 and abs(L.hess_cpu_atan2f(1.0, 0.0) - np.pi / 2) < 1e-6
    assert np.abs(got).max() <= np.float32(np.pi)  # keeps floor(theta*5.7296) within 36 bins


def test_sincosf_abs_error():
    L = lib()
    a = (np.random.RandomState(4).rand(40000) * 2 * np.pi).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    gs, gc = np.zeros_like(a), np.zeros_like(a)
    for i, v in enumerate(a):
        L.hess_cpu_sincosf(float(v), C.byref(s), C.byref(c))
        gs[i], gc[i] = s.value, c.value
    # __sincosf is a fast intrinsic (abs error ~2^-21.4 in [-pi,pi]); this model is tighter
    assert np.abs(gs - np.sin(a.astype(np.float64))).max() < 2.0e-7
    assert np.abs(gc - np.cos(a.astype(np.float64))).max() < 2.0e-7


def test_u8_to_unit_three_instruction_form_is_the_ieee_quotient():
    """The HIP kernels convert u8 luminance with q0 = b * fl(1/255); e = fma(q0, -255, b); q = fma(e, fl(1/255), q0)
    (hess_devmath.h: dm_u8_unit) instead of the division the reference and the oracle write (GLTexImage.cpp:828).
    Exhaustive check that the two agree bit for bit (also for the 16-bit form with 65535); the fused multiply-adds are
    evaluated exactly in binary64 here (the products of these small integers fit) and rounded once."""
    for top in (255.0, 65535.0):
        b = np.arange(int(top) + 1, dtype=np.float32)
        r = np.float32(1.0) / np.float32(top)
        q0 = (b * r).astype(np.float32)
        e = (q0.astype(np.float64) * -top + b.astype(np.float64)).astype(np.float32)
        q = (e.astype(np.float64) * np.float64(r) + q0.astype(np.float64)).astype(np.float32)
        assert np.array_equal(q.view(np.uint32), (b / np.float32(top)).view(np.uint32))
