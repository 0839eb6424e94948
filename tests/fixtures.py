"""Input fixtures: the reference's own data/ images (copied as data under tests/golden/data) and
the synthetic generator of SURVEY.md section 8(d) / BASELINE.json configs[1]."""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")


def load_rgb(name):
    """Decoded u8 RGB [H,W,3], the layout DevIL hands SetImageData (GLTexImage.cpp:1136-1142)."""
    from PIL import Image

    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(_DATA, name)).convert("RGB")))


def list640():
    with open(os.path.join(_DATA, "list640.txt")) as f:
        return f.read().split()


# ---------------------------------------------------------------------------------------------
# Synthetic "blobs" generator, SURVEY.md 8(d) config 2: deterministic, no <random> distributions.
_MASK = (1 << 64) - 1


class _XorShift64Star:
    def __init__(self, seed):
        self.s = seed & _MASK or 0x9E3779B97F4A7C15

    def u64(self):
        x = self.s
        x ^= x >> 12
        x ^= (x << 25) & _MASK
        x ^= x >> 27
        self.s = x
        return (x * 0x2545F4914F6CDD1D) & _MASK

    def unit(self):  # [0,1) with 53 bits
        return (self.u64() >> 11) / float(1 << 53)


def synthetic_blobs(width=1920, height=1080, index=0, cell=14, big_per_mpix=96.45):
    """u8 luminance [H,W] of BASELINE.json configs[1] ("1920x1080 synthetic blobs").

    Background 0.5; one isotropic Gaussian blob per `cell` x `cell` grid square, jittered inside the
    square, sigma 1.8*2^(1.6u), amplitude +-(0.3+0.4u) with a random sign (adjacent blobs of
    opposite sign give saddle points); 200 large blobs per 1920x1080 (sigma 8*2^(2.5u), amplitude
    +-(0.1+0.15u)) to populate the higher octaves; +-1/512 uniform dither; clipped, quantised.
    All randomness from one xorshift64* stream seeded 0x9E3779B97F4A7C15 ^ index.
    Yields about 16k raw extrema at 1080p with the default threshold (>= 3 x top-K 4096)."""
    rng = _XorShift64Star(0x9E3779B97F4A7C15 ^ index)
    img = np.full((height, width), 0.5, dtype=np.float64)

    def add(cx, cy, sb, amp):
        rad = 4.0 * sb
        x0, x1 = max(0, int(cx - rad)), min(width, int(cx + rad) + 1)
        y0, y1 = max(0, int(cy - rad)), min(height, int(cy + rad) + 1)
        if x0 >= x1 or y0 >= y1:
            return
        dx = np.arange(x0, x1, dtype=np.float64) - cx
        dy = np.arange(y0, y1, dtype=np.float64) - cy
        k = -1.0 / (2.0 * sb * sb)
        img[y0:y1, x0:x1] += amp * np.outer(np.exp(k * dy * dy), np.exp(k * dx * dx))

    for gy in range(0, height, cell):
        for gx in range(0, width, cell):
            cx = gx + cell * (0.25 + 0.5 * rng.unit())
            cy = gy + cell * (0.25 + 0.5 * rng.unit())
            sb = 1.8 * 2.0 ** (rng.unit() * 1.6)
            amp = (0.3 + 0.4 * rng.unit()) * (1.0 if (rng.u64() >> 40) & 1 else -1.0)
            add(cx, cy, sb, amp)
    nbig = int(round(big_per_mpix * width * height / 1e6))
    for j in range(nbig):
        cx, cy = rng.unit() * width, rng.unit() * height
        sb = 8.0 * 2.0 ** (rng.unit() * 2.5)
        amp = (0.1 + 0.15 * rng.unit()) * (1.0 if j & 1 else -1.0)
        add(cx, cy, sb, amp)
    # dither: xorshift64* words -> 8 noise bytes each
    n = width * height
    words = (n + 7) // 8
    noise = np.empty(words, dtype=np.uint64)
    s = rng.s
    for i in range(words):
        s ^= s >> 12
        s ^= (s << 25) & _MASK
        s ^= s >> 27
        noise[i] = (s * 0x2545F4914F6CDD1D) & _MASK
    d = noise.view(np.uint8)[:n].astype(np.float64).reshape(height, width)
    img += (d / 255.0 - 0.5) / 256.0
    return np.clip(np.rint(np.clip(img, 0.0, 1.0) * 255.0), 0, 255).astype(np.uint8)
