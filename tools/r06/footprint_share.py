#!/usr/bin/env python3
"""Share of the 64 x 32 tiles of the gradient/theta planes (levels 1-3 of every octave) that lie under the descriptor
footprint (its bounding square, the orientation disc inside it) of at least one selected feature -- what a pipeline that
computed the planes by need would have to compute.  Keypoints from the CPU oracle (the checker), -topk 4096.
  python tools/r06/footprint_share.py            (container; prints one line per image)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from oracle_lib import OracleSession
from PIL import Image


def share(img, topk=4096):
    o = OracleSession(threads=8, keep_levels=False, truncate_method=3, feature_count_threshold=topk)
    o.run(img[None])
    keys, _ = o.fetch(0)
    o.close()
    h, w = img.shape
    w &= ~3
    marked = total = 0
    px_marked = px_total = 0
    oct_of = keys["level"] // 3
    for oc in range(int(oct_of.max()) + 1 if len(keys) else 0):
        ow, oh = max(w >> oc, 1), max(h >> oc, 1)
        ow = (ow + 3) & ~3
        tx, ty = (ow + 63) // 64, (oh + 31) // 32
        for lv in range(3):
            m = np.zeros((ty, tx), bool)
            sel = keys[(oct_of == oc) & (keys["level"] % 3 == lv)]
            s_oct = sel["s"] / 2.0 ** oc
            r = 2.5 * 3.0 * s_oct * np.sqrt(2.0) + 2.0
            x, y = (sel["x"] - 0.5) / 2.0 ** oc + 0.5, (sel["y"] - 0.5) / 2.0 ** oc + 0.5
            for xi, yi, ri in zip(x, y, r):
                x0, x1 = int(max(0, xi - ri)) // 64, int(min(ow - 1, xi + ri)) // 64
                y0, y1 = int(max(0, yi - ri)) // 32, int(min(oh - 1, yi + ri)) // 32
                m[y0:y1 + 1, x0:x1 + 1] = True
            marked += int(m.sum()); total += m.size
            px_marked += int(m.sum()) * 64 * 32; px_total += ow * oh
    return len(keys), marked / max(total, 1), min(1.0, px_marked / max(px_total, 1))


def main():
    data = os.path.join(ROOT, "tests", "golden", "data")
    rows = [("synthetic blobs 1920x1080 #0 (bench)", fixtures.synthetic_blobs(1920, 1080, 0))]
    for n in ("1600.jpg", "640-1.jpg", "640-3.jpg", "800-1.jpg", "800-4.jpg"):
        rows.append((n, np.ascontiguousarray(np.asarray(Image.open(os.path.join(data, n)).convert("L")))))
    for name, img in rows:
        n, f, fp = share(img)
        print(f"{name:40s} {img.shape[1]}x{img.shape[0]}  features {n:6d}  tiles under a footprint {f:5.2f}  (pixel share {fp:4.2f})")


if __name__ == "__main__":
    main()
