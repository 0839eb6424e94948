#!/usr/bin/env python3
"""Lane occupancy of descriptor sampling schemes on the bench workload's features (CPU only: the features come from the
test oracle; nothing here is product code).

VERDICT r2, item 1b: "measure what fraction of live-iteration lanes could be filled if the 16 cells shared one raster
over the union box".  For every feature of image 0 of the bench workload (1920x1080 synthetic, top-K 4096) this
reproduces the sample geometry of ComputeDescriptor_Kernel (ProgramCU.cu:1692-1745: cell centres, boxes clamped to
[1.5, W-1.5], window test abs(nx), abs(ny) < 1) in double precision and counts, per scheme, the lane slots a
64-lane wavefront spends and how many of them carry a (cell, sample) pair that passes the window test:

  cells      the shipped scheme: lane = cell*4 + q, the four lanes of a cell take four consecutive samples of the cell's
             own box per iteration, iterations in which no lane hits are skipped by a wave-uniform test;
  union4     one raster over the union box of the 16 cells, 16 pixels per iteration, the four lanes of a pixel take the
             (at most 2 x 2) cells whose window contains it;
  union1     the same raster, 64 pixels per iteration and lane, each lane then loops over its (at most four) cells: the
             slots are (iteration, lane, cell pass) triples, a pass being skipped when no lane has a cell left.

In every scheme a cell's samples arrive in the reference's raster order (all boxes lie on the same half-integer lattice),
which is what keeps the per-bin sums bit-identical; what differs is how many slots are wasted.  The per-pair arithmetic
(window coordinates, exp, bilinear weights) cannot be shared between the cells of a pixel: the reference forms
nx = fma(c/spt, x - ptx, (s/spt) * (y - pty)) from each cell's own rounded centre, so the four values differ in their last
bits."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from hessgpu_amd import _abi
from oracle_lib import OracleSession  # test infrastructure: supplies the feature list only

W, H = 1920, 1080


def main():
    img = fixtures.synthetic_blobs(W, H, 0)
    o = OracleSession(threads=8, keep_levels=False, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
    o.run(img[None])
    keys, _ = o.fetch(0)
    o.close()
    tot = {k: np.zeros(3) for k in ("cells", "union4", "union1")}  # slots spent, slots hit, iterations (or passes)
    skipped = 0
    for k in keys:
        octv = int(k["level"]) // 3
        scale = float(2 ** octv)
        wo, ho = ((W >> octv) + 3) // 4 * 4, H >> octv
        x = (float(k["x"]) - 0.5) / scale + 0.5
        y = (float(k["y"]) - 0.5) / scale + 0.5
        s = float(k["s"]) / scale
        ang = 2 * np.pi - float(k["o"])          # the un-mirrored angle the kernel gets (PyramidCU.cpp:903)
        spt = abs(s * 3.0)
        c, sn = np.cos(ang), np.sin(ang)
        bsz = abs(c * spt) + abs(sn * spt)
        cells = []
        for iy in range(4):
            for ix in range(4):
                offx, offy = ix - 1.5, iy - 1.5
                ptx = c * spt * offx - sn * spt * offy + x
                pty = c * spt * offy + sn * spt * offx + y
                xmin = max(1.5, np.floor(ptx - bsz) + 0.5)
                ymin = max(1.5, np.floor(pty - bsz) + 0.5)
                xmax = min(wo - 1.5, np.floor(ptx + bsz) + 0.5)
                ymax = min(ho - 1.5, np.floor(pty + bsz) + 0.5)
                cells.append((ptx, pty, xmin, ymin, xmax, ymax))
        # ---- shipped scheme ----
        hits_per_cell = []
        nmax = 0
        for (ptx, pty, xmin, ymin, xmax, ymax) in cells:
            nxs = int(xmax - xmin) + 1 if xmax >= xmin else 0
            nys = int(ymax - ymin) + 1 if ymax >= ymin else 0
            n = nxs * nys
            nmax = max(nmax, n)
            if n == 0:
                hits_per_cell.append(np.zeros(0, bool))
                continue
            t = np.arange(n)
            px, py = xmin + t % nxs, ymin + t // nxs
            dx, dy = px - ptx, py - pty
            nx = (c * dx + sn * dy) / spt
            ny = (c * dy - sn * dx) / spt
            hits_per_cell.append((np.abs(nx) < 1) & (np.abs(ny) < 1))
        nit = (nmax + 3) // 4
        grid = np.zeros((nit, 16, 4), bool)
        for ci, hc in enumerate(hits_per_cell):
            pad = np.zeros(nit * 4, bool)
            pad[:len(hc)] = hc
            grid[:, ci, :] = pad.reshape(nit, 4)
        live = grid.reshape(nit, 64).any(axis=1)
        tot["cells"] += (live.sum() * 64, grid[live].sum(), live.sum())
        skipped += nit - live.sum()
        # ---- union raster ----
        uxmin = min(cl[2] for cl in cells); uymin = min(cl[3] for cl in cells)
        uxmax = max(cl[4] for cl in cells); uymax = max(cl[5] for cl in cells)
        nxs, nys = int(uxmax - uxmin) + 1, int(uymax - uymin) + 1
        t = np.arange(nxs * nys)
        px, py = uxmin + t % nxs, uymin + t // nxs
        npairs = np.zeros(len(t), int)
        for (ptx, pty, xmin, ymin, xmax, ymax) in cells:
            dx, dy = px - ptx, py - pty
            nx = (c * dx + sn * dy) / spt
            ny = (c * dy - sn * dx) / spt
            npairs += ((np.abs(nx) < 1) & (np.abs(ny) < 1) & (px >= xmin) & (px <= xmax) & (py >= ymin) & (py <= ymax))
        n16 = (len(t) + 15) // 16
        p16 = np.zeros(n16 * 16, int); p16[:len(t)] = npairs
        live16 = p16.reshape(n16, 16).any(axis=1)
        tot["union4"] += (live16.sum() * 64, p16.reshape(n16, 16)[live16].sum(), live16.sum())
        n64 = (len(t) + 63) // 64
        p64 = np.zeros(n64 * 64, int); p64[:len(t)] = npairs
        passes = p64.reshape(n64, 64).max(axis=1)  # cell passes per iteration = the busiest lane's cell count
        tot["union1"] += (passes.sum() * 64, p64.sum(), passes.sum())
    n = len(keys)
    print(f"{n} features of image 0 of the bench workload")
    for name, (slots, hit, its) in tot.items():
        print(f"  {name:7s}: {its / n:7.1f} iterations (passes) per feature, {slots / n:8.0f} lane slots, {hit / n:7.0f} of them with a "
              f"(cell, sample) pair inside its window = {hit / slots:.3f}")
    print(f"  cells: {skipped / n:.1f} all-miss iterations per feature skipped ({skipped / (skipped + tot['cells'][2]):.3f} of all)")


if __name__ == "__main__":
    main()
