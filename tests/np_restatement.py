"""Independent NumPy / pure-Python restatement of the reference's Hessian + SIFT path.

Written from the textual description in SURVEY.md section 8(a) and the reference source, NOT from
oracle/hess_oracle.c: vectorised float64 arithmetic for the dense stages, plain Python loops for
the per-keypoint stages (small cases only).  It shares no code with the oracle or the product;
tests/test_oracle_vs_numpy.py uses it to cross-check the oracle within tolerances (the oracle
itself is bit-exact only against the HIP path).  Reference lines are cited per function.
"""
import math

import numpy as np

PI = 3.14159265358979323846


# ---- schedule: SiftGPU.cpp:482-563,1422-1425 ----------------------------------------------------
def schedule(dog=3, sigma0=1.6, sigman=0.5, first_octave_ds=0):
    k = 2.0 ** (1.0 / dog)
    dsigma0 = sigma0 * math.sqrt(k * k - 1.0)
    inter = [dsigma0 * k ** i for i in range(dog + 1)]          # blur taking level i to i+1
    level_sigma = [sigma0 * 2.0 ** (l / dog) for l in range(dog + 2)]
    sb = sigman / 2.0 ** first_octave_ds
    init = math.sqrt(sigma0 * sigma0 - sb * sb)
    return init, inter, level_sigma


# ---- ProgramCU::CreateFilterKernel, ProgramCU.cu:423-453 ------------------------------------------
def filter_taps(sigma, factor=4.0):
    sz = int(math.ceil(factor * sigma - 0.5))
    width = min(max(2 * sz + 1, 5), 33)
    sz = width // 2
    i = np.arange(-sz, sz + 1, dtype=np.float64)
    k = np.exp(-0.5 * i * i / (sigma * sigma))
    return k / k.sum()


# ---- input conversion, GLTexImage.cpp:802-862 ------------------------------------------------------
def luminance(img):
    a = np.asarray(img)
    if a.ndim == 2:
        if a.dtype == np.uint8:
            return a.astype(np.float64) / 255.0
        if a.dtype == np.uint16:
            return a.astype(np.float64) / 65535.0
        return a.astype(np.float64)
    r, g, b = (a[..., i].astype(np.float64) for i in range(3))
    if a.dtype == np.uint8:
        return (19595.0 * r + 38470.0 * g + 7471.0 * b) / (65535.0 * 255.0)
    if a.dtype == np.uint16:
        return (19595.0 * r + 38470.0 * g + 7471.0 * b) / (65535.0 * 65535.0)
    return 0.299 * r + 0.587 * g + 0.114 * b


# ---- FilterH / FilterV with replicated borders, ProgramCU.cu:117-231 -------------------------------
def gaussian(img, taps):
    r = len(taps) // 2
    p = np.pad(img, ((0, 0), (r, r)), mode="edge")
    h = sum(taps[i] * p[:, i:i + img.shape[1]] for i in range(len(taps)))
    p = np.pad(h, ((r, r), (0, 0)), mode="edge")
    return sum(taps[i] * p[i:i + img.shape[0], :] for i in range(len(taps)))


# ---- DownsampleKernel, ProgramCU.cu:312-326 + 4-aligned widths, PyramidCU.cpp:274-309 --------------
def downsample(src, dst_w_aligned, dst_h):
    sw = src.shape[1]
    cols = np.minimum(np.arange(dst_w_aligned) * 2, sw - 1)
    return src[np.arange(dst_h) * 2][:, cols]


def octave_geometry(w, h, octave_num=-1):
    w &= ~3
    nmax = max(1, int(math.floor(math.log(min(w, h)) / math.log(2.0))) - 3)
    n = octave_num if 1 <= octave_num < nmax else nmax
    out = []
    for _ in range(n):
        out.append((((w + 3) // 4) * 4, h))
        w >>= 1
        h >>= 1
    return out


def build_pyramid(lum, dog=3, octave_num=-1):
    init, inter, _ = schedule(dog)
    geo = octave_geometry(lum.shape[1], lum.shape[0], octave_num)
    lum = lum[:, :geo[0][0]]
    pyr = []
    for o, (wa, h) in enumerate(geo):
        if o == 0:
            levels = [gaussian(lum, filter_taps(init))]
        else:
            levels = [downsample(pyr[o - 1][dog], wa, h)]
        for l in range(1, dog + 2):
            levels.append(gaussian(levels[-1], filter_taps(inter[l - 1])))
        pyr.append(levels)
    return pyr


# ---- ComputeHessian_Kernel, ProgramCU.cu:523-595: 1-D index neighbours, zero outside the plane -----
def hessian_planes(g, sigma):
    h, w = g.shape
    flat = np.concatenate([np.zeros(w + 1), g.ravel(), np.zeros(w + 1)])
    base = w + 1
    n = h * w

    def nb(off):
        return flat[base + off: base + off + n].reshape(h, w)

    v11, v12, v13 = nb(-w - 1), nb(-w), nb(-w + 1)
    v21, v22, v23 = nb(-1), nb(0), nb(1)
    v31, v32, v33 = nb(w - 1), nb(w), nb(w + 1)
    lxx = v21 - 2.0 * v22 + v23
    lyy = v12 - 2.0 * v22 + v32
    lxy = (v13 - v11 + v31 - v33) * 0.25
    deth = (lxx * lyy - lxy * lxy) * sigma ** 4
    dx, dy = v23 - v21, v32 - v12
    grad = 0.5 * np.sqrt(dx * dx + dy * dy)
    theta = np.where(grad == 0.0, 0.0, np.arctan2(dy, dx))
    return deth, grad, theta


# ---- ComputeKEY_Kernel, ProgramCU.cu:702-882 (pure Python, one pixel) -----------------------------
def key_test(C, P, N, G, row, col, T, edge=10.0, subpixel=True):
    """-> None or (response, type, dx, dy, ds).  C/P/N: det-H of the level / previous / next."""
    thr0 = (0.8 if subpixel else 1.0) * T
    edge_t = (edge + 1.0) ** 2 / edge
    r = C[row, col]
    if abs(r) <= thr0:
        return None
    left, right = C[row, col - 1], C[row, col + 1]
    nmax, nmin = max(left, right), min(left, right)
    if nmin <= r <= nmax:
        return None
    state = {"nmax": nmax, "nmin": nmin}

    def triple(plane, rr):
        vals = [plane[rr, col - 1], plane[rr, col], plane[rr, col + 1]]
        if r > state["nmax"]:
            state["nmax"] = max([state["nmax"]] + vals)
            return not (r < state["nmax"] or r < 0)
        state["nmin"] = min([state["nmin"]] + vals)
        return not (r > state["nmin"] or r > 0)

    if not triple(C, row - 1) or not triple(C, row + 1):
        return None
    fxx = left + right - 2 * r
    fyy = C[row - 1, col] + C[row + 1, col] - 2 * r
    fxy = 0.25 * (C[row + 1, col + 1] + C[row - 1, col - 1] - C[row + 1, col - 1] - C[row - 1, col + 1])
    det = fxx * fyy - fxy * fxy
    if det <= 0 or (fxx + fyy) ** 2 > edge_t * det:
        return None
    for plane in (P, N):
        for rr in (row - 1, row, row + 1):
            if not triple(plane, rr):
                return None
    dx = dy = ds = 0.0
    resp = r
    if subpixel:
        fx = 0.5 * (right - left)
        fy = 0.5 * (C[row + 1, col] - C[row - 1, col])
        fs = 0.5 * (N[row, col] - P[row, col])
        fss = N[row, col] + P[row, col] - 2 * r
        fxs = 0.25 * (N[row, col + 1] + P[row, col - 1] - N[row, col - 1] - P[row, col + 1])
        fys = 0.25 * (N[row + 1, col] + P[row - 1, col] - N[row - 1, col] - P[row + 1, col])
        rows = [[fxx, fxy, fxs, -fx], [fxy, fyy, fys, -fy], [fxs, fys, fss, -fs]]
        rows = [rw if rw[0] > 0 else [-v for v in rw] for rw in rows]
        maxa = max(rw[0] for rw in rows)
        if maxa >= 1e-10:
            if maxa == rows[1][0]:
                rows[0], rows[1] = rows[1], rows[0]
            elif maxa == rows[2][0]:
                rows[0], rows[2] = rows[2], rows[0]
            a0 = [rows[0][0]] + [v / rows[0][0] for v in rows[0][1:]]
            a1 = [rows[1][0]] + [rows[1][j] - rows[1][0] * a0[j] for j in (1, 2, 3)]
            a2 = [rows[2][0]] + [rows[2][j] - rows[2][0] * a0[j] for j in (1, 2, 3)]
            if abs(a2[1]) > abs(a1[1]):
                a1, a2 = a2, a1
            if abs(a1[1]) >= 1e-10:
                a1 = a1[:2] + [a1[2] / a1[1], a1[3] / a1[1]]
                a2 = a2[:2] + [a2[2] - a2[1] * a1[2], a2[3] - a2[1] * a1[3]]
                if abs(a2[2]) >= 1e-10:
                    ds = a2[3] / a2[2]
                    dy = a1[3] - ds * a1[2]
                    dx = a0[3] - ds * a0[2] - dy * a0[1]
                    resp = r + 0.5 * (dx * fx + dy * fy + ds * fs)
                    if not (abs(resp) > T and abs(ds) < 1 and abs(dx) < 1 and abs(dy) < 1):
                        return None
    if resp < 0:
        typ = 2
    else:
        typ = 0 if (G[row, col - 1] - 2 * G[row, col] + G[row, col + 1]) > 0 else 1
    return resp, typ, dx, dy, ds


# ---- ComputeOrientation_Kernel, ProgramCU.cu:1221-1605 (multi-orientation branch) -----------------
def orientations(grad, theta, x, y, s, half=False, gaussian_factor=1.5, window_factor=2.0):
    """-> list of up to 4 rotations in bin units (rot in [0,36)), strongest first."""
    h, w = grad.shape
    gs = s * gaussian_factor
    win = abs(s) * gaussian_factor * window_factor
    factor = -0.5 / (gs * gs)
    xmin, ymin = max(1.5, math.floor(x - win) + 0.5), max(1.5, math.floor(y - win) + 0.5)
    xmax, ymax = min(w - 1.5, math.floor(x + win) + 0.5), min(h - 1.5, math.floor(y + win) + 0.5)
    vote = [0.0] * 37
    yy = ymin
    while yy <= ymax:
        xx = xmin
        while xx <= xmax:
            d2 = (xx - x) ** 2 + (yy - y) ** 2
            if d2 < win * win + 0.5:
                g, t = grad[int(yy), int(xx)], theta[int(yy), int(xx)]
                b = int(math.floor(t * 5.7295779513082320876798154814105))
                if b < 0:
                    b += 36
                vote[b] += g * math.exp(d2 * factor)
            xx += 1.0
        yy += 1.0
    for _ in range(6):
        old = vote[:36]
        for j in range(36):
            vote[j] = (old[j - 1] + old[j] + old[(j + 1) % 36]) / 3.0
    vote[36] = vote[0]
    if half:
        for i in range(18):
            vote[i] += vote[i + 18]
            vote[i + 18] = 0.0
    mx = max(vote[:36])
    peaks = []
    for i in range(36):
        pre, nxt = vote[i - 1] if i else vote[35], vote[i + 1]
        if vote[i] > 0.8 * mx and vote[i] > pre and vote[i] > nxt:
            di = 0.5 * (nxt - pre) / (2 * vote[i] - nxt - pre)
            peaks.append((vote[i], i + di + 0.5))
    peaks.sort(key=lambda p: -p[0])  # stable: equal weights keep bin order
    return [p[1] for p in peaks[:4]]


# ---- ComputeDescriptor_Kernel + NormalizeDescriptor_Kernel, ProgramCU.cu:1650-2054 -----------------
def descriptor(grad, theta, x, y, s, angle, half=False, window_factor=3.0):
    h, w = grad.shape
    spt = abs(s * window_factor)
    sn, cs = math.sin(angle), math.cos(angle)
    anglef = angle - 2 * PI if angle > PI else angle
    out = []
    for cell in range(16):
        ix, iy = cell & 3, cell >> 2
        ox, oy = ix - 1.5, iy - 1.5
        px = cs * spt * ox - sn * spt * oy + x
        py = cs * spt * oy + sn * spt * ox + y
        bsz = abs(cs * spt) + abs(sn * spt)
        xmin, ymin = max(1.5, math.floor(px - bsz) + 0.5), max(1.5, math.floor(py - bsz) + 0.5)
        xmax, ymax = min(w - 1.5, math.floor(px + bsz) + 0.5), min(h - 1.5, math.floor(py + bsz) + 0.5)
        des = [0.0] * 9
        yy = ymin
        while yy <= ymax:
            xx = xmin
            while xx <= xmax:
                dx, dy = xx - px, yy - py
                nx = (cs * dx + sn * dy) / spt
                ny = (cs * dy - sn * dx) / spt
                if abs(nx) < 1 and abs(ny) < 1:
                    g, t = grad[int(yy), int(xx)], theta[int(yy), int(xx)]
                    wgt = math.exp(-0.125 * ((nx + ox) ** 2 + (ny + oy) ** 2)) * (1 - abs(nx)) * (1 - abs(ny)) * g
                    th = (anglef - t) * 4.0 / PI
                    if th < 0:
                        th += 8.0
                    fo = math.floor(th)
                    if 0 <= fo < 8:
                        des[int(fo)] += (fo + 1 - th) * wgt
                        des[int(fo) + 1] += (th - fo) * wgt
                xx += 1.0
            yy += 1.0
        des[0] += des[8]
        out.extend([des[k] + des[k + 4] for k in range(4)] if half else des[:8])
    d = np.array(out)
    d = np.minimum(0.2, d / np.sqrt((d * d).sum()))
    return d / np.sqrt((d * d).sum())
