"""Stage-by-stage comparison HIP vs oracle with printed diagnostics (developer tool, GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fixtures
import hessgpu_amd
from hessgpu_amd import _abi
from oracle_lib import OracleSession

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "640-1.jpg"
    kw = {}
    if name == "synth":
        img = fixtures.synthetic_blobs(1920, 1080, 0)
        kw = dict(truncate_method=3, feature_count_threshold=4096)
    else:
        img = fixtures.load_rgb(name)
    g = hessgpu_amd.HessContext(0, verbose=1, **kw)
    o = OracleSession(threads=8, **kw)
    g.keep_levels()
    t = time.time(); ng = g.run(img[None]); print("gpu run", ng, "%.1f ms" % ((time.time() - t) * 1e3), flush=True)
    t = time.time(); ng = g.run(img[None]); print("gpu run2", ng, "%.1f ms" % ((time.time() - t) * 1e3), g.timing().round(3), flush=True)
    no = o.run(img[None]); print("oracle", no, flush=True)
    print("geometry", g.geometry(), o.geometry())
    nlev = o.params.dog_level_num + 2
    bad = 0
    for oc in range(len(o.geometry())):
        for what, nm, rng in ((_abi.DBG_GAUSS, "gauss", range(nlev)), (_abi.DBG_DETH, "deth", range(nlev)), (_abi.DBG_GOT, "got", range(1, nlev - 1))):
            for l in rng:
                a, r = g.level(0, oc, l, what), o.level(0, oc, l, what)
                nd = int(np.sum(a.view(np.uint32) != r.view(np.uint32)))
                if nd:
                    bad += 1
                    idx = np.argwhere(a.view(np.uint32) != r.view(np.uint32))[:5]
                    print(f"DIFF {nm} oct {oc} lvl {l}: {nd} of {a.size} differ, max {np.nanmax(np.abs(a - r)):.3g}, first at {idx.tolist()}")
    print("stage planes with differences:", bad)
    gl, ol = g.rawlist(0), o.rawlist(0)
    print("list", len(gl), len(ol), "equal" if gl.tobytes() == ol.tobytes() else "DIFFER")
    if gl.tobytes() != ol.tobytes():
        m = min(len(gl), len(ol))
        for f in gl.dtype.names:
            d = np.flatnonzero(gl[f][:m] != ol[f][:m])
            if len(d): print("  field", f, len(d), "differ; first", d[:5], gl[f][d[:3]], ol[f][d[:3]])
    gk, gd = g.fetch(0); ok, od = o.fetch(0)
    print("features", len(gk), len(ok))
    if len(gk) == len(ok):
        for f in gk.dtype.names:
            d = np.flatnonzero(gk[f] != ok[f])
            if len(d): print("  key field", f, len(d), "differ; first", d[:5], gk[f][d[:3]], ok[f][d[:3]])
        dd = np.flatnonzero(np.any(gd.view(np.uint32) != od.view(np.uint32), axis=1))
        print("  descriptor rows differing:", len(dd), "max abs", float(np.nanmax(np.abs(gd - od))) if gd.size else 0.0)
        if len(dd): print("   first rows", dd[:10])
    print("DONE")

main()

def orient_debug():
    name = sys.argv[1] if len(sys.argv) > 1 else "640-1.jpg"
    img = fixtures.load_rgb(name)
    g = hessgpu_amd.HessContext(0)
    g.keep_levels(True); o = OracleSession(threads=8)
    g.run(img[None]); o.run(img[None])
    gk, _ = g.fetch(0); ok, _ = o.fetch(0)
    from collections import OrderedDict
    def grp(k):
        d = OrderedDict()
        for r in k:
            d.setdefault((int(r["level"]), float(r["x"]), float(r["y"])), []).append(float(r["o"]))
        return d
    a, b = grp(gk), grp(ok)
    print("locations", len(a), len(b))
    n = 0
    for key in b:
        if key not in a: print("missing on gpu", key); continue
        if a[key] != b[key]:
            print(key, "gpu", a[key], "oracle", b[key]); n += 1
            if n > 12: break
if os.environ.get("ORIENT_DEBUG"): orient_debug()
