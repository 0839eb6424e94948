"""The copier thread's error paths: a copy that never completes or that ROCr reports
as failed ends the batch with HESS_ERR_DEVICE instead of hanging hess_wait or handing out garbage; the context then
delivers through the stream-copy fallback.  The fault switches are read once per process, so every case runs in a
child process (one at a time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path[:0] = [{root!r}, {root!r} + "/tests"]
import numpy as np
import fixtures, hessgpu_amd
from oracle_lib import OracleSession

pinned = {pinned}
expect = {expect!r}
lum = np.ascontiguousarray(fixtures.load_rgb("640-1.jpg")[..., 1])
batch = np.stack([lum, lum[::-1].copy(), lum[:, ::-1].copy(), lum])
o = OracleSession(threads=8, keep_levels=False)
want = o.run(batch)
g = hessgpu_amd.HessContext(0, dev_switches=True)   # the developer build: HESS_COPIER_FAULT / _ENGINE exist there only
keep = None
def run_once():
    global keep
    if pinned:
        import torch
        keep = torch.from_numpy(batch).pin_memory()
        g.submit_host(ptr=keep.data_ptr(), batch=4, height=lum.shape[0], width=lum.shape[1])
        g.wait()
        return [g.count(i) for i in range(4)]
    return g.run(batch)
if expect:
    try:
        run_once()
    except hessgpu_amd.HessError as e:
        assert e.code == -3 and expect in str(e), str(e)
        print("FIRST: error as expected:", e)
    else:
        raise SystemExit("the injected fault did not fail the batch")
    try:
        g.count(0)
    except hessgpu_amd.HessError:
        pass
    else:
        raise SystemExit("a failed batch left results behind")
got = run_once()
assert got == want, (got, want)
for i in range(4):
    gk, gd = g.fetch(i)
    ok, od = o.fetch(i)
    assert gk.tobytes() == ok.tobytes() and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), i
g.close()
print("CHILD OK")
"""


def _child(env_extra, pinned, expect):
    env = dict(os.environ)
    env.update(env_extra)
    env["HESS_DELIVERY"] = "dma"
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, pinned=pinned, expect=expect)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHILD OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_failed_result_copy_is_an_error_then_the_fallback_delivers():
    out = _child({"HESS_COPIER_FAULT": "error"}, False, "device->host copy of the results failed")
    assert "FIRST: error as expected" in out


def test_lost_result_copy_times_out_instead_of_hanging():
    out = _child({"HESS_COPIER_FAULT": "timeout"}, False, "did not complete in time")
    assert "FIRST: error as expected" in out


def test_failed_upload_never_runs_the_kernels_on_garbage():
    out = _child({"HESS_COPIER_FAULT": "error"}, True, "host->device upload of the pixels failed")
    assert "FIRST: error as expected" in out


def test_invalid_copy_engine_falls_back():
    _child({"HESS_COPIER_ENGINE": "0x80000000"}, False, "")


CHILD_POISONED = r"""
import sys
sys.path[:0] = [{root!r}, {root!r} + "/tests"]
import numpy as np
import fixtures, hessgpu_amd

lum = np.ascontiguousarray(fixtures.load_rgb("640-1.jpg")[..., 1])
batch = np.stack([lum, lum[::-1].copy(), lum[:, ::-1].copy(), lum])
g = hessgpu_amd.HessContext(0, dev_switches=True)   # the developer build: HESS_COPIER_FAULT / _ENGINE exist there only
try:
    g.run(batch)
except hessgpu_amd.HessError as e:
    assert e.code == -3 and "did not complete in time" in str(e), str(e)
else:
    raise SystemExit("the injected fault did not fail the batch")
for call in (lambda: g.run(batch), lambda: g.reserve(lum.shape[1], lum.shape[0], 4), lambda: g.run(batch[:1])):
    try:
        call()
    except hessgpu_amd.HessError as e:
        assert e.code == -3 and "poisoned" in str(e), str(e)
    else:
        raise SystemExit("a poisoned context accepted another run")
g.close()            # leaves the copy's buffers and signals alone (they may still be written); must not crash
g2 = hessgpu_amd.HessContext(0, dev_switches=True)   # a new context of the same process works
n = g2.run(batch)
assert min(n) > 100, n
g2.close()
print("CHILD OK")
"""


def test_a_really_lost_copy_poisons_the_context():
    """after a REAL timeout the copy may still be in flight: the context must neither reuse nor free its
    targets.  HESS_COPIER_FAULT=poisoned reports the first wait as a real timeout (the copy has in fact completed)."""
    env = dict(os.environ, HESS_COPIER_FAULT="poisoned", HESS_DELIVERY="dma")
    r = subprocess.run([sys.executable, "-c", CHILD_POISONED.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHILD OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
