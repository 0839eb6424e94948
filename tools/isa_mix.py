#!/usr/bin/env python3
"""Instruction mix of a gfx950 kernel from hipcc's assembly, per basic block.

    python tools/isa_mix.py hessgpu_amd/csrc/k_feature.hip descriptor_kernel [--flags "-fno-slp-vectorize"] [--blocks]

Compiles the file with the flags of hessgpu_amd/build.py (device code only, -S), finds the kernel whose mangled name
contains the given substring and prints: registers / LDS / occupancy as the compiler reports them, the instruction
count per class for the whole kernel, and (with --blocks) for every basic block that ends in a backward branch or
holds more than 20 instructions -- the loops.  Classes:
  valu      v_* except the ones below          trans   v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_sin/v_cos (quarter rate)
  div       v_div_scale/v_div_fmas/v_div_fixup (the IEEE division sequence around v_rcp)
  pk        v_pk_* (packed FP32: half issue rate on gfx950, tools/micro/README.md)
  dpp       VALU instructions with a DPP / row / quad_perm modifier (counted in valu as well)
  salu      s_* except waitcnt/branch/nop      lds     ds_*          vmem    global_/buffer_/flat_
  wait      s_waitcnt                          branch  s_cbranch/s_branch
Static counts: a loop body's count times its trip count is what the SQ_INSTS_VALU counter sees.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
DIV = ("v_div_scale", "v_div_fmas", "v_div_fixup")


def classify(op, rest):
    out = []
    if op.startswith("v_"):
        if op.startswith(TRANS):
            out.append("trans")
        elif op.startswith(DIV):
            out.append("div")
        elif op.startswith("v_pk_"):
            out.append("pk")
        elif op.startswith("v_mfma") or op.startswith("v_smfma"):
            out.append("mfma")
        else:
            out.append("valu")
        if "quad_perm" in rest or "row_" in rest or "wave_" in rest or "_dpp" in op:
            out.append("dpp")
    elif op.startswith("ds_"):
        out.append("lds")
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        out.append("vmem")
    elif op == "s_waitcnt":
        out.append("wait")
    elif op.startswith(("s_cbranch", "s_branch")):
        out.append("branch")
    elif op.startswith("s_"):
        out.append("salu")
    else:
        out.append("other")
    return out


def assemble(src, extra):
    from hessgpu_amd import build as hb

    flags = [f for f in hb.CXXFLAGS if f != "-fPIC"] + hb.FILE_FLAGS.get(os.path.basename(src), []) + extra
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hb.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, src], check=True,
                       stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_text(asm, name):
    m = None
    for mm in re.finditer(r"^(_Z\w*%s\w*):" % re.escape(name), asm, re.M):
        m = mm
        break
    if not m:
        raise SystemExit(f"kernel *{name}* not found")
    start = m.end()
    end = asm.index("s_endpgm", start)
    tail = asm[end:end + 6000]
    meta = {}
    for key in ("NumVgprs", "NumSgprs", "ScratchSize", "LDSByteSize", "Occupancy"):
        mm = re.search(r"; %s: (\d+)" % key, tail)
        if mm:
            meta[key] = int(mm.group(1))
    return m.group(1), asm[start:end], meta


def mix(body):
    blocks, cur, label = [], {}, "entry"
    order = []
    n = 0
    labels_seen = {"entry": 0}
    for line in body.splitlines():
        line = line.split(";")[0].strip()
        if not line or line.startswith("."):
            if line.startswith(".LBB") and line.endswith(":"):
                pass
            else:
                continue
        mm = re.match(r"^(\.LBB\w+):", line)
        if mm:
            blocks.append((label, cur, n))
            label, cur, n = mm.group(1), {}, 0
            labels_seen[label] = len(blocks)
            continue
        parts = line.split(None, 1)
        op, rest = parts[0], parts[1] if len(parts) > 1 else ""
        for c in classify(op, rest):
            cur[c] = cur.get(c, 0) + 1
        n += 1
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = rest.strip()
            cur.setdefault("_targets", []).append(tgt)
    blocks.append((label, cur, n))
    return blocks, labels_seen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel")
    ap.add_argument("--flags", default="")
    ap.add_argument("--blocks", action="store_true")
    a = ap.parse_args()
    asm = assemble(os.path.join(ROOT, a.source) if not os.path.isabs(a.source) else a.source, a.flags.split())
    name, body, meta = kernel_text(asm, a.kernel)
    blocks, seen = mix(body)
    total = {}
    for _, c, _n in blocks:
        for k, v in c.items():
            if not k.startswith("_"):
                total[k] = total.get(k, 0) + v
    keys = ["valu", "trans", "div", "pk", "mfma", "dpp", "salu", "lds", "vmem", "wait", "branch", "other"]
    print(f"kernel {name}")
    print("  " + "  ".join(f"{k}={v}" for k, v in meta.items()))
    print("  static total: " + "  ".join(f"{k}={total.get(k, 0)}" for k in keys if total.get(k)))
    if a.blocks:
        for i, (label, c, n) in enumerate(blocks):
            back = [t for t in c.get("_targets", []) if t in seen and seen[t] <= i]
            if n > 20 or back:
                tag = f" loop->{back[0]}" if back else ""
                print(f"  {label:12s} n={n:4d}{tag:16s} " + "  ".join(f"{k}={c.get(k, 0)}" for k in keys if c.get(k)))


if __name__ == "__main__":
    main()
