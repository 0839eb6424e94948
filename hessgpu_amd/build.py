"""Builds the product's native libraries in-tree with hipcc for gfx950 (MI355X).

  hessgpu_amd/libhessgpu.so   HIP kernels + the C ABI of include/hess_abi.h
  hessgpu_amd/libsiftgpu.so   SiftGPU C++ plugin surface on top of the C ABI (if its source exists)
  hessgpu_amd/dev/libhessgpu.so   the developer build (-DHESS_DEV_SWITCHES): same kernels, environment switches compiled in

hipcc cross-compiles without a GPU.  -ffp-contract=off: every fused multiply-add in the kernels
is an explicit fmaf() (see csrc/hess_devmath.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
            "-Wno-unused-function", f"--offload-arch={ARCH}"] + os.environ.get("HESS_EXTRA_FLAGS", "").split()

# ROCr itself, beside the HIP runtime: the copier thread hands its device->host copies straight to an SDMA engine
# (hsa_amd_memory_async_copy_on_engine, csrc/hess_copier.hip)
# -Bsymbolic: the library's own references to its functions bind inside it.  The product and the developer build are
# loaded side by side by the tests; without it the second one's internal calls would resolve to the first one's code.
LINK_LIBS = ["-lhsa-runtime64", "-lrt", "-Wl,-Bsymbolic"]
KERNEL_SOURCES = ["k_gauss.hip", "k_detect.hip", "k_feature.hip", "hess_plan.hip", "hess_schedule.hip", "hess_copier.hip",
                  "hess_shared.hip", "hess_abi.hip", "hess_match.hip"]
DEV_SWITCH_SOURCES = ("hess_abi.hip", "hess_copier.hip", "hess_shared.hip")  # the files that call dev_env()
# Per-file flags.  k_feature.hip: the SLP vectoriser turns pairs of FP32 operations into packed
# instructions (v_pk_add/mul/fma_f32), which on gfx950 issue at half rate (tools/micro/README.md) and need
# extra register moves to form the pairs: without it the descriptor kernel runs 10 % faster (1.37 -> 1.23 ms
# per 16x1080p step), results bit-identical.  The Gaussian kernel is 3 % faster WITH it, so it stays on there.
# hess_match.hip: MFMA results straight into vector registers (no v_accvgpr_read per accumulator before the folds): matcher
# + 1.6 % (213 -> 216 TMAC/s at 8192^2, same call, profiles/r06_experiments/matcher.txt).
FILE_FLAGS = {"k_feature.hip": ["-fno-slp-vectorize"], "hess_match.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def sources_digest():
    """First 16 hex digits of the SHA-256 over the sources of libhessgpu.so (csrc/*.hip, *.h, sorted by name) and the
    compiler flags: what a committed profile (profiles/*.json, `kernel_sources_sha16`) was measured on.  bench.py compares it
    with the sources it runs and marks numbers copied from a profile of other sources as stale."""
    import hashlib

    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode() + b"\0")
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
    h.update(" ".join(CXXFLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(FILE_FLAGS.items())]).encode())
    return h.hexdigest()[:16]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    return r


def _jpeg_flags():
    """libjpeg's header, if one is installed next to its shared library (siftgpu_api.cpp decodes JPEG files through the
    library looked up at run time, but needs the version's own struct layout at compile time).  -idirafter: the directory
    is searched LAST, so it adds jpeglib.h and nothing that a system directory provides."""
    for prefix in ("/usr", "/usr/local", "/opt/conda"):
        inc = os.path.join(prefix, "include")
        if os.path.exists(os.path.join(inc, "jpeglib.h")):
            libdirs = [d for d in (os.path.join(prefix, "lib"), os.path.join(prefix, "lib", "x86_64-linux-gnu"), os.path.join(prefix, "lib64"))
                       if any(f.startswith("libjpeg.so") for f in (os.listdir(d) if os.path.isdir(d) else []))]
            if libdirs:
                return ["-idirafter", inc, f'-DHESS_JPEG_LIBDIR="{libdirs[0]}"']
    return []


def build_variant(name, extra_flags, verbose=False):
    """Developer A/B builds: tools/_variants/<name>/libhessgpu.so compiled with extra flags (e.g. -DHESS_DESC_WAVES=6);
    select it at run time with HESS_LIB=<path>.  Not part of the product build."""
    vdir = os.path.join(HERE, "..", "tools", "_variants", name)
    os.makedirs(vdir, exist_ok=True)
    objs, jobs = [], []
    for src in KERNEL_SOURCES:
        o = os.path.join(vdir, src + ".o")
        objs.append(o)
        jobs.append([HIPCC] + CXXFLAGS + FILE_FLAGS.get(src, []) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", o])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_run, jobs))
    lib = os.path.join(vdir, "libhessgpu.so")
    _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + LINK_LIBS)
    for o in objs:
        os.remove(o)
    if verbose:
        print("built variant:", lib)
    return lib


def build_all(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "hess_abi.h"))
    objs, jobs = [], []
    for src in KERNEL_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or _newer([s] + headers, o):
            jobs.append([HIPCC] + CXXFLAGS + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_run, jobs))
    lib = os.path.join(HERE, "libhessgpu.so")
    if force or jobs or not os.path.exists(lib):
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + LINK_LIBS)
    built = [lib]
    # The developer build of the same sources: hessgpu_amd/dev/libhessgpu.so with -DHESS_DEV_SWITCHES (csrc/hess_ctx.h:
    # schedule A/B switches, fault injection and the other test hooks read from the environment).  Only the files that
    # read a switch are compiled again; the kernels are the product's objects.
    dev_dir = os.path.join(HERE, "dev")
    os.makedirs(dev_dir, exist_ok=True)
    dev_objs, dev_jobs = [], []
    for src, o in zip(KERNEL_SOURCES, objs):
        if src in DEV_SWITCH_SOURCES:
            s = os.path.join(CSRC, src)
            od = os.path.join(OBJ, src + ".dev.o")
            dev_objs.append(od)
            if force or _newer([s] + headers, od):
                dev_jobs.append([HIPCC] + CXXFLAGS + FILE_FLAGS.get(src, []) + ["-DHESS_DEV_SWITCHES", "-c", s, "-o", od])
        else:
            dev_objs.append(o)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_run, dev_jobs))
    dev_lib = os.path.join(dev_dir, "libhessgpu.so")
    if force or jobs or dev_jobs or not os.path.exists(dev_lib):
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", dev_lib] + dev_objs + LINK_LIBS)
    built.append(dev_lib)
    api_src = os.path.join(CSRC, "siftgpu_api.cpp")
    if os.path.exists(api_src):
        api = os.path.join(HERE, "libsiftgpu.so")
        api_hdr = os.path.join(HERE, "..", "include", "SiftGPU.h")
        if force or _newer([api_src, api_hdr] + headers, api) or jobs:
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", os.path.join(HERE, "..", "include")] + _jpeg_flags() +
                 [api_src, "-o", api, "-L", HERE, "-lhessgpu", "-ldl", "-pthread", "-Wl,-rpath,$ORIGIN"])
        built.append(api)
        apps_dir = os.path.join(HERE, "..", "apps")
        bindir = os.path.join(HERE, "bin")
        os.makedirs(bindir, exist_ok=True)
        for app in ("hess", "speed", "multithread"):
            src = os.path.join(apps_dir, app + ".cpp")
            exe = os.path.join(bindir, app)
            if os.path.exists(src) and (force or _newer([src, api_hdr, api], exe)):
                _run(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(HERE, "..", "include"), src, "-o", exe,
                      "-L", HERE, "-lsiftgpu", "-lhessgpu", "-pthread", "-Wl,-rpath,$ORIGIN/.."])
            if os.path.exists(exe):
                built.append(exe)
        # the C++ multi-GPU driver: C ABI + HIP runtime + RCCL (host code only)
        src = os.path.join(apps_dir, "multigpu.cpp")
        exe = os.path.join(bindir, "multigpu")
        if os.path.exists(src) and (force or _newer([src, os.path.join(HERE, "..", "include", "hess_abi.h"), lib], exe)):
            _run(["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(HERE, "..", "include"),
                  "-I", "/opt/rocm/include", src, "-o", exe, "-L", HERE, "-lhessgpu", "-L", "/opt/rocm/lib", "-lamdhip64",
                  "-lrccl", "-pthread", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"])
        if os.path.exists(exe):
            built.append(exe)
        # bench.py's step loop in C++ (C ABI + HIP runtime, host code only)
        src = os.path.join(apps_dir, "pipeline.cpp")
        exe = os.path.join(bindir, "pipeline")
        if os.path.exists(src) and (force or _newer([src, os.path.join(HERE, "..", "include", "hess_abi.h"), lib], exe)):
            _run(["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(HERE, "..", "include"),
                  "-I", "/opt/rocm/include", src, "-o", exe, "-L", HERE, "-lhessgpu", "-L", "/opt/rocm/lib", "-lamdhip64",
                  "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"])
        if os.path.exists(exe):
            built.append(exe)
    if verbose:
        print("built:", *built)
    return built


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # python -m hessgpu_amd.build --variant NAME [flags...]
        build_variant(sys.argv[2], sys.argv[3:], verbose=True)
    else:
        build_all(force="--force" in sys.argv, verbose=True)
