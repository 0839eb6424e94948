"""Bind a rank's host threads to the CPUs next to its GPU -- before any GPU call.

The reference has no notion of this (one process, one thread per device, TestWin/MultiThreadSIFT.cpp:231-244); with one
process per GPU the staging copies, the copier thread's DMA submissions and the pinned result buffers should sit on the
NUMA node the GPU's PCIe root hangs off, or every host<->device byte crosses the socket interconnect.

Plumbing only, sysfs only (no HIP call: the binding has to be in place before the runtime creates its threads and
pinned allocations):
  /sys/class/kfd/kfd/topology/nodes/<n>/properties   simd_count > 0 marks a GPU node; `domain`, `location_id`
                                                     (bus << 8 | device << 3 | function) give its PCI address
  /sys/bus/pci/devices/<dddd:bb:dd.f>/local_cpulist  the CPUs local to that device
GPU ordinal = position among the GPU nodes, after HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (lists of ordinals) if set.
Everything is best effort: anything unreadable leaves the affinity as it is and returns None.
"""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_pci_addresses(root="/"):
    """PCI addresses of the KFD GPU nodes, in node order."""
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    out = []
    for n in sorted((d for d in os.listdir(base) if d.isdigit()), key=int):
        props = {}
        try:
            with open(os.path.join(base, n, "properties")) as f:
                for line in f:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = int(kv[1])
        except OSError:
            # a GPU of the host that this process may not use (a container that was given some of the host's GPUs:
            # the read is refused): it is not a HIP device here either, so it does not take an ordinal
            continue
        if props.get("simd_count", 0) <= 0:
            continue
        loc, dom = props.get("location_id", 0), props.get("domain", 0)
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return out


def visible_ordinal(local_rank, env=None):
    """Physical GPU ordinal of HIP device `local_rank` under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES."""
    env = os.environ if env is None else env
    ordinal = local_rank
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):   # HIP's list indexes into ROCr's
        v = env.get(var, "").strip()
        if v:
            ids = [int(t) for t in v.split(",") if t.strip().lstrip("-").isdigit()]
            if ordinal >= len(ids):
                return None
            ordinal = ids[ordinal]
    return ordinal


def local_cpus(local_rank, root="/", env=None):
    """CPUs local to the GPU that HIP device `local_rank` maps to, or None."""
    try:
        ordinal = visible_ordinal(local_rank, env)
        gpus = gpu_pci_addresses(root)
        if ordinal is None or ordinal >= len(gpus):
            return None
        with open(os.path.join(root, "sys/bus/pci/devices", gpus[ordinal], "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError):
        return None


def bind_to_gpu(local_rank, root="/", env=None):
    """Restrict this process to the CPUs local to its GPU (intersected with the CPUs it may use at all).
    Returns the CPU list that was set, or None if nothing was changed."""
    cpus = local_cpus(local_rank, root, env)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = os.sched_getaffinity(0)
        want = sorted(allowed.intersection(cpus))
        if not want or len(want) == len(allowed):
            return None
        os.sched_setaffinity(0, want)
        return want
    except OSError:
        return None
