"""hessgpu_amd/numa.py: sysfs parsing on a synthetic tree (no GPU needed)."""
import os

from hessgpu_amd import numa


def _tree(tmp_path, gpus):
    """gpus: list of (bus, device, function, cpulist); node 0 is a CPU node."""
    base = tmp_path / "sys/class/kfd/kfd/topology/nodes"
    (base / "0").mkdir(parents=True)
    (base / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (bus, dev, fn, cpus) in enumerate(gpus, start=1):
        (base / str(i)).mkdir()
        loc = (bus << 8) | (dev << 3) | fn
        (base / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {loc}\ndomain 0\n")
        d = tmp_path / "sys/bus/pci/devices" / ("0000:%02x:%02x.%x" % (bus, dev, fn))
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpus + "\n")
    return str(tmp_path)


def test_parse_cpulist():
    assert numa.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert numa.parse_cpulist("") == []


def test_local_cpus_follow_the_pci_address_and_the_visible_list(tmp_path):
    root = _tree(tmp_path, [(0x05, 0, 0, "0-7"), (0x26, 0, 0, "8-15"), (0xc5, 0, 0, "64-71,192-199")])
    assert numa.gpu_pci_addresses(root) == ["0000:05:00.0", "0000:26:00.0", "0000:c5:00.0"]
    assert numa.local_cpus(0, root, env={}) == list(range(0, 8))
    assert numa.local_cpus(2, root, env={}) == list(range(64, 72)) + list(range(192, 200))
    assert numa.local_cpus(0, root, env={"HIP_VISIBLE_DEVICES": "1"}) == list(range(8, 16))
    assert numa.local_cpus(1, root, env={"ROCR_VISIBLE_DEVICES": "2,0"}) == list(range(0, 8))
    assert numa.local_cpus(3, root, env={}) is None and numa.local_cpus(1, root, env={"HIP_VISIBLE_DEVICES": "0"}) is None
    assert numa.local_cpus(0, str(tmp_path / "nothing"), env={}) is None


def test_bind_is_a_no_op_without_topology(tmp_path):
    before = os.sched_getaffinity(0)
    assert numa.bind_to_gpu(0, str(tmp_path / "nothing"), env={}) is None
    assert os.sched_getaffinity(0) == before


def test_gpus_of_the_host_that_the_process_may_not_read_take_no_ordinal(tmp_path):
    """A container that was given one of the host's GPUs sees all KFD nodes, but reading the others' properties is
    refused (seen on the GPU pool): HIP device 0 is the one readable GPU node."""
    import shutil
    root = _tree(tmp_path, [(0x05, 0, 0, "0-7"), (0x26, 0, 0, "8-15"), (0xc5, 0, 0, "64-71")])
    base = tmp_path / "sys/class/kfd/kfd/topology/nodes"
    for n in ("1", "3"):   # unreadable: a directory in place of the file raises an OSError on open, like EPERM does on read
        os.unlink(base / n / "properties")
        (base / n / "properties").mkdir()
    assert numa.gpu_pci_addresses(root) == ["0000:26:00.0"]
    assert numa.local_cpus(0, root, env={}) == list(range(8, 16))
    assert numa.local_cpus(1, root, env={}) is None
    shutil.rmtree(base / "1" / "properties")
