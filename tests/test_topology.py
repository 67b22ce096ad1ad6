"""Topology-aware staging ("report which GPUs hang off which root complex"): avifgpu_init_devices reads each device's
PCI bus id and, from sysfs, its NUMA node and that node's CPUs; the device's workers pin themselves there and allocate the pinned
staging.  The sysfs half is plain file parsing and runs here against a fake tree; the device half runs on the GPU box."""
import ctypes
import os
import re

import pytest

import harness

pkg = harness.pkg


def _tree(root, bdf, node, node_cpus=None, local_cpus=None):
    d = root / "bus" / "pci" / "devices" / bdf
    d.mkdir(parents=True)
    (d / "numa_node").write_text(f"{node}\n")
    if local_cpus is not None:
        (d / "local_cpulist").write_text(local_cpus + "\n")
    if node_cpus is not None:
        n = root / "devices" / "system" / "node" / f"node{node}"
        n.mkdir(parents=True)
        (n / "cpulist").write_text(node_cpus + "\n")


def _probe(root, bdf):
    lib = pkg.load()
    node = ctypes.c_int32(-7)
    buf = ctypes.create_string_buffer(256)
    rc = lib.avifgpu_topology_probe(str(root).encode() if root is not None else None, bdf.encode() if bdf is not None else None,
                                    ctypes.byref(node), buf, len(buf))
    return rc, node.value, buf.value.decode()


def test_probe_reads_node_and_cpus_from_sysfs(tmp_path):
    # an 8-GPU MI355X node: four GPUs per socket, SMT siblings in the second half of the CPU numbering
    _tree(tmp_path, "0000:05:00.0", 0, node_cpus="0-63,128-191")
    _tree(tmp_path, "0000:c1:00.0", 1, node_cpus="64-127,192-255")
    assert _probe(tmp_path, "0000:05:00.0") == (128, 0, "0-63,128-191")
    assert _probe(tmp_path, "0000:C1:00.0") == (128, 1, "64-127,192-255")            # HIP prints hex digits either way; sysfs is lower case
    # a single-node host: node -1, the device's own local_cpulist is all there is
    _tree(tmp_path, "0000:03:00.0", -1, local_cpus="0-15")
    assert _probe(tmp_path, "0000:03:00.0") == (16, -1, "0-15")
    # ... or nothing at all: known device, no CPU list -> nothing to pin to
    _tree(tmp_path, "0000:04:00.0", -1)
    assert _probe(tmp_path, "0000:04:00.0") == (0, -1, "")


def test_two_socket_eight_device_node_gets_four_worker_sets_per_socket(tmp_path):
    """The tree of an 8-GPU OAM node (four MI355X behind each socket; bus numbers as rocm-smi shows them on such hosts): the
    placement avifgpu_init_devices would make -- per device its socket's CPUs, AVIFGPU_LANES workers pinned there, and one helper
    pool / pinned-allocation domain per socket: 2 domains, four worker sets in each."""
    lib = pkg.load()
    node0, node1 = "0-63,128-191", "64-127,192-255"
    bdfs = ["0000:05:00.0", "0000:15:00.0", "0000:65:00.0", "0000:75:00.0", "0000:85:00.0", "0000:95:00.0", "0000:e5:00.0", "0000:f5:00.0"]
    for i, b in enumerate(bdfs):
        d = tmp_path / "bus" / "pci" / "devices" / b
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{i // 4}\n")
    for n, cpus in ((0, node0), (1, node1)):
        nd = tmp_path / "devices" / "system" / "node" / f"node{n}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cpus + "\n")
    arr = (ctypes.c_char_p * 8)(*[b.encode() for b in bdfs])
    out = (pkg.DeviceInfo * 8)()
    lanes = int(os.environ.get("AVIFGPU_LANES", "2"))
    assert lib.avifgpu_topology_plan(str(tmp_path).encode(), arr, 8, out) == 2          # two NUMA domains
    per_node = {0: [], 1: []}
    for i in range(8):
        assert out[i].pci_bus_id.decode() == bdfs[i] and out[i].workers == lanes and out[i].workers_pinned == 1
        assert out[i].cpulist.decode() == (node0 if i < 4 else node1)
        per_node[out[i].numa_node].append(i)
    assert per_node == {0: [0, 1, 2, 3], 1: [4, 5, 6, 7]}                               # four worker sets per socket
    # one socket only: one domain; a device the tree does not know: error, nothing half-planned is used
    assert lib.avifgpu_topology_plan(str(tmp_path).encode(), arr, 4, out) == 1
    bad = (ctypes.c_char_p * 2)(bdfs[0].encode(), b"0000:aa:00.0")
    assert lib.avifgpu_topology_plan(str(tmp_path).encode(), bad, 2, out) == pkg.readErr
    assert lib.avifgpu_topology_plan(str(tmp_path).encode(), None, 2, out) == pkg.formatBadParameters


def test_probe_errors(tmp_path):
    _tree(tmp_path, "0000:05:00.0", 0, node_cpus="0-3,x")
    assert _probe(tmp_path, "0000:05:00.0")[0] == pkg.readErr                         # malformed list
    assert _probe(tmp_path, "0000:99:00.0")[0] == pkg.readErr                         # no such device in this tree
    assert _probe(tmp_path, None)[0] == pkg.formatBadParameters
    lib = pkg.load()
    info = pkg.DeviceInfo()
    assert lib.avifgpu_device_topology(0, ctypes.byref(info)) == pkg.formatBadParameters or lib.avifgpu_device_count() > 0


@pytest.mark.gpu
def test_bound_devices_report_their_place_in_the_host(tmp_path):
    gpu = pkg.AvifGpu(0)
    topo = gpu.topology()
    assert len(topo) == 1 and topo[0]["device"] == 0
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-7]", topo[0]["pci_bus_id"]), topo
    assert topo[0]["workers"] == int(os.environ.get("AVIFGPU_LANES", "2"))
    print("topology:", topo)
    # the same device under a sysfs tree that places it on node 1 with exactly the CPUs this process may run on: the workers pin
    # themselves there, and conversions still come out right (the pinned tile buffers are now allocated by those workers)
    cpus = sorted(os.sched_getaffinity(0))
    half = cpus[:max(1, len(cpus) // 2)]
    runs, start = [], half[0]                                   # "0-63,128-191": the kernel's own list format
    for a, b in zip(half, half[1:] + [None]):
        if b != a + 1:
            runs.append(f"{start}-{a}" if a != start else f"{a}")
            start = b
    cpulist = ",".join(runs)
    assert len(cpulist) < 250
    _tree(tmp_path, topo[0]["pci_bus_id"].lower(), 1, node_cpus=cpulist)
    old = {k: os.environ.get(k) for k in ("AVIFGPU_SYSFS_ROOT", "AVIFGPU_SLOTS")}
    try:
        os.environ["AVIFGPU_SYSFS_ROOT"] = str(tmp_path)
        os.environ["AVIFGPU_SLOTS"] = "3"                       # a changed knob re-binds the same device list
        gpu2 = pkg.AvifGpu(0)
        t2 = gpu2.topology()[0]
        assert t2["numa_node"] == 1 and t2["cpulist"] == cpulist and t2["workers_pinned"], t2
        d = pkg.WriteDesc(width=640, height=96, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                          chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601)
        src = harness.make_write_source(d, seed=5)
        want = harness.oracle_write(d, src)
        got = harness.gpu_write(gpu2, d, src, mem="host")
        assert harness.compare_write(d, want, got)["max_abs"] == 0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        pkg.AvifGpu(0)                                          # back to the default binding for the tests that follow
