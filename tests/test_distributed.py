"""N>1 path on CPU: two ranks over gloo exercise the sharding + gather + max-over-ranks plumbing that
bench.py / decode_batch use with RCCL on the GPUs (there is no data-path collective to test: images
are independent)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import jpeg_decoder_amd.distributed as D

    dist = D.init(backend="gloo")
    mine = D.shard(n_images, rank, world)
    # stand-in for decoded pixels: image i -> 5 bytes of value i (padded to the largest shard)
    per = -(-n_images // world)
    local = torch.zeros(per * 5, dtype=torch.uint8)
    for k, i in enumerate(mine):
        local[k * 5:(k + 1) * 5] = i
    gathered = D.gather_pixels(local, dst=0)
    tmax = D.max_over_ranks([float(rank + 1), 10.0 - rank])
    ok = True
    if rank == 0:
        seen = []
        for r, t in enumerate(gathered):
            for k, i in enumerate(D.shard(n_images, r, world)):
                ok = ok and bool((t[k * 5:(k + 1) * 5] == i).all())
                seen.append(i)
        ok = ok and seen == list(range(n_images))
    else:
        ok = gathered is None
    ok = ok and tmax == [float(world), 10.0]
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_everything_once():
    sys.path.insert(0, ROOT)
    import jpeg_decoder_amd.distributed as D
    for n in (0, 1, 7, 8, 4096):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in D.shard(n, r, world)]
            assert got == list(range(n))
            sizes = [len(D.shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_multi_rank_init_refuses_to_guess_a_port(monkeypatch):
    """no constant rendezvous port (two jobs on one node would collide).  One rank picks a free port for itself;
    several ranks without MASTER_PORT are a launcher error, not something to paper over."""
    sys.path.insert(0, ROOT)
    import jpeg_decoder_amd.distributed as D
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        D.init(backend="gloo")
    a, b = D.free_port(), D.free_port()
    assert 1024 < a < 65536 and 1024 < b < 65536
    assert "29511" not in open(D.__file__).read()


@pytest.mark.timeout(120)
def test_two_ranks_gloo_shard_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


# ---- CPU shares by NUMA node (VERDICT r4 #2b): the Python rule (bench.py's ranks) and the library's (jpgpu_pipeline_create_multi)
# against fake sysfs trees ------------------------------------------------------------------------------------------------------------
def _fake_sysfs(root, gpu_nodes, node_cpus):
    """gpu_nodes: {bdf: node}, node_cpus: {node: "cpulist"}"""
    for bdf, node in gpu_nodes.items():
        d = os.path.join(root, "bus", "pci", "devices", bdf)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "numa_node"), "w").write(f"{node}\n")
    for node, cpus in node_cpus.items():
        d = os.path.join(root, "devices", "system", "node", f"node{node}")
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "cpulist"), "w").write(cpus + "\n")


def _library_shares(sysfs, bdfs, allowed):
    import ctypes as C

    import jpeg_decoder_amd as J
    J.build()
    L = J.lib()
    ids = (C.c_char_p * len(bdfs))(*[b.encode() for b in bdfs])
    al = (C.c_int * len(allowed))(*allowed)
    dev_of = (C.c_int * len(allowed))()
    nodes = (C.c_int * len(bdfs))()
    assert L.jpgpu_plan_cpu_shares(sysfs.encode(), ids, len(bdfs), al, len(allowed), dev_of, nodes) == 0
    shares = [[allowed[i] for i in range(len(allowed)) if dev_of[i] == k] for k in range(len(bdfs))]
    return shares, list(nodes)


EIGHT = [f"0000:{b:02x}:00.0" for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xE5, 0xF5)]


@pytest.mark.parametrize("case", ["two_sockets", "two_sockets_16_granted", "one_node_unknown", "node_too_small", "fewer_cpus_than_devices", "upper_case_ids", "all_on_node_1"])
def test_cpu_shares_follow_the_numa_node_of_each_gpu(tmp_path, case):
    sys.path.insert(0, ROOT)
    import jpeg_decoder_amd.distributed as D

    root = str(tmp_path)
    nodes = {b: (0 if k < 4 else 1) for k, b in enumerate(EIGHT)}
    cpus = {0: "0-63,128-191", 1: "64-127,192-255"}
    allowed = list(range(256))
    bdfs = list(EIGHT)
    if case == "two_sockets_16_granted":
        allowed = list(range(0, 8)) + list(range(64, 72))  # what an affinity mask of 16 CPUs over both sockets looks like
    elif case == "one_node_unknown":
        nodes[EIGHT[3]] = -1
    elif case == "node_too_small":
        allowed = list(range(0, 2)) + list(range(64, 128))  # two allowed CPUs on node 0 for four devices
    elif case == "fewer_cpus_than_devices":
        allowed = [0, 1, 2]
    elif case == "upper_case_ids":
        bdfs = [b.upper() for b in EIGHT]
    elif case == "all_on_node_1":
        nodes = {b: 1 for b in EIGHT}
    _fake_sysfs(root, nodes, cpus)
    py_shares, py_nodes = D.cpu_shares(allowed, bdfs, root)
    c_shares, c_nodes = _library_shares(root, bdfs, allowed)
    assert py_shares == c_shares and py_nodes == c_nodes
    flat = [c for s in py_shares for c in s]
    assert len(flat) == len(set(flat)) and set(flat) <= set(allowed)  # disjoint, nothing invented
    node_cpu_sets = {0: set(range(0, 64)) | set(range(128, 192)), 1: set(range(64, 128)) | set(range(192, 256))}
    if case in ("two_sockets", "two_sockets_16_granted", "upper_case_ids", "all_on_node_1"):
        for k, s in enumerate(py_shares):  # every device's CPUs are on ITS node, and the devices of a node share it evenly
            assert s and set(s) <= node_cpu_sets[py_nodes[k]]
        sizes = [len(s) for s in py_shares]
        assert max(sizes) - min(sizes) <= 1
        if case == "two_sockets":
            assert py_shares[0] == list(range(0, 32)) and py_shares[4] == list(range(64, 96))
        if case == "two_sockets_16_granted":
            assert py_shares == [[0, 1], [2, 3], [4, 5], [6, 7], [64, 65], [66, 67], [68, 69], [70, 71]]
        if case == "all_on_node_1":
            assert sorted(flat) == sorted(node_cpu_sets[1])
    elif case in ("one_node_unknown", "node_too_small"):  # contiguous slices of the allowed list for everybody
        n = len(bdfs)
        assert py_shares == [allowed[len(allowed) * k // n: len(allowed) * (k + 1) // n] for k in range(n)]
    else:
        assert py_shares == [[] for _ in bdfs]  # nobody is pinned


def test_rank_cpu_share_uses_the_gpu_topology(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    root = str(tmp_path)
    _fake_sysfs(root, {b: (0 if k < 4 else 1) for k, b in enumerate(EIGHT)}, {0: "0-3", 1: "4-7"})
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(8)))
    shares = [bench.rank_cpu_share(r, 8, EIGHT, root)[0] for r in range(8)]
    assert shares == [[0], [1], [2], [3], [4], [5], [6], [7]]
    # unknown topology (no ids: dry runs, one-socket hosts): the contiguous rule of round 4
    assert [bench.rank_cpu_share(r, 2)[0] for r in range(2)] == [[0, 1, 2, 3], [4, 5, 6, 7]]
