"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
in the CPU tests).  Images are independent, so a batch shards by image with NO data-path collective;
the only optional collective is the final gather of the pixel buffers to one rank."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def free_port():
    """A TCP port that is free on 127.0.0.1 right now (the launcher hands it to every rank of a job)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard(n_items, rank, world):
    """Contiguous shard of `n_items` images for `rank`: sizes differ by at most one, order preserved."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def _parse_cpulist(txt):
    """"0-63,128-191" -> [0, ..., 63, 128, ..., 191]"""
    out = []
    for part in txt.replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        try:
            a = int(lo)
            b = int(hi) if hi else a
        except ValueError:
            continue
        out.extend(range(a, b + 1))
    return out


def numa_node_of(bdf, sysfs="/sys"):
    """NUMA node of the PCI device `bdf` ("0000:c1:00.0"); -1 if unknown (one socket, a VM, no such file)."""
    if not bdf:
        return -1
    try:
        return int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf.lower(), "numa_node")).read().strip())
    except (OSError, ValueError):
        return -1


def cpu_shares(allowed, bdfs, sysfs="/sys"):
    """Deal the CPUs `allowed` to the devices `bdfs` (PCI bus ids, one per rank / device): a device's feeder threads belong on the
    socket its PCIe root hangs off.  Devices of one NUMA node split the node's allowed CPUs evenly, in list order; if any
    device's node is unknown, or a node has fewer allowed CPUs than devices, EVERY device gets a contiguous slice of the allowed list
    instead (shares stay disjoint).  Fewer CPUs than devices: nobody is pinned (empty shares).  The same rule as the library's
    jpgpu_plan_cpu_shares (csrc/pipeline.cpp), which tests/test_distributed.py checks against this one on fake sysfs trees.
    -> (list of CPU lists, list of NUMA nodes)"""
    n = len(bdfs)
    nodes = [numa_node_of(b, sysfs) for b in bdfs]
    if n == 0 or len(allowed) < n:
        return [[] for _ in range(n)], nodes
    by, by_node = [], all(x >= 0 for x in nodes)
    for k in range(n):
        if not by_node:
            break
        same = [j for j in range(n) if nodes[j] == nodes[k]]
        try:
            node_cpus = set(_parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{nodes[k]}", "cpulist")).read()))
        except OSError:
            node_cpus = set()
        cand = [c for c in allowed if c in node_cpus]
        if len(cand) < len(same):
            by_node = False
            break
        r = same.index(k)
        by.append(cand[len(cand) * r // len(same): len(cand) * (r + 1) // len(same)])
    if by_node:
        return by, nodes
    return [list(allowed[len(allowed) * k // n: len(allowed) * (k + 1) // n]) for k in range(n)], nodes


def init(backend=None, force=False):
    """Initialise the default process group from the torchrun environment (no-op for world size 1 unless `force`:
    a one-rank group — the collective library's set-up and collectives exercised on a single GPU)."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank_world()
    if world == 1 and not force:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        # One rank can pick any free port for itself.  Several ranks must AGREE on one, and only their launcher can tell them
        # (bench.py's launcher_command passes --master-port free_port(); torchrun exports MASTER_PORT): a constant here would
        # make two jobs on one node collide, so a multi-rank start without a port is refused.
        if world > 1:
            raise RuntimeError("jpeg_decoder_amd.distributed.init: WORLD_SIZE > 1 but MASTER_PORT is not set — launch the ranks through "
                               "torch.distributed.run (--master-port P) or bench.py --gpus N, which pick a free port for the whole job")
        os.environ["MASTER_PORT"] = str(free_port())
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(values, device="cpu"):
    """Element-wise MAX of a list of floats over all ranks (bench timing contract)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_initialized():  # (a one-rank group too: bench.py --force-dist runs the collective on the device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def min_over_ranks(value, device="cpu"):
    """MIN of one float over all ranks (e.g. the job size every rank can hold)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t[0])


def gather_pixels(local, dst=0):
    """Final gather of equal-sized per-rank pixel arenas to rank `dst` (north_star's only collective).
    Returns the list of per-rank tensors on `dst`, None elsewhere.  On a fully connected xGMI node each
    peer->root transfer rides its own link; there is nothing to reduce, so gather (not all-gather)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    rank, world = dist.get_rank(), dist.get_world_size()
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out
