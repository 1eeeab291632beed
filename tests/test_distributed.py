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
