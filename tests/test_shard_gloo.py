"""Multi-process (gloo, world_size 2 and 3, CPU) check of the block-list sharding + frame reassembly
used by bench.py --gpus N.  The per-rank compressor is stood in for by the oracle (test-only): what
is under test is the product's partitioning / size exchange / gather logic."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from cases import text


def _worker(rank, world, port, n, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, H.ROOT)
        from zstdmt_amd.shard import exchange_segment_sizes, gather_segments, shard_range
        data = text(n)
        nchunks = max(1, (n + chunk - 1) // chunk)
        lo, hi = shard_range(nchunks, rank, world)
        part = data[lo * chunk:hi * chunk]
        seg = H.oracle_compress(part, chunk) if hi > lo else b""
        sizes, my_off = exchange_segment_sizes(len(seg))
        full = gather_segments(torch.frombuffer(bytearray(seg), dtype=torch.uint8) if seg
                               else torch.empty(0, dtype=torch.uint8), sizes, dst=0)
        # the no-collective variant: every rank writes its segment at its offset of ONE shared host buffer
        from zstdmt_amd.shard import SharedHostStream
        name = "zstdmt_amd_test_%d" % port
        if rank == 0:
            shm = SharedHostStream(name, sum(sizes), owner=True)
        dist.barrier()
        if rank != 0:
            shm = SharedHostStream(name, sum(sizes), owner=False)
        shm.write_at(my_off, seg)
        dist.barrier()
        placed = bytes(shm.map[:sum(sizes)]) if rank == 0 else None
        dist.barrier()
        shm.close()
        q.put((rank, lo, hi, my_off, sizes, bytes(full.numpy().tobytes()) if rank == 0 else None, placed))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 5 * 131072 + 777), (3, 2 * 131072), (2, 131072)])
def test_sharded_stream_equals_single_stream(world, n):
    chunk = 131072
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = H.oracle_compress(text(n), chunk)
    assert res[0][5] == whole      # grouped send/recv gatherv
    assert res[0][6] == whole      # every rank's own copy into the shared host buffer
    # ranges tile the chunk list, offsets are the exclusive scan of the sizes
    assert res[0][1] == 0 and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    assert [r[3] for r in res] == [sum(res[0][4][:i]) for i in range(world)]


def test_shard_range_properties():
    sys.path.insert(0, H.ROOT)
    from zstdmt_amd.shard import shard_range
    for n in (0, 1, 7, 8, 65536, 65537):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1
