"""Block-list sharding across GPUs (one process per GPU, torch.distributed; "nccl" = RCCL on ROCm).

Chunks are independent (each is its own frame, /root/reference/lib/lz4-mt_compress.c:279-283), so
rank r takes the contiguous chunk range [r*N/W, (r+1)*N/W): its output is one contiguous segment of
the final MT stream and order is preserved by concatenating segments in rank order -- the
multi-GPU form of pt_write's in-order flush (lib/lz4-mt_compress.c:178-205).  The only exchange the
path needs is the segment sizes (-> every rank's byte offset in the final stream); gathering the
bytes themselves to one rank is optional (`gather_segments`), each rank can just as well write its
segment at its offset.
"""
import torch
import torch.distributed as dist


def shard_range(n_chunks: int, rank: int, world: int):
    """Contiguous, balanced: the first (n_chunks % world) ranks get one extra chunk."""
    base, extra = divmod(n_chunks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def exchange_segment_sizes(my_bytes: int, device="cpu"):
    """all-gather of one int64 per rank -> (sizes list, my byte offset in the final stream)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([int(my_bytes)], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine)
    sizes = [int(x) for x in sizes.tolist()]
    return sizes, sum(sizes[:rank])


def gather_segments(segment: torch.Tensor, sizes, dst: int = 0):
    """gatherv of variable-length uint8 segments to rank `dst` (grouped send/recv).
    Returns the concatenated stream on dst, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert segment.dtype == torch.uint8 and segment.numel() == sizes[rank]
    if rank != dst:
        if sizes[rank]:
            dist.send(segment, dst=dst)
        return None
    full = torch.empty(sum(sizes), dtype=torch.uint8, device=segment.device)
    off, reqs = 0, []
    for r in range(world):
        view = full[off:off + sizes[r]]
        if r == dst:
            view.copy_(segment)
        elif sizes[r]:
            reqs.append(dist.irecv(view, src=r))
        off += sizes[r]
    for q in reqs:
        q.wait()
    return full
