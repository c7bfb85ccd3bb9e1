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
    """gatherv of variable-length uint8 segments to rank `dst` as ONE group of point-to-point operations
    (`dist.batch_isend_irecv` = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd under RCCL): the root's
    receives are posted together, so the W - 1 transfers run side by side over the root's W - 1 xGMI links
    instead of one after the other on one stream.  Returns the concatenated stream on dst,
    None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert segment.dtype == torch.uint8 and segment.numel() == sizes[rank]
    if rank != dst:
        if sizes[rank]:
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, segment, dst)]):
                q.wait()
        return None
    full = torch.empty(sum(sizes), dtype=torch.uint8, device=segment.device)
    off, ops = 0, []
    for r in range(world):
        view = full[off:off + sizes[r]]
        if r == dst:
            view.copy_(segment)
        elif sizes[r]:
            ops.append(dist.P2POp(dist.irecv, view, r))
        off += sizes[r]
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return full


class SharedHostStream:
    """The no-collective reassembly: ONE host buffer of the whole stream, mapped by every rank of the
    node (POSIX shared memory), into which each rank copies its own segment at its own offset -- W device-to-host
    copies over W PCIe links in parallel, no inter-GPU traffic at all.  `name` must be the same on every rank;
    rank `owner` creates and finally unlinks it."""

    def __init__(self, name: str, total_bytes: int, owner: bool):
        import mmap
        import os
        self.path = "/dev/shm/" + name
        self.total = max(1, int(total_bytes))
        self.owner = owner
        nofollow = getattr(os, "O_NOFOLLOW", 0)
        if owner:
            # a stale segment of a crashed run is removed, never reused; O_EXCL | O_NOFOLLOW: a file or symbolic link
            # planted under the (predictable) name is neither followed nor truncated -- the open fails instead
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            fd = os.open(self.path, os.O_CREAT | os.O_EXCL | os.O_RDWR | nofollow, 0o600)
            os.ftruncate(fd, self.total)
        else:
            fd = os.open(self.path, os.O_RDWR | nofollow)
            st = os.fstat(fd)
            if st.st_uid != os.getuid() or st.st_size < self.total:
                os.close(fd)
                raise OSError("shared segment %s is not the owner rank's (uid / size mismatch)" % self.path)
        try:
            self.map = mmap.mmap(fd, self.total)
        finally:
            os.close(fd)

    def address(self) -> int:
        import ctypes
        return ctypes.addressof(ctypes.c_char.from_buffer(self.map))

    def write_at(self, off: int, data: bytes):
        """CPU path of the same placement (tests; the GPU path copies device -> self.address() + off)"""
        self.map[off:off + len(data)] = data

    def close(self):
        import os
        self.map.close()
        if self.owner:
            try:
                os.unlink(self.path)
            except OSError:
                pass
