"""Multi-GPU layer: targets shard by sequence, hit records are gathered to every rank.

The reference distributes (target, query-range) work items from an MPI master to workers and
collects packed hit lists per chunk of assays (tntblast_master.cpp:438-461,811-828,
tntblast_worker.cpp:173-186,602).  Targets are independent, so here the database is cut into
contiguous index ranges balanced by base count (global `seq_index` is preserved), the assay
tables are replicated, and there is exactly one collective after the search: an all-gather of the
per-rank hit counts followed by an all-gather of the padded POD records (RCCL over xGMI with the
"nccl" backend on GPUs; the same code runs on "gloo" for the CPU tests).
"""
import ctypes as C

import numpy as np

from . import abi


def shard_ranges(offsets, world):
    """Contiguous target ranges [begin, end) per rank, balanced by number of bases."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    total = int(offsets[-1])
    cuts = [0]
    for r in range(1, world):
        want = total * r // world
        cuts.append(int(np.searchsorted(offsets, want, side="left")))
    cuts.append(n)
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def take_shard(codes, offsets, rng):
    b, e = rng
    offsets = np.asarray(offsets, dtype=np.uint64)
    return codes[int(offsets[b]):int(offsets[e])], (offsets[b:e + 1] - offsets[b]).astype(np.uint64)


class _DeviceBytes:
    """A span of device memory for torch.as_tensor (zero copy, __cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


_pinned = {}


def _pinned_bytes(nbytes):
    """A reusable page-locked receive buffer on the root (the D2H of the gathered records runs at link speed)."""
    import torch
    buf = _pinned.get("buf")
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        _pinned["buf"] = buf
    return buf


class _Staged:
    """First half of a gather: counts known everywhere, this rank's records in a private send buffer."""

    def __init__(self, hits, counts, buf, device, root):
        self.hits, self.counts, self.buf, self.device, self.root = hits, counts, buf, device, root


def stage_hits(hits, device="cpu", device_hits=None, root=0, always_collective=False):
    """All-gather of the per-rank counts and a copy of this rank's records into a send buffer of the common (padded)
    size.  After it returns the caller may start the next search: the library's hit buffers are no longer needed.

    device_hits = (device pointer, count) from Engine.hits_device(): the records are taken where the pair kernel left
    them, without a round trip through the host."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (always_collective and dist.is_initialized()):  # always_collective: tests run the collective path on one rank
        return _Staged(hits, [len(hits)], None, device, root)
    hb = C.sizeof(abi.tnt_hit)
    n_mine = len(hits)
    n = torch.tensor([n_mine], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    nmax = max(1, max(counts))
    buf = torch.empty(nmax * hb, dtype=torch.uint8, device=device)
    if n_mine:
        if device_hits is not None and device != "cpu" and device_hits[1] == n_mine:
            buf[:n_mine * hb].copy_(torch.as_tensor(_DeviceBytes(device_hits[0], n_mine * hb), device=device))
        else:
            buf[:n_mine * hb].copy_(torch.from_numpy(np.frombuffer(hits, dtype=np.uint8)))
    if device != "cpu":
        torch.cuda.current_stream().synchronize()  # the copy is done before the library reuses its buffers
    return _Staged(hits, counts, buf, device, root)


def finish_gather(st, copy=True):
    """ONE gather of the padded record arrays to the root (RCCL on GPUs, gloo in the CPU tests) and, on the root, their
    copy into a page-locked host buffer.  Returns (hits, counts); on the root `hits` holds all records in rank order,
    elsewhere it is the rank's own list.

    copy=True (default) returns records the caller owns.  copy=False returns a VIEW of the module's reusable page-locked
    receive buffer: it is overwritten by the next gather on this process (bench.py, which only counts the records)."""
    import torch
    import torch.distributed as dist
    if st.buf is None:
        return st.hits, st.counts
    world, rank = dist.get_world_size(), dist.get_rank()
    hb = C.sizeof(abi.tnt_hit)
    out = [torch.empty_like(st.buf) for _ in range(world)] if rank == st.root else None
    dist.gather(st.buf, out, dst=st.root)
    if rank != st.root:
        if st.device != "cpu":
            torch.cuda.current_stream().synchronize()  # the send buffer is free again
        return st.hits, st.counts
    total = sum(st.counts)
    host = torch.empty(max(1, total * hb), dtype=torch.uint8) if st.device == "cpu" else _pinned_bytes(total * hb)
    o = 0
    for r in range(world):
        if st.counts[r]:
            host[o:o + st.counts[r] * hb].copy_(out[r][:st.counts[r] * hb], non_blocking=True)
            o += st.counts[r] * hb
    if st.device != "cpu":
        torch.cuda.current_stream().synchronize()
    if not total:
        return (abi.tnt_hit * 0)(), st.counts
    raw = host.numpy()[:total * hb]
    res = (abi.tnt_hit * total).from_buffer_copy(raw) if copy else (abi.tnt_hit * total).from_buffer(raw)
    return res, st.counts


def gather_hits(hits, device="cpu", device_hits=None, root=0, always_collective=False, copy=True):
    """The hit records of every rank, collected on `root` (stage_hits + finish_gather in one call)."""
    return finish_gather(stage_hits(hits, device, device_hits, root, always_collective), copy=copy)
