"""Multi-GPU layer: targets shard by sequence, hit records are gathered to rank 0.

The reference distributes (target, query-range) work items from an MPI master to workers and
collects packed hit lists per chunk of assays (tntblast_master.cpp:438-461,811-828,
tntblast_worker.cpp:173-186,602).  Targets are independent, so here the database is cut into
contiguous index ranges balanced by base count (global `seq_index` is preserved), the assay
tables are replicated, and there is exactly one collective after the search: an all-gather of the
per-rank hit counts followed by the transfer of the records to rank 0 (RCCL over xGMI with the "nccl"
backend on GPUs; the same code runs on "gloo" for the CPU tests).  Two forms of the transfer:
  * stage_hits / finish_gather: ONE gather of the full tnt_hit arrays, padded to the largest rank (1 464 bytes per hit);
  * stage_compact / finish_compact: the compact records of tnt_search_hits_compact (numbers + text lengths + text bytes,
    about 350 bytes per hit for 20-mers).  mode "padded" (default, what bench.py runs): ONE `dist.gather` of the blobs
    padded to the largest rank's byte count -- a plain collective, the form that has run on RCCL; with the same workload
    on every rank the padding is a few per cent.  mode "exact": every rank sends exactly its bytes to rank 0 (isend /
    irecv sized from the all-gathered byte counts); if posting or completing those fails on any rank, all ranks agree
    on it (one all-reduce) and repeat the transfer in the padded form.
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


# ------------------------------------------------------------------------------------------------ compact records
HIT_POD_BYTES = 120  # the numbers of a tnt_hit (everything before its three texts)


def pack_hits(hits):
    """Host-side packer of the compact layout (include/tnt_mi355x.h, tnt_search_hits_compact): for ranks whose hits come
    from the CPU checkers (tests) and for checking the device packer."""
    out = [b""]
    total = 8
    for h in hits:
        raw = bytes(C.string_at(C.addressof(h), HIT_POD_BYTES))
        texts = [bytes(h.forward_align), bytes(h.reverse_align), bytes(h.probe_align)]
        rec = raw + np.array([len(t) for t in texts] + [0], dtype="<u2").tobytes() + b"".join(texts)
        rec += b"\0" * (-len(rec) % 4)
        out.append(rec)
        total += len(rec)
    out[0] = np.array([len(hits), total], dtype="<u4").tobytes()
    return b"".join(out)


class _StagedCompact:
    def __init__(self, mine, sizes, counts, device, root, reqs, recv, mode="exact", padded=None, failed=False):
        self.mine, self.sizes, self.counts, self.device, self.root, self.reqs, self.recv = mine, sizes, counts, device, root, reqs, recv
        self.mode, self.padded, self.failed = mode, padded, failed


def _as_tensor(blob, device):
    """This rank's compact records as a uint8 tensor on `device`: bytes (CPU) or (device pointer, n_bytes)."""
    import torch
    if isinstance(blob, tuple):
        nbytes = int(blob[1])
        return torch.as_tensor(_DeviceBytes(blob[0], max(1, nbytes)), device=device)[:nbytes], nbytes
    nbytes = len(blob)
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8) if nbytes else torch.empty(0, dtype=torch.uint8)
    if device != "cpu":
        mine = mine.to(device)
    return mine, nbytes


def stage_compact(blob, n_hits, device="cpu", root=0, always_collective=False, mode="padded"):
    """`blob` = this rank's compact records: bytes (CPU) or (device pointer, n_bytes) from Engine.hits_compact().
    All-gather of (hit count, byte count); then, mode "padded": a copy of the records into a send buffer of the largest
    rank's size (the gather itself runs in finish_compact); mode "exact": every other rank posts the send of exactly its
    bytes and the root the matching receives.  Either way the caller may start the next search right away: the blob is
    only rewritten by the next hits_compact() call (padded: it has been copied; exact: the send reads it, so collect
    the next search only after finish_compact)."""
    import torch
    import torch.distributed as dist
    if mode not in ("padded", "exact"):
        raise ValueError("mode must be 'padded' or 'exact'")
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine, nbytes = _as_tensor(blob, device)
    if world == 1 and not (always_collective and dist.is_initialized()):
        return _StagedCompact(mine, [nbytes], [int(n_hits)], device, root, [], [], mode)
    me = torch.tensor([int(n_hits), nbytes], dtype=torch.int64, device=device)
    every = [torch.zeros_like(me) for _ in range(world)]
    dist.all_gather(every, me)
    counts = [int(e[0].item()) for e in every]
    sizes = [int(e[1].item()) for e in every]
    rank = dist.get_rank()
    if mode == "padded":
        return _StagedCompact(mine, sizes, counts, device, root, [], [], mode, padded=_pad(mine, nbytes, max(1, max(sizes)), device))
    reqs, recv, failed = [], [], False
    try:
        if rank == root:
            for r in range(world):
                if r == root or sizes[r] == 0:
                    recv.append(None)
                    continue
                buf = torch.empty(sizes[r], dtype=torch.uint8, device=device)
                recv.append(buf)
                reqs.append(dist.irecv(buf, src=r))
        elif nbytes:
            reqs.append(dist.isend(mine, dst=root))
    except Exception:  # agreed on by all ranks in finish_compact, which then runs the padded form
        failed = True
    return _StagedCompact(mine, sizes, counts, device, root, reqs, recv, mode, failed=failed)


def _pad(mine, nbytes, nmax, device):
    import torch
    buf = torch.zeros(nmax, dtype=torch.uint8, device=device)
    if nbytes:
        buf[:nbytes].copy_(mine[:nbytes])
    if device != "cpu":
        torch.cuda.current_stream().synchronize()  # the copy is done before the library rewrites the blob
    return buf


def finish_compact(st):
    """Completes the transfers.  On the root returns (list of per-rank blobs as bytes-like host arrays, counts): the
    device buffers are copied into page-locked host memory.  Elsewhere returns ([own blob], counts)."""
    import torch
    import torch.distributed as dist
    world = len(st.sizes)
    rank = dist.get_rank() if dist.is_initialized() else 0
    collective = dist.is_initialized() and (world > 1 or st.padded is not None or st.reqs or st.recv)
    parts = None
    if collective and st.mode == "exact":
        # The ranks first agree on whether every transfer could be POSTED, before any of them waits: a rank whose isend
        # raised leaves the root's irecv without a partner, and the root would sit in wait() in front of the agreement
        # (gloo for ever, RCCL until its watchdog).  What this cannot repair: the orphaned irecv stays posted, so a
        # one-sided failure is only survivable where the backend lets a later collective overtake it (gloo does; with RCCL
        # point-to-point operations and collectives of a communicator are sequenced and the run ends at the watchdog).  The
        # padded gather is the default for that reason; the exact form is an option for uniform, healthy fabrics.
        posted = torch.tensor([1 if st.failed else 0], dtype=torch.int64, device=st.device)
        dist.all_reduce(posted, op=dist.ReduceOp.MAX)
        failed = bool(int(posted.item()))
        if not failed:
            for r in st.reqs:
                try:
                    r.wait()
                except Exception:
                    failed = True
        flag = torch.tensor([1 if failed else 0], dtype=torch.int64, device=st.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):  # some rank could not send or receive: every rank repeats the transfer as one padded gather
            st.padded = _pad(st.mine, st.sizes[rank], max(1, max(st.sizes)), st.device)
            st.mode = "padded (after a failed exact transfer)"
        elif rank == st.root:
            parts = [st.mine if r == st.root else st.recv[r] for r in range(world)] if st.recv else [st.mine]
    if collective and st.padded is not None:
        out = [torch.empty_like(st.padded) for _ in range(world)] if rank == st.root else None
        dist.gather(st.padded, out, dst=st.root)
        if rank == st.root:
            parts = out
    if rank != st.root:
        if st.device != "cpu":
            torch.cuda.current_stream().synchronize()
        return [st.mine], st.counts
    if parts is None:
        parts = [st.mine]
    total = sum(st.sizes)
    host = torch.empty(max(1, total), dtype=torch.uint8) if st.device == "cpu" else _pinned_bytes(total)
    out, o = [], 0
    for r, p in enumerate(parts):
        n = st.sizes[r]
        if n and p is not None:
            host[o:o + n].copy_(p[:n], non_blocking=True)
        out.append((o, n))
        o += n
    if st.device != "cpu":
        torch.cuda.current_stream().synchronize()
    raw = host.numpy()
    return [raw[o:o + n] for o, n in out], st.counts


def unpack_all(blobs):
    """Per-rank compact blobs -> one ctypes array of tnt_hit in rank order (host only)."""
    from .engine import unpack_hits
    parts = [unpack_hits(bytes(b)) for b in blobs if len(b) >= 8]
    total = sum(len(p) for p in parts)
    out = (abi.tnt_hit * total)()
    o = 0
    for p in parts:
        if len(p):
            C.memmove(C.addressof(out) + o * C.sizeof(abi.tnt_hit), p, len(p) * C.sizeof(abi.tnt_hit))
            o += len(p)
    return out
