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


def gather_hits(hits, device="cpu"):
    """All ranks contribute their tnt_hit arrays; returns (all hits as one ctypes array, counts)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return hits, [len(hits)]
    hb = C.sizeof(abi.tnt_hit)
    n = torch.tensor([len(hits)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    nmax = max(1, max(counts))
    buf = torch.zeros(nmax * hb, dtype=torch.uint8, device=device)
    if len(hits):
        raw = np.frombuffer(hits, dtype=np.uint8).copy()
        buf[:raw.size] = torch.from_numpy(raw).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    total = sum(counts)
    res = (abi.tnt_hit * total)()
    o = 0
    for r in range(world):
        if counts[r]:
            chunk = out[r][:counts[r] * hb].cpu().numpy().tobytes()
            C.memmove(C.byref(res, o * hb), chunk, len(chunk))
            o += counts[r]
    return res, counts
