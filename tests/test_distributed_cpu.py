"""CPU, world_size 2 on gloo: the sharding + hit-gather layer used by the multi-GPU runs.
Each rank searches its shard with the ORACLE (there is no GPU here) and the gathered result must equal
the single-process result: same partitioning / seq_index / collective code path as on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

import tntcheck as tc
from thermonucleotideblast_amd import shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = synth.make_workload(3, n_targets=240, n_assays=16)
        # ragged lengths so that the balance-by-bases cut is not the midpoint
        lens = np.diff(wl.offsets.astype(np.int64))
        keep = np.ones(len(wl.codes), dtype=bool)
        for t in range(0, 60):
            keep[int(wl.offsets[t]) + 500:int(wl.offsets[t + 1])] = False
            lens[t] = 500
        codes = wl.codes[keep]
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        ranges = shard.shard_ranges(offsets, world)
        b, e = ranges[rank]
        c, o = shard.take_shard(codes, offsets, ranges[rank])
        ora = tc.OracleImpl()
        local = ora.search(wl.options, wl.expanded, c, o, first=b, threads=1)
        allhits, counts = shard.gather_hits(local)
        keys = tc.sorted_hit_keys(allhits)
        # the compact form (what bench.py sends): exact byte counts per rank, unpacked on the root by the library
        for mode in ("padded", "exact"):
            blobs, counts2 = shard.finish_compact(shard.stage_compact(shard.pack_hits(local), len(local), mode=mode))
            if rank == 0:
                assert counts2 == counts
                assert tc.sorted_hit_keys(shard.unpack_all(blobs)) == keys
                assert sum(len(b) for b in blobs) < sum(counts) * 1464 // 3
        if rank == 0:
            whole = tc.sorted_hit_keys(ora.search(wl.options, wl.expanded, codes, offsets, threads=2))
            q.put((ranges, counts, keys == whole, len(whole)))
        else:
            q.put((rank, len(keys)))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather(checkers):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = [r for r in results if isinstance(r[0], list)][0]
    ranges, counts, same, n = r0
    assert same and n > 0
    assert sum(counts) == n
    assert ranges[0][1] == ranges[1][0] and ranges[0][0] == 0 and ranges[1][1] == 240
    assert ranges[0][1] != 120  # balanced by bases, not by count


def test_shard_ranges_cover_and_balance():
    offs = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(10, 3000, 1000))]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        r = shard.shard_ranges(offs, world)
        assert r[0][0] == 0 and r[-1][1] == 1000
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        sizes = [int(offs[e] - offs[b]) for b, e in r]
        assert max(sizes) - min(sizes) <= 2 * 3000
    assert shard.shard_ranges(np.array([0], dtype=np.uint64), 4) == [(0, 0)] * 4


def test_bench_launcher_starts_ranks():
    """`python bench.py --gpus 2` must start two ranks itself (the parent never touches a GPU): the launch-only mode
    runs the rendezvous, the rank count and the staged hit gather on gloo and prints rank 0's JSON line last."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-only"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["gather_ok"] is True
    assert len(line["ms_per_step_per_rank"]) == 2


def test_bench_launcher_eight_ranks():
    """The driver's 8-GPU form of the job: eight ranks started by the launcher, both forms of the compact gather, and the
    exact form with a failing send (all ranks agree and deliver through the padded gather)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--launch-only"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["gather_ok"] is True
    assert line["gather_modes"] == {"padded": "padded", "exact": "exact", "exact-failing": "padded (after a failed exact transfer)"}


def test_strong_scaling_deals_every_piece_once():
    """bench.py --scaling strong: the configuration's full database in 8 pieces, contiguous runs per rank."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for cfg, total in ((4, 10_000_000), (5, 5_000_000)):
        for world in (1, 2, 4, 8):
            seen = []
            for rank in range(world):
                per_piece, pieces, full, full_total = bench.deal_pieces(cfg, "strong", world, rank)
                assert per_piece == total // bench.PIECES and full_total == total
                assert pieces == list(range(pieces[0], pieces[0] + bench.PIECES // world))
                seen += pieces
            assert seen == list(range(bench.PIECES))
        with pytest.raises(SystemExit):
            bench.deal_pieces(cfg, "strong", 3, 0)
    per_piece, pieces, full, _ = bench.deal_pieces(4, "weak", 8, 5)
    assert (per_piece, pieces) == (1_250_000, [5]) and full == 1_250_000
    assert bench.deal_pieces(2, "weak", 1, 0, targets=5000)[:2] == (5000, [0])


def test_gather_returns_owned_records(checkers):
    """finish_gather(copy=True) hands out records that a later gather cannot overwrite (single process: no collective)."""
    from thermonucleotideblast_amd import abi
    a = (abi.tnt_hit * 2)()
    a[0].seq_index, a[1].seq_index = 5, 6
    got, counts = shard.gather_hits(a)
    assert counts == [2] and [h.seq_index for h in got] == [5, 6]
