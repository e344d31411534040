#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: oligo x target-base evaluations / second.

`python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no rank environment this process only LAUNCHES:
it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (it never touches the GPU
itself), relays the ranks' output and prints rank 0's JSON line last.  Started by torch.distributed.run (the
driver's form) it is one rank of N.  The reference distributes (target, query-range) work from an MPI master
to workers and collects their hit lists (tntblast_master.cpp:438-461,811-828); here targets shard by sequence,
there is no data-path collective, and the hit records of every step are gathered to rank 0 over RCCL.

A step = one pass of the hot path (seed -> cull -> bind -> pair) over targets that are already packed and
resident in HBM.
  --scaling weak   every rank owns an independent shard of the per-GPU size of the configuration (default)
  --scaling strong one database of the configuration's full size (configs 4 / 5 as BASELINE.json words them: 10 M x
                   2 kb / 5 M x 1 kb), generated in 8 pieces and dealt to the ranks in contiguous runs (default for
                   --config 4 / 5 with --gpus > 1)
Workload at N = 1: BASELINE.json configs[1], "96 PCR primer pairs vs 1M x 1kb synthetic targets".
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_BASE = 0.25      # 2-bit packed targets read once
VALU_PEAK_GUIDE = 1024 * 2.4e9 / 2.0  # guide: 1 024 SIMDs, one wave-instruction per 2 cycles, 2.4 GHz
# measured on the same hardware (scripts/ubench_valu.hip, profiles/r04q_ubench_valu.txt, 2+ wavefronts per SIMD): plain add / logic
# ops sustain 1.07e12 wave-instructions/s, v_max / v_cmp / v_cndmask / v_alignbit / three-operand, SDWA and packed-16 forms 5.95e11
VALU_RATE_MEASURED = {"full_rate_ops": 1.07e12, "half_rate_ops": 5.95e11}
INT32_LANE_OPS_PEAK = VALU_PEAK_GUIDE * 64  # the same issue rate in lane operations: 7.86e13 int32 lane-ops/s
PIECES = 8                     # the strong-scaling database is generated in this many pieces


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config 1..5 (default 2)")
    ap.add_argument("--targets", type=int, default=None, help="override targets per GPU (weak) / in total (strong)")
    ap.add_argument("--assays", type=int, default=None)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: strong for configs 4 / 5 on more than one GPU (BASELINE.json words them as ONE database sharded over the GPUs), else weak")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the bounded CPU baseline sample")
    ap.add_argument("--parity-targets", type=int, default=65536, help="cap of the parity sample (targets checked against the CPU checkers)")
    ap.add_argument("--gather", choices=("padded", "exact"), default="padded",
                    help="N > 1: transfer of the compact hit records to rank 0 -- one dist.gather of blobs padded to the largest rank "
                         "(default) or exact-size isend / irecv (falls back to the padded gather when a transfer fails)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective-selftest", action="store_true",
                    help="single GPU: run the multi-GPU step (RCCL process group of one rank, staged gather) instead of the plain one")
    ap.add_argument("--launch-only", action="store_true",
                    help="no GPU: start the ranks (gloo), count them, run the hit gather on made-up records and print the line")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if (args.config in (4, 5) and args.gpus > 1) else "weak"
    return args


def csrc_hash():
    """sha256 over the kernel / host sources the library is built from (thermonucleotideblast_amd/csrc, sorted by name), 16 hex
    digits: what a PMC file must carry (profiles/bind_pmc.json: csrc_sha16) for its counters to describe THIS build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "thermonucleotideblast_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch(args):
    """Parent of an N-rank run.  Makes no torch.cuda / HIP call: the ranks are fresh child processes."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    line = None
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s  # rank 0's result: printed last
        else:
            sys.stdout.write(out)
            sys.stdout.flush()
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        rc = 1
    sys.exit(rc)


def deal_pieces(config, scaling, world, rank, targets=None):
    """(targets per piece, pieces of this rank).  Weak scaling: every rank owns one piece of the per-GPU size of the
    configuration.  Strong scaling: one database of the configuration's full size, generated in PIECES pieces that are
    dealt to the ranks in contiguous runs (rank r gets pieces [r * PIECES / world, (r + 1) * PIECES / world))."""
    full = {1: 1000, 2: 1_000_000, 3: 1_000_000, 4: 1_250_000, 5: 625_000}[config]
    full_total = {1: 1000, 2: 1_000_000, 3: 1_000_000, 4: 10_000_000, 5: 5_000_000}[config]
    if scaling == "strong":
        if PIECES % world:
            raise SystemExit("strong scaling deals %d pieces: --gpus must divide it" % PIECES)
        return max(1, (targets or full_total) // PIECES), list(range(rank * PIECES // world, (rank + 1) * PIECES // world)), full, full_total
    return targets or full, [rank], full, full_total


def cpu_baseline(wl, budget_s, max_targets=None):
    """Bounded sample of the same workload on the host cores: the reference's own objects
    (oracle/_ref, kind 'reference') when they are present, else this repo's restatement ('port').
    Returns (record, hits of the sample, number of targets in the sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tntcheck as tc
    cores = os.cpu_count() or 1
    if tc.have_ref():
        impl, kind = tc.RefImpl(), "reference"
    elif os.path.exists(tc.ORACLE_SO):
        impl, kind = tc.OracleImpl(), "port"
    else:
        return None, [], 0
    n = min(wl.n_targets, 256 * cores)
    total_t, total_n = 0.0, 0
    first = 0
    hits = []
    limit = min(wl.n_targets, max_targets) if max_targets else wl.n_targets
    while total_t < budget_s and first < limit:
        last = min(limit, first + n)
        codes = wl.codes[int(wl.offsets[first]):int(wl.offsets[last])]
        offs = wl.offsets[first:last + 1] - wl.offsets[first]
        t0 = time.perf_counter()
        got = impl.search(wl.options, wl.expanded, codes, offs, first=first, threads=cores)
        total_t += time.perf_counter() - t0
        hits.extend(tc.hit_key(h) for h in got)
        total_n += last - first
        first = last
        if total_t < budget_s / 4:
            n *= 2
    bases = int(wl.offsets[first])
    rec = {"value": wl.n_oligos * bases / total_t, "unit": "evals/s", "cores": cores, "kind": kind,
           "targets_per_s": total_n / total_t,
           "sample": "first %d targets of the same workload, %d OpenMP threads, %.1f s" % (total_n, cores, total_t)}
    return rec, hits, total_n


def parity_sample(gpu_hits, wl, n_sample, ref_keys, threads):
    """the step's hits on the first `n_sample` targets against the CPU checkers: equality of the hit
    multisets (every field, floats as bit patterns, alignment text) and max |Tm| / |dG| differences over the hits
    that pair up by their integer fields.

    The checker is the oracle (deterministic).  The reference's own objects are compared too and reported beside it:
    cull_oligo_match reads the mask of the element FOLLOWING a valid one even at end() (amplicon_search.cpp:748-755,
    SURVEY Q4) -- a byte of the caller's stack frame -- and the strand it binds first hangs on that count, so single
    amplicons of low-complexity primers come and go between runs of the reference itself
    (profiles/r02a_reference_q4_nondeterminism.log)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tntcheck as tc
    from thermonucleotideblast_amd import abi
    codes = wl.codes[:int(wl.offsets[n_sample])]
    offs = wl.offsets[:n_sample + 1]
    c = sorted(tc.hit_key(h) for h in tc.OracleImpl().search(wl.options, wl.expanded, codes, offs, threads=threads))
    g = sorted(tc.hit_key(h) for h in gpu_hits if h.seq_index < n_sample)
    names = [n for n, _ in abi.tnt_hit._fields_]
    fl = [i for i, (n, t) in enumerate(abi.tnt_hit._fields_) if t is C.c_float]
    txt = [i for i, n in enumerate(names) if n.endswith("_align")]
    ints = [i for i in range(len(names)) if i not in fl and i not in txt]
    target_t = float(wl.options.target_t)

    def split(keys):
        d = {}
        for k in keys:
            d.setdefault(tuple(k[i] for i in ints), []).append(k)
        return d
    dg_, dc_ = split(g), split(c)
    tm_err = dg_err = 0.0
    paired = 0
    f32 = lambda bits: float(np.uint32(bits).view(np.float32))  # noqa: E731
    for ik, gl in dg_.items():
        for a, b in zip(sorted(gl), sorted(dc_.get(ik, []))):
            paired += 1
            for o in ("forward", "reverse", "probe"):
                tm_a, tm_b = f32(a[names.index(o + "_tm")]), f32(b[names.index(o + "_tm")])
                tm_err = max(tm_err, abs(tm_a - tm_b))
                ga = f32(a[names.index(o + "_dH")]) - target_t * f32(a[names.index(o + "_dS")])
                gb = f32(b[names.index(o + "_dH")]) - target_t * f32(b[names.index(o + "_dS")])
                dg_err = max(dg_err, abs(ga - gb))
    out = {"checker": "oracle (CPU restatement, pinned against the reference)", "targets": n_sample, "hits_gpu": len(g), "hits_cpu": len(c),
           "paired": paired, "hit_multiset_equal": g == c, "tm_max_abs_err": tm_err, "dg_max_abs_err": dg_err}
    if ref_keys is not None:
        r = sorted(ref_keys)
        odd = sorted({k[2] for k in set(r) ^ set(g)})
        rec = {"hits": len(r), "hit_multiset_equal": r == g, "targets_that_differ": odd[:16],
               "note": "the reference's first cull reads one element past the end of a std::list (Q4): its "
                       "output varies between runs on such targets"}
        if odd:
            # Where the threaded run of the reference differs, run its objects again on those targets alone, on ONE thread
            # (the stack byte the cull reads past end() is then the same every time): that pins the Q4 fence to the
            # reference where the reference is deterministic.
            ref = tc.RefImpl()
            same = True
            for t in odd[:16]:
                c1 = wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])]
                o1 = np.array([0, len(c1)], dtype=np.uint64)
                one = sorted(tc.hit_key(h) for h in ref.search(wl.options, wl.expanded, c1, o1, first=t, threads=1))
                same = same and one == [k for k in g if k[2] == t]
            rec["reference_1thread_equal"] = same
            rec["reference_1thread_targets"] = len(odd[:16])
        out["reference_objects"] = rec
    return out


def launch_only(args, rank, world):
    """CPU plumbing check of the N-rank path: rendezvous, rank count, the staged gather of hit records (gloo)."""
    import torch
    import torch.distributed as dist
    from thermonucleotideblast_amd import abi, shard
    dist.init_process_group("gloo")
    assert dist.get_world_size() == world == args.gpus, "ranks started (%d) != --gpus (%d)" % (dist.get_world_size(), args.gpus)
    one = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(one)
    mine = (abi.tnt_hit * (rank + 2))()
    for i, h in enumerate(mine):
        h.seq_index, h.assay_id = rank * 1000 + i, i
        h.forward_align = b"5' ACGT 3'"
    t0 = time.perf_counter()
    allhits, counts = shard.gather_hits(mine)
    # the compact records in both forms of the transfer, and the exact form with a send that fails on the last rank: all
    # ranks must agree on the failure and deliver through the padded gather
    got = {}
    for mode in ("padded", "exact", "exact-failing"):
        real_isend = dist.isend
        if mode == "exact-failing" and world > 1 and rank == world - 1:
            def broken(*a, **k):
                raise RuntimeError("simulated transport failure")
            dist.isend = broken
        try:
            st = shard.stage_compact(shard.pack_hits(mine), len(mine), mode="exact" if mode != "padded" else "padded")
            if mode == "exact-failing" and world > 1 and rank == 0:
                # the root's receive from the failing rank would never complete: it is abandoned like a transport error
                class Dead:
                    def wait(self):
                        raise RuntimeError("simulated transport failure")
                st.reqs = [Dead() if i == len(st.reqs) - 1 else r for i, r in enumerate(st.reqs)]
            got[mode] = shard.finish_compact(st) + (st.mode,)
        finally:
            dist.isend = real_isend
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = [torch.zeros_like(el) for _ in range(world)]
    dist.all_gather(per_rank, el)
    ok = True
    modes = {k: v[2] for k, v in got.items()}
    if rank == 0:
        want = [r * 1000 + i for r in range(world) for i in range(r + 2)]
        ok = counts == [r + 2 for r in range(world)] and [h.seq_index for h in allhits] == want
        for mode, (blobs, counts2, _) in got.items():
            unpacked = shard.unpack_all(blobs)
            ok = ok and counts2 == counts and [h.seq_index for h in unpacked] == want and all(h.forward_align == b"5' ACGT 3'" for h in unpacked)
        ok = ok and (world == 1 or modes["exact-failing"].startswith("padded (after"))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launch-only check of the %d-rank path (no GPU work)" % world, "value": 0.0, "unit": "evals/s",
                          "n_gpus": int(one.item()), "rccl_ranks": world, "backend": "gloo", "launch_only": True,
                          "gather_ok": bool(ok), "gather_modes": modes, "steps": 0, "warmup": 0, "ms_per_step": 0.0,
                          "ms_per_step_per_rank": [1000.0 * float(x.item()) for x in per_rank]}), flush=True)
    if not ok:
        sys.exit(1)


def roofline_of_bind_stage(hbm_achieved, traffic, valu, n_binds, bind_ms_step, n_cells=0, pmc_state=None):
    """The dominant stage (pre-filter + bind kernels).  What bounds it is VALU issue (integer DP: DESIGN.md 4.2, 4.2a), so where
    the instruction counts of the workload are on file (profiles/bind_pmc.json, config 2 at full size) the line prices it against
    the VALU issue peak of the guide; the algorithmic-bytes figure against HBM (tiny by construction) stays beside it."""
    kernel = "bind stage: filter_kernel<L> (pre-filter, on the queues the seed kernel filled) + bind_kernel<L> on the survivors; phase B through bucket_kernel"
    hbm = {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
           "note": "algorithmic bytes = 0.25 B/target-base (2-bit packed targets read once) over the stage time (HIP events on the library's stream)"}
    src = ("profiles/bind_pmc.json: rocprofv3 --pmc passes of this workload in a run of their own (scripts/pmc_collect.sh), not measured by this process"
           if traffic is not None else None)
    dps = n_binds / (bind_ms_step * 1e-3) if bind_ms_step > 0 else 0.0
    # SURVEY 8(d) asks for ALGORITHMIC work over the measured time: the windows this run handed to the bind stage (counted on the
    # device: tnt_timing.n_cells = sum of oligo length x window width) x the three states align_dimer updates per cell
    # (nuc_cruc.cpp:492-696), against the int32 lane-operation rate of the chip (one lane-op per state update at the very least)
    su = 3.0 * n_cells / (bind_ms_step * 1e-3) if bind_ms_step > 0 else 0.0
    algorithmic = {"state_updates_per_s": su, "peak_int32_lane_ops": INT32_LANE_OPS_PEAK, "frac": su / INT32_LANE_OPS_PEAK,
                   "dp_per_step": n_binds, "cells_per_step": n_cells,
                   "note": "3 x DP cells of this run's windows (device counter) / bind-stage time (HIP events) / (1 024 SIMDs x 2.4 GHz / 2 x 64 lanes); "
                           "the pre-filter decides 94 % of the windows with 2 states per cell, so the work actually executed is less than this"}
    if valu:
        return {"bound": "valu", "kernel": kernel, "achieved": su, "peak": INT32_LANE_OPS_PEAK, "unit": "DP state updates/s vs int32 lane-ops/s", "frac": su / INT32_LANE_OPS_PEAK,
                "algorithmic": algorithmic, "pmc": pmc_state,
                "traffic": traffic, "traffic_source": src, "dp_per_s": dps, "hbm": hbm, "valu_issue": dict(valu, frac=valu["frac_guide"]),
                "note": "bound by VALU issue, not by HBM: achieved / frac = ALGORITHMIC DP state updates of this run (3 x cells, device counter) over "
                        "the stage time (HIP events) against 7.86e13 int32 lane-ops/s (1 024 SIMDs x 2.4 GHz / 2 x 64 lanes); valu_issue = the pipe's "
                        "utilisation (PMC SQ_INSTS_VALU of this workload and build, profiles/bind_pmc.json, over this run's time: it credits every "
                        "instruction issued); traffic = PMC 2*FETCH_SIZE + WRITE_SIZE per step"}
    return {"bound": "valu", "kernel": kernel, "achieved": su, "peak": INT32_LANE_OPS_PEAK, "unit": "DP state updates/s vs int32 lane-ops/s", "frac": su / INT32_LANE_OPS_PEAK,
            "algorithmic": algorithmic, "pmc": pmc_state, "traffic": traffic, "traffic_source": src, "dp_per_s": dps, "hbm": hbm, "valu_issue": None,
            "note": "bound by VALU issue, not by HBM (integer DP); achieved = algorithmic DP state updates of this run over the stage time (HIP events); "
                    "no PMC file for this build and workload: traffic and issue utilisation not reported"}


def main():
    args = parse()
    in_rank_env = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_rank_env:
        launch(args)  # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and in_rank_env and world > 1:
        raise SystemExit("bench.py: --gpus %d but torch.distributed.run started %d ranks" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.launch_only:
        return launch_only(args, rank, world)

    import torch  # first: the HIP runtime torch loads is the one libtnt_mi355x.so binds to
    import torch.distributed as dist
    from thermonucleotideblast_amd import shard, synth
    from thermonucleotideblast_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    selftest = args.collective_selftest and world == 1
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif selftest:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 1

    # ---- workload: same assays everywhere; the targets of this rank are resident before the timed region
    per_piece, my_pieces, full, full_total = deal_pieces(args.config, args.scaling, world, rank, args.targets)
    eng = None
    dbs = []
    pack_s = 0.0
    wl0 = None
    n_bases_rank = 0
    for pc in my_pieces:
        wl = synth.make_workload(args.config, n_targets=per_piece, n_assays=args.assays, shard=pc)
        if eng is None:
            eng = Engine(wl.options, wl.expanded, device=local_rank)
        t0 = time.perf_counter()
        dbs.append(eng.pack(wl.codes, wl.offsets, first_seq_index=pc * per_piece))
        pack_s += time.perf_counter() - t0
        n_bases_rank += wl.n_bases
        if wl0 is None and rank == 0:
            wl0 = wl  # the CPU baseline and the parity sample use rank 0's first piece
        else:
            del wl
    n_oligos = sum((1 if e.forward else 0) + (1 if e.reverse else 0) + (1 if e.probe else 0) for e in eng.expanded)
    n_targets_rank = per_piece * len(my_pieces)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last = {"hits": None}

    def run(n):
        """n steps.  One step = one search of every resident piece of this rank + the gather of the hit records to
        rank 0 (shard.stage_hits / finish_gather: all-gather of the counts, ONE RCCL gather of the records taken from
        device memory, page-locked receive buffer on rank 0; the same code runs on gloo in tests/test_distributed_cpu.py).
        The gather of one search travels while the next search runs; every search and every gather of the n steps is
        complete when this returns."""
        acc = {"seed": 0.0, "bind": 0.0, "pair": 0.0, "n_binds": 0, "n_cells": 0, "n_filtered": 0, "n_bound": 0, "n_hits": 0, "gather_bytes": 0, "gather_mode": None}
        if n <= 0:
            return acc
        todo = [db for _ in range(n) for db in dbs]
        eng.enqueue(todo[0])
        step_hits = step_binds = step_cells = step_filtered = step_bound = 0
        for i, db in enumerate(todo):
            if world == 1 and not selftest:
                # the hits reach the host in compact form (9 MB per search of config 2 instead of 47 MB of 1 464-byte records)
                addr, nbytes, nh = eng.collect_compact()
                tm = eng.timing()
                if i + 1 < len(todo):
                    eng.enqueue(todo[i + 1])
                else:
                    last["blob"] = (addr, nbytes, nh)  # (the parity sample unpacks the records behind the timed region)
            else:
                eng.collect_device()  # the records stay on the device ...
                tm = eng.timing()
                comp = eng.hits_compact()  # ... where they are packed: ~290 bytes per hit instead of 1 464 (windowed targets' hits included)
                st = shard.stage_compact((comp[0], comp[1]), comp[2], device="cuda", always_collective=selftest, mode=args.gather)
                acc["gather_mode"] = st.mode
                if i + 1 < len(todo):
                    eng.enqueue(todo[i + 1])
                _, counts = shard.finish_compact(st)
                acc["gather_mode"] = st.mode  # ("padded (after a failed exact transfer)" when the exact form failed)
                nh = int(sum(counts))
                acc["gather_bytes"] = int(sum(st.sizes))
            acc["seed"] += tm.seed_ms
            acc["bind"] += tm.bind_ms
            acc["pair"] += tm.pair_ms
            step_hits += nh
            step_binds += tm.n_binds
            step_cells += tm.n_cells
            step_filtered += tm.n_filtered
            step_bound += tm.n_bound
            if (i + 1) % len(dbs) == 0:
                acc["n_hits"], acc["n_binds"], acc["n_cells"] = step_hits, step_binds, step_cells  # of the last step
                acc["n_filtered"], acc["n_bound"] = step_filtered, step_bound
                step_hits = step_binds = step_cells = step_filtered = step_bound = 0
        return acc

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    acc = run(args.steps)
    bind_ms, seed_ms, pair_ms, n_binds, n_hits = acc["bind"], acc["seed"], acc["pair"], acc["n_binds"], acc["n_hits"]
    barrier()
    my_elapsed = elapsed = time.perf_counter() - t0
    if last.get("blob"):
        last["hits"] = eng.unpack(*last["blob"])  # the blob sits in page-locked memory of the context until its next search
    per_rank_ms = [1000.0 * my_elapsed / args.steps]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_ms = [1000.0 * float(x.item()) / args.steps for x in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tot = torch.tensor([n_bases_rank, n_targets_rank], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        n_bases_all, n_targets_all = int(tot[0].item()), int(tot[1].item())
    else:
        n_bases_all, n_targets_all = n_bases_rank, n_targets_rank

    if rank == 0:
        evals_per_step = n_oligos * n_bases_all
        value = evals_per_step * args.steps / elapsed
        ms_per_step = 1000.0 * elapsed / args.steps
        steps = max(1, args.steps)
        bind_ms_step, seed_ms_step = bind_ms / steps, seed_ms / steps
        alg_bytes = ALG_BYTES_PER_BASE * n_bases_rank  # this rank's kernels stream this rank's targets
        achieved = alg_bytes / (bind_ms_step * 1e-3) / 1e9 if bind_ms_step > 0 else 0.0
        seed_achieved = alg_bytes / (seed_ms_step * 1e-3) / 1e9 if seed_ms_step > 0 else 0.0
        traffic = None
        valu = None
        seed_issue = None
        pmc = os.path.join(ROOT, "profiles", "bind_pmc.json")
        pmc_state = None
        if os.path.exists(pmc) and args.config == 2 and per_piece == full and args.scaling == "weak":  # measured for this workload only
            try:
                rec = json.load(open(pmc))
                # counters from a file describe the build they were collected on: used only when the file carries the hash of
                # the sources this library was built from
                pmc_state = "current" if rec.get("csrc_sha16") == csrc_hash() else "stale"
                if pmc_state == "stale":
                    raise ValueError("profiles/bind_pmc.json was collected on another build")
                traffic = rec.get("hbm_bytes_per_step")
                # What bounds these kernels is instruction issue and LDS bank conflicts, not HBM:
                # VALU wave-instructions per step from the same PMC file over this run's bind time, against the guide's plain
                # rate: 1 024 SIMDs x 2.4 GHz / 2 cycles per wave-instruction
                si = rec.get("seed_kernel_insts_per_step")
                if si and seed_ms_step > 0:
                    # the seed stage is a chain of short dependent per-target steps: ~3 700 wave-instructions per target (round 2: 5 200)
                    n_all, n_v = sum(si.values()), si["SQ_INSTS_VALU"]
                    seed_issue = {"wave_instructions_per_step": n_all, "valu_wave_instructions_per_step": n_v, "unit": "wave-instructions/s",
                                  "achieved_all": n_all / (seed_ms_step * 1e-3), "achieved_valu": n_v / (seed_ms_step * 1e-3),
                                  "peak_guide": VALU_PEAK_GUIDE, "frac_guide_valu": n_v / (seed_ms_step * 1e-3) / VALU_PEAK_GUIDE}
                n_valu = rec.get("SQ_INSTS_VALU_per_step")
                if n_valu and bind_ms_step > 0:
                    ach = n_valu / (bind_ms_step * 1e-3)
                    valu = {"achieved": ach, "unit": "VALU wave-instructions/s", "peak_guide": VALU_PEAK_GUIDE,
                            "frac_guide": ach / VALU_PEAK_GUIDE, "wave_instructions_per_step": n_valu,
                            "rate_measured": VALU_RATE_MEASURED,
                            "note": "the filter's sweep is packed-16 maxima / saturating adds, the fill's v_max3 / v_alignbit / v_sub: mostly "
                                    "half-rate forms, whose measured rate (scripts/ubench_valu.hip) is 48 % of the guide's peak"}
            except Exception:
                traffic = None
        # the streaming part of the seed stage alone (SURVEY 8(d) means the HBM roofline for THIS scan): its share of the
        # kernel's cycles from the phase stamps of a diagnostic build (profiles/seed_phases.json) x this run's stage time
        scan_only = None
        if world == 1 and len(dbs) == 1:
            # the scan of the seed stage on its own (tnt_seed_scan_only: every k-mer of every target looked up in the query-word
            # table of every assay chunk, candidates counted, nothing behind it), timed with HIP events on the library's stream:
            # the launch SURVEY 8(d) prices against the HBM roofline (0.25 B per target base and chunk)
            try:
                scan_ms, n_cand = eng.scan_only(dbs[0], repeats=10)
                gbs = alg_bytes / (scan_ms * 1e-3) / 1e9
                scan_only = {"ms_per_pass": scan_ms, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "candidates_per_pass": int(n_cand),
                             "candidates_per_target_base": n_cand / float(n_bases_rank),
                             "source": "measured: tnt_seed_scan_only, 10 passes behind a warm-up, HIP events on the library's stream "
                                       "(achieved = 0.25 B/base x bases / time of one pass over all assay chunks; rocprofv3 summary of the same launch: profiles/)"}
            except Exception as ex:  # noqa: BLE001
                scan_only = {"error": str(ex)}
        tl = n_bases_rank // max(1, n_targets_rank)
        line = {
            "metric": "oligo x target-base evaluations per second (seed + NN-thermodynamic bind + pairing)",
            "value": value, "unit": "evals/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_per_rank": per_rank_ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "i32 DP + f32 scoring", "data": "synthetic",
            "config": {"workload": "config%d: %d expanded assays (%d oligos) x %d x %d-base synthetic targets %s"
                                   % (args.config, len(eng.expanded), n_oligos, n_targets_rank if args.scaling == "weak" else n_targets_all,
                                      tl, "per GPU" if args.scaling == "weak" else "in total, dealt to the GPUs in %d pieces" % PIECES),
                       "sharding": ("targets by sequence, no data-path collective, compact hit records gathered to rank 0 over RCCL every step (%s)"
                                    % acc["gather_mode"]) if (world > 1 or selftest) else "one GPU: no gather (N > 1: targets by sequence, hit records gathered to rank 0 over RCCL)",
                       "thresholds": "-e 40 -E 40, all other options default"},
            "targets_per_s": n_targets_all * args.steps / elapsed,
            "hits_per_step": n_hits,
            "windows_per_step": {"handed_to_the_bind_stage": n_binds, "rejected_by_the_pre_filter": acc["n_filtered"], "bound": acc["n_bound"]},
            "gather_bytes_per_search": acc["gather_bytes"],
            "pack_and_h2d_s": pack_s,
            "stage_ms_per_step": {"seed": seed_ms_step, "bind": bind_ms_step, "pair": pair_ms / steps},
            "roofline": roofline_of_bind_stage(achieved, traffic, valu, n_binds, bind_ms_step, acc["n_cells"], pmc_state),
            # the streaming stage, for which SURVEY 8(d) means the HBM roofline: the seed kernels of a step (scan + cull)
            "roofline_seed": {"bound": "hbm", "kernel": "seed stage: seed2_kernel (scan + diagonal de-duplication + first cull + emission into the buckets' queues)", "achieved": seed_achieved,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": seed_achieved / HBM_PEAK_GBS,
                              "note": "a chain of short dependent per-target steps (scan, de-duplication, counting sort, first cull, "
                                      "emission), bound by instruction issue and LDS round trips, not by the 0.25 B/base it streams "
                                      "(DESIGN.md 4.1); issue = PMC instruction counts of the stage (profiles/bind_pmc.json) over its time",
                              "issue": seed_issue, "scan_only": scan_only},
        }
        if world == 1 and not args.no_cpu_baseline and wl0 is not None:
            try:
                cb, cpu_keys, n_sample = cpu_baseline(wl0, args.cpu_seconds, args.parity_targets)
                if cb is not None and last["hits"] is not None and len(dbs) == 1:
                    line["parity_sample"] = parity_sample(last["hits"], wl0, n_sample, cpu_keys if cb["kind"] == "reference" else None,
                                                          os.cpu_count() or 1)
                    for k in ("tm_max_abs_err", "dg_max_abs_err", "hit_multiset_equal"):
                        line[k] = line["parity_sample"][k]
            except Exception as e:  # the baseline is reporting, never the product
                cb = {"error": str(e)}
            line["cpu_baseline"] = cb
    # RCCL writes its version banner to the C-level stdout, which is only flushed at exit: every rank pushes it out now,
    # so that rank 0's JSON line is the last line of the job's output
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1 or selftest:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
