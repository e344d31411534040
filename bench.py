#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: oligo x target-base evaluations / second.

`python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU
through torch.distributed.run (RCCL).  A step = one pass of the hot path (seed -> cull -> bind ->
pair) over one batch of synthetic targets that is already packed and resident in HBM.  Targets
shard by sequence: every rank owns an independent shard of the same size (weak scaling), there is
no data-path collective, and the hit records are gathered to rank 0 over RCCL after every step.

Workload at N = 1: BASELINE.json configs[1], "96 PCR primer pairs vs 1M x 1kb synthetic targets".
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_BASE = 0.25      # 2-bit packed targets read once (configs 1-3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config 1..5 (default 2)")
    ap.add_argument("--targets", type=int, default=None, help="override targets per GPU (default: full config size)")
    ap.add_argument("--assays", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the bounded CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective-selftest", action="store_true",
                    help="single GPU: run the multi-GPU step (RCCL process group of one rank, staged gather) instead of the plain one")
    return ap.parse_args()


def cpu_baseline(wl, budget_s):
    """Bounded sample of the same workload on the host cores: the reference's own objects
    (oracle/_ref, kind 'reference') when they are present, else this repo's restatement ('port')."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tntcheck as tc
    cores = os.cpu_count() or 1
    if tc.have_ref():
        impl, kind = tc.RefImpl(), "reference"
    elif os.path.exists(tc.ORACLE_SO):
        impl, kind = tc.OracleImpl(), "port"
    else:
        return None
    n = min(wl.n_targets, 256 * cores)
    total_t, total_n = 0.0, 0
    first = 0
    while total_t < budget_s and first < wl.n_targets:
        last = min(wl.n_targets, first + n)
        codes = wl.codes[int(wl.offsets[first]):int(wl.offsets[last])]
        offs = wl.offsets[first:last + 1] - wl.offsets[first]
        t0 = time.perf_counter()
        impl.search(wl.options, wl.expanded, codes, offs, first=first, threads=cores)
        total_t += time.perf_counter() - t0
        total_n += last - first
        first = last
        if total_t < budget_s / 4:
            n *= 2
    bases = int(wl.offsets[first])
    return {"value": wl.n_oligos * bases / total_t, "unit": "evals/s", "cores": cores, "kind": kind,
            "targets_per_s": total_n / total_t,
            "sample": "first %d targets of the same workload, %d OpenMP threads, %.1f s" % (total_n, cores, total_t)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch  # first: the HIP runtime torch loads is the one libtnt_mi355x.so binds to
    import torch.distributed as dist
    import numpy as np
    from thermonucleotideblast_amd import shard, synth
    from thermonucleotideblast_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    selftest = args.collective_selftest and world == 1
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif selftest:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))

    # ---- workload: same assays everywhere, one shard of targets per rank (resident before timing)
    full = {1: 1000, 2: 1_000_000, 3: 1_000_000, 4: 1_250_000, 5: 625_000}[args.config]
    n_targets = args.targets or full
    wl = synth.make_workload(args.config, n_targets=n_targets, n_assays=args.assays, shard=rank)
    eng = Engine(wl.options, wl.expanded, device=local_rank)
    t0 = time.perf_counter()
    db = eng.pack(wl.codes, wl.offsets, first_seq_index=rank * n_targets)
    pack_s = time.perf_counter() - t0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n):
        """n steps.  One step = one search of the resident shard + the gather of its hit records to rank 0
        (shard.stage_hits / finish_gather: all-gather of the counts, ONE RCCL gather of the padded records taken from
        device memory, page-locked receive buffer on rank 0; the same code runs on gloo in tests/test_distributed_cpu.py).
        The gather of step i travels while the search of step i + 1 runs; every search and every gather of the n steps is
        complete when this returns."""
        acc = {"seed": 0.0, "bind": 0.0, "pair": 0.0, "n_binds": 0, "n_hits": 0}
        if n <= 0:
            return acc
        eng.enqueue(db)
        for i in range(n):
            hits = eng.collect(copy=False)  # view of the library's pinned buffer
            tm = eng.timing()
            if world == 1 and not selftest:
                acc["n_hits"] = len(hits)
                if i + 1 < n:
                    eng.enqueue(db)
            else:
                st = shard.stage_hits(hits, device="cuda", device_hits=eng.hits_device(), always_collective=selftest)
                if i + 1 < n:
                    eng.enqueue(db)
                _, counts = shard.finish_gather(st)
                acc["n_hits"] = int(sum(counts))
            acc["seed"] += tm.seed_ms
            acc["bind"] += tm.bind_ms
            acc["pair"] += tm.pair_ms
            acc["n_binds"] = tm.n_binds  # DP evaluations of the last step on this rank
        return acc

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    acc = run(args.steps)
    bind_ms, seed_ms, pair_ms, n_binds, n_hits = acc["bind"], acc["seed"], acc["pair"], acc["n_binds"], acc["n_hits"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        evals_per_step = wl.n_oligos * wl.n_bases * world  # every rank holds a shard of the same size
        value = evals_per_step * args.steps / elapsed
        ms_per_step = 1000.0 * elapsed / args.steps
        bind_ms_step = bind_ms / max(1, args.steps)
        alg_bytes = ALG_BYTES_PER_BASE * wl.n_bases
        achieved = alg_bytes / (bind_ms_step * 1e-3) / 1e9 if bind_ms_step > 0 else 0.0
        traffic = None
        valu = None
        pmc = os.path.join(ROOT, "profiles", "bind_pmc.json")
        if os.path.exists(pmc) and args.config == 2 and n_targets == full:  # measured for this workload only
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_step")
                # What actually bounds the kernel: VALU issue.  Wave-instructions per step from the same
                # PMC file over this run's bind time, against 1 024 SIMDs x the sustained shader clock / the mean issue
                # cost of the fill's instruction mix (scripts/microbench/valu_rate.hip: 2.25 / 4.3 cycles by class).
                n_valu = rec.get("SQ_INSTS_VALU_per_step")
                if n_valu and bind_ms_step > 0:
                    peak = 1024 * 1.97e9 / 3.15
                    ach = n_valu / (bind_ms_step * 1e-3)
                    valu = {"achieved": ach, "peak": peak, "unit": "VALU wave-instructions/s", "frac": ach / peak}
            except Exception:
                traffic = None
        line = {
            "metric": "oligo x target-base evaluations per second (seed + NN-thermodynamic bind + pairing)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32 DP + f32 scoring", "data": "synthetic",
            "config": {"workload": "config%d: %d expanded assays (%d oligos) x %d x %d-base synthetic targets per GPU"
                                   % (args.config, len(wl.expanded), wl.n_oligos, wl.n_targets, wl.n_bases // max(1, wl.n_targets)),
                       "sharding": "targets by sequence, one shard per GPU, hit records gathered to rank 0 over RCCL every step",
                       "thresholds": "-e 40 -E 40, all other options default"},
            "targets_per_s": wl.n_targets * world * args.steps / elapsed,
            "hits_per_step": n_hits,
            "pack_and_h2d_s": pack_s,
            "stage_ms_per_step": {"seed": seed_ms / max(1, args.steps), "bind": bind_ms_step, "pair": pair_ms / max(1, args.steps)},
            "roofline": {"bound": "hbm", "kernel": "bind_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "algorithmic bytes = 0.25 B/target-base (2-bit packed targets read once); the bind "
                                 "stage is bound by integer VALU issue, not by HBM (DESIGN.md 4.2); traffic = PMC "
                                 "2*FETCH_SIZE + WRITE_SIZE per step of the full-size config (profiles/bind_pmc.json)",
                         "dp_per_s": n_binds * world / (bind_ms_step * 1e-3) if bind_ms_step > 0 else 0.0,
                         "valu_issue": valu},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(wl, args.cpu_seconds)
            except Exception as e:  # the baseline is reporting, never the product
                cb = {"error": str(e)}
            line["cpu_baseline"] = cb
    # RCCL writes its version banner to the C-level stdout, which is only flushed at exit: every rank pushes it out now,
    # so that rank 0's JSON line is the last line of the job's output
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
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
