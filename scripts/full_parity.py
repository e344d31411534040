#!/usr/bin/env python3
"""All targets of a benchmark configuration against the oracle, once (on the GPU box, through gpurun):
    python scripts/full_parity.py [CONFIG] [OUT.json]
The library searches the whole database of `bench.py --config CONFIG` (per-GPU shard size); the oracle (CPU restatement, pinned
against the reference: tests/test_oracle_vs_reference.py) searches the same targets in slices on all host cores; the complete hit
multisets -- every field, floats as bit patterns, alignment text -- are compared slice by slice.  bench.py's own parity sample
covers the first 65 536 targets of config 2 in every run; this is the one-off check of all of them."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
import tntcheck as tc  # noqa: E402
from thermonucleotideblast_amd import synth  # noqa: E402
from thermonucleotideblast_amd.engine import Engine  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "full_parity_config%d.json" % cfg)
    n_targets = int(os.environ.get("NT", "0")) or bench.deal_pieces(cfg, "weak", 1, 0)[0]
    wl = synth.make_workload(cfg, n_targets=n_targets)
    eng = Engine(wl.options, wl.expanded, device=0)
    db = eng.pack(wl.codes, wl.offsets)
    t0 = time.perf_counter()
    hits = eng.search(db)
    gpu_s = time.perf_counter() - t0
    by_slice = {}
    step = int(os.environ.get("SLICE", "65536"))
    for h in hits:
        by_slice.setdefault(h.seq_index // step, []).append(tc.hit_key(h))
    oracle = tc.OracleImpl()
    cores = os.cpu_count() or 1
    n_cpu = n_equal = 0
    differ = []
    t_cpu = 0.0
    for s, first in enumerate(range(0, wl.n_targets, step)):
        last = min(wl.n_targets, first + step)
        codes = wl.codes[int(wl.offsets[first]):int(wl.offsets[last])]
        offs = wl.offsets[first:last + 1] - wl.offsets[first]
        t1 = time.perf_counter()
        want = sorted(tc.hit_key(h) for h in oracle.search(wl.options, wl.expanded, codes, offs, first=first, threads=cores))
        t_cpu += time.perf_counter() - t1
        got = sorted(by_slice.get(s, []))
        n_cpu += len(want)
        if got == want:
            n_equal += 1
        else:
            differ.append({"slice": s, "first_target": first, "hits_gpu": len(got), "hits_oracle": len(want),
                           "targets": sorted({k[2] for k in set(got) ^ set(want)})[:16]})
    rec = {"config": cfg, "targets": wl.n_targets, "expanded_assays": len(wl.expanded), "hits_gpu": len(hits), "hits_oracle": n_cpu,
           "slices": (wl.n_targets + step - 1) // step, "slices_equal": n_equal, "hit_multiset_equal": not differ and len(hits) == n_cpu,
           "differ": differ, "gpu_search_s": gpu_s, "oracle_s": t_cpu, "oracle_threads": cores,
           "csrc_sha16": bench.csrc_hash(),
           "compared": "complete hit records: integer fields, float32 bit patterns of Tm / dH / dS, alignment texts"}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))
    db.close()
    eng.close()
    sys.exit(0 if rec["hit_multiset_equal"] else 1)


if __name__ == "__main__":
    main()
