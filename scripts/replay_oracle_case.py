"""Replays one case of scripts/fuzz_oracle.py (oracle against the reference's objects, build container): options, hit counts of the
oracle and of three runs of the reference, where they differ.   FUZZ_LONG=.. FUZZ_DINK=.. python scripts/replay_oracle_case.py <seed> <case>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/scripts")
import tntcheck as tc, fuzz_parity
from thermonucleotideblast_amd import abi
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    cfg, wl, codes, offsets = fuzz_parity.random_case(rng, case, max_code=16, dangle=(case % 4 == 0))
print("cfg", cfg, {n: getattr(wl.options, n) for n, _ in abi.tnt_options._fields_})
print("targets", len(offsets) - 1, "longest", int(np.diff(offsets.astype(np.int64)).max()), "assays", len(wl.expanded))
ora, ref = tc.OracleImpl(), tc.RefImpl()
t = time.time(); a = tc.sorted_hit_keys(ora.search(wl.options, wl.expanded, codes, offsets)); print("oracle", len(a), time.time() - t)
for th in (1, 1, 8):
    t = time.time(); b = tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, codes, offsets, threads=th)); print("ref threads", th, len(b), time.time() - t)
sa, sb = set(a), set(b)
print("only oracle", len(sa - sb), "only ref", len(sb - sa))
from collections import Counter
print("only-ref by (assay, target):", Counter((k[0], k[2]) for k in (sb - sa)).most_common(5))
print("only-oracle by (assay, target):", Counter((k[0], k[2]) for k in (sa - sb)).most_common(5))
x = sorted(sb - sa)[:2]
for k in x: print(k[:12])
