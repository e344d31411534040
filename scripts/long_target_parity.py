"""Parity of the long-target path at genome sizes: records of tens of megabases (seeded in windows; lists and first cull on the
device, pairing on host threads) against the oracle on the same input -- complete hit multisets, bit-exact.  One JSON line
per case.  Usage: python scripts/long_target_parity.py [case ...]   with case = cfg:megabases:records:assays"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tntcheck as tc  # the checker (oracle/): test infrastructure
from thermonucleotideblast_amd import synth
from thermonucleotideblast_amd.engine import Engine

cases = sys.argv[1:] or ["2:20:1:24", "3:20:1:24", "2:60:3:48", "5:10:1:40", "2:100:1:16"]
ora = tc.OracleImpl()
for case in cases:
    cfg, mb, n, na = case.split(":")
    cfg, mb, n, na = int(cfg), float(mb), int(n), int(na)
    wl = synth.make_workload(cfg, n_targets=int(mb * 1000), n_assays=na, seed=97)
    per = (len(wl.offsets) - 1) // n
    offsets = np.array([wl.offsets[i * per] for i in range(n)] + [wl.offsets[-1]], dtype=np.uint64)
    t0 = time.perf_counter()
    want = tc.sorted_hit_keys(ora.search(wl.options, wl.expanded, wl.codes, offsets))
    t_ora = time.perf_counter() - t0
    eng = Engine(wl.options, wl.expanded)
    db = eng.pack(wl.codes, offsets)
    eng.search(db)
    t0 = time.perf_counter()
    got = tc.sorted_hit_keys(eng.search(db))
    t_lib = time.perf_counter() - t0
    print(json.dumps({"config": cfg, "megabases": mb, "records": n, "expanded_assays": len(wl.expanded), "hits_oracle": len(want), "hits_library": len(got),
                      "hit_multiset_equal": got == want, "oracle_s": round(t_ora, 2), "library_search_s": round(t_lib, 4)}), flush=True)
    db.close()
    eng.close()
