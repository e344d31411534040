#!/usr/bin/env python3
"""Build container only: the same random cases as scripts/fuzz_parity.py, oracle (restatement) against the reference's
own objects (oracle/_ref).   python scripts/fuzz_oracle.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import tntcheck as tc  # noqa: E402
import fuzz_parity  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ora, ref = tc.OracleImpl(), tc.RefImpl()
    bad, total, skipped, t0 = 0, 0, 0, time.time()
    for case in range(n_cases):
        cfg, wl, codes, offsets = fuzz_parity.random_case(rng, case, max_code=16, dangle=(case % 4 == 0))
        res = []
        print('case', case, 'cfg', cfg, 'k', wl.options.hash_word_size, flush=True) if os.environ.get('TRACE') else None
        dangle = False
        for impl in (ora, ref):
            try:
                hits = impl.search(wl.options, wl.expanded, codes, offsets, threads=(1 if impl is ref else 8))
                if impl is ora:
                    dangle = any(b"$" in bytes(getattr(h, f)) for h in hits for f in ("forward_align", "reverse_align", "probe_align"))
                res.append(tc.sorted_hit_keys(hits))
            except AssertionError as e:
                res.append(("error", str(e)[:80]))
        # The reference cannot report a hit whose alignment holds a dangling base: compress.h:deflate_dna_seq has no
        # code for '$' and throws out of the search (amplicon_search.cpp:503, padlock_search.cpp:199); such cases
        # are checked at the binding level instead (tntref_bind_batch keeps the text uncompressed).
        if dangle and isinstance(res[1], tuple) and "Unknown symbol" in res[1][1]:
            skipped += 1
            continue
        if res[0] != res[1] and not (isinstance(res[0], tuple) and isinstance(res[1], tuple)):
            bad += 1
            print("CASE %d (seed %d) cfg %d: oracle %s vs reference %s" % (case, seed, cfg, len(res[0]) if isinstance(res[0], list) else res[0],
                                                                            len(res[1]) if isinstance(res[1], list) else res[1]))
        elif isinstance(res[0], list):
            total += len(res[0])
    print("%d cases (%d where the reference aborts on a dangle), %d hits compared, %d differ, %.0f s" % (n_cases, skipped, total, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
