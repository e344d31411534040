#!/usr/bin/env python3
"""Is the CPU checker deterministic under many OpenMP threads?  Runs the reference objects (oracle/_ref) and the oracle
on a prefix of a bench workload several times with different thread counts and prints the hits that differ between
runs, then re-runs each differing target alone on one thread."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--targets", type=int, default=262144)
    ap.add_argument("--full", type=int, default=1000000)
    a = ap.parse_args()
    import tntcheck as tc
    from thermonucleotideblast_amd import synth
    wl = synth.make_workload(a.config, n_targets=a.full)
    n = a.targets
    codes = wl.codes[:int(wl.offsets[n])]
    offs = wl.offsets[:n + 1]
    cores = os.cpu_count()
    runs = {}
    for name, impl in (("oracle", tc.OracleImpl()), ("ref", tc.RefImpl())):
        for thr in (cores, cores, max(1, cores // 8)):
            key = "%s/%d#%d" % (name, thr, len(runs))
            runs[key] = {tc.hit_key(h) for h in impl.search(wl.options, wl.expanded, codes, offs, threads=thr)}
            print(key, len(runs[key]), flush=True)
    union = set().union(*runs.values())
    common = set.intersection(*runs.values())
    odd = sorted(union - common)
    print("hits not reported by every run:", len(odd))
    ref1 = tc.RefImpl()
    for k in odd:
        t = k[2]
        c = codes[int(offs[t]):int(offs[t + 1])]
        o = np.array([0, len(c)], dtype=np.uint64)
        alone = {tc.hit_key(h) for h in ref1.search(wl.options, wl.expanded, c, o, first=t, threads=1)}
        print("target", t, "assay", k[0], "range", k[3:5], "in:", [r for r in runs if k in runs[r]], "| reference alone on one thread reports it:", k in alone)


if __name__ == "__main__":
    main()
