#!/usr/bin/env python3
"""Randomised differential test on the GPU box: HIP path vs the oracle over random option sets, assay mixes and
target shapes (lengths, IUPAC content, repeats).  Every case compares the complete hit multisets (coordinates, counts,
float32 bit patterns, alignment text).   python scripts/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import longwl  # noqa: E402
import tntcheck as tc  # noqa: E402
from thermonucleotideblast_amd import abi, synth  # noqa: E402

SHARED_FRACTION = float(os.environ.get("FUZZ_SHARED", "0.15"))  # share of the cases whose assay list uses an oligo under two classes (row a6)
DINK_FRACTION = float(os.environ.get("FUZZ_DINK", "0"))  # share of the cases run with --dinkelbach (any oligo length)
LONG_FRACTION = float(os.environ.get("FUZZ_LONG", "0"))  # share of the cases that use oligos of 65..128 bases (tests/longwl.py)


def random_case(rng, case, max_code=18, dangle=True):
    """max_code: exclusive upper bound of the target codes sprinkled in (the reference itself throws on DB_GAP /
    DB_UNKNOWN inside an amplicon, seq.h hash_base_to_ascii, so the oracle-vs-reference runs stay below 16)."""
    cfg = int(rng.choice([1, 2, 3, 3, 4, 5, 5]))
    nt = int(rng.integers(40, 260))
    na = None if cfg == 1 else int(rng.integers(4, 40))
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=int(rng.integers(1, 1 << 30)))
    if LONG_FRACTION > 0.0:  # (its own stream: the sequence of ordinary cases does not depend on the share)
        lrng = np.random.default_rng([case, 977])
        if lrng.random() < LONG_FRACTION:
            kind = ("pcr", "probe", "padlock", "beyond128", "pcr>128", "taqman>128", "padlock>128")[int(lrng.integers(0, 7))]
            cfg = 2 if kind.startswith("pcr") else (3 if kind.startswith("taqman") else 5)
            if kind.endswith(">128"):  # primers, TaqMan probes, padlock arms of more than 128 bases: segments that feed one list
                wl = longwl.make_primers_beyond_128(kind[:-4], int(lrng.integers(1, 1 << 30)), n_targets=8, tl=2200)
            elif kind == "beyond128":  # probes of 129 to 600 bases: the wavefront-per-window kernel, seeds in segments, long texts
                wl = longwl.make_beyond_128(int(lrng.integers(1, 1 << 30)), lengths=tuple(int(x) for x in lrng.integers(129, 600, 3)), n_targets=8, tl=1800)
            else:
                wl = longwl.make(kind, int(lrng.integers(1, 1 << 30)), n_targets=min(nt, 120))
            nt = wl.n_targets
    o = wl.options
    pick = lambda *v: v[int(rng.integers(0, len(v)))]  # noqa: E731
    o.min_primer_tm = float(pick(0.0, 30.0, 40.0, 45.0, 52.0))
    o.min_probe_tm = float(pick(0.0, 35.0, 40.0, 48.0))
    o.max_primer_tm = float(pick(9999.0, 9999.0, 65.0))
    o.max_primer_dg = float(pick(0.0, 0.0, -6.0, -10.0))
    o.max_probe_dg = float(pick(0.0, -5.0, -9.0))
    o.salt = float(pick(0.05, 0.05, 0.2, 0.01, 0.9))
    o.primer_strand = float(pick(9e-7, 9e-7, 2e-7, 5e-6))
    o.probe_strand = float(pick(2.5e-7, 9e-7, 1e-8))
    o.asymmetric_strand_ratio = float(pick(1.0, 1.0, 0.2, 5.0))
    o.target_t = float(pick(310.15, 310.15, 300.0, 330.0))
    o.primer_clamp = int(pick(0, 0, 1, 3, 6))
    o.min_max_primer_clamp = int(pick(-1, -1, 2, 8))
    o.probe_clamp_5 = int(pick(0, 0, 2))
    o.probe_clamp_3 = int(pick(0, 0, 3))
    o.max_gap = int(pick(999, 999, 0, 1))
    o.max_mismatch = int(pick(999, 999, 1, 4))
    o.max_poly_degen = int(pick(3, 3, 0, 8))
    o.max_len = int(pick(2000, 2000, 150, 400, 0 if cfg == 5 else 800))
    o.hash_word_size = int(pick(7, 7, 7, 5, 6, 8, 7, 7, 4, 3))  # (3 and 4: every target is seeded in windows, the seed-list repairs run)
    o.single_primer_pcr = int(pick(1, 1, 0))
    o.allow_dangle_5 = int(pick(0, 0, 1))
    o.allow_dangle_3 = int(pick(0, 0, 1))
    if DINK_FRACTION > 0.0 and np.random.default_rng([case, 313]).random() < DINK_FRACTION:  # (its own stream, as above)
        o.use_dinkelbach = 1
    if not dangle:  # same random stream, options off
        o.allow_dangle_5 = o.allow_dangle_3 = 0
    o.target_strand = int(pick(abi.STRAND_BOTH, abi.STRAND_BOTH, abi.STRAND_PLUS, abi.STRAND_MINUS))
    if cfg == 5:
        o.assay_format = int(pick(abi.ASSAY_PADLOCK, abi.ASSAY_PADLOCK, abi.ASSAY_MIPS))
    codes, offsets = wl.codes, wl.offsets
    shape = int(rng.integers(0, 5))
    if shape == 2 and nt <= 21:  # (the few targets of a case with probes beyond 128 bases stay as they are)
        shape = 0
    if shape == 1:   # ragged lengths incl. empty and shorter-than-word targets
        lens = np.minimum(rng.integers(0, 1000, nt), np.diff(offsets).astype(np.int64))
        lens[rng.integers(0, nt, 3)] = [0, 3, 6]
        keep = np.zeros(len(codes), dtype=bool)
        for t in range(nt):
            keep[int(offsets[t]):int(offsets[t]) + int(lens[t])] = True
        codes = codes[keep]
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    elif shape == 2:  # a few long targets: seeded in windows, culled and paired whole (no guards on the options: the lists
        # of a windowed target are the whole target's lists)
        k = int(rng.integers(20, nt))
        offsets = np.concatenate([[0], offsets[k:]]).astype(np.uint64)
    elif shape == 3:  # IUPAC codes, N-runs, unknown and gap codes sprinkled in
        codes = codes.copy()
        n = int(rng.integers(10, 400))
        codes[rng.integers(0, len(codes), n)] = rng.integers(4, max_code, n).astype(np.uint8)
        p = int(rng.integers(0, len(codes) - 40))
        codes[p:p + 30] = 15
    elif shape == 4:  # low complexity: tandem repeats pasted over some targets
        codes = codes.copy()
        for _ in range(6):
            unit = rng.integers(0, 4, int(rng.integers(1, 5))).astype(np.uint8)
            p = int(rng.integers(0, len(codes) - 200))
            codes[p:p + 120] = np.resize(unit, 120)
    if o.hash_word_size <= 4:
        # -W 3 / -W 4 seed every position dozens of times: a smaller case keeps the CPU checkers (the single-threaded
        # reference in scripts/fuzz_oracle.py above all) at seconds per case
        nt4 = min(len(offsets) - 1, 6 if o.hash_word_size == 3 else 16)
        # (a glued 100-kb target at -W 3 keeps the CPU checker busy for half an hour, sixteen 6-kb ones for minutes)
        lens = np.minimum(np.diff(offsets[:nt4 + 1].astype(np.int64)), 1200 if o.hash_word_size == 3 else 3000)
        keep = np.zeros(len(codes), dtype=bool)
        for t in range(nt4):
            keep[int(offsets[t]):int(offsets[t]) + int(lens[t])] = True
        codes = codes[keep]
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        wl.expanded = wl.expanded[:6]
    return cfg, wl, codes, offsets


def share_oligos(case, wl, offsets):
    """Row a6: some assays get a sibling that uses one of their oligos under another strand concentration / threshold set
    (the probe as a primer, the same pair at another Ct divisor, padlock arms swapped) -- the reference's per-target
    BindCache then decides, and the library's cache replay has to follow it.  Own random stream; whole-target work units
    only (the replay does not pair windowed targets)."""
    rng = np.random.default_rng([case, 4243])
    if rng.random() >= SHARED_FRACTION or int(np.diff(offsets.astype(np.int64)).max(initial=0)) > 20000 or wl.options.hash_word_size < 6:
        return False
    out, did = [], 0
    for e in wl.expanded:
        sib = None
        if e.forward and e.reverse and rng.random() < 0.5:
            r = rng.random()
            if e.probe and r < 0.5:
                sib = synth.ExpandedAssay(e.id, 0, e.probe, e.reverse, None, 1, e.reverse_degen, 1)
            elif r < 0.75:
                sib = synth.ExpandedAssay(e.id, 0, e.forward, e.reverse, e.probe, e.forward_degen * 2, e.reverse_degen, e.probe_degen)
            else:
                sib = synth.ExpandedAssay(e.id, 0, e.reverse, e.forward, None, e.reverse_degen, e.forward_degen, 1)
        for x in ((sib, e) if (sib is not None and rng.random() < 0.5) else (e, sib)):
            if x is not None:
                out.append(synth.ExpandedAssay(x.id, did, x.forward, x.reverse, x.probe, x.forward_degen, x.reverse_degen, x.probe_degen))
                did += 1
    wl.expanded = out
    return True


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None  # replay one case of the sequence
    rng = np.random.default_rng(seed)
    ora = tc.OracleImpl()
    bad = 0
    limited = 0
    shared = 0
    t0 = time.time()
    total_hits = 0
    for case in range(n_cases):
        cfg, wl, codes, offsets = random_case(rng, case)
        if only is not None and case != only:
            continue
        shared += 1 if share_oligos(case, wl, offsets) else 0
        t_case = time.time()
        if case and case % 50 == 0:  # (progress, flushed: a run cut short still tells how far it got)
            print("... %d cases, %d hits compared, %d differ, %.0f s" % (case, total_hits, bad, time.time() - t0), flush=True)
        try:
            want = tc.sorted_hit_keys(ora.search(wl.options, wl.expanded, codes, offsets))
        except AssertionError as e:  # the oracle refuses (e.g. word size vs oligo length): both must refuse
            want = ("error", str(e)[:60])
        t_ora = time.time() - t_case
        try:
            from thermonucleotideblast_amd.engine import Engine
            eng = Engine(wl.options, wl.expanded)
            db = eng.pack(codes, offsets)
            got = tc.sorted_hit_keys(eng.search(db), text_of=eng.long_text)  # (alignments beyond the 448-byte fields: cut where the oracle cuts)
            db.close()
            eng.close()
        except Exception as e:
            got = ("error", str(e)[:60])
        t_gpu = time.time() - t_case - t_ora
        if os.environ.get("FUZZ_SLOW") and time.time() - t_case > float(os.environ["FUZZ_SLOW"]):
            print("slow case %d: cfg %d k %d, %d targets (longest %d), oracle %.1f s, library %.1f s" % (
                case, cfg, wl.options.hash_word_size, len(offsets) - 1, int(np.diff(offsets.astype(np.int64)).max()), t_ora, t_gpu), flush=True)
        ok = got == want or (isinstance(got, tuple) and isinstance(want, tuple))
        if not ok and isinstance(got, tuple) and ("overflowed the LDS buffer" in got[1] or "seeds of one assay on one target" in got[1]):
            # the documented capacity limit (more seeds of one assay chunk inside one seed work unit than
            # the LDS lists hold -- word size 5 on long targets; more than 1 048 576 seeds of one assay on one target after the first cull):
            # reported as TNT_E_LIMIT, never a silent difference
            limited += 1
            ok = True
        if isinstance(want, list):
            total_hits += len(want)
        if not ok:
            bad += 1
            print("CASE %d (seed %d) cfg %d DIFFERS: hip %s vs oracle %s; options %s" % (
                case, seed, cfg, len(got) if isinstance(got, list) else got, len(want) if isinstance(want, list) else want,
                {n: getattr(wl.options, n) for n, _ in abi.tnt_options._fields_}))
            if isinstance(got, list) and isinstance(want, list):
                sg, sw = set(got), set(want)
                print("   only hip:", list(sg - sw)[:1], "\n   only oracle:", list(sw - sg)[:1])
                if only is not None:
                    import pickle
                    pickle.dump({"options": {n: getattr(wl.options, n) for n, _ in abi.tnt_options._fields_}, "expanded": [(e.id, e.degen_id, e.forward, e.reverse, e.probe, e.forward_degen, e.reverse_degen, e.probe_degen) for e in wl.expanded],
                                 "codes": codes, "offsets": offsets, "only_hip": list(sg - sw), "only_oracle": list(sw - sg)},
                                open(os.path.join(ROOT, "gpurun_out", "fuzz_case.pkl"), "wb"))
    print("%d cases (%d with an oligo shared across classes), %d hits compared, %d differ, %d refused at the seed-list limit, %.0f s" % (
        n_cases, shared, total_hits, bad, limited, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
