"""CPU, build container only: the oracle against the reference itself (oracle/_ref) on fresh seeded inputs.
Skipped where oracle/_ref is absent (it cannot be built without /root/reference)."""
import numpy as np
import pytest

import tntcheck as tc
from thermonucleotideblast_amd import abi, synth

pytestmark = pytest.mark.ref


@pytest.mark.parametrize("T,na", [(310.15, 0.05), (333.15, 0.2), (300.0, 1.0), (350.0, 0.001)])
def test_tables(ref, oracle, T, na):
    opt = abi.options_default(target_t=T, salt=na)
    a, b = ref.tables(opt), oracle.tables(opt)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("cfg,nt,na,seed", [(2, 150, 12, 11), (3, 150, 12, 12), (5, 120, 16, 13), (4, 100, 30, 14)])
def test_seeds_and_binds(ref, oracle, cfg, nt, na, seed):
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
    oligos = sorted({o for e in wl.expanded for o in (e.forward, e.reverse, e.probe) if o})
    sa = ref.seeds(wl.options, oligos, wl.codes, wl.offsets)
    assert sa == oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
    reqs = tc.make_reqs((s[0], s[1], s[2], s[3], s[4], i % 2, 9e-7 if i % 3 else 2.5e-7) for i, s in enumerate(sa[:8000]))
    ra = ref.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    rb = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    bad = [(i, tc.bind_compare_fields(ra[i], rb[i])) for i in range(len(reqs))]
    bad = [b for b in bad if b[1]]
    assert not bad, bad[:3]


@pytest.mark.parametrize("cfg,nt,na,seed,kw", [
    (1, 400, None, 21, {}),
    (2, 400, 48, 22, {}),
    (3, 400, 48, 23, {}),
    (4, 200, 40, 24, {}),
    (5, 300, 40, 25, {}),
    (5, 200, 30, 26, {"assay_format": abi.ASSAY_MIPS, "max_len": 30}),
    (5, 200, 30, 27, {"target_strand": abi.STRAND_PLUS}),
    (3, 300, 30, 28, {"single_primer_pcr": 0, "primer_clamp": 2, "probe_clamp_5": 1, "min_max_primer_clamp": 4}),
    (2, 300, 30, 29, {"min_primer_tm": 0.0, "max_primer_dg": -8.0, "max_mismatch": 2, "max_gap": 0}),
    (2, 300, 30, 30, {"hash_word_size": 6, "max_len": 250}),
])
def test_search(ref, oracle, cfg, nt, na, seed, kw):
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    a = tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    b = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    assert a == b
    assert len(a) > 0


@pytest.mark.parametrize("cfg,nt,na,seed,kw", [(2, 300, 24, 31, {}), (3, 300, 24, 32, {}), (4, 150, 30, 33, {}),
                                                (3, 300, 24, 34, {"max_len": 150, "single_primer_pcr": 0}),
                                                (2, 300, 24, 35, {"hash_word_size": 5, "max_len": 400})])
def test_first_cull(ref, oracle, cfg, nt, na, seed, kw):
    """Row a4 at unit level: the oracle's restatement of amplicon()'s early exits and cull_oligo_match against the reference's
    own functions (match_oligo_to_*_strand + cull_oligo_match): the same seeds survive.  (The strand that is bound first
    comes from counts the reference takes one node past its list, SURVEY Q4: compared where they happen to agree.)"""
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    a = oracle.first_cull(wl.options, wl.expanded, wl.codes, wl.offsets)
    b = ref.first_cull(wl.options, wl.expanded, wl.codes, wl.offsets)
    assert sorted(map(tuple, a[:, :6])) == sorted(map(tuple, b[:, :6])) and len(a) > 100


LONG_RECORD_CASES = [(2, 2000, 24, {}), (3, 1500, 24, {}), (3, 1000, 16, {"max_len": 400, "single_primer_pcr": 0}), (5, 600, 24, {}),
                     (2, 300, 16, {"hash_word_size": 5})]


def _long_record(cfg, nt, na, kw):
    """A workload whose first nt - 5 targets are one record (0.3 to 2 Mb) in front of five short ones."""
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=140 + cfg)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    lens = np.diff(wl.offsets.astype(np.int64))
    cut = nt - 5
    offsets = np.array([0, int(lens[:cut].sum())] + [int(lens[:cut + i + 1].sum()) for i in range(5)], dtype=np.uint64)
    return wl, offsets


@pytest.mark.parametrize("cfg,nt,na,kw", LONG_RECORD_CASES)
def test_search_long_records(ref, oracle, cfg, nt, na, kw):
    """The oracle against the reference where the lists are long (thousands of seeds per (target, assay): the first cull's
    successor counts, Q4, and the merge order of std::list::sort under sort_by_oligo_loc, Q5, act on whole-record lists) --
    the link between the reference and the GPU parity tests of the long-target path, which compare with the oracle."""
    wl, offsets = _long_record(cfg, nt, na, kw)
    a = tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, offsets))
    b = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, offsets))
    assert a == b
    assert len(a) > 20


@pytest.mark.parametrize("cfg,nt,na,kw", [c for c in LONG_RECORD_CASES if c[0] != 5])
def test_first_cull_long_records(ref, oracle, cfg, nt, na, kw):
    """cull_oligo_match on the lists of a long record: the same seeds survive, and the strand bound first agrees (on these inputs
    the element the reference reads one node past the list, Q4, does not decide)."""
    wl, offsets = _long_record(cfg, nt, na, kw)
    a = oracle.first_cull(wl.options, wl.expanded, wl.codes, offsets)
    b = ref.first_cull(wl.options, wl.expanded, wl.codes, offsets)
    assert sorted(map(tuple, a[:, :6])) == sorted(map(tuple, b[:, :6])) and len(a) > 5000
    assert sorted(map(tuple, a)) == sorted(map(tuple, b))


@pytest.mark.parametrize("kind,seed", [("probe", 1), ("probe", 2), ("padlock", 3), ("pcr", 4), ("pcr", 5)])
def test_long_oligos(ref, oracle, kind, seed):
    """Oligos of 65..128 bases (the library's long instances): seeds, binds, hits and self-structure of the oracle against
    the reference."""
    import longwl
    wl = longwl.make(kind, seed)
    oligos = sorted({o for e in wl.expanded for o in (e.forward, e.reverse, e.probe) if o})
    assert max(len(o) for o in oligos) > 64
    sa = ref.seeds(wl.options, oligos, wl.codes, wl.offsets)
    assert sa == oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
    reqs = tc.make_reqs((s[0], s[1], s[2], s[3], s[4], i % 2, 9e-7 if i % 3 else 2.5e-7) for i, s in enumerate(sa[:6000]))
    ra = ref.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    rb = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    bad = [b for b in ((i, tc.bind_compare_fields(ra[i], rb[i])) for i in range(len(reqs))) if b[1]]
    assert not bad, bad[:3]
    assert sum(1 for r in ra if r.status == 0) > 5
    a = tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    assert a == tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets)) and len(a) > 10
    for o in oligos[:4]:
        assert ref.structure(wl.options, o, oligos[0]) == oracle.structure(wl.options, o, oligos[0])


@pytest.mark.parametrize("kind,seed", [("pcr", 0), ("pcr", 1), ("taqman", 2), ("taqman", 3), ("padlock", 4), ("padlock", 5)])
def test_primers_beyond_128_bases(ref, oracle, kind, seed):
    """PCR primers, TaqMan probes and padlock arms of 129..1 024 bases (nuc_cruc.h:80: every role may be that long): seeds,
    binds and hits of the oracle against the reference."""
    import longwl
    wl = longwl.make_primers_beyond_128(kind, seed)
    oligos = sorted({o for e in wl.expanded for o in (e.forward, e.reverse, e.probe) if o})
    assert max(len(o) for o in oligos) > 128
    sa = ref.seeds(wl.options, oligos, wl.codes, wl.offsets)
    assert sa == oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
    reqs = tc.make_reqs((s[0], s[1], s[2], s[3], s[4], i % 2, 9e-7 if i % 3 else 2.5e-7) for i, s in enumerate(sa[::max(1, len(sa) // 1500)]))
    ra = ref.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    rb = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    bad = [b for b in ((i, tc.bind_compare_fields(ra[i], rb[i])) for i in range(len(reqs))) if b[1]]
    assert not bad, bad[:3]
    a = tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    assert a == tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets)) and len(a) >= 4
    if kind != "padlock":
        ca, cb = ref.first_cull(wl.options, wl.expanded, wl.codes, wl.offsets), oracle.first_cull(wl.options, wl.expanded, wl.codes, wl.offsets)
        assert np.array_equal(np.asarray(ca), np.asarray(cb))


@pytest.mark.parametrize("kw", [{}, {"allow_dangle_5": 1, "allow_dangle_3": 1}, {"salt": 1.0, "target_t": 300.0}])
def test_oligo_structure(ref, oracle, kw):
    """hairpin / homodimer / heterodimer Tm of oligos (tntblast_local.cpp:656-686) on fresh random + designed oligos."""
    rng = np.random.default_rng(77)
    oligos = [synth._rand_oligo(rng, 12, 45) for _ in range(120)]
    for _ in range(120):
        stem = synth._rand_oligo(rng, 3, 10)
        loop = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(3, 9))))
        oligos.append(synth._rand_oligo(rng, 1, 5)[:int(rng.integers(0, 5))] + stem + loop + synth.revcomp(stem) + synth._rand_oligo(rng, 1, 4)[:int(rng.integers(0, 4))])
    opt = abi.options_default(**kw)
    nz = 0
    for i, o in enumerate(oligos):
        p = oligos[(i * 11 + 5) % len(oligos)]
        a = ref.structure(opt, o, p, 9e-7, 2e-7)
        assert a == oracle.structure(opt, o, p, 9e-7, 2e-7), (o, p)
        nz += sum(1 for v in a if v)
    assert nz > 200


def test_expand_matches_reference(checkers):
    import ctypes as C
    if not checkers.have_ref():
        pytest.skip("no oracle/_ref")
    lib = C.CDLL(checkers.REF_SO)
    buf = C.create_string_buffer(1 << 20)
    n, d = C.c_int(), C.c_int()
    for s in ("ACGT", "TATTCTCYCTACAAAAGCATTGGG", "ARYNKMSWACGI", "BDHVN", "NNN"):
        assert lib.tntref_expand(s.encode(), buf, len(buf), C.byref(n), C.byref(d)) == 0
        assert buf.value.decode().split("\n")[:-1] == synth.expand_oligo(s)
        assert d.value == synth.degeneracy(s) == n.value


def test_random_option_sets(checkers):
    """A slice of scripts/fuzz_oracle.py (random options x target shapes, the generator of the GPU fuzz)."""
    import os
    import subprocess
    import sys
    if not checkers.have_ref():
        pytest.skip("no oracle/_ref")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_oracle.py"), "24", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 differ" in r.stdout


def _conflict_workload(kind, seed):
    """Assay sets in which one oligo string is evaluated under two strand concentrations / threshold sets on the same
    target: the inputs on which the reference's per-target BindCache (a6) changes results."""
    if kind == "taqman":  # the probe of a TaqMan assay is the forward primer of a PCR assay listed behind / in front of it
        wl = synth.make_workload(3, n_targets=300, n_assays=12, seed=seed)
        extra = [synth.Assay("swap%d" % i, a.probe, a.reverse, None) for i, a in enumerate(wl.assays)]
        assays = [x for i, pair in enumerate(zip(wl.assays, extra)) for x in (pair if i % 2 == 0 else pair[::-1])]
        wl.options.min_primer_tm, wl.options.min_probe_tm = 45.0, 30.0
        wl.options.primer_clamp, wl.options.probe_clamp_5 = 2, 1
    elif kind == "degenerate":  # an explicit expansion listed next to its degenerate parent: another Ct divisor
        wl = synth.make_workload(4, n_targets=200, n_assays=20, seed=seed)
        assays = []
        for i, a in enumerate(wl.assays):
            assays.append(a)
            if synth.degeneracy(a.forward) > 1 or synth.degeneracy(a.reverse) > 1:
                assays.append(synth.Assay("expl%d" % i, synth.expand_oligo(a.forward)[0], synth.expand_oligo(a.reverse)[-1], None))
        assert len(assays) > len(wl.assays)
    else:  # padlock: the upstream arm of one assay is the downstream arm of the next (5' clamp against 3' clamp)
        wl = synth.make_workload(5, n_targets=200, n_assays=20, seed=seed)
        pads = [a for a in wl.assays if a.forward and a.reverse]
        extra = [synth.Assay("swap%d" % i, a.reverse, a.forward, None) for i, a in enumerate(pads)]
        assays = list(wl.assays) + extra
        wl.options.probe_clamp_5, wl.options.probe_clamp_3 = 3, 1
    return wl, synth.expand_assays(assays)


@pytest.mark.parametrize("kind,seed", [("taqman", 61), ("taqman", 62), ("degenerate", 63), ("padlock", 64)])
def test_search_with_bind_cache_conflicts(ref, oracle, kind, seed):
    """Row a6: oracle == reference on assay sets whose results hang on the BindCache (first evaluation of an (oligo, window)
    per target and strand wins, thresholds re-applied to its partial record on a hit, bind_oligo.cpp:508-512,750-803)."""
    wl, ex = _conflict_workload(kind, seed)
    b = tc.sorted_hit_keys(oracle.search(wl.options, ex, wl.codes, wl.offsets))
    c = tc.sorted_hit_keys(oracle.search(wl.options, ex, wl.codes, wl.offsets, bind_cache=False))
    # The reference is not deterministic on these inputs: amplicon()'s first cull reads the strand bits of the node behind its
    # list (stack contents), that decides which strand is bound first, and here -- unlike on ordinary assay sets --
    # which evaluation of a shared oligo comes first decides hits.  About one single-thread run in six of the taqman-61 case
    # comes out with another strand choice on one target.  The oracle reads "nothing behind the list": it has to equal the
    # reference's outcome of some run out of a few.
    a = None
    for attempt in range(8):
        a = tc.sorted_hit_keys(ref.search(wl.options, ex, wl.codes, wl.offsets, threads=1))
        if a == b:
            break
    assert a == b and len(a) > 0
    if kind != "padlock":
        assert b != c  # the memo does change these results


def test_bind_cache_on_a_long_record(ref, oracle):
    """Row a6 on records of 100 kb (the library seeds them in windows): the memo is per target whatever its length
    (tntblast_local.cpp:503-506) -- oracle == reference on the TaqMan / PCR assay set that shares its probes with primers."""
    wl, ex = _conflict_workload("taqman", 61)
    offsets = np.concatenate([wl.offsets[:-1:100], wl.offsets[-1:]]).astype(np.uint64)
    b = tc.sorted_hit_keys(oracle.search(wl.options, ex, wl.codes, offsets))
    a = None
    for attempt in range(8):  # (the reference's strand choice is not deterministic, see above)
        a = tc.sorted_hit_keys(ref.search(wl.options, ex, wl.codes, offsets, threads=1))
        if a == b:
            break
    assert a == b and len(a) > 20
    assert b != tc.sorted_hit_keys(oracle.search(wl.options, ex, wl.codes, offsets, bind_cache=False))


@pytest.mark.parametrize("seed", [1, 2])
def test_oligos_beyond_128_bases(ref, oracle, seed):
    """Hybridisation probes of 150 to 1 000 bases (the reference's MAX_SEQUENCE_LENGTH is 1 024): seeds, binds, hits and
    self-structure of the oracle against the reference -- what pins the checker for the wavefront-per-window kernel's
    own lengths."""
    import longwl
    wl = longwl.make_beyond_128(seed, lengths=(150, 1000) if seed == 1 else (130, 260, 520), n_targets=4, tl=1500)
    oligos = sorted({e.probe for e in wl.expanded})
    assert max(len(o) for o in oligos) >= 520
    sa = ref.seeds(wl.options, oligos, wl.codes, wl.offsets)
    assert sa == oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
    pick = sa[::max(1, len(sa) // 40)]
    reqs = tc.make_reqs((s[0], s[1], s[2], s[3], s[4], 1, 2.5e-7) for s in pick)
    ra = ref.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    rb = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    bad = [b for b in ((i, tc.bind_compare_fields(ra[i], rb[i])) for i in range(len(reqs))) if b[1]]
    assert not bad, bad[:3]
    a = tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, wl.offsets, threads=2))
    assert a == tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets)) and len(a) >= 3
    for o in oligos[:2]:
        assert ref.structure(wl.options, o, oligos[-1]) == oracle.structure(wl.options, o, oligos[-1])


@pytest.mark.parametrize("kind,seed", [("probe", 1), ("pcr", 4), ("beyond", 2)])
def test_dinkelbach_with_long_oligos(ref, oracle, kind, seed):
    """--dinkelbach with oligos of 65 to 520 bases: binds and hits of the oracle against the reference (pins the checker
    for the wavefront-per-window kernel's Dinkelbach form)."""
    import longwl
    wl = longwl.make(kind, seed) if kind != "beyond" else longwl.make_beyond_128(seed, lengths=(129, 200, 520), n_targets=10, tl=2400)
    wl.options.use_dinkelbach = 1
    oligos = sorted({o for e in wl.expanded for o in (e.forward, e.reverse, e.probe) if o})
    sa = ref.seeds(wl.options, oligos, wl.codes, wl.offsets)
    pick = sa[::max(1, len(sa) // 300)]
    reqs = tc.make_reqs((s[0], s[1], s[2], s[3], s[4], i % 2, 2.5e-7) for i, s in enumerate(pick))
    ra = ref.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    rb = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, reqs)
    bad = [b for b in ((i, tc.bind_compare_fields(ra[i], rb[i])) for i in range(len(reqs))) if b[1]]
    assert not bad, bad[:3]
    a = tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, wl.offsets, threads=2))
    assert a == tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets)) and len(a) >= 5
