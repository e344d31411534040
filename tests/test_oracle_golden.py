"""CPU: the oracle (oracle/tnt_oracle.cpp) against the committed golden vectors produced by the reference."""
import numpy as np
import pytest

import golden_util as gu
import tntcheck as tc
from thermonucleotideblast_amd import abi, synth


def test_g1_tables(oracle):
    fx = gu.load("g1_tables.json")
    for key, dg in fx["delta_g"].items():
        T, na = (float(x) for x in key.split("_"))
        t = oracle.tables(abi.options_default(target_t=T, salt=na))
        assert np.array_equal(t["delta_g"], np.array(dg, dtype=np.int32)), key
    t = oracle.tables(abi.options_default())
    assert np.array_equal(t["H"], np.array(fx["param_H_bits"], dtype=np.uint32))
    assert np.array_equal(t["S"], np.array(fx["param_S_bits"], dtype=np.uint32))
    assert np.array_equal(t["loop_S"], np.array(fx["loop_S_bits"], dtype=np.uint32))
    assert np.array_equal(t["bulge_S"], np.array(fx["bulge_S_bits"], dtype=np.uint32))
    assert np.array_equal(t["scalars"], np.array(fx["scalars_bits"], dtype=np.uint32))
    # spot values, SURVEY appendix B
    dg = np.array(fx["delta_g"]["310.15_0.05"])
    AT, CG, GC, TA = 3, 9, 15, 21
    assert (dg[AT * 49 + AT], dg[AT * 49 + CG], dg[CG * 49 + GC], dg[TA * 49 + AT]) == (-6580, -10980, -18280, -2380)
    assert ((dg < 0).sum(), (dg > 0).sum(), (dg == 0).sum()) == (97, 2304, 0)


def test_g2_binds(oracle):
    fx = gu.load("g2_binds.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    reqs = tc.make_reqs(fx["requests"])
    for case in fx["cases"]:
        out = oracle.bind_batch(gu.options_from(case["options"]), fx["oligos"], codes, offsets, reqs)
        bad = [(i, gu.compare_bind(e, gu.bind_record(out[i]))) for i, e in enumerate(case["expected"])]
        bad = [b for b in bad if b[1]]
        assert not bad, "%s: %d of %d differ, first %r" % (case["name"], len(bad), len(case["expected"]), bad[:2])


def test_g3_structure(oracle):
    g = gu.load("g3_structure.json.gz")
    so = g["oligos"]
    for case in g["cases"]:
        opt = gu.options_from(case["options"])
        for i, o in enumerate(so):
            got = oracle.structure(opt, o, so[(i * 7 + 3) % len(so)], g["ct_a"], g["ct_b"])
            assert list(got) == case["expected"][i], (case["name"], o)


def test_g4_seeds(oracle):
    fx = gu.load("g4_seeds.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    for k, expected in fx["seeds"].items():
        n_t = 100 if k == "7" else 12
        got = oracle.seeds(abi.options_default(hash_word_size=int(k)), fx["oligos"], codes[:int(offsets[n_t])], offsets[:n_t + 1])
        assert [list(s) for s in got] == expected, "word size %s" % k


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_g5_search(oracle, cfg):
    fx = gu.load("g5_search_cfg%d.json.gz" % cfg)
    expanded, codes, offsets = gu.g5_inputs(fx)  # stored in the fixture, not regenerated from RNG seeds
    hits = oracle.search(gu.options_from(fx["options"]), expanded, codes, offsets)
    assert len(hits) == fx["n_hits"] > 0
    assert gu.hits_as_keys(hits) == fx["hits"]


def test_g5_readme_example(oracle):
    """README.md:130-203 (gibb-marburg): dG/dH/dS of the README still hold, Tm per the current code."""
    fx = gu.load("g5_readme.json.gz")
    ex = synth.expand_assays([synth.Assay("gibb-marburg", *fx["assay"])])
    t = fx["target"]
    hits = oracle.search(gu.options_from(fx["options"]), ex, synth.encode(t), np.array([0, len(t)], dtype=np.uint64))
    assert gu.hits_as_keys(hits) == fx["hits"]
    h = hits[0]
    assert (h.amplicon_start, h.amplicon_stop, h.probe_start, h.probe_stop) == (8, 154, 30, 52)
    assert abs(h.forward_dH - -135.5) < 1e-4 and abs(h.forward_dS - -0.382533) < 1e-6
    assert abs(h.reverse_dH - -146.5) < 1e-4 and abs(h.reverse_dS - -0.414653) < 1e-6
    assert abs(h.probe_dH - -180.2) < 1e-4 and abs(h.probe_dS - -0.506923) < 1e-6
    assert abs(h.forward_tm - 57.1794) < 1e-4 and abs(h.reverse_tm - 58.061) < 1e-4 and abs(h.probe_tm - 62.3354) < 1e-4


def test_threads_do_not_change_results(oracle):
    wl = synth.make_workload(3, n_targets=300, n_assays=20)
    a = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets, threads=1))
    b = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets, threads=4))
    assert a == b and len(a) > 0


def test_g8_reference_bind_cache_quirk_is_pinned(oracle):
    """G8 (reference-generated): one oligo as a primer of one assay and as the TaqMan probe of another.  The reference's
    per-target BindCache gives the probe the primer's evaluation (its Tm at the primer's strand concentration; a record of
    zeros where the primer's threshold cascade had stopped early), so its hits differ from the uncached computation.  The
    oracle models the cache since round 4: oracle == reference; with the memo off it computes what the library computes
    under TNT_MI355X_UNCACHED_SHARED_OLIGOS (the library refuses such an assay set otherwise, tests/test_gpu_parity.py)."""
    g = gu.load("g8_bindcache.json.gz")
    assays = [synth.Assay(a[0], a[1], a[2], a[3]) for a in g["assays"]]
    opt = gu.options_from(g["options"])
    codes, offsets = gu.pack_targets(g["targets"])
    ref = g["hits"]
    got = [list(k) for k in tc.sorted_hit_keys(oracle.search(opt, synth.expand_assays(assays), codes, offsets))]
    assert len(ref) == 113 and got == ref
    # the uncached computation: what round 3 pinned the quirk with
    got = [list(k) for k in tc.sorted_hit_keys(oracle.search(opt, synth.expand_assays(assays), codes, offsets, bind_cache=False))]
    both = [h for h in got if h in ref]
    assert len(got) == 135 and len(both) == 57
    # the PCR assay (its oligos are evaluated first on every target) is not affected
    assert sorted(h for h in got if h[0] == 0) == sorted(h for h in ref if h[0] == 0)
    # where both report the same TaqMan amplicon, only the probe's Tm differs (it was computed at the primer's Ct)
    names = [n for n, _ in abi.tnt_hit._fields_]
    ints = [i for i, (n, t) in enumerate(abi.tnt_hit._fields_) if t is not __import__("ctypes").c_float and not n.endswith("_align")]
    key = lambda h: tuple(h[i] for i in ints)  # noqa: E731
    refmap = {key(h): h for h in ref if h[0] == 1}
    differing = [(h, refmap[key(h)]) for h in got if h[0] == 1 and key(h) in refmap and h != refmap[key(h)]]
    assert differing
    for a, b in differing:
        diff = [names[i] for i in range(len(names)) if a[i] != b[i]]
        assert diff == ["probe_tm"], diff


def test_g9_dinkelbach(oracle):
    """G9: the oracle's Dinkelbach iteration against the reference's (binds of every third G2 request, self-structure Tm)."""
    fx, g = gu.load("g2_binds.json.gz"), gu.load("g9_dinkelbach.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    reqs = tc.make_reqs([tuple(fx["requests"][i]) for i in g["g2_request_indices"]])
    so = gu.load("g3_structure.json.gz")["oligos"]
    for case in g["cases"]:
        opt = gu.options_from(case["options"])
        out = oracle.bind_batch(opt, fx["oligos"], codes, offsets, reqs)
        bad = [(i, gu.compare_bind(x, gu.bind_record(out[i]))) for i, x in enumerate(case["expected"])]
        bad = [b for b in bad if b[1]]
        assert not bad, (case["name"], len(bad), bad[:2])
        for i, o in enumerate(so):
            assert list(oracle.structure(opt, o, so[(i * 7 + 3) % len(so)], g["ct_a"], g["ct_b"])) == case["structure"][i], (case["name"], o)


def test_g4_g5_short_hash_words(oracle):
    """-W 3 and -W 4 (options.cpp:639): reference-generated seed lists and hit sets (tests/gen_golden_w34.py)."""
    fx = gu.load("g4_seeds_w34.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    for k, expected in fx["seeds"].items():
        got = sorted(tuple(s) for s in oracle.seeds(abi.options_default(hash_word_size=int(k)), fx["oligos"], codes, offsets))
        assert got == sorted(tuple(s) for s in expected), "word size %s" % k
    for case in gu.load("g5_search_w34.json.gz")["cases"]:
        expanded, codes, offsets = gu.g5_inputs(case)
        hits = oracle.search(gu.options_from(case["options"]), expanded, codes, offsets)
        assert gu.hits_as_keys(hits) == case["hits"] and len(hits) > 0, (case["config"], case["options"]["hash_word_size"])


def test_g10_long_records(oracle):
    """Records of 0.25-0.6 Mb (the library seeds them in windows): the reference's hit sets over whole-record lists and the
    seeds its first cull keeps, with the strand bound first (tests/gen_golden_long.py)."""
    for case in gu.load("g10_long_records.json.gz")["cases"]:
        expanded, codes, offsets = gu.g5_inputs(case)
        opt = gu.options_from(case["options"])
        hits = oracle.search(opt, expanded, codes, offsets)
        assert gu.hits_as_keys(hits) == case["hits"] and len(hits) > 20, case["config"]
        if "first_cull" in case:
            rows = sorted([int(v) for v in r] for r in oracle.first_cull(opt, expanded, codes, offsets))
            assert rows == case["first_cull"] and len(rows) > 2000, case["config"]


def test_g11_oligos_beyond_128_bases(oracle):
    """G11 (reference-generated): hybridisation probes of 130, 200 and 1 000 bases -- NucCruc evaluations of a sample of their
    seeds, the hit set of hybrid(), hairpin / homodimer / heterodimer Tm: oracle == reference."""
    g = gu.load("g11_beyond128.json.gz")
    opt = gu.options_from(g["options"])
    codes, offsets = gu.pack_targets(g["targets"])
    reqs = tc.make_reqs((r[0], r[1], r[2], r[3], r[4], 1, 2.5e-7) for r in g["requests"])
    out = oracle.bind_batch(opt, g["oligos"], codes, offsets, reqs)
    bad = [(i, gu.compare_bind(x, gu.bind_record(out[i]))) for i, x in enumerate(g["binds"])]
    assert not [b for b in bad if b[1]], [b for b in bad if b[1]][:2]
    ex = synth.expand_assays([synth.Assay(a[0], a[1], a[2], a[3]) for a in g["assays"]])
    assert [list(k) for k in tc.sorted_hit_keys(oracle.search(opt, ex, codes, offsets))] == g["hits"]
    for i, o in enumerate(g["oligos"]):
        assert list(oracle.structure(opt, o, g["oligos"][(i + 1) % len(g["oligos"])], 2.5e-7, 2.5e-7)) == g["structure"][i]
