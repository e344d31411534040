"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle, the committed golden
vectors and (where oracle/_ref travelled) the reference itself.  Integer / index / text results must
be identical; Tm, dH, dS, dG are compared as float32 BIT PATTERNS (tolerance 0; the north-star bar is
1e-6 C / 1e-6 kcal/mol)."""
import ctypes as C

import numpy as np
import pytest

import golden_util as gu
import tntcheck as tc
from thermonucleotideblast_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    import __graft_entry__ as g
    g.build_hip()
    from thermonucleotideblast_amd.engine import Engine as E
    return E


def _oligo_set(expanded):
    return sorted({o for e in expanded for o in (e.forward, e.reverse, e.probe) if o})


def test_native_library_is_loaded(Engine):
    from thermonucleotideblast_amd import engine
    lib = engine.load_library()
    assert lib.tnt_abi_version() == 7
    assert any("libtnt_mi355x.so" in line for line in open("/proc/self/maps"))


def test_g1_delta_g(Engine):
    fx = gu.load("g1_tables.json")
    for key, dg in fx["delta_g"].items():
        T, na = (float(x) for x in key.split("_"))
        e = Engine(abi.options_default(target_t=T, salt=na), synth.expand_assays([synth.Assay("a", "ACGTACGTACGTACGTAC", "TGCATGCATGCATGCATG")]))
        assert np.array_equal(e.delta_g(), np.array(dg, dtype=np.int32)), key
        e.close()


@pytest.mark.parametrize("route", [0, 4])
def test_g2_binds_golden(Engine, route):
    """10k reference NucCruc evaluations per option set: mismatches, bulges, loops, IUPAC, inosine, clipped windows, ties.
    Route 0: the unit-level instance of the lane-per-window kernel; 4: of the wavefront-per-window kernel (tnt_wbind.hip.h)."""
    fx = gu.load("g2_binds.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    ex = [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(fx["oligos"])]
    for case in fx["cases"]:
        e = Engine(gu.options_from(case["options"]), ex)
        om = {o: i for i, o in enumerate(e.oligos)}
        reqs = tc.make_reqs((r[0], om[fx["oligos"][r[1]]], r[2], r[3], r[4], r[5], r[6]) for r in fx["requests"])
        db = e.pack(codes, offsets)
        out = e.bind_batch(db, reqs, route=route)
        bad = [(i, gu.compare_bind(x, gu.bind_record(out[i]))) for i, x in enumerate(case["expected"])]
        bad = [b for b in bad if b[1]]
        assert not bad, "%s: %d of %d differ, first %r" % (case["name"], len(bad), len(case["expected"]), bad[:2])
        db.close()
        e.close()


ROUTED_FIELDS = [3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 21]  # tm dH dS dg a5 a3 mm gap loc5 loc3 align: what a throughput instance keeps


@pytest.mark.parametrize("route", [1, 2, 3])
def test_g2_binds_golden_production_instances(Engine, route):
    """The G2 cases through the kernels tnt_search runs (bucket_kernel + one launch per length bucket: exact / plain
    ladder / general instances, 256- and 768-thread; route 2: everything through the longest general instance; route 3:
    every bucket through the wavefront-per-window kernel): the status of every request and the record + text of every
    bound one must equal the reference's."""
    fx = gu.load("g2_binds.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    ex = [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(fx["oligos"])]
    for case in fx["cases"]:
        e = Engine(gu.options_from(case["options"]), ex)
        om = {o: i for i, o in enumerate(e.oligos)}
        reqs = tc.make_reqs((r[0], om[fx["oligos"][r[1]]], r[2], r[3], r[4], r[5], r[6]) for r in fx["requests"])
        db = e.pack(codes, offsets)
        out = e.bind_batch(db, reqs, route=route)
        bad, n_bound = [], 0
        for i, x in enumerate(case["expected"]):
            got = gu.bind_record(out[i])
            idx = [0] + (ROUTED_FIELDS if x[0] == 0 else [])
            n_bound += x[0] == 0
            d = [(k, x[k], got[k]) for k in idx if x[k] != got[k]]
            if d:
                bad.append((i, d))
        assert n_bound > 100
        assert not bad, "%s route %d: %d of %d differ, first %r" % (case["name"], route, len(bad), len(case["expected"]), bad[:2])
        db.close()
        e.close()


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(min_primer_tm=0.0, min_probe_tm=0.0, max_primer_dg=-2.0, max_probe_dg=-2.0)),
                                     (3, dict(salt=0.9, target_t=300.0, min_primer_tm=20.0, min_probe_tm=20.0)),
                                     (4, dict(allow_dangle_5=1, allow_dangle_3=1, min_primer_tm=30.0, min_probe_tm=30.0, primer_clamp=2, max_mismatch=4, max_gap=1))])
def test_random_windows_production_vs_oracle(Engine, oracle, seed, kw):
    """Random windows -- oligos of 12..64 bases, sites with substitutions, deletions, insertions, IUPAC codes, clipped at
    the target ends -- through every route of the bind stage against the oracle."""
    rng = np.random.default_rng([77, seed])
    oligos = sorted({synth._rand_oligo(rng, 12, 64) for _ in range(60)} | {synth._rand_oligo(rng, 17, 32) for _ in range(60)})
    targets, rows = [], []
    for t in range(1500):
        o = int(rng.integers(0, len(oligos)))
        q = oligos[o]
        minus = rng.random() < 0.5  # the minus-strand binding site reads like the oligo, the plus-strand one like its reverse complement
        site = list(q) if minus else list(synth.revcomp(q))
        for _ in range(int(rng.integers(0, 5))):
            k = int(rng.integers(0, len(site)))
            r = rng.random()
            if r < 0.6:
                site[k] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.75:
                site[k] = "RYKMSWNBDHV"[int(rng.integers(0, 11))]
            elif r < 0.9 and len(site) > 10:
                del site[k]
            else:
                site.insert(k, "ACGT"[int(rng.integers(0, 4))])
        left = int(rng.integers(0, 12))
        right = int(rng.integers(0, 12))
        seq = "".join("ACGT"[i] for i in rng.integers(0, 4, left)) + "".join(site) + "".join("ACGT"[i] for i in rng.integers(0, 4, right))
        targets.append(seq)
        # a seed somewhere on the site's diagonal: query_loc indexes the word list of the strand-specific oligo
        ql = int(rng.integers(0, max(1, len(q) - 7)))
        rows.append((t, o, abi.MINUS if minus else abi.PLUS, ql, min(len(seq) - 1, left + ql), int(rng.integers(0, 2)), 9.0e-7 if t % 3 else 2.5e-7))
    codes, offsets = gu.pack_targets(targets)
    opt = abi.options_default(**dict(dict(min_primer_tm=40.0, min_probe_tm=40.0), **kw))
    e = Engine(opt, [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(oligos)])
    om = {o: i for i, o in enumerate(e.oligos)}
    reqs = tc.make_reqs((r[0], om[oligos[r[1]]], r[2], r[3], r[4], r[5], r[6]) for r in rows)
    oreqs = tc.make_reqs(rows)
    want = oracle.bind_batch(opt, oligos, codes, offsets, oreqs)
    db = e.pack(codes, offsets)
    n_bound = sum(1 for w in want if w.status == 0)
    assert n_bound > 50, n_bound
    for route in (0, 1, 2, 3, 4):
        out = e.bind_batch(db, reqs, route=route)
        bad = []
        for i, w in enumerate(want):
            x, got = gu.bind_record(w), gu.bind_record(out[i])
            if route in (0, 4):
                d = gu.compare_bind(x, got)
            else:
                d = [(k, x[k], got[k]) for k in ([0] + (ROUTED_FIELDS if x[0] == 0 else [])) if x[k] != got[k]]
            if d:
                bad.append((i, d))
        assert not bad, "route %d: %d of %d differ, first %r" % (route, len(bad), len(want), bad[:2])
    db.close()
    e.close()


@pytest.mark.parametrize("k", [0, 1, 2, 4, 5, 6, 8, 9, 11, 100, 102, 1000, 1001, 1004, 1006, 1009])
def test_filter_kernel_equals_its_checker(Engine, oracle, k, monkeypatch):
    """The pre-filter kernels (every row-count instance the oligo lengths pick) against the oracle-side restatement whose
    soundness tests/test_filter_bound.py checks: same bound, same threshold, for every window.  Plain windows of oligos of up
    to 64 bases go through the regrouping and ufilter_kernel (what a search does with them); k >= 1000: the same sets
    with TNT_MI355X_NO_REGROUP=1, i.e. filter_kernel on the queue as it is."""
    import test_filter_bound as fbt
    if k >= 1000:
        monkeypatch.setenv("TNT_MI355X_NO_REGROUP", "1")
        k -= 1000
    else:
        monkeypatch.delenv("TNT_MI355X_NO_REGROUP", raising=False)
    long_oligos = k >= 100  # (100, 102: a Tm and a dG set for oligos of 65 to 128 bases -- the 128-row instances)
    k %= 100
    name, kw = fbt.LONG_OLIGO_SETS[k] if long_oligos else fbt.OPTION_SETS[k]
    rng = np.random.default_rng([99, k])
    opt = abi.options_default(**kw)
    # (sets 1, 5, 9 -- a Tm, a dG and a dangle set -- on windows full of inosines: the column value ANY)
    if long_oligos:
        oligos, targets, rows = fbt.make_windows(rng, 6000, lo=65, hi=128)
    else:
        oligos, targets, rows = fbt.make_windows(rng, 6000, lo=12, hi=64, **(dict(degen=0.8, inosine=0.6) if k in (1, 5, 9) else {}))
    if k == 2:  # some degenerate target bases: those windows are outside the filter
        targets = [t if i % 7 else t[:5] + "N" + t[6:] for i, t in enumerate(targets)]
    rows = [r[:6] + ((opt.probe_strand if r[5] else opt.primer_strand),) for r in rows]
    codes, offsets = gu.pack_targets(targets)
    e = Engine(opt, [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(oligos)])
    om = {o: i for i, o in enumerate(e.oligos)}
    db = e.pack(codes, offsets)
    got = e.filter_bound(db, tc.make_reqs((r[0], om[oligos[r[1]]], r[2], r[3], r[4], r[5], r[6]) for r in rows))
    want = oracle.filter_bound(opt, oligos, codes, offsets, tc.make_reqs(rows))
    assert (want[:, 0] >= 0).sum() > 4000
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "%s: %d of %d differ, first %r got %r want %r" % (name, len(bad), len(rows), rows[bad[0]], got[bad[0]], want[bad[0]])
    db.close()
    e.close()


@pytest.mark.parametrize("k", [0, 1, 3, 5, 8, 9])
def test_second_tier_kernel_equals_its_checker(Engine, oracle, k):
    """sbound_kernel / sbound_degen_kernel (the pre-filter's second tier: a trace-free fill of align_dimer's recurrence, V = score << 11 |
    credit; the second for windows that hold an IUPAC code or an inosine: a column's code resolved against the query base of every
    row) against the oracle-side restatement, integer by integer: the credit of the maximal-score paths, INT32_MAX -> 32 767 where
    the guard bit fired or the window is not served (oligos beyond 40 bases).  Minimum-Tm, dG-only and dangling-end sets."""
    import test_filter_bound as fbt
    name, kw = fbt.OPTION_SETS[k]
    rng = np.random.default_rng([77, k])
    opt = abi.options_default(**kw)
    oligos, targets, rows = fbt.make_windows(rng, 6000, lo=12, hi=48, degen=0.3, inosine=0.15)
    rows = [r[:6] + ((opt.probe_strand if r[5] else opt.primer_strand),) for r in rows]
    codes, offsets = gu.pack_targets(targets)
    e = Engine(opt, [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(oligos)])
    om = {o: i for i, o in enumerate(e.oligos)}
    db = e.pack(codes, offsets)
    got = e.filter_bound_tier2(db, tc.make_reqs((r[0], om[oligos[r[1]]], r[2], r[3], r[4], r[5], r[6]) for r in rows))
    want = oracle.filter_bound(opt, oligos, codes, offsets, tc.make_reqs(rows), tier=30)
    served = got[:, 1] != np.iinfo(np.int32).min
    degenerate = np.array([any(ch not in "ACGT" for ch in targets[r[0]]) for r in rows])
    assert served.sum() > 3000 and (~served).sum() > 300 and (served & degenerate).sum() > 800
    g0 = np.minimum(got[:, 0], 32767)
    bad = np.nonzero(g0 != want[:, 0])[0]
    assert len(bad) == 0, "%s: %d of %d differ, first %r got %r want %r" % (name, len(bad), len(rows), rows[bad[0]], got[bad[0]], want[bad[0]])
    assert (got[served, 1] == want[served, 1]).all()
    assert (g0[served] < got[served, 1]).sum() > 300  # the tier rejects windows
    db.close()
    e.close()


def test_the_bind_stage_s_scheduling_changes_no_result(Engine, oracle, monkeypatch):
    """The buckets' launch chains of a bind phase run on side streams, dealt by expected / measured work (round 6).  Which
    stream takes which bucket -- one stream, three, eight, round robin, the measured times of an earlier search -- moves no
    hit, and neither does the seed kernel's instance (short targets, large lists, general)."""
    cases = [(2, 20000, 96), (3, 12000, 48), (5, 6000, 120)]
    for cfg, nt, na in cases:
        wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=37)
        want = None
        for env in ({}, {"TNT_MI355X_BIND_LANES": "1"}, {"TNT_MI355X_BIND_LANES": "8"}, {"TNT_MI355X_LANES_ROUND_ROBIN": "1"}, {"TNT_MI355X_LANES_STATIC": "1"},
                    {"TNT_MI355X_SEED2_GENERAL": "1"}, {"TNT_MI355X_SEED2_STD_LISTS": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            e = Engine(wl.options, wl.expanded)
            db = e.pack(wl.codes, wl.offsets)
            got = tc.sorted_hit_keys(e.search(db))
            again = tc.sorted_hit_keys(e.search(db))  # (the second search deals the chains by the first one's times)
            db.close()
            e.close()
            for k in env:
                monkeypatch.delenv(k)
            if want is None:
                want = got
                assert len(want) > 5
                if cfg == 2:
                    assert want == tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
            assert got == want and again == want, env


def test_filter_changes_no_result(Engine, oracle, monkeypatch):
    """The same searches with the pre-filter on, off and in audit mode (everything is bound and the bind kernel raises an
    error if it accepts a window the filter marked): identical hits, and the filter does reject most windows."""
    for cfg, nt, na, kw in [(2, 30000, 96, {}), (3, 20000, 96, dict(min_primer_tm=50.0, min_probe_tm=45.0)), (5, 6000, 200, {}),
                            # dG-only thresholds: the bound is taken against the maximum dG at the target temperature
                            (2, 20000, 96, dict(min_primer_tm=0.0, max_primer_dg=-10.0)),
                            (3, 12000, 48, dict(min_primer_tm=0.0, min_probe_tm=0.0, max_primer_dg=-9.0, max_probe_dg=-11.0)),
                            # dangling-end options: the threshold carries the most a dangling / frayed end can add
                            (2, 20000, 96, dict(allow_dangle_5=1, allow_dangle_3=1)),
                            # inosine in the targets (one base in 300): the windows that hold one are filtered like IUPAC windows
                            (2, 20000, 96, dict(inosine=1))]:
        wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=31)
        if kw.pop("inosine", 0):
            where = np.random.default_rng(8).random(len(wl.codes)) < 1.0 / 300.0
            wl.codes[where] = abi.DB_CODES.index("I")
            kw = {}
            inosine = True
        else:
            inosine = False
        for k_, v_ in kw.items():
            setattr(wl.options, k_, v_)
        res = {}
        for mode in ("on", "off", "audit", "tier1"):
            monkeypatch.delenv("TNT_MI355X_NO_FILTER", raising=False)
            monkeypatch.delenv("TNT_MI355X_FILTER_AUDIT", raising=False)
            monkeypatch.delenv("TNT_MI355X_NO_TIER2", raising=False)
            if mode == "tier1":  # (the first tier alone: TNT_MI355X_NO_TIER2)
                monkeypatch.setenv("TNT_MI355X_NO_TIER2", "1")
            if mode == "off":
                monkeypatch.setenv("TNT_MI355X_NO_FILTER", "1")
            if mode == "audit":
                monkeypatch.setenv("TNT_MI355X_FILTER_AUDIT", "1")
            e = Engine(wl.options, wl.expanded)
            db = e.pack(wl.codes, wl.offsets)
            res[mode] = tc.sorted_hit_keys(e.search(db))
            tm = e.timing()
            if mode in ("on", "tier1"):
                assert tm.n_filtered > tm.n_binds // (2 if set(kw) <= {"min_primer_tm", "min_probe_tm"} else 5), (cfg, tm.n_filtered, tm.n_binds)  # (dG / dangle sets: looser bounds)
                res[mode + "_filtered"] = tm.n_filtered
            else:
                assert tm.n_filtered == 0
            db.close()
            e.close()
        assert res["on"] == res["off"] == res["audit"] == res["tier1"] and len(res["on"]) > 0
        assert res["on_filtered"] > res["tier1_filtered"]  # the second tier rejects windows the first lets through
        if cfg == 2 or inosine:
            assert res["on"] == tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))


@pytest.mark.parametrize("route", [0, 2])
def test_g9_dinkelbach_golden(Engine, route):
    """--dinkelbach: the bind kernel's temperature iteration (unit-level instance and the throughput instance a search
    runs) against the reference's NucCruc, and the self-structure Tm printed with every hit."""
    fx, g = gu.load("g2_binds.json.gz"), gu.load("g9_dinkelbach.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    ex = [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(fx["oligos"])]
    so = gu.load("g3_structure.json.gz")["oligos"]
    for case in g["cases"]:
        e = Engine(gu.options_from(case["options"]), ex)
        om = {o: i for i, o in enumerate(e.oligos)}
        rows = [fx["requests"][i] for i in g["g2_request_indices"]]
        reqs = tc.make_reqs((r[0], om[fx["oligos"][r[1]]], r[2], r[3], r[4], r[5], r[6]) for r in rows)
        db = e.pack(codes, offsets)
        out = e.bind_batch(db, reqs, route=route)
        bad = []
        for i, x in enumerate(case["expected"]):
            got = gu.bind_record(out[i])
            d = gu.compare_bind(x, got) if route == 0 else [(k, x[k], got[k]) for k in ([0] + (ROUTED_FIELDS if x[0] == 0 else [])) if x[k] != got[k]]
            if d:
                bad.append((i, d))
        assert not bad, "%s route %d: %d of %d differ, first %r" % (case["name"], route, len(bad), len(case["expected"]), bad[:2])
        if route == 0:
            for i, o in enumerate(so[:120]):
                got = e.oligo_structure(o, so[(i * 7 + 3) % len(so)], g["ct_a"], g["ct_b"])
                assert [tc.f32_bits(v) for v in got] == case["structure"][i], (case["name"], o)
        db.close()
        e.close()


def test_dinkelbach_search_vs_oracle(Engine, oracle):
    for cfg, nt, na in [(3, 500, 32), (5, 400, 40), (4, 300, 30)]:
        wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=77)
        wl.options.use_dinkelbach = 1
        e = Engine(wl.options, wl.expanded)
        db = e.pack(wl.codes, wl.offsets)
        got = tc.sorted_hit_keys(e.search(db))
        want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
        assert len(want) > 0 and got == want, (cfg, len(got), len(want))
        db.close()
        e.close()


def test_g3_structure_golden(Engine):
    """Row a17: hairpin / homodimer / heterodimer Tm of oligos through the unit-level bind kernel instance."""
    g = gu.load("g3_structure.json.gz")
    so = g["oligos"]
    for case in g["cases"]:
        e = Engine(gu.options_from(case["options"]), [synth.ExpandedAssay(0, 0, None, None, "ACGTACGTACGTACGTAC")])
        for i, o in enumerate(so):
            got = e.oligo_structure(o, so[(i * 7 + 3) % len(so)], g["ct_a"], g["ct_b"])
            assert [tc.f32_bits(v) for v in got] == case["expected"][i], (case["name"], o)
        h, d, x = e.oligo_structure(so[0])
        assert x is None and [tc.f32_bits(h), tc.f32_bits(d)] == case["expected"][0][:2]
        e.close()


def test_structure_vs_oracle(Engine, oracle):
    rng = np.random.default_rng(5)
    oligos = [synth._rand_oligo(rng, 8, 64) for _ in range(200)] + ["A" * 64, "AT" * 32, "GC" * 32, "G" * 20 + "TTTT" + "C" * 20]
    opt = abi.options_default(allow_dangle_3=1, salt=0.1)
    e = Engine(opt, [synth.ExpandedAssay(0, 0, None, None, "ACGTACGTACGTACGTAC")])
    for i, o in enumerate(oligos):
        p = oligos[(i * 13 + 1) % len(oligos)]
        got = e.oligo_structure(o, p, 5e-7, 9e-7)
        assert tuple(tc.f32_bits(v) for v in got) == oracle.structure(opt, o, p, 5e-7, 9e-7), (o, p)
    for o in ("A" * 65, "AT" * 64, synth._rand_oligo(rng, 100, 128)):  # the long unit-level instance
        got = e.oligo_structure(o, oligos[3], 5e-7, 9e-7)
        assert tuple(tc.f32_bits(v) for v in got) == oracle.structure(opt, o, oligos[3], 5e-7, 9e-7), o
    for o in (synth._rand_oligo(rng, 129, 129), synth._rand_oligo(rng, 300, 300), "GC" * 100 + "TTTT" + "GC" * 100):  # the wavefront-per-window kernel
        got = e.oligo_structure(o, oligos[3], 5e-7, 9e-7)
        assert tuple(tc.f32_bits(v) for v in got) == oracle.structure(opt, o, oligos[3], 5e-7, 9e-7), o
    with pytest.raises(Exception):
        e.oligo_structure("A" * 1025)
    with pytest.raises(Exception):
        e.oligo_structure("ACGTNACGT")
    e.close()


@pytest.mark.parametrize("cfg,nt,na,seed,kw", [(2, 2000, 48, 31, {}), (3, 2000, 48, 32, {}), (4, 600, 60, 33, {}),
                                                (3, 1500, 24, 34, {"max_len": 150, "single_primer_pcr": 0}),
                                                (2, 800, 24, 35, {"hash_word_size": 5, "max_len": 400}),
                                                (1, 1000, None, 36, {}),
                                                # targets beyond the reach of the cull (max_len + 50 < target length): the general loop
                                                (2, 600, 24, 37, {"max_len": 300}),
                                                (3, 600, 24, 38, {"max_len": 400}),
                                                (2, 1000, 48, 39, {"single_primer_pcr": 0}),
                                                (3, 1000, 48, 40, {"single_primer_pcr": 0})])
def test_first_cull_vs_oracle(Engine, oracle, cfg, nt, na, seed, kw):
    """Row a4 at unit level: the seeds the seed kernel hands to the bind stage (amplicon()'s early exits + the first
    cull_oligo_match, decided per seed without sorting) and the strand it binds first, against the oracle's list-based
    restatement (which tests/test_oracle_vs_reference.py::test_first_cull holds against the reference's own functions)."""
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    got = e.first_cull(db)
    want = oracle.first_cull(wl.options, wl.expanded, wl.codes, wl.offsets)
    assert sorted(map(tuple, got)) == sorted(map(tuple, want)) and len(want) > 100
    db.close()
    e.close()


@pytest.mark.parametrize("cfg,kw,host", [(2, {}, False), (3, {}, False), (3, {"max_len": 400}, False), (2, {"single_primer_pcr": 0}, False),
                                         (3, {"single_primer_pcr": 0}, False), (3, {}, True)])
def test_first_cull_of_windowed_targets_vs_oracle(Engine, oracle, cfg, kw, host, monkeypatch):
    """The same rows for targets that are seeded in windows: their lists are assembled and culled on the device
    (nearest-element scans over the whole list, tnt_sort.hip; `host`: the replay of cull_oligo_match on host threads) --
    every kept seed and the strand its list binds first, against the oracle's list-based restatement."""
    wl = synth.make_workload(cfg, n_targets=700, n_assays=32, seed=44)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    lens = np.diff(wl.offsets.astype(np.int64))
    # a 400-kb record, a 250-kb record, fifty short ones
    cuts = [0, 400, 650] + list(range(651, 701))
    offsets = np.array([int(lens[:c].sum()) for c in cuts], dtype=np.uint64)
    if host:
        monkeypatch.setenv("TNT_MI355X_LONG_HOST_CULL", "1")
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    got = e.first_cull(db)
    want = oracle.first_cull(wl.options, wl.expanded, wl.codes, offsets)
    assert sorted(map(tuple, got)) == sorted(map(tuple, want))
    assert len([r for r in want if r[0] == 0]) > 100 and len([r for r in want if r[0] == 1]) > 100 and len([r for r in want if r[0] > 1]) > 5
    db.close()
    e.close()


@pytest.mark.parametrize("kind,seed", [("probe", 1), ("probe", 2), ("padlock", 3), ("pcr", 4), ("pcr", 5)])
def test_long_oligos_vs_oracle(Engine, oracle, kind, seed, monkeypatch):
    """Oligos of 65..128 bases run through the 128-row instances (rows in scratch memory): binds through all three routes and
    complete searches against the oracle (itself held against the reference on the same workloads,
    tests/test_oracle_vs_reference.py::test_long_oligos); 129 bases are refused.  The 128-row pre-filter is used when the
    context's calibration finds that it rejects a quarter of its sample: the search is repeated with it forced on, forced
    on in audit mode (every window bound; an accepted window the filter had marked is an error) and forced off."""
    import longwl
    wl = longwl.make(kind, seed)
    oligos = sorted({o for x in wl.expanded for o in (x.forward, x.reverse, x.probe) if o})
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    for force, audit in (("1", True), ("1", False), ("0", False)):
        monkeypatch.setenv("TNT_MI355X_FILTER_LONG", force)
        if audit:
            monkeypatch.setenv("TNT_MI355X_FILTER_AUDIT", "1")
        e2 = Engine(wl.options, wl.expanded)
        db2 = e2.pack(wl.codes, wl.offsets)
        assert tc.sorted_hit_keys(e2.search(db2)) == want, (force, audit)
        db2.close()
        e2.close()
        monkeypatch.delenv("TNT_MI355X_FILTER_AUDIT", raising=False)
    monkeypatch.delenv("TNT_MI355X_FILTER_LONG")
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    assert tc.sorted_hit_keys(e.search(db)) == want and len(want) > 10
    sa = oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
    rows = [(s[0], s[1], s[2], s[3], s[4], i % 2, 9e-7 if i % 3 else 2.5e-7) for i, s in enumerate(sa[:5000])]
    ow = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, tc.make_reqs(rows))
    om = {o: i for i, o in enumerate(e.oligos)}
    reqs = tc.make_reqs((r[0], om[oligos[r[1]]], r[2], r[3], r[4], r[5], r[6]) for r in rows)
    assert sum(1 for w in ow if w.status == 0) > 5
    for route in (0, 1, 2):
        out = e.bind_batch(db, reqs, route=route)
        bad = []
        for i, w in enumerate(ow):
            x, got = gu.bind_record(w), gu.bind_record(out[i])
            if route == 0:
                d = gu.compare_bind(x, got)
            else:
                d = [(k, x[k], got[k]) for k in ([0] + (ROUTED_FIELDS if x[0] == 0 else [])) if x[k] != got[k]]
            if d:
                bad.append((i, rows[i], d))
        assert not bad, (route, len(bad), bad[:3])
    db.close()
    e.close()
    too_long = [synth.ExpandedAssay(0, 0, None, None, "ACGT" * 256 + "A")]  # beyond the reference's MAX_SEQUENCE_LENGTH too
    with pytest.raises(Exception):
        Engine(wl.options, too_long)


def test_g6_fasta_ingest_golden(Engine):
    """FASTA text parsed on the GPU == what the reference's reader returns (codes, deflines)."""
    g = gu.load("g6_fasta.json.gz")
    e = Engine(abi.options_default(), [synth.ExpandedAssay(0, 0, None, None, "ACGTACGTACGTACGTAC")])
    db = e.pack_fasta(g["text_latin1"].encode("latin-1"), first_seq_index=7)
    assert db.n_targets == len(g["deflines"])
    assert db.n_bases == sum(len(t) for t in g["targets"])
    for t, want in enumerate(g["targets"]):
        assert db.defline(t) == g["deflines"][t]
        assert synth.decode(db.target_codes(t)) == want, t
    tm = db.ingest_timing()
    assert tm.bases == db.n_bases and tm.text_bytes == len(g["text_latin1"])
    db.close()
    empty = e.pack_fasta(b"no record here\n")
    assert empty.n_targets == 0 and len(e.search(empty)) == 0
    empty.close()
    with pytest.raises(Exception):
        e.pack_fasta(b">truncated defline")
    e.close()


@pytest.mark.parametrize("cfg,nt,na", [(5, 400, 40), (2, 600, 48)])
def test_fasta_ingest_search_parity(Engine, oracle, cfg, nt, na, tmp_path):
    """A database ingested from FASTA text (or a gzip file) gives the same hits as one packed from codes."""
    import gzip
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=61)
    text = synth.to_fasta(wl.codes, wl.offsets, width=61)
    e = Engine(wl.options, wl.expanded)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    assert len(want) > 0
    a = e.pack(wl.codes, wl.offsets)
    b = e.pack_fasta(text)
    path = tmp_path / "t.fna.gz"
    with gzip.open(path, "wb") as f:
        f.write(text)
    c = e.load_fasta(str(path))
    for db in (a, b, c):
        assert db.n_targets == wl.n_targets and db.n_bases == wl.n_bases
        hits = e.search(db)
        assert tc.sorted_hit_keys(hits) == want
        if len(hits):  # amplicon text comes from the host copy / the rows fetched back from the device
            assert e.hit_sequence(a, hits[0]) == e.hit_sequence(db, hits[0])
    for t in (0, nt // 2, nt - 1):
        assert np.array_equal(b.target_codes(t), wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])])
        assert c.defline(t) == "seq%d synthetic target" % t
    for db in (a, b, c):
        db.close()
    e.close()


def test_g4_seeds_golden(Engine):
    fx = gu.load("g4_seeds.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    ex = [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(fx["oligos"])]
    for k, expected in fx["seeds"].items():
        n_t = 100 if k == "7" else 12
        e = Engine(abi.options_default(hash_word_size=int(k)), ex)
        db = e.pack(codes[:int(offsets[n_t])], offsets[:n_t + 1])
        got = sorted(e.seeds(db))
        want = sorted((s[0], fx["oligos"][s[1]], s[2], s[3], s[4]) for s in expected)
        assert got == want, "word size %s" % k
        db.close()
        e.close()


def test_g4_g5_short_hash_words_golden(Engine):
    """-W 3 and -W 4: reference-generated seed lists and hit sets (tests/gen_golden_w34.py)."""
    fx = gu.load("g4_seeds_w34.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    ex = [synth.ExpandedAssay(i, i, None, None, o) for i, o in enumerate(fx["oligos"])]
    for k, expected in fx["seeds"].items():
        e = Engine(abi.options_default(hash_word_size=int(k)), ex)
        db = e.pack(codes, offsets)
        got = sorted(e.seeds(db))
        want = sorted((s[0], fx["oligos"][s[1]], s[2], s[3], s[4]) for s in expected)
        assert got == want, "word size %s" % k
        db.close()
        e.close()
    for case in gu.load("g5_search_w34.json.gz")["cases"]:
        expanded, codes, offsets = gu.g5_inputs(case)
        e = Engine(gu.options_from(case["options"]), expanded)
        db = e.pack(codes, offsets)
        assert gu.hits_as_keys(e.search(db)) == case["hits"], (case["config"], case["options"]["hash_word_size"])
        db.close()
        e.close()


def test_g10_long_records_golden(Engine):
    """Reference-generated hit sets and first-cull rows of records of 0.25-0.6 Mb (tests/gen_golden_long.py): the long-target
    path -- windows, lists and first cull on the device, pairing on host threads -- against the reference itself."""
    for case in gu.load("g10_long_records.json.gz")["cases"]:
        expanded, codes, offsets = gu.g5_inputs(case)
        e = Engine(gu.options_from(case["options"]), expanded)
        db = e.pack(codes, offsets)
        assert gu.hits_as_keys(e.search(db)) == case["hits"], case["config"]
        if "first_cull" in case:
            assert sorted([int(v) for v in r] for r in e.first_cull(db)) == case["first_cull"], case["config"]
        db.close()
        e.close()


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_g5_search_golden(Engine, cfg):
    fx = gu.load("g5_search_cfg%d.json.gz" % cfg)
    expanded, codes, offsets = gu.g5_inputs(fx)  # stored in the fixture, not regenerated from RNG seeds
    e = Engine(gu.options_from(fx["options"]), expanded)
    db = e.pack(codes, offsets)
    hits = e.search(db)
    assert gu.hits_as_keys(hits) == fx["hits"]
    db.close()
    e.close()


def test_g8_bind_cache_semantics(Engine, oracle, monkeypatch):
    """Row a6 where it changes results (golden G8, reference-generated): one oligo as a primer of one assay and as the
    TaqMan probe of another.  The reference's per-target BindCache hands the first evaluation of an (oligo string, window)
    to every later caller, whatever its strand concentration and thresholds; the library replays that (cache replay,
    tnt_pair.hip.h): its hits equal the reference's.  With TNT_MI355X_UNCACHED_SHARED_OLIGOS=1 every use is bound on its
    own (the oracle with the memo off)."""
    g = gu.load("g8_bindcache.json.gz")
    ex = synth.expand_assays([synth.Assay(a[0], a[1], a[2], a[3]) for a in g["assays"]])
    opt = gu.options_from(g["options"])
    codes, offsets = gu.pack_targets(g["targets"])
    monkeypatch.delenv("TNT_MI355X_UNCACHED_SHARED_OLIGOS", raising=False)
    e = Engine(opt, ex)
    db = e.pack(codes, offsets)
    got = [list(k) for k in tc.sorted_hit_keys(e.search(db))]
    assert got == g["hits"] and len(got) == 113
    assert got == [list(k) for k in tc.sorted_hit_keys(e.search(db))]  # (the replay's buffers are reused)
    assert got == [list(k) for k in tc.sorted_hit_keys(oracle.search(opt, ex, codes, offsets))]
    db.close()
    e.close()
    monkeypatch.setenv("TNT_MI355X_UNCACHED_SHARED_OLIGOS", "1")
    e = Engine(opt, ex)
    db = e.pack(codes, offsets)
    uncached = tc.sorted_hit_keys(e.search(db))
    assert uncached == tc.sorted_hit_keys(oracle.search(opt, ex, codes, offsets, bind_cache=False)) and len(uncached) == 135
    db.close()
    e.close()


def _shared_oligo_workload(kind, seed):
    """Assay sets in which one oligo string is evaluated under two strand concentrations / threshold sets on one target
    (the inputs of tests/test_oracle_vs_reference.py::test_search_with_bind_cache_conflicts, oracle == reference there)."""
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
    else:  # padlock: the upstream arm of one assay is the downstream arm of the next (5' clamp against 3' clamp)
        wl = synth.make_workload(5, n_targets=200, n_assays=20, seed=seed)
        pads = [a for a in wl.assays if a.forward and a.reverse]
        assays = list(wl.assays) + [synth.Assay("swap%d" % i, a.reverse, a.forward, None) for i, a in enumerate(pads)]
        wl.options.probe_clamp_5, wl.options.probe_clamp_3 = 3, 1
    return wl, synth.expand_assays(assays)


@pytest.mark.parametrize("kind,seed", [("taqman", 61), ("taqman", 62), ("degenerate", 63), ("padlock", 64), ("taqman", 65)])
def test_shared_oligos_across_classes_follow_the_bind_cache(Engine, oracle, kind, seed):
    """Row a6: searches of assay sets whose results hang on the reference's BindCache == the oracle (== the reference,
    CPU suite), through the cache replay; the other assays of the same context go the ordinary way."""
    wl, ex = _shared_oligo_workload(kind, seed)
    want = tc.sorted_hit_keys(oracle.search(wl.options, ex, wl.codes, wl.offsets))
    e = Engine(wl.options, ex)
    db = e.pack(wl.codes, wl.offsets)
    got = tc.sorted_hit_keys(e.search(db))
    assert got == want and len(want) > 0
    # the same in two batches of targets (the memo is per target)
    half = wl.n_targets // 2
    parts = []
    for lo, hi in ((0, half), (half, wl.n_targets)):
        d = e.pack(wl.codes[int(wl.offsets[lo]):int(wl.offsets[hi])], wl.offsets[lo:hi + 1] - wl.offsets[lo], first_seq_index=lo)
        parts += list(tc.sorted_hit_keys(e.search(d)))
        d.close()
    assert sorted(parts) == want
    db.close()
    e.close()


@pytest.mark.parametrize("kind,seed,per", [("taqman", 61, 100), ("taqman", 65, 300), ("degenerate", 63, 50), ("padlock", 64, 100)])
def test_bind_cache_on_windowed_targets(Engine, oracle, kind, seed, per):
    """Row a6 on targets that are seeded in windows (records of 50 to 300 kb made of the targets of the cases above): the
    reference's memo is per target whatever its length (tntblast_local.cpp:503-506).  The groups of the assays that share an
    oligo are bound in full by the unit-level instance and replayed per target in assay order on a host thread (pair_group<N,
    true>, the memo as a map); the other assays' lists are paired the ordinary way beside them.  == oracle (== reference:
    tests/test_oracle_vs_reference.py::test_bind_cache_on_a_long_record)."""
    wl, ex = _shared_oligo_workload(kind, seed)
    offsets = np.concatenate([wl.offsets[:-1:per], wl.offsets[-1:]]).astype(np.uint64)
    assert int(np.diff(offsets.astype(np.int64)).max()) > 40000
    want = tc.sorted_hit_keys(oracle.search(wl.options, ex, wl.codes, offsets))
    uncached = tc.sorted_hit_keys(oracle.search(wl.options, ex, wl.codes, offsets, bind_cache=False))
    e = Engine(wl.options, ex)
    db = e.pack(wl.codes, offsets)
    got = tc.sorted_hit_keys(e.search(db))
    assert got == want and len(want) > 20
    assert got == tc.sorted_hit_keys(e.search(db))  # (again: nothing of the replay is left over)
    if kind == "taqman":
        assert want != uncached  # the memo decides hits here
    db.close()
    e.close()


def test_g5_readme_example(Engine):
    fx = gu.load("g5_readme.json.gz")
    ex = synth.expand_assays([synth.Assay("gibb-marburg", *fx["assay"])])
    t = fx["target"]
    e = Engine(gu.options_from(fx["options"]), ex)
    db = e.pack(synth.encode(t), np.array([0, len(t)], dtype=np.uint64))
    hits = e.search(db)
    assert gu.hits_as_keys(hits) == fx["hits"]
    assert e.hit_sequence(db, hits[0]) == t[8:155]
    db.close()
    e.close()


@pytest.mark.parametrize("cfg,nt,na,seed,kw", [
    (2, 3000, 96, 41, {}),
    (3, 3000, 96, 42, {}),
    (4, 800, 120, 43, {}),
    (5, 1500, 120, 44, {}),
    (5, 600, 40, 45, {"assay_format": abi.ASSAY_MIPS, "max_len": 30}),
    (5, 600, 40, 46, {"target_strand": abi.STRAND_MINUS}),
    (3, 800, 40, 47, {"single_primer_pcr": 0, "primer_clamp": 2, "probe_clamp_5": 1, "min_max_primer_clamp": 4}),
    (2, 800, 40, 48, {"min_primer_tm": 0.0, "max_primer_dg": -8.0, "max_mismatch": 2, "max_gap": 0}),
    (2, 800, 40, 49, {"hash_word_size": 6, "max_len": 250}),
    (2, 800, 40, 50, {"target_t": 330.0, "salt": 0.2, "primer_strand": 2e-7}),
    # the shortest hash words (options.cpp:639): seed lists so dense that 1-kb targets are seeded in windows and the
    # chunk-halving / window-shortening repairs run
    (2, 120, 12, 51, {"hash_word_size": 4}),
    (3, 60, 8, 52, {"hash_word_size": 3}),
    (5, 60, 8, 53, {"hash_word_size": 4}),
    # targets beyond the reach of the first cull, with and without probes
    # dG-only thresholds with probes (the pre-filter's bound against the maximum dG)
    (3, 800, 40, 56, {"min_primer_tm": 0.0, "min_probe_tm": 0.0, "max_primer_dg": -9.0, "max_probe_dg": -8.0}),
    (3, 600, 40, 54, {"max_len": 350}),
    (2, 600, 40, 55, {"max_len": 500, "single_primer_pcr": 0}),
])
def test_search_vs_oracle(Engine, oracle, cfg, nt, na, seed, kw):
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets, first_seq_index=1000)
    got = tc.sorted_hit_keys(e.search(db))
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets, first=1000))
    assert got == want
    assert len(want) > 0
    # the unit-level entries agree too (probe / padlock assays only seed the strands --target-strand allows)
    oligos = e.oligos
    allowed = {abi.MINUS, abi.PLUS}
    if wl.options.assay_format != abi.ASSAY_PCR:
        if not (wl.options.target_strand & abi.STRAND_MINUS):
            allowed.discard(abi.MINUS)
        if not (wl.options.target_strand & abi.STRAND_PLUS):
            allowed.discard(abi.PLUS)
    so = sorted((s[0], oligos[s[1]], s[2], s[3], s[4]) for s in oracle.seeds(wl.options, oligos, wl.codes, wl.offsets, first=1000)
                if s[2] in allowed)
    assert sorted(e.seeds(db)) == so
    db.close()
    e.close()


def test_search_vs_reference_where_present(Engine, checkers):
    if not checkers.have_ref():
        pytest.skip("oracle/_ref did not travel")
    ref = checkers.RefImpl()
    for cfg, nt, na in ((3, 1200, 60), (5, 800, 60)):
        wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=77 + cfg)
        e = Engine(wl.options, wl.expanded)
        db = e.pack(wl.codes, wl.offsets)
        assert tc.sorted_hit_keys(e.search(db)) == tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, wl.offsets))
        db.close()
        e.close()


def test_edge_cases(Engine, oracle):
    opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=40.0)
    F, R = "TTCCCCTTTGGAGGCATC", "GGAGGATCCAACAGCAAGG"
    ex = synth.expand_assays([synth.Assay("a", F, R, None)])
    amp = F + "ACGTTGCAAGGCTTAACCGGTTAACC" + synth.revcomp(R)
    targets = ["", "ACG", "ACGTAC", amp, "N" * 50, amp[:30], "ACGT" * 3 + amp + "TTGA", "A" * 200, amp + "X" + amp]
    codes, offsets = gu.pack_targets(targets)
    e = Engine(opt, ex)
    # empty database
    db0 = e.pack(np.zeros(0, np.uint8), np.array([0], dtype=np.uint64))
    assert len(e.search(db0)) == 0
    db0.close()
    db = e.pack(codes, offsets)
    got = tc.sorted_hit_keys(e.search(db))
    want = tc.sorted_hit_keys(oracle.search(opt, ex, codes, offsets))
    assert got == want and len(want) >= 3
    db.close()
    e.close()
    # the longest oligos of the register-resident instances (64) and of the long ones (128) work, 1 025 fail loudly
    rng = np.random.default_rng(5)
    for n_long in (64, 128):
        long_o = "".join("ACGT"[i] for i in rng.integers(0, 4, n_long))
        ex = [synth.ExpandedAssay(0, 0, None, None, long_o)]
        t = "ACGTACGTAC" + synth.revcomp(long_o) + "GGATCCAT" + long_o[:40] + "T" + long_o[40:] + "ACGT"
        codes, offsets = gu.pack_targets([t])
        e = Engine(abi.options_default(min_probe_tm=40.0), ex)
        db = e.pack(codes, offsets)
        got = tc.sorted_hit_keys(e.search(db))
        want = tc.sorted_hit_keys(oracle.search(abi.options_default(min_probe_tm=40.0), ex, codes, offsets))
        assert got == want and len(want) >= 2
        db.close()
        e.close()
    from thermonucleotideblast_amd.engine import TntError
    with pytest.raises(TntError):
        Engine(abi.options_default(), [synth.ExpandedAssay(0, 0, None, None, "ACGT" * 256 + "A")])  # 1 025 bases
    with pytest.raises(TntError):
        Engine(abi.options_default(), [synth.ExpandedAssay(0, 0, None, None, "ACGTNACGTACGTACGT")])


def _run_cli(args):
    import subprocess
    import __graft_entry__ as g
    return subprocess.run([g.CLI] + args, capture_output=True, text=True)


def test_cli_output_golden(Engine, tmp_path):
    """assay file + FASTA database through tntblast-mi355x give, byte for byte, the output file
    the reference binary wrote (PCR, TaqMan, degenerate oligos, padlock + probes, option variants)."""
    g = gu.load("g7_cli.json.gz")
    for case in g["cases"]:
        d = tmp_path / case["name"]
        d.mkdir()
        (d / "assays.txt").write_text(case["assays"])
        (d / "db.fna").write_text(case["fasta"])
        r = _run_cli(["-i", str(d / "assays.txt"), "-d", str(d / "db.fna"), "-o", str(d / "out.txt")] + case["flags"])
        assert r.returncode == 0, r.stderr
        # every file the run writes (-m 2: .sif / .atr, -n T: one per query) and the closing statistics on stdout
        got = {f.name[len("out.txt"):]: f.read_text() for f in sorted(d.iterdir()) if f.name.startswith("out.txt")}
        assert sorted(got) == sorted(case["files"]), case["name"]
        for suffix, text in case["files"].items():
            assert got[suffix] == text, (case["name"], suffix)
        import os
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
        import cli_compare
        assert cli_compare.closing_lines(r.stdout) == case["closing"], case["name"]


@pytest.mark.parametrize("cfg,kw", [(3, {}), (5, {}), (2, {"max_len": 300})])
def test_long_targets_are_searched_in_windows(Engine, oracle, cfg, kw):
    """Targets far longer than one seed work unit (windows of k-mer positions overlapping by the reach of the first
    cull): same hits as the unfragmented search of the oracle, every hit once."""
    rng = np.random.default_rng(100 + cfg)
    wl = synth.make_workload(cfg, n_targets=48, n_assays=24, seed=80 + cfg)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    # glue the 1-kb targets into a few long ones (30 kb, 12 kb, ...) next to short ones: the planted sites end up anywhere,
    # also across window boundaries
    lens = np.diff(wl.offsets.astype(np.int64))
    groups = [30, 12, 1, 1, 1, 1, 1, 1]
    assert sum(groups) == 48
    offsets = np.concatenate([[0], np.cumsum([int(lens[sum(groups[:i]):sum(groups[:i + 1])].sum()) for i in range(len(groups))])]).astype(np.uint64)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, offsets))
    assert len(want) > 10
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    got = tc.sorted_hit_keys(e.search(db))
    assert got == want
    sa = e.seeds(db)
    oligos = _oligo_set(wl.expanded)
    so = oracle.seeds(wl.options, oligos, wl.codes, offsets)
    allowed = wl.options.target_strand
    assert sorted((s[0], s[1], s[2], s[3], s[4]) for s in sa) == sorted((s[0], oligos[s[1]], s[2], s[3], s[4]) for s in so) or allowed != abi.STRAND_BOTH
    db.close()
    e.close()


def test_genome_sized_target(Engine, oracle):
    """One 1.5 Mb target (366 windows) beside short ones, PCR + probe assays with planted sites."""
    wl = synth.make_workload(3, n_targets=1500, n_assays=12, seed=93)
    lens = np.diff(wl.offsets.astype(np.int64))
    offsets = np.array([0, int(lens[:1496].sum())] + [int(lens[:1496 + i + 1].sum()) for i in range(4)], dtype=np.uint64)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, offsets))
    assert len(want) > 20
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    assert tc.sorted_hit_keys(e.search(db)) == want
    db.close()
    e.close()


@pytest.mark.parametrize("cfg,kw", [(2, {}), (3, {}), (2, {"single_primer_pcr": 1}), (2, {"max_len": 300}), (4, {})])
def test_pcr_lists_of_windowed_targets_are_paired_on_the_device(Engine, oracle, cfg, kw, monkeypatch):
    """Row (f)-4: the bind / cull sequence and the pairing loops of the PCR lists of a windowed target run on the device
    (tnt_sort.hip, long_pair_device); TNT_MI355X_LONG_HOST_PAIR=1 leaves them to the host threads' replay of the reference's
    list surgery.  Same hits in the same order either way, equal to the oracle's; tnt_long_pair_stats says who did what."""
    wl = synth.make_workload(cfg, n_targets=900, n_assays=14, seed=71 + cfg)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    lens = np.diff(wl.offsets.astype(np.int64))
    cut = len(lens) - 3  # the first `cut` targets become one long record, three short ones stay
    offsets = np.array([0, int(lens[:cut].sum())] + [int(lens[:cut + i + 1].sum()) for i in range(3)], dtype=np.uint64)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, offsets))
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    got = [tc.hit_key(h) for h in e.search(db)]
    on_device, handed_back, host_lists, device_hits = e.long_pair_stats()
    monkeypatch.setenv("TNT_MI355X_LONG_HOST_PAIR", "1")
    host = [tc.hit_key(h) for h in e.search(db)]
    assert e.long_pair_stats()[0] == 0 and e.long_pair_stats()[3] == 0
    monkeypatch.delenv("TNT_MI355X_LONG_HOST_PAIR")
    assert len(want) > 5
    assert got == host
    assert sorted(got) == want
    assert on_device > 0 and device_hits > 0
    assert on_device + handed_back + host_lists >= 1
    db.close()
    e.close()


@pytest.mark.parametrize("fmt,max_len", [(abi.ASSAY_PADLOCK, 0), (abi.ASSAY_MIPS, 40), (None, 0)])
def test_probe_and_padlock_lists_of_windowed_targets_on_the_device(Engine, oracle, fmt, max_len, monkeypatch):
    """The lists of probe / padlock assays of a windowed target -- the hash overload's lists: the bound seeds by (strand, arm),
    sorted by oligo_info::operator<, one per footprint, then hybrid()'s / padlock()'s loops -- run on the device
    (tnt_sort.hip, long_hash_device); TNT_MI355X_LONG_HOST_HASH_LISTS=1 leaves them to the host threads.  Same hits in the same
    order either way, equal to the oracle's.  fmt None: config 5's mix of padlock and hybridisation-probe assays."""
    if fmt is None:
        wl = synth.make_workload(5, n_targets=700, n_assays=14, seed=83)
        opt, ex, codes = wl.options, wl.expanded, wl.codes
        glued = np.array([0, int(wl.offsets[-1])], dtype=np.uint64)
    else:
        rng = np.random.default_rng(5150 + int(fmt))
        n, tl = 90, 1000
        codes, _ = synth._targets(rng, n, tl)
        arms = [(synth._rand_oligo(rng, 20, 24), synth._rand_oligo(rng, 20, 24)) for _ in range(3)]
        for k in range(150):
            F, R = arms[k % 3]
            gap = "" if fmt == abi.ASSAY_PADLOCK else "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(0, 60))))
            site = synth._mutate(rng, F, 1) + gap + synth._mutate(rng, R, 1)
            if rng.random() < 0.5:
                site = synth.revcomp(site)
            synth._put(codes, 0, int(rng.integers(10, n * tl - 200)), site)
        glued = np.array([0, n * tl], dtype=np.uint64)
        opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=40.0, assay_format=fmt, max_len=max_len)
        ex = synth.expand_assays([synth.Assay("pad%d" % i, F, R, None) for i, (F, R) in enumerate(arms)])
    want = tc.sorted_hit_keys(oracle.search(opt, ex, codes, glued))
    e = Engine(opt, ex)
    db = e.pack(codes, glued)
    got = [tc.hit_key(h) for h in e.search(db)]
    on_device, handed_back, host_lists, device_hits = e.long_pair_stats()
    monkeypatch.setenv("TNT_MI355X_LONG_HOST_HASH_LISTS", "1")
    host = [tc.hit_key(h) for h in e.search(db)]
    stats_host = e.long_pair_stats()
    monkeypatch.delenv("TNT_MI355X_LONG_HOST_HASH_LISTS")
    assert len(want) > 20
    assert got == host
    assert sorted(got) == want
    assert on_device > 0 and device_hits == len(got) and handed_back == 0 and host_lists == 0
    assert stats_host[0] == 0 and stats_host[2] == on_device
    db.close()
    e.close()


def test_a_list_beyond_the_host_replay_s_million_entries_is_paired_on_the_device(Engine, monkeypatch):
    """The host replay keeps a (target, assay) list in arrays of 1 048 576 entries; the device pairing has no such bound.  A
    40-Mb record against one primer pair at -W 5 leaves some 2.8 million seeds behind the first cull: the search succeeds with
    the list on the device (and is refused with the host threads forced).  Without an oracle run of that size, the hits are held
    against the two halves of the record searched as records of their own: a hit needs its two primers bound within reach of
    each other, whatever else the list holds, so every hit of a half is a hit of the whole record."""
    rng = np.random.default_rng(20261)
    n_bases = 40_000_000
    codes = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    F, R = synth._rand_oligo(rng, 22, 22), synth._rand_oligo(rng, 22, 22)
    planted = []
    for k in range(120):
        amp = synth._mutate(rng, F, 1) + "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(30, 300)))) + synth.revcomp(synth._mutate(rng, R, 1))
        if rng.random() < 0.5:
            amp = synth.revcomp(amp)
        pos = int(rng.integers(1000, n_bases - 1000))
        if abs(pos - n_bases // 2) < 5000:
            continue  # (not across the cut)
        synth._put(codes, 0, pos, amp)
        planted.append(pos)
    opt = abi.options_default(min_primer_tm=45.0, hash_word_size=5)
    ex = synth.expand_assays([synth.Assay("pair", F, R, None)])
    e = Engine(opt, ex)
    whole = e.pack(codes, np.array([0, n_bases], dtype=np.uint64))
    hits = e.search(whole)
    on_device, handed_back, host_lists, device_hits = e.long_pair_stats()
    assert on_device == 1 and handed_back == 0 and device_hits == len(hits) and len(hits) >= 60
    monkeypatch.setenv("TNT_MI355X_LONG_HOST_PAIR", "1")
    with pytest.raises(Exception, match="1048576"):
        e.search(whole)
    monkeypatch.delenv("TNT_MI355X_LONG_HOST_PAIR")
    whole.close()
    halves = e.pack(codes, np.array([0, n_bases // 2, n_bases], dtype=np.uint64))
    part = e.search(halves)

    def place(h, shift):
        k = list(tc.hit_key(h))
        names = [n for n, _ in abi.tnt_hit._fields_]
        for n in ("amplicon_start", "amplicon_stop"):
            k[names.index(n)] += shift
        k[names.index("seq_index")] = 0
        return tuple(k)
    got = set(tc.hit_key(h) for h in hits)
    moved = [place(h, (n_bases // 2) * int(h.seq_index if not isinstance(h, dict) else h["seq_index"])) for h in part]
    assert len(moved) >= 60 and all(m in got for m in moved)
    halves.close()
    e.close()


def test_lists_whose_footprints_cross_go_back_to_the_host_threads(Engine, oracle):
    """sort_by_oligo_loc (amplicon_search.cpp:12-26) orders two bound entries by their footprints and any other pair by
    target_loc: where the two disagree on a list, what std::list::sort leaves depends on its merge tree, which only the host
    replay restates.  Here the two primers of an assay overlap (R = X[0:22], F = X[2:24]) and every planted site carries a
    mismatch that puts both seeds on one target position with different footprints: the device flags the list, the host
    threads pair it, the hits equal the oracle's."""
    rng = np.random.default_rng(9173)
    n, tl = 300, 1000
    codes, offsets = synth._targets(rng, n, tl)
    X = synth._rand_oligo(rng, 26, 26)
    R, F = X[0:22], X[2:24]
    other = [(synth._rand_oligo(rng, 20, 24), synth._rand_oligo(rng, 20, 24)) for _ in range(3)]
    for k in range(60):
        site = list(X)
        site[3] = "ACGT"[("ACGT".index(site[3]) + 1 + int(rng.integers(0, 3))) % 4]  # base 3 of R = base 1 of F
        site = "".join(site)
        amp = site + "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(20, 120)))) + synth.revcomp(X)
        if rng.random() < 0.5:
            amp = synth.revcomp(amp)
        synth._put(codes, 0, int(rng.integers(10, n * tl - 400)), amp)
    for k in range(40):  # ordinary amplicons of the other assays: their lists stay on the device
        A, B = other[k % 3]
        amp = synth._mutate(rng, A, 1) + "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(20, 120)))) + synth.revcomp(synth._mutate(rng, B, 1))
        if rng.random() < 0.5:
            amp = synth.revcomp(amp)
        synth._put(codes, 0, int(rng.integers(10, n * tl - 400)), amp)
    glued = np.array([0, n * tl], dtype=np.uint64)  # one 300-kb record: seeded in windows
    opt = abi.options_default(min_primer_tm=40.0, max_len=400, single_primer_pcr=1)
    ex = synth.expand_assays([synth.Assay("overlap", F, R, None)] + [synth.Assay("pcr%d" % i, A, B, None) for i, (A, B) in enumerate(other)])
    want = tc.sorted_hit_keys(oracle.search(opt, ex, codes, glued))
    e = Engine(opt, ex)
    db = e.pack(codes, glued)
    got = tc.sorted_hit_keys(e.search(db))
    on_device, handed_back, host_lists, device_hits = e.long_pair_stats()
    assert got == want and len(want) > 30
    assert handed_back >= 1 and on_device >= 1, (on_device, handed_back, host_lists)
    db.close()
    e.close()


@pytest.mark.parametrize("cfg,kw", [(2, {}), (3, {}), (2, {"single_primer_pcr": 1}), (5, {})])
def test_long_target_first_cull_on_the_device_equals_the_host_replay(Engine, cfg, kw, monkeypatch):
    """The first cull of the lists of windowed targets runs on the device (nearest-element scans, tnt_sort.hip);
    TNT_MI355X_LONG_HOST_CULL=1 replays cull_oligo_match list by list on host threads instead: same hits, same order."""
    wl = synth.make_workload(cfg, n_targets=600, n_assays=16, seed=57)
    for k, v in kw.items():
        setattr(wl.options, k, v)
    lens = np.diff(wl.offsets.astype(np.int64))
    cut = len(lens) - 3  # the first `cut` targets become one long record, three short ones stay
    offsets = np.array([0, int(lens[:cut].sum())] + [int(lens[:cut + i + 1].sum()) for i in range(3)], dtype=np.uint64)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    got = [tc.hit_key(h) for h in e.search(db)]
    monkeypatch.setenv("TNT_MI355X_LONG_HOST_CULL", "1")
    host = [tc.hit_key(h) for h in e.search(db)]
    monkeypatch.delenv("TNT_MI355X_LONG_HOST_CULL")
    assert len(got) > 5
    assert got == host
    db.close()
    e.close()


def test_lists_of_a_chromosome_sized_target(Engine, monkeypatch):
    """A 30-Mb record: ~110 000 seeds per (target, assay) list before the first cull, ~20 000 behind it (round 2 refused
    lists of more than 32 768 seeds).  The device cull and the host replay of cull_oligo_match give the same hits."""
    wl = synth.make_workload(2, n_targets=30000, n_assays=8, seed=61)
    offsets = np.array([0, int(wl.offsets[-1])], dtype=np.uint64)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    got = [tc.hit_key(h) for h in e.search(db)]
    monkeypatch.setenv("TNT_MI355X_LONG_HOST_CULL", "1")
    host = [tc.hit_key(h) for h in e.search(db)]
    monkeypatch.delenv("TNT_MI355X_LONG_HOST_CULL")
    assert len(got) > 50
    assert got == host
    # the amplicons of the 30 000 pieces searched one by one are there too (but for the few whose primer windows the piece ends cut)
    db2 = e.pack(wl.codes, wl.offsets)
    pieces = e.search(db2)
    starts = wl.offsets.astype(np.int64)
    moved = set((h.assay_id, h.degen_id, int(starts[h.seq_index]) + h.amplicon_start, int(starts[h.seq_index]) + h.amplicon_stop) for h in pieces)
    whole = set((h.assay_id, h.degen_id, h.amplicon_start, h.amplicon_stop) for h in e.search(db))
    assert len(moved) > 20 and len(moved & whole) >= 0.95 * len(moved)
    db.close()
    db2.close()
    e.close()


def test_randomised_options_and_shapes(Engine, checkers):
    """A slice of scripts/fuzz_parity.py: random option sets (thresholds, clamps, salt, Ct, temperature, word size 5-8,
    strands, dangles, assay formats) x target shapes (ragged, long, IUPAC-laden, repetitive) against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "40", "11"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-500:]
    assert " 0 differ" in r.stdout


def test_fasta_streaming_reader(Engine, oracle, tmp_path):
    """Batches of whole records (tnt_fasta_open / tnt_fasta_next): same hits as one resident database, global seq_index."""
    import gzip
    from thermonucleotideblast_amd.engine import FastaReader
    wl = synth.make_workload(3, n_targets=500, n_assays=32, seed=71)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    assert len(want) > 5
    path = tmp_path / "db.fna.gz"
    with gzip.open(path, "wb") as f:
        f.write(b"text before the first record\n" + synth.to_fasta(wl.codes, wl.offsets))
    e = Engine(wl.options, wl.expanded)
    for max_bytes in (1 << 16, 150_000, 1 << 30):
        got, n_t, n_batches = [], 0, 0
        for db in FastaReader(e, str(path), max_bytes=max_bytes):
            assert db.defline(0) == "seq%d synthetic target" % n_t
            got += [tc.hit_key(h) for h in e.search(db)]
            n_t += db.n_targets
            n_batches += 1
            db.close()
        assert n_t == wl.n_targets and sorted(got) == want
        assert (n_batches > 3) == (max_bytes < 200_000)
    e.close()


def test_cli_streams_in_batches(Engine, tmp_path, monkeypatch):
    """The front end with a tiny batch size writes the same file."""
    g = gu.load("g7_cli.json.gz")
    case = g["cases"][1]
    (tmp_path / "assays.txt").write_text(case["assays"])
    (tmp_path / "db.fna").write_text(case["fasta"])
    monkeypatch.setenv("TNT_MI355X_CHUNK_BYTES", "20000")
    r = _run_cli(["-i", str(tmp_path / "assays.txt"), "-d", str(tmp_path / "db.fna"), "-o", str(tmp_path / "out.txt")] + case["flags"])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "out.txt").read_text() == case["files"][""]


def test_cli_vs_reference_binary_where_present(Engine, checkers, tmp_path):
    """Fresh, larger inputs through both binaries (the reference binary only travels with oracle/_ref)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "tntblast")):
        pytest.skip("no oracle/_ref/tntblast")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "cli_compare.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]


def test_hit_buffer_grows(Engine, oracle, monkeypatch):
    """More hits than the hit buffer holds: the library grows the buffer and repeats the search."""
    wl = synth.make_workload(5, n_targets=300, n_assays=30, seed=9)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    assert len(want) > 64
    monkeypatch.setenv("TNT_MI355X_HIT_CAP", "16")
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    assert tc.sorted_hit_keys(e.search(db)) == want
    assert tc.sorted_hit_keys(e.search(db)) == want
    db.close()
    e.close()


def test_size_independent_properties(Engine):
    """Full-size-style checks that need no oracle: idempotence, batch-split invariance (targets are
    independent), and reverse-complement symmetry of the hit coordinates."""
    wl = synth.make_workload(2, n_targets=60000, n_assays=96, seed=99)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    a = tc.sorted_hit_keys(e.search(db))
    b = tc.sorted_hit_keys(e.search(db))
    assert a == b and len(a) > 10
    # split into three ragged batches with global seq_index preserved
    cuts = [0, 1234, 40000, wl.n_targets]
    parts = []
    for i in range(3):
        c = wl.codes[int(wl.offsets[cuts[i]]):int(wl.offsets[cuts[i + 1]])]
        o = wl.offsets[cuts[i]:cuts[i + 1] + 1] - wl.offsets[cuts[i]]
        d = e.pack(c, o, first_seq_index=cuts[i])
        parts += list(tc.sorted_hit_keys(e.search(d)))
        d.close()
    assert sorted(parts) == a
    db.close()
    # reverse-complement every target: each amplicon maps to [len-1-stop, len-1-start] on the other strand
    hits_fwd = e.search(e.pack(wl.codes[:int(wl.offsets[3000])], wl.offsets[:3001]))
    comp = np.array([3, 2, 1, 0] + list(range(4, 18)), dtype=np.uint8)
    rc = np.concatenate([comp[wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])][::-1]] for t in range(3000)])
    hits_rc = e.search(e.pack(rc, wl.offsets[:3001]))
    L = 1000
    # F/R amplicons flip strand; single-primer amplicons (same oligo on both sides, amplicon_search.cpp:453-465)
    # keep theirs and swap which site is reported in the forward / reverse slot
    def key(h, rc):
        single = h.forward_oligo == h.reverse_oligo
        tms = (tc.f32_bits(h.forward_tm), tc.f32_bits(h.reverse_tm))
        if single:
            tms = tuple(sorted(tms))
        if rc:
            return (h.assay_id, h.seq_index, L - 1 - h.amplicon_stop, L - 1 - h.amplicon_start) + tms + (h.primer_strand,)
        strand = h.primer_strand if single else 1 - h.primer_strand
        return (h.assay_id, h.seq_index, h.amplicon_start, h.amplicon_stop) + tms + (strand,)
    fwd = sorted(key(h, False) for h in hits_fwd)
    rcs = sorted(key(h, True) for h in hits_rc)
    assert fwd == rcs and len(fwd) > 0
    e.close()


def test_torch_coexists_with_the_library(Engine):
    """torch (device memory / streams / RCCL plumbing) and the HIP library share one HIP runtime."""
    import torch
    assert torch.cuda.is_available()
    x = torch.arange(1024, device="cuda").sum().item()
    assert x == 1023 * 512
    wl = synth.make_workload(1, n_targets=200)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    n1 = len(e.search(db))
    y = (torch.ones(4096, device="cuda") * 2).sum().item()
    assert y == 8192 and n1 == len(e.search(db))
    db.close()
    e.close()


def test_long_targets_in_several_passes(Engine, monkeypatch):
    """A database with several windowed records goes through the long-target path in passes of whole records (bounded device
    memory: TNT_MI355X_LONG_PASS_SEEDS); one record per pass gives the hits of one pass over all of them, in the same order --
    also for the rows of tnt_first_cull."""
    wl = synth.make_workload(3, n_targets=1600, n_assays=24, seed=52)
    lens = np.diff(wl.offsets.astype(np.int64))
    cuts = [0, 500, 900, 1200, 1500] + list(range(1501, 1601))  # records of 500, 400, 300 and 300 kb, a hundred short ones
    offsets = np.array([int(lens[:c].sum()) for c in cuts], dtype=np.uint64)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    def parts(hits):  # (hits of the four windowed records in the order they come, the others -- the pair kernel's, in no order -- sorted)
        return [tc.hit_key(h) for h in hits if h.seq_index < 4], sorted(tc.hit_key(h) for h in hits if h.seq_index >= 4)
    first = e.search(db)
    whole = parts(first)
    n_records_hit = len({h.seq_index for h in first})
    rows = sorted(map(tuple, e.first_cull(db)))
    monkeypatch.setenv("TNT_MI355X_LONG_PASS_SEEDS", "1")
    assert parts(e.search(db)) == whole
    assert sorted(map(tuple, e.first_cull(db))) == rows
    monkeypatch.setenv("TNT_MI355X_LONG_PASS_SEEDS", "60000")  # two or three records per pass
    assert parts(e.search(db)) == whole
    assert n_records_hit > 4 and len(whole[0]) > 20
    db.close()
    e.close()


def test_hits_of_windowed_targets_have_a_device_copy(Engine):
    """Targets seeded in windows are paired on host threads; their hits are copied behind the device's own, so that the
    device view and the compact form (what the multi-GPU gather sends) hold every hit of such a database too."""
    import torch
    from thermonucleotideblast_amd import shard
    wl = synth.make_workload(2, n_targets=3000, n_assays=24, seed=33)
    lens = np.diff(wl.offsets.astype(np.int64))
    cut = 2000  # one 2-Mb record in front of 1 000 short ones
    offsets = np.array([0, int(lens[:cut].sum())] + [int(lens[:cut + i + 1].sum()) for i in range(len(lens) - cut)], dtype=np.uint64)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    e.enqueue(db)
    hits = e.collect(copy=False)
    assert len([h for h in hits if h.seq_index == 0]) > 5 and len([h for h in hits if h.seq_index > 0]) > 5
    dev = e.hits_device()
    assert dev is not None and dev[1] == len(hits)
    nbytes = len(hits) * C.sizeof(abi.tnt_hit)
    t = torch.as_tensor(shard._DeviceBytes(dev[0], nbytes), device="cuda")
    assert bytes(t.cpu().numpy().tobytes()) == bytes(np.frombuffer(hits, dtype=np.uint8).tobytes())
    comp = e.hits_compact()
    assert comp is not None and comp[2] == len(hits)
    blob = bytes(torch.as_tensor(shard._DeviceBytes(comp[0], comp[1]), device="cuda").cpu().numpy().tobytes())
    assert blob == shard.pack_hits(hits)
    db.close()
    e.close()


def test_hits_device_view_matches_host_copy(Engine):
    """tnt_search_hits_device: the records the multi-GPU gather takes from device memory are the host records."""
    import torch
    from thermonucleotideblast_amd import shard
    wl = synth.make_workload(3, n_targets=400, n_assays=24, seed=31)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    e.enqueue(db)
    hits = e.collect(copy=False)
    assert len(hits) > 10
    dev = e.hits_device()
    assert dev is not None and dev[1] == len(hits)
    nbytes = len(hits) * C.sizeof(abi.tnt_hit)
    t = torch.as_tensor(shard._DeviceBytes(dev[0], nbytes), device="cuda")
    assert t.is_cuda and t.numel() == nbytes
    assert bytes(t.cpu().numpy().tobytes()) == bytes(np.frombuffer(hits, dtype=np.uint8).tobytes())
    # single process: gather_hits is the identity
    same, counts = shard.gather_hits(hits, device="cuda", device_hits=dev)
    assert counts == [len(hits)] and same is hits
    # the collective path itself (RCCL, one rank): device records -> gather -> page-locked buffer on the root
    import socket
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    # the compact records the benchmark gathers: packed on the device == packed on the host, and unpacking gives the hits back
    comp = e.hits_compact()
    assert comp is not None and comp[2] == len(hits)
    blob = bytes(torch.as_tensor(shard._DeviceBytes(comp[0], comp[1]), device="cuda").cpu().numpy().tobytes())
    assert blob == shard.pack_hits(hits)
    assert len(blob) < nbytes // 3
    back = shard.unpack_all([np.frombuffer(blob, dtype=np.uint8)])
    assert tc.sorted_hit_keys(back) == tc.sorted_hit_keys(hits) and [h.seq_index for h in back] == [h.seq_index for h in hits]
    try:
        for use_dev in (dev, None):
            got, counts = shard.gather_hits(hits, device="cuda", device_hits=use_dev, always_collective=True)
            assert counts == [len(hits)] and len(got) == len(hits)
            assert bytes(np.frombuffer(got, dtype=np.uint8).tobytes()) == bytes(np.frombuffer(hits, dtype=np.uint8).tobytes())
        blobs, counts = shard.finish_compact(shard.stage_compact((comp[0], comp[1]), comp[2], device="cuda", always_collective=True))
        assert counts == [len(hits)] and bytes(blobs[0].tobytes()) == blob
    finally:
        dist.destroy_process_group()
    db.close()
    e.close()


def test_fastq_ingest_and_streaming(Engine, oracle, tmp_path):
    """FASTQ databases (detected by their first character): the device parser against the oracle's restatement of
    load_fastq / read_bio_seq_fastq_slow, whole and in batches cut at record starts."""
    from thermonucleotideblast_amd.engine import FastaReader
    e = Engine(abi.options_default(), [synth.ExpandedAssay(0, 0, None, None, "ACGTACGTACGTACGTAC")])
    for seed in (1, 2):
        text = synth.quirky_fastq(np.random.default_rng(seed), n_records=120)
        codes, offsets, names = oracle.parse_fasta(text)
        db = e.pack_fasta(text)
        assert db.n_targets == len(names) and db.n_bases == int(offsets[-1])
        for t in range(len(names)):
            assert db.defline(t) == names[t]
            assert np.array_equal(db.target_codes(t), codes[int(offsets[t]):int(offsets[t + 1])]), t
        db.close()
        path = tmp_path / ("reads%d.fastq" % seed)
        path.write_bytes(text)
        t = 0
        n_batches = 0
        for db in FastaReader(e, str(path), max_bytes=1 << 16):
            for k in range(db.n_targets):
                assert db.defline(k) == names[t + k]
                assert np.array_equal(db.target_codes(k), codes[int(offsets[t + k]):int(offsets[t + k + 1])])
            t += db.n_targets
            n_batches += 1
            db.close()
        assert t == len(names)
    # large enough for several batches
    big = b"".join(synth.quirky_fastq(np.random.default_rng(10 + i), n_records=200) + b"\n" for i in range(12))
    codes, offsets, names = oracle.parse_fasta(big)
    path = tmp_path / "big.fastq"
    path.write_bytes(big)
    t, n_batches = 0, 0
    for db in FastaReader(e, str(path), max_bytes=1 << 16):
        for k in range(0, db.n_targets, 7):
            assert db.defline(k) == names[t + k]
            assert np.array_equal(db.target_codes(k), codes[int(offsets[t + k]):int(offsets[t + k + 1])])
        t += db.n_targets
        n_batches += 1
        db.close()
    assert t == len(names) and n_batches > 2
    with pytest.raises(Exception):
        e.pack_fasta(b"@truncated")
    e.close()


def test_shared_oligos_are_bound_once(Engine, oracle, monkeypatch):
    """SURVEY a6 (the reference's per-target BindCache): assays that share oligos -- degenerate expansion, multiplex
    combinations -- bind every (oligo, window) once; the hit set does not change."""
    wl = synth.make_workload(3, n_targets=300, n_assays=12, seed=41)
    # multiplex-like list: every forward primer with every reverse primer and every probe of the first four assays
    base = wl.expanded[:4]
    combos = []
    for i, f in enumerate(base):
        for j, r in enumerate(base):
            for k, p in enumerate(base):
                combos.append(synth.ExpandedAssay(len(combos), len(combos), f.forward, r.reverse, p.probe))
    want = tc.sorted_hit_keys(oracle.search(wl.options, combos, wl.codes, wl.offsets))
    assert len(want) > 10
    # (behind the pre-filter the marking is off by default -- a repeated window costs one cheap sweep there --: ask for it)
    monkeypatch.setenv("TNT_MI355X_DEDUPE", "1")
    e = Engine(wl.options, combos)
    db = e.pack(wl.codes, wl.offsets)
    assert tc.sorted_hit_keys(e.search(db)) == want
    shared = e.timing().n_binds
    db.close()
    e.close()
    monkeypatch.delenv("TNT_MI355X_DEDUPE")
    e = Engine(wl.options, combos)  # the default
    db = e.pack(wl.codes, wl.offsets)
    assert tc.sorted_hit_keys(e.search(db)) == want
    db.close()
    e.close()
    monkeypatch.setenv("TNT_MI355X_NO_DEDUPE", "1")
    e = Engine(wl.options, combos)
    db = e.pack(wl.codes, wl.offsets)
    assert tc.sorted_hit_keys(e.search(db)) == want
    every = e.timing().n_binds
    db.close()
    e.close()
    assert shared * 3 < every * 2, (shared, every)


def test_seed_kernels_agree(Engine, oracle, monkeypatch):
    """The round-3 seed kernel (tnt_seed2.hip.h: private regions, phase-A items straight into their buckets' queues) and
    the round-2 one (TNT_MI355X_SEED_V2=1: sharded lists + bucket pass) hand the bind stage the same work and end in the
    same hits as the oracle."""
    for cfg, nt, na, kw in ((2, 1500, 96, {}), (3, 1500, 96, {}), (5, 800, 60, {}), (4, 500, 90, {}), (3, 500, 30, {"max_len": 300})):
        wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=60 + cfg)
        for k, v in kw.items():
            setattr(wl.options, k, v)
        want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
        res = []
        for old in (False, True):
            if old:
                monkeypatch.setenv("TNT_MI355X_SEED_V2", "1")
            else:
                monkeypatch.delenv("TNT_MI355X_SEED_V2", raising=False)
            e = Engine(wl.options, wl.expanded)
            db = e.pack(wl.codes, wl.offsets)
            hits = tc.sorted_hit_keys(e.search(db))
            t = e.timing()
            cull = sorted(map(tuple, e.first_cull(db))) if wl.options.assay_format == abi.ASSAY_PCR else []
            res.append((hits, cull, (t.n_seeds, t.n_binds, t.n_bound, t.n_hits)))
            db.close()
            e.close()
        monkeypatch.delenv("TNT_MI355X_SEED_V2", raising=False)
        assert res[0][0] == want and res[1][0] == want and len(want) > 0, cfg
        assert res[0][1] == res[1][1], cfg
        assert res[0][2] == res[1][2], (cfg, res[0][2], res[1][2])


def test_window_overflow_in_a_mixed_database(Engine, oracle):
    """A low-complexity stretch in a long target overflows the seed list of its window even with one assay per pass: the
    windows shrink and the long targets are searched again -- the short targets of the same database, whose hits were
    collected before, must be reported once (the split into whole and windowed targets does not move)."""
    wl = synth.make_workload(2, n_targets=64, n_assays=4, seed=97)
    codes = wl.codes.copy()
    lens = np.diff(wl.offsets.astype(np.int64))
    # targets 0..39 glued into one 40-kb target, the rest stay short; a poly-A run inside the long one
    offsets = np.array([0, int(lens[:40].sum())] + [int(lens[:40 + i + 1].sum()) for i in range(24)], dtype=np.uint64)
    codes[5000:6200] = 0
    assays = wl.assays[:3] + [synth.Assay("polyA", "AAAAAAAAAAAAAAAAAAAAAAAA", "TTTTTTTTTTTTTTTTTTTTTTCG", None)]
    ex = synth.expand_assays(assays)
    wl.options.hash_word_size = 5
    want = tc.sorted_hit_keys(oracle.search(wl.options, ex, codes, offsets))
    e = Engine(wl.options, ex)
    db = e.pack(codes, offsets)
    got = tc.sorted_hit_keys(e.search(db))
    assert got == want and len(want) > 0
    assert len(set(got)) == len(got) or len(set(want)) != len(want)
    db.close()
    e.close()


def test_skipped_targets_are_not_searched(Engine, oracle):
    """tnt_db_skip_targets: what the front end does with records beyond --max-target-len (only their fragments are
    searched, tntblast_local.cpp:282-289,438-468)."""
    wl = synth.make_workload(3, n_targets=300, n_assays=16, seed=98)
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    all_hits = tc.sorted_hit_keys(e.search(db))
    mask = np.zeros(300, dtype=np.uint8)
    mask[::3] = 1
    db.skip_targets(mask)
    got = tc.sorted_hit_keys(e.search(db))
    want = [h for h in all_hits if h[2] % 3 != 0]
    assert got == want and 0 < len(want) < len(all_hits)
    db.skip_targets(np.zeros(300, dtype=np.uint8))
    assert tc.sorted_hit_keys(e.search(db)) == all_hits
    db.close()
    e.close()


def test_pipelined_batches_give_the_same_hits(Engine, oracle, monkeypatch):
    """TNT_MI355X_PIPELINE=n: a large search cut into n batches on two buffer sets, the seed stage of one batch on its own
    stream beside the bind stage of the batch before it (off by default: measured slower) -- same hits."""
    wl = synth.make_workload(3, n_targets=110000, n_assays=6, seed=77)
    res = {}
    for n in ("1", "3"):
        monkeypatch.setenv("TNT_MI355X_PIPELINE", n)
        e = Engine(wl.options, wl.expanded)
        db = e.pack(wl.codes, wl.offsets)
        res[n] = [tc.sorted_hit_keys(e.search(db)) for _ in range(2)]  # (twice: the buffer sets are reused)
        db.close()
        e.close()
    monkeypatch.delenv("TNT_MI355X_PIPELINE", raising=False)
    assert res["1"][0] == res["1"][1] == res["3"][0] == res["3"][1] and len(res["1"][0]) > 100
    sample = 4000
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes[:int(wl.offsets[sample])], wl.offsets[:sample + 1]))
    assert [h for h in res["3"][0] if h[2] < sample] == want


def test_queue_region_overflow_is_repaired(Engine, oracle, monkeypatch):
    """The region of a length bucket inside the phase-A queue overflows (TNT_MI355X_QUEUE_SLOTS: a test hook that caps every
    region): the seed kernel raises ERR_ITEM_OVERFLOW, the filter / bind launches of that batch stay inside the bucket's
    own region (they used to be capped by the whole buffer: the next bucket's entries, of another oligo length, or memory
    behind the buffer), and the search is repeated with smaller batches until it fits -- same hits as the oracle."""
    wl = synth.make_workload(5, n_targets=3000, n_assays=40, seed=123)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    monkeypatch.setenv("TNT_MI355X_QUEUE_SLOTS", "8192")
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    got = tc.sorted_hit_keys(e.search(db))
    assert e.timing().n_binds > 8 * 8192  # more windows than the capped regions hold at once: the repair ran
    assert got == want and len(want) > 50
    assert tc.sorted_hit_keys(e.search(db)) == want
    db.close()
    e.close()


def test_gang_gathers_the_hits_of_several_contexts(Engine, oracle):
    """tnt_gang_*: the hits of searches collected on several contexts reach the host as compact blobs, rank by rank.  One
    context = an RCCL communicator of one rank (its blob reaches the receive buffer on the same device, the all-reduce of
    the hit counts runs); three contexts on ONE device = the host-merged transport (RCCL wants one rank per device).
    Either way the records equal those of plain searches and of the oracle."""
    from thermonucleotideblast_amd.engine import Gang
    wl = synth.make_workload(3, n_targets=900, n_assays=24, seed=201)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    cuts = [0, 250, 600, 900]
    for n_rank in (1, 3):
        engines = [Engine(wl.options, wl.expanded) for _ in range(n_rank)]
        gang = Gang(engines)
        assert gang.transport == "rccl" if n_rank == 1 else gang.transport.startswith("host")
        got = []
        for rnd in range(2):  # two rounds: the buffers of the gang are reused
            dbs = []
            for r, e in enumerate(engines):
                lo, hi = (cuts[r], cuts[r + 1]) if n_rank == 3 else (0, 900)
                c = wl.codes[int(wl.offsets[lo]):int(wl.offsets[hi])]
                o = wl.offsets[lo:hi + 1] - wl.offsets[lo]
                dbs.append(e.pack(c, o, first_seq_index=lo))
                e.enqueue(dbs[-1])
            counts = [e.collect_device() for e in engines]
            per_rank = gang.gather()
            assert [len(h) for h in per_rank] == counts
            got = sorted(k for h in per_rank for k in tc.sorted_hit_keys(h))
            assert got == want and len(want) > 20, (n_rank, rnd)
            # a rank may sit a round out
            if n_rank == 3:
                some = gang.gather([1, 0, 1])
                assert some[1] is None and len(some[0]) == counts[0] and len(some[2]) == counts[2]
            for d in dbs:
                d.close()
        gang.close()
        for e in engines:
            e.close()


def test_cli_deals_batches_to_several_contexts(Engine, tmp_path, monkeypatch):
    """tntblast-mi355x --gpus N: the streamed batches are dealt to N contexts (here three on one device: host-merged
    gather), searched by one thread each and filed in batch order -- the output file does not depend on N."""
    import os
    import subprocess
    import __graft_entry__ as g
    wl = synth.make_workload(3, n_targets=700, n_assays=16, seed=202)
    fa, assay = tmp_path / "db.fna", tmp_path / "assays.txt"
    with open(fa, "w") as f:
        for t in range(wl.n_targets):
            f.write(">t%d\n%s\n" % (t, synth.decode(wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])])))
    with open(assay, "w") as f:
        for a in wl.assays:
            f.write("%s\t%s\t%s\t%s\n" % (a.name, a.forward, a.reverse, a.probe))
    outs = {}
    for label, env, extra in (("one", {}, ["--gpus", "1"]), ("three", {"TNT_MI355X_DEVICES": "0,0,0", "TNT_MI355X_CHUNK_BYTES": "60000", "TNT_MI355X_VERBOSE_GANG": "1"}, ["--gpus", "3"]),
                              ("default", {"TNT_MI355X_CHUNK_BYTES": "200000"}, [])):
        out = tmp_path / ("out_%s.txt" % label)
        r = subprocess.run([g.CLI, "-i", str(assay), "-d", str(fa), "-o", str(out), "-e", "40", "-E", "40"] + extra, capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
        outs[label] = open(out).read()
        if label == "three":
            assert "3 device context(s), gather over host" in r.stderr, r.stderr
    assert outs["one"] == outs["three"] == outs["default"] and outs["one"].count("amplicon range") > 20


@pytest.mark.parametrize("cfg", [3, 4, 5])
def test_full_size_configs_through_the_bench(Engine, cfg):
    """BASELINE.json configs 3, 4, 5 at the size of one GPU's shard (1 M x 1 kb x 96 TaqMan triplets; 1.25 M x 2 kb x 1 000
    assays = 2 043 expanded; 625 k targets with degenerate bases x padlock + probe mix), one warm-up + one step of bench.py
    with a parity sample of 16 384 targets against the oracle: the capacity repairs, multi-chunk seed passes and
    multi-million-hit buffers of that scale run under the driver's eyes, not only in profiles/."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", str(cfg), "--steps", "1", "--warmup", "1", "--cpu-seconds", "4",
                        "--parity-targets", "16384"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    print("config %d per-GPU shard: %.1f ms per step, %.3g evals/s, %d hits per step; parity sample %s" % (
        cfg, line["ms_per_step"], line["value"], line["hits_per_step"], line["parity_sample"]))
    assert line["hit_multiset_equal"] is True and line["tm_max_abs_err"] == 0.0 and line["dg_max_abs_err"] == 0.0
    assert line["parity_sample"]["hits_gpu"] == line["parity_sample"]["hits_cpu"] > 0
    assert line["config"]["workload"] and line["n_gpus"] == 1


def test_integration_stub_drops_in(Engine, checkers, tmp_path):
    """INTEGRATION.md's binding compiled for real (oracle/build_stub.py: the reference's tntblast_local.cpp with the work loop
    of its OpenMP region replaced by tnt_ctx_create / tnt_db_create / tnt_search / tnt_hit -> hybrid_sig, linked with the
    reference's other objects and libtnt_mi355x.so): its output files and closing statistics equal the reference binary's on
    every front-end case that needs neither masking nor fragmentation."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "tntblast")) or not os.path.exists(os.path.join(root, "oracle", "_ref", "tntblast_stub")):
        pytest.skip("no oracle/_ref/tntblast or tntblast_stub")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "cli_compare.py"), str(tmp_path), "stub"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-500:]
    assert r.stdout.count("identical") >= 20


def _bind_record_resolved(e, b):
    """gu.bind_record with the alignment read through tnt_long_text and cut at the field's size (where the CPU checkers cut theirs)."""
    rec = gu.bind_record(b) if not bytes(b.align).startswith(b"\x7fLT") else None
    if rec is None:
        raw = bytes(b.align)
        b2 = abi.tnt_bind.from_buffer_copy(b)
        b2.align = b"?"
        rec = gu.bind_record(b2)
        rec[21] = e.long_text(raw)[:abi.TNT_ALIGN_TEXT - 1]
    return rec


def test_g11_oligos_beyond_128_bases_golden(Engine):
    """G11 (reference-generated): probes of 130, 200 and 1 000 bases through the wavefront-per-window kernel (query rows across
    the lanes, 4 / 16 rows per lane, trace words in a slab per wavefront) -- its unit-level form on a sample of seeds, the
    search (seeds in segments of 128 query words, hits of the segments merged), the self-structure values: == the reference.
    Alignment texts beyond the 448-byte fields are read through tnt_long_text."""
    g = gu.load("g11_beyond128.json.gz")
    opt = gu.options_from(g["options"])
    codes, offsets = gu.pack_targets(g["targets"])
    ex = synth.expand_assays([synth.Assay(a[0], a[1], a[2], a[3]) for a in g["assays"]])
    e = Engine(opt, ex)
    om = {o: i for i, o in enumerate(e.oligos)}
    db = e.pack(codes, offsets)
    reqs = tc.make_reqs((r[0], om[g["oligos"][r[1]]], r[2], r[3], r[4], 1, 2.5e-7) for r in g["requests"])
    out = e.bind_batch(db, reqs)
    bad = [(i, gu.compare_bind(x, _bind_record_resolved(e, out[i]))) for i, x in enumerate(g["binds"])]
    assert not [b for b in bad if b[1]], [b for b in bad if b[1]][:2]
    assert sum(1 for x in g["binds"] if x[0] == 0 and len(x[21]) == abi.TNT_ALIGN_TEXT - 1) > 0  # (texts that fill the field: the long ones)
    hits = e.search(db)
    assert [list(k) for k in tc.sorted_hit_keys(hits, text_of=e.long_text)] == g["hits"] and len(g["hits"]) > 5
    for i, o in enumerate(g["oligos"]):
        got = e.oligo_structure(o, g["oligos"][(i + 1) % len(g["oligos"])], 2.5e-7, 2.5e-7)
        assert [tc.f32_bits(v) for v in got] == g["structure"][i]
    db.close()
    e.close()


@pytest.mark.parametrize("seed,lengths", [(1, (150, 220, 400, 1000)), (2, (129, 135, 300, 640)), (3, (257, 513, 1024))])
def test_oligos_beyond_128_bases_vs_oracle(Engine, oracle, seed, lengths):
    """Probe-only assays with probes of 129 to 1 024 bases: search == oracle (== reference, CPU suite), a sample of their
    seeds through the unit-level route == oracle, full alignment texts through tnt_long_text."""
    import longwl
    wl = longwl.make_beyond_128(seed, lengths=lengths, n_targets=8, tl=2600)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    hits = e.search(db)
    assert tc.sorted_hit_keys(hits, text_of=e.long_text) == want and len(want) >= 6
    full = [e.long_text(bytes(h.probe_align)) for h in hits]
    assert max(len(t) for t in full) > 2 * abi.TNT_ALIGN_TEXT  # (three lines of hundreds of columns: far beyond the 448-byte field)
    oligos = sorted({x.probe for x in wl.expanded})
    seeds = oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
    pick = seeds[::max(1, len(seeds) // 120)]
    om = {o: i for i, o in enumerate(e.oligos)}
    want_b = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, tc.make_reqs((s[0], s[1], s[2], s[3], s[4], 1, 2.5e-7) for s in pick))
    got_b = e.bind_batch(db, tc.make_reqs((s[0], om[oligos[s[1]]], s[2], s[3], s[4], 1, 2.5e-7) for s in pick))
    bad = [(i, gu.compare_bind(gu.bind_record(w), _bind_record_resolved(e, got_b[i]))) for i, w in enumerate(want_b)]
    assert not [b for b in bad if b[1]], [b for b in bad if b[1]][:2]
    db.close()
    e.close()


@pytest.mark.parametrize("kind,seed,kw", [("pcr", 0, {}), ("pcr", 1, {}), ("pcr", 6, {"every": 1, "n_targets": 16}), ("taqman", 2, {}), ("taqman", 3, {}), ("padlock", 4, {}),
                                          ("padlock", 5, {})])
def test_primers_beyond_128_bases_vs_oracle(Engine, oracle, kind, seed, kw):
    """PCR primers, TaqMan probes and padlock arms of 129 to 1 000 bases (the reference takes 1 024 in every role,
    nuc_cruc.h:80): seeded in segments of 128 query words whose seeds join ONE list per (target, assay) -- assembled on the
    host, seeds that two segments found on one diagonal reduced to the first -- then culled, bound (wavefront-per-window
    kernel) and paired as the lists of targets that are seeded in windows are.  Search and first cull == oracle (== reference:
    tests/test_oracle_vs_reference.py::test_primers_beyond_128_bases)."""
    import longwl
    wl = longwl.make_primers_beyond_128(kind, seed, **kw)
    if seed == 6:  # one primer of a thousand bases among them
        rng = np.random.default_rng(66)
        F, R = synth._rand_oligo(rng, 1000, 1000), synth._rand_oligo(rng, 22, 22)
        wl.assays.append(synth.Assay("kilo", F, R, None))
        site = longwl._edit(rng, F + "ACGTTGCAAGGCTTAACCGGTTAAGC" + synth.revcomp(R), 30, 2)
        synth._put(wl.codes, int(wl.offsets[3]), 200, site)
        synth._put(wl.codes, int(wl.offsets[5]), 700, synth.revcomp(site))
        wl.expanded = synth.expand_assays(wl.assays)
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    hits = e.search(db)
    assert tc.sorted_hit_keys(hits, text_of=e.long_text) == want and len(want) >= 4
    if kind != "padlock":
        got = e.first_cull(db)
        wantc = oracle.first_cull(wl.options, wl.expanded, wl.codes, wl.offsets)
        assert sorted(map(tuple, got)) == sorted(map(tuple, wantc)) and len(wantc) > 10
    db.close()
    e.close()


def test_primers_beyond_128_bases_beside_ordinary_assays(Engine, oracle):
    """Long primer pairs among ordinary assays (and a probe-only assay with a long probe): every assay of the context takes
    the route of the windowed targets then; same hits as the oracle."""
    import longwl
    wl = longwl.make_primers_beyond_128("pcr", 1)
    base = synth.make_workload(2, n_targets=10, n_assays=12, seed=5)
    rng = np.random.default_rng(8)
    codes = np.concatenate([wl.codes, base.codes])
    offsets = np.concatenate([wl.offsets, base.offsets[1:] + wl.offsets[-1]]).astype(np.uint64)
    P = synth._rand_oligo(rng, 180, 180)
    synth._put(codes, int(offsets[2]), 900, longwl._edit(rng, P, 6, 1))
    assays = list(wl.assays) + list(base.assays) + [synth.Assay("longprobe", None, None, P)]
    ex = synth.expand_assays(assays)
    want = tc.sorted_hit_keys(oracle.search(wl.options, ex, codes, offsets))
    e = Engine(wl.options, ex)
    db = e.pack(codes, offsets)
    got = tc.sorted_hit_keys(e.search(db), text_of=e.long_text)
    assert got == want and len(want) > 20
    db.close()
    e.close()


@pytest.mark.parametrize("kind,seed", [("probe", 1), ("padlock", 3), ("pcr", 4), ("beyond", 2)])
def test_dinkelbach_with_long_oligos_vs_oracle(Engine, oracle, kind, seed):
    """--dinkelbach with oligos of more than 64 bases: the lane-per-window Dinkelbach instances stop at 64 rows, longer
    oligos take the wavefront-per-window kernel's Dinkelbach form (passes over the same window with the per-pass scores
    rebuilt in LDS).  Search and unit-level binds == oracle (== reference, tests/test_oracle_vs_reference.py)."""
    import longwl
    wl = longwl.make(kind, seed) if kind != "beyond" else longwl.make_beyond_128(seed, lengths=(129, 200, 520), n_targets=10, tl=2400)
    wl.options.use_dinkelbach = 1
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    assert tc.sorted_hit_keys(e.search(db), text_of=e.long_text) == want and len(want) >= 5
    oligos = sorted({o for x in wl.expanded for o in (x.forward, x.reverse, x.probe) if o})
    seeds = oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
    pick = seeds[::max(1, len(seeds) // 300)]
    om = {o: i for i, o in enumerate(e.oligos)}
    want_b = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, tc.make_reqs((s[0], s[1], s[2], s[3], s[4], i % 2, 2.5e-7) for i, s in enumerate(pick)))
    got_b = e.bind_batch(db, tc.make_reqs((s[0], om[oligos[s[1]]], s[2], s[3], s[4], i % 2, 2.5e-7) for i, s in enumerate(pick)))
    assert sum(1 for w in want_b if w.status == 0) > 3
    bad = [(i, gu.compare_bind(gu.bind_record(w), _bind_record_resolved(e, got_b[i]))) for i, w in enumerate(want_b)]
    assert not [b for b in bad if b[1]], [b for b in bad if b[1]][:2]
    db.close()
    e.close()


def test_dinkelbach_many_windows_of_long_probes(Engine, oracle):
    """--dinkelbach with probes of 200 and 300 bases on enough targets that each of the two buckets holds thousands of windows:
    the Dinkelbach form of the wavefront kernel runs one wavefront per workgroup with a trace slab per wavefront, and its
    grid must stay inside the slabs its lane stream owns while the other bucket's launch runs beside it (the round-4
    advisor's finding: a grid of 6 workgroups per CU against slabs for 4)."""
    import longwl
    wl = longwl.make_beyond_128(11, lengths=(200, 300), n_targets=48, tl=2400)
    wl.options.use_dinkelbach = 1
    want = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, wl.offsets)
    for _ in range(2):
        assert tc.sorted_hit_keys(e.search(db), text_of=e.long_text) == want and len(want) >= 4
    assert e.timing().n_binds > 4 * 4 * 256, e.timing().n_binds  # more windows than either grid has wavefronts
    db.close()
    e.close()


def test_deferred_finish_changes_no_result(Engine, oracle, monkeypatch):
    """TNT_MI355X_DEFER=1: the throughput bind instances park the windows that pass the Tm / dG tests and bind_finish_kernel
    writes their records and alignment texts (built and measured in round 5, no gain, opt-in): same hits as the default path
    and as the oracle -- PCR, TaqMan, padlock + probes on targets with IUPAC codes -- and the same statuses / records through
    the production instances of the unit-level route."""
    for cfg, nt, na in [(2, 20000, 96), (3, 12000, 48), (5, 4000, 120)]:
        wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=77)
        res = {}
        for mode in ("default", "deferred"):
            monkeypatch.delenv("TNT_MI355X_DEFER", raising=False)
            if mode == "deferred":
                monkeypatch.setenv("TNT_MI355X_DEFER", "1")
            e = Engine(wl.options, wl.expanded)
            db = e.pack(wl.codes, wl.offsets)
            res[mode] = tc.sorted_hit_keys(e.search(db))
            if mode == "deferred" and cfg == 2:
                oligos = sorted({o for x in wl.expanded for o in (x.forward, x.reverse, x.probe) if o})
                seeds = oracle.seeds(wl.options, oligos, wl.codes, wl.offsets)
                pick = seeds[::max(1, len(seeds) // 4000)]
                om = {o: i for i, o in enumerate(e.oligos)}
                want_b = oracle.bind_batch(wl.options, oligos, wl.codes, wl.offsets, tc.make_reqs((s[0], s[1], s[2], s[3], s[4], 0, 9.0e-7) for s in pick))
                got_b = e.bind_batch(db, tc.make_reqs((s[0], om[oligos[s[1]]], s[2], s[3], s[4], 0, 9.0e-7) for s in pick), route=1)
                bad = []
                for i, w in enumerate(want_b):
                    x, got = gu.bind_record(w), gu.bind_record(got_b[i])
                    d = [(k, x[k], got[k]) for k in ([0] + (ROUTED_FIELDS if x[0] == 0 else [])) if x[k] != got[k]]
                    if d:
                        bad.append((i, d))
                assert not bad, bad[:2]
                assert sum(1 for w in want_b if w.status == 0) > 10
            db.close()
            e.close()
        assert res["default"] == res["deferred"] and len(res["default"]) > 0
        if cfg != 5:
            assert res["default"] == tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, wl.offsets))


def test_q11_regime_is_reported(Engine):
    """Thresholds that accept Tm = 0 and dG = 0 (no -e / -E, no negative -g / -G) put a context in the regime of SURVEY Q11, where
    the reference can report hits built from stale alignment state (nuc_cruc.h:360-371): tnt_ctx_warnings says so; with a
    minimum Tm above 0 C it is silent."""
    wl = synth.make_workload(3, n_targets=16, n_assays=4, seed=5)
    e = Engine(wl.options, wl.expanded)
    assert e.warnings() == ""
    e.close()
    wl.options.min_probe_tm = 0.0
    e = Engine(wl.options, wl.expanded)
    w = e.warnings()
    assert "probe thresholds" in w and "nuc_cruc.h:360-371" in w and "primer and" not in w
    e.close()
    wl.options.min_primer_tm = 0.0
    e = Engine(wl.options, wl.expanded)
    assert "primer and probe thresholds" in e.warnings()
    e.close()
    wl.options.max_primer_dg = -5.0
    wl.options.max_probe_dg = -5.0
    e = Engine(wl.options, wl.expanded)
    assert e.warnings() == ""
    e.close()


def test_standalone_front_end_gathers_over_rccl(tmp_path):
    """tntblast-mi355x --gpus 1 with TNT_MI355X_GATHER=rccl: the C++ front end, no torch in the process, has to find librccl
    by itself (tnt_gang_create: dlopen by name, then /opt/rocm/lib, then ROCM_PATH) and says which transport it uses; the run
    fails instead of falling back when RCCL is asked for and missing.  Also: the Q11 warning on stderr."""
    import os
    import subprocess
    import __graft_entry__ as g
    wl = synth.make_workload(2, n_targets=300, n_assays=8, seed=77)
    fa, assay = tmp_path / "db.fna", tmp_path / "assays.txt"
    with open(fa, "w") as f:
        for t in range(wl.n_targets):
            f.write(">t%d\n%s\n" % (t, synth.decode(wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])])))
    with open(assay, "w") as f:
        for a in wl.assays:
            f.write("%s\t%s\t%s\n" % (a.name, a.forward, a.reverse))
    env = dict(os.environ, TNT_MI355X_GATHER="rccl", TNT_MI355X_VERBOSE_GANG="1")
    env.pop("LD_PRELOAD", None)
    out = tmp_path / "out.txt"
    r = subprocess.run([g.CLI, "-i", str(assay), "-d", str(fa), "-o", str(out), "-e", "40", "--gpus", "1"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    assert "1 device context(s), gather over rccl" in r.stderr, r.stderr
    assert "warning" not in r.stderr
    assert open(out).read().count("amplicon range") > 3
    # no minimum Tm, no maximum dG: the Q11 regime is named once on stderr
    r = subprocess.run([g.CLI, "-i", str(assay), "-d", str(fa), "-o", str(tmp_path / "out2.txt"), "-z", "-40", "--gpus", "1"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    assert r.stderr.count("warning: the primer thresholds accept Tm = 0") == 1, r.stderr


def test_cache_replay_of_crowded_targets(Engine, oracle):
    """Row a6 at its capacity edges: ninety assays that share one primer with the probe of a TaqMan assay (every one of them goes
    through the cache replay) all find their amplicons on the same targets -- more groups per target than one lane used to
    hold --, and a low-complexity target gives one of those groups more than 64 items (the replay's instance for crowded
    groups).  Search == oracle."""
    rng = np.random.default_rng(2718)
    F = synth._rand_oligo(rng, 20, 20)
    Rs = [synth._rand_oligo(rng, 19, 22) for _ in range(90)]
    P, F2, R2 = synth._rand_oligo(rng, 24, 24), synth._rand_oligo(rng, 20, 20), synth._rand_oligo(rng, 20, 20)
    assays = [synth.Assay("pcr%d" % i, F, r, None) for i, r in enumerate(Rs[:45])]
    assays.append(synth.Assay("taq", F2, R2, F))  # F as a probe: another strand concentration, other thresholds
    assays += [synth.Assay("pcr%d" % (45 + i), F, r, None) for i, r in enumerate(Rs[45:])]
    ex = synth.expand_assays(assays)
    opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=30.0, primer_clamp=1, probe_clamp_5=2)
    filler = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    targets = []
    for t in range(6):
        seq = filler(30) + F + filler(40)
        for r in Rs:
            seq += synth.revcomp(r if t % 2 == 0 else synth._mutate(rng, r, 1)) + filler(int(rng.integers(5, 30)))
        seq += F2 + filler(20) + synth.revcomp(F) + filler(25) + synth.revcomp(R2) + filler(10)
        targets.append(seq)
    # a crowded group: the shared primer's 3' half over and over (every copy seeds the primer), one reverse site behind them
    targets.append(filler(20) + (F[8:] + "T") * 80 + filler(30) + synth.revcomp(Rs[0]) + filler(20))
    codes, offsets = gu.pack_targets(targets)
    want = tc.sorted_hit_keys(oracle.search(opt, ex, codes, offsets))
    e = Engine(opt, ex)
    db = e.pack(codes, offsets)
    got = tc.sorted_hit_keys(e.search(db))
    assert got == want and len(want) > 200
    db.close()
    e.close()


def test_cache_replay_of_a_list_beyond_512_seeds(Engine, oracle):
    """Row a6 with more than 512 seeds of one assay on one target (the capacity of the device replay's crowded-group instance and of
    a work unit's seed list): the target is seeded in windows and its lists -- up to 2^20 entries -- are replayed on a host
    thread with the memo as a map.  Round 4 refused this input (TNT_E_LIMIT / TNT_E_UNSUPPORTED).  Search == oracle."""
    rng = np.random.default_rng(31415)
    F = synth._rand_oligo(rng, 20, 20)
    R, R1 = synth._rand_oligo(rng, 21, 21), synth._rand_oligo(rng, 20, 20)
    P, F2, R2 = synth._rand_oligo(rng, 24, 24), synth._rand_oligo(rng, 20, 20), synth._rand_oligo(rng, 20, 20)
    assays = [synth.Assay("pcr0", F, R, None), synth.Assay("taq", F2, R2, F), synth.Assay("pcr1", F, R1, None)]
    ex = synth.expand_assays(assays)
    opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=30.0, primer_clamp=1, probe_clamp_5=2)
    filler = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    crowded = filler(20) + (F[8:] + "T") * 700 + filler(30) + synth.revcomp(R) + filler(20) + F2 + filler(15) + synth.revcomp(F) + filler(20) + synth.revcomp(R2) + filler(10)
    plain = filler(40) + F + filler(60) + synth.revcomp(R1) + filler(30) + F2 + filler(20) + F + filler(25) + synth.revcomp(R2) + filler(10)
    codes, offsets = gu.pack_targets([plain, crowded, plain[::-1]])
    want = tc.sorted_hit_keys(oracle.search(opt, ex, codes, offsets))
    e = Engine(opt, ex)
    db = e.pack(codes, offsets)
    got = tc.sorted_hit_keys(e.search(db))
    assert got == want and len(want) > 3
    db.close()
    e.close()


def test_long_alignment_texts_of_windowed_targets(Engine, oracle):
    """Probes of 150 to 300 bases (alignment texts beyond the 448-byte field: long-text arena) on targets that are seeded in
    windows (their binds run in passes of their own behind the whole-target part of the search): the arena keeps the texts of
    the whole search, tnt_long_text resolves the references of either part.  Search == oracle, texts included."""
    import longwl
    wl = longwl.make_beyond_128(9, lengths=(150, 220, 300), n_targets=32, tl=3000)
    # the last twelve targets glued into one record of 36 000 bases (beyond a seed work unit: seeded in windows), the others stay whole
    offsets = np.concatenate([wl.offsets[:21], wl.offsets[-1:]]).astype(np.uint64)
    want_glued = tc.sorted_hit_keys(oracle.search(wl.options, wl.expanded, wl.codes, offsets))
    e = Engine(wl.options, wl.expanded)
    db = e.pack(wl.codes, offsets)
    hits = e.search(db)
    assert tc.sorted_hit_keys(hits, text_of=e.long_text) == want_glued and len(want_glued) >= 8
    assert any(bytes(h.probe_align)[:3] == b"\x7fLT" for h in hits if h.seq_index == 20)  # a reference made by the windowed part
    assert any(bytes(h.probe_align)[:3] == b"\x7fLT" for h in hits if h.seq_index < 20)
    db.close()
    e.close()


@pytest.mark.parametrize("fmt,max_len", [(abi.ASSAY_PADLOCK, 0), (abi.ASSAY_MIPS, 40)])
def test_many_bound_arms_on_one_long_record(Engine, oracle, fmt, max_len):
    """padlock() / the pairing loop of amplicon() on lists with dozens of bound oligos (a 60-kb record that carries eighty
    ligation sites of two padlock assays, arms side by side or up to 60 bases apart, either strand; and eighty amplicons of a
    PCR assay in a second context): the pairing walks only the stretch of the list a pair can come from -- the same hits, in
    the oracle's multiset, as the reference's walk over every pair."""
    rng = np.random.default_rng(4711 + int(fmt))
    n, tl = 60, 1000
    codes, offsets = synth._targets(rng, n, tl)
    arms = [(synth._rand_oligo(rng, 20, 24), synth._rand_oligo(rng, 20, 24)) for _ in range(2)]
    for k in range(80):
        F, R = arms[k % 2]
        gap = "" if fmt == abi.ASSAY_PADLOCK else "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(0, 60))))
        site = synth._mutate(rng, F, 1) + gap + synth._mutate(rng, R, 1)
        if rng.random() < 0.5:
            site = synth.revcomp(site)
        synth._put(codes, 0, int(rng.integers(10, n * tl - 200)), site)
    glued = np.array([0, n * tl], dtype=np.uint64)  # one 60-kb record: seeded in windows, paired whole
    opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=40.0, assay_format=fmt, max_len=max_len)
    ex = synth.expand_assays([synth.Assay("pad%d" % i, F, R, None) for i, (F, R) in enumerate(arms)])
    want = tc.sorted_hit_keys(oracle.search(opt, ex, codes, glued))
    e = Engine(opt, ex)
    db = e.pack(codes, glued)
    assert tc.sorted_hit_keys(e.search(db)) == want and len(want) > 30
    db.close()
    e.close()
    # the same sites as PCR amplicons (forward primer, 0..60 bases, reverse complement of the other arm)
    opt2 = abi.options_default(min_primer_tm=40.0, max_len=200)
    codes2, _ = synth._targets(rng, n, tl)
    for k in range(80):
        F, R = arms[k % 2]
        amp = synth._mutate(rng, F, 1) + "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(5, 60)))) + synth.revcomp(synth._mutate(rng, R, 1))
        if rng.random() < 0.5:
            amp = synth.revcomp(amp)
        synth._put(codes2, 0, int(rng.integers(10, n * tl - 200)), amp)
    ex2 = synth.expand_assays([synth.Assay("pcr%d" % i, F, R, None) for i, (F, R) in enumerate(arms)])
    want2 = tc.sorted_hit_keys(oracle.search(opt2, ex2, codes2, glued))
    e = Engine(opt2, ex2)
    db = e.pack(codes2, glued)
    assert tc.sorted_hit_keys(e.search(db)) == want2 and len(want2) > 30
    db.close()
    e.close()
