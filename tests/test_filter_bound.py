"""CPU: soundness of the bind pre-filter (thermonucleotideblast_amd/csrc/tnt_filter.hip.h), checked on its oracle-side
restatement (tntor_filter_bound; the GPU suite holds the kernel against that restatement integer by integer).

Property: a window the reference's algorithm accepts (valid alignment, Tm >= the minimum) always has bound >= threshold,
so the filter never rejects it.  Checked in the stronger form "the bound is an upper bound of -(dH - Tk dS) + constant of
the best alignment of EVERY window that has one", which gives every random window a say, not only the accepted ones.
Windows: random oligos against sites with substitutions, deletions and insertions (bulges, internal loops), clipped at
the target ends, both strands; option sets with different minimum Tm, salt, temperature and strand concentrations.

The in-suite size is TNT_FILTER_PROPERTY_N windows per option set (default 40 000); scripts/filter_property.py runs the same
check on >= 1 000 000 per set and logs it under profiles/."""
import os

import numpy as np
import pytest

import tntcheck as tc
from thermonucleotideblast_amd import abi, synth

OPTION_SETS = [
    ("default -e 40 -E 40", dict(min_primer_tm=40.0, min_probe_tm=40.0)),
    ("-e 55 -E 48, 0.2 M salt", dict(min_primer_tm=55.0, min_probe_tm=48.0, salt=0.2)),
    ("-e 30 -E 35, 0.9 M salt, 300 K", dict(min_primer_tm=30.0, min_probe_tm=35.0, salt=0.9, target_t=300.0)),
    ("-e 20 -E 62, 10 mM salt, 333 K", dict(min_primer_tm=20.0, min_probe_tm=62.0, salt=0.01, target_t=333.15)),
    ("-e 45 -E 45, asymmetric Ct", dict(min_primer_tm=45.0, min_probe_tm=45.0, primer_strand=2.0e-6, probe_strand=5.0e-8)),
    # dG-only runs (options.cpp:833-916 accepts -z / -g without -e): the bound is taken against the maximum dG
    ("-g -8 -G -6, no minimum Tm", dict(min_primer_tm=0.0, min_probe_tm=0.0, max_primer_dg=-8.0, max_probe_dg=-6.0)),
    ("-g -12 -G -10, 0.2 M salt, 300 K", dict(min_primer_tm=0.0, min_probe_tm=0.0, max_primer_dg=-12.0, max_probe_dg=-10.0, salt=0.2, target_t=300.0)),
    ("-g -5 -G -7, 10 mM salt, 330 K", dict(min_primer_tm=0.0, min_probe_tm=0.0, max_primer_dg=-5.0, max_probe_dg=-7.0, salt=0.01, target_t=330.0)),
    ("-e 45 for primers, -G -9 for probes", dict(min_primer_tm=45.0, min_probe_tm=0.0, max_probe_dg=-9.0)),
    # dangling-end options (nuc_cruc.cpp:1068-1137): one more column per end, covered by a constant in the threshold
    ("-e 40 -E 40, --dangle5 T --dangle3 T", dict(min_primer_tm=40.0, min_probe_tm=40.0, allow_dangle_5=1, allow_dangle_3=1)),
    ("-e 50 -E 45, --dangle3 T, 0.2 M salt", dict(min_primer_tm=50.0, min_probe_tm=45.0, allow_dangle_3=1, salt=0.2)),
    ("-g -9 -G -8, --dangle5 T, 10 mM salt", dict(min_primer_tm=0.0, min_probe_tm=0.0, max_primer_dg=-9.0, max_probe_dg=-8.0, allow_dangle_5=1, salt=0.01)),
]


def make_windows(rng, n, lo=12, hi=40, degen=0.3, inosine=0.04):
    """`degen`: share of the targets that get one to three IUPAC letters (compatible with the base they replace or not)
    or -- a share `inosine` of those letters -- an inosine (a column the filter reads like an IUPAC code: the maximum
    over A C G T I)."""
    oligos = sorted({synth._rand_oligo(rng, lo, hi) for _ in range(64)})
    targets, rows = [], []
    for t in range(n):
        o = int(rng.integers(0, len(oligos)))
        q = oligos[o]
        minus = rng.random() < 0.5
        site = list(q) if minus else list(synth.revcomp(q))
        kind = rng.random()
        n_edit = 0 if kind < 0.15 else int(rng.integers(1, 7))
        if kind > 0.9:  # unrelated sequence with a short seed of the site
            keep = int(rng.integers(0, len(site) - 7))
            site = ["ACGT"[i] for i in rng.integers(0, 4, len(site))]
            site[keep:keep + 8] = (list(q) if minus else list(synth.revcomp(q)))[keep:keep + 8]
        for _ in range(n_edit):
            k = int(rng.integers(0, len(site)))
            r = rng.random()
            if r < 0.55:
                site[k] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.8 and len(site) > 10:
                del site[k:k + int(rng.integers(1, 4))]
            else:
                site[k:k] = ["ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 4)))]
        left, right = int(rng.integers(0, 10)), int(rng.integers(0, 10))
        seq = "".join("ACGT"[i] for i in rng.integers(0, 4, left)) + "".join(site) + "".join("ACGT"[i] for i in rng.integers(0, 4, right))
        if rng.random() < degen:
            seq = list(seq)
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, len(seq)))
                r = rng.random()
                if r < inosine:
                    seq[k] = "I"
                elif r < 0.5:  # a code that contains the base it replaces
                    seq[k] = str(rng.choice([c for c in "MRSVWYHKDBN" if seq[k] in synth.expand_oligo(c)] or ["N"]))
                else:
                    seq[k] = "MRSVWYHKDBN"[int(rng.integers(0, 11))]
            seq = "".join(seq)
        targets.append(seq)
        ql = int(rng.integers(0, max(1, len(q) - 7)))
        role = int(rng.integers(0, 2))
        rows.append((t, o, abi.MINUS if minus else abi.PLUS, ql, min(len(seq) - 1, left + ql), role, None))
    return oligos, targets, rows


def check_option_set(oracle, name, kw, n, seed, tier=1, **window_kw):
    rng = np.random.default_rng([2024, seed])
    opt = abi.options_default(**kw)
    oligos, targets, rows = make_windows(rng, n, **window_kw)
    rows = [r[:6] + ((opt.probe_strand if r[5] else opt.primer_strand),) for r in rows]
    codes = np.concatenate([synth.encode(t) for t in targets])
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    reqs = tc.make_reqs(rows)
    out = oracle.bind_batch(opt, oligos, codes, offsets, reqs)
    fb = oracle.filter_bound(opt, oligos, codes, offsets, reqs, tier=tier)
    role = np.array([r[5] for r in rows])
    min_tm = np.where(role == 1, opt.min_probe_tm, opt.min_primer_tm).astype(np.float64)
    max_dg = np.where(role == 1, opt.max_probe_dg, opt.max_primer_dg).astype(np.float64)
    valid = np.array([b.valid for b in out]) > 0
    tm = np.array([b.tm for b in out], dtype=np.float64)
    dH = np.array([b.dH for b in out], dtype=np.float64)
    dS = np.array([b.dS for b in out], dtype=np.float64)
    dg = np.array([b.dg for b in out], dtype=np.float64)
    bound, theta = fb[:, 0].astype(np.int64), fb[:, 1].astype(np.int64)
    # the threshold the class's bound is taken against: the minimum Tm where there is one, else the maximum dG
    by_tm = min_tm > 0.0
    accepted = valid & np.where(by_tm, tm >= min_tm, dg <= max_dg)
    rejected = (bound < theta) & (bound < 32767)  # a saturated bound is never rejected
    assert accepted.sum() > n // 50, (name, int(accepted.sum()))
    bad = np.nonzero(accepted & rejected)[0]
    assert len(bad) == 0, "%s: the filter rejects %d accepted windows, e.g. request %r bound %d theta %d tm %.3f" % (
        name, len(bad), rows[bad[0]], bound[bad[0]], theta[bad[0]], tm[bad[0]])
    # stronger: the bound exceeds its threshold by at least as much as the scored alignment exceeds the reference's -- i.e.
    # bound >= -(dH - Tk dS) + G_init + min(0, 2 G_AT) (- the dangling-end slack) for every window with an alignment.
    # Phi = -(dH - Tk dS); the reference accepts iff Phi >= -Tk R ln(Ct) (minimum Tm) resp. Phi >= -max_dg (maximum dG); the
    # checker's threshold carries those terms, a margin of 0.02 kcal/mol and one unit of rounding.
    Tk = np.where(by_tm, min_tm + 273.15, float(opt.target_t))
    ct = np.array([r[6] for r in rows], dtype=np.float64)
    ref_threshold = np.where(by_tm, -Tk * 1.9872e-3 * np.log(ct.astype(np.float32)).astype(np.float64), -max_dg)
    phi = -(dH - Tk * dS)
    slack = np.where(bound >= 32767, 1.0e3, (bound - theta) / 1.0e3 - (phi - ref_threshold))  # (saturated: no statement)
    worst = slack[valid].min()
    assert worst > 0.0, "%s: bound below the scored alignment by %.4f kcal/mol beyond the margin" % (name, -worst)
    worst -= 0.02  # (reported without the margin, like round 2's logs)
    return {"name": name, "windows": n, "with_alignment": int(valid.sum()), "accepted": int(accepted.sum()),
            "rejected_by_filter": int(rejected.sum()), "min_slack_kcal": float(worst),
            "reject_rate_of_non_accepted": float(rejected[~accepted].mean())}


@pytest.mark.parametrize("k", range(len(OPTION_SETS)))
def test_filter_never_rejects_an_accepted_window(oracle, k):
    n = int(os.environ.get("TNT_FILTER_PROPERTY_N", "40000"))
    name, kw = OPTION_SETS[k]
    res = check_option_set(oracle, name, kw, n, k)
    assert res["rejected_by_filter"] > 0  # the filter does something


@pytest.mark.parametrize("k", range(len(OPTION_SETS)))
def test_second_tier_never_rejects_an_accepted_window(oracle, k):
    """The pre-filter's second tier (tnt_sbound.hip.h; restated in tntor_filter_bound under tier 3 = the smaller of both tiers'
    bounds): the credit of align_dimer's maximal-score paths is an upper bound of Phi - const of the alignment NucCruc scores,
    for every window that has one -- and it rejects more than the first tier does."""
    n = int(os.environ.get("TNT_FILTER_PROPERTY_N", "40000")) // 2
    name, kw = OPTION_SETS[k]
    one = check_option_set(oracle, name, kw, n, 400 + k, tier=1)
    two = check_option_set(oracle, name + " (both tiers)", kw, n, 400 + k, tier=3)  # asserts "accepted => not rejected" and bound >= Phi
    assert two["rejected_by_filter"] > one["rejected_by_filter"]


@pytest.mark.parametrize("k", [0, 3, 5])
def test_second_tier_candidate_is_sound_and_gains_little(oracle, k):
    """Round 5 tried a second tier on the pre-filter's survivors: the same decomposition of an alignment at its Watson-Crick columns,
    but a terminal stack charged AS IT IS instead of dropped when unfavourable (two states per diagonal cell: counted column /
    inside a region; restated in tntor_filter_bound under tntor_set_filter_tier(2)).  It is sound by the first tier's argument and
    never above the first tier's bound -- and rejects almost nothing more: an unfavourable terminal stack is dodged by counting
    the helix one pair short, which is what evaluate_alignment itself would do for an alignment that leaves the pair unpaired
    (a symmetric loop with one mismatch costs nothing, loop_S[2] = 0).  Not built as a kernel; the measurement is kept."""
    n = int(os.environ.get("TNT_FILTER_PROPERTY_N", "40000")) // 2
    name, kw = OPTION_SETS[k]
    one = check_option_set(oracle, name, kw, n, 300 + k, tier=1)
    two = check_option_set(oracle, name + " (second tier)", kw, n, 300 + k, tier=2)  # asserts "accepted => not rejected" and the stronger bound
    assert two["rejected_by_filter"] >= one["rejected_by_filter"]
    assert two["reject_rate_of_non_accepted"] - one["reject_rate_of_non_accepted"] < 0.05


@pytest.mark.parametrize("k", [0, 5, 9])
def test_filter_with_inosine_in_the_target(oracle, k):
    """Windows with one to three inosines (DB_I, seq.h:19) in the target: the column takes the maximum over A C G T I."""
    n = int(os.environ.get("TNT_FILTER_PROPERTY_N", "40000"))
    name, kw = OPTION_SETS[k]
    res = check_option_set(oracle, name + ", inosine in the targets", kw, n, 100 + k, degen=1.0, inosine=0.7)
    assert res["rejected_by_filter"] > 0


# thresholds a run with long hybridisation probes / MIPS arms would set (at -e 40 an unrelated 120-base window already reaches the
# bound: 2 % rejected; at -e 60 three quarters of the windows that fail are rejected)
LONG_OLIGO_SETS = [
    ("-e 60 -E 60, oligos of 65-128 bases", dict(min_primer_tm=60.0, min_probe_tm=60.0)),
    ("-e 40 -E 70, oligos of 65-128 bases", dict(min_primer_tm=40.0, min_probe_tm=70.0)),
    ("-g -25 -G -25, no minimum Tm, oligos of 65-128 bases", dict(min_primer_tm=0.0, min_probe_tm=0.0, max_primer_dg=-25.0, max_probe_dg=-25.0)),
]


@pytest.mark.parametrize("k", range(len(LONG_OLIGO_SETS)))
def test_filter_with_oligos_of_65_to_128_bases(oracle, k):
    """The same bound (bulges up to the longest window, loops up to 2 x 128 + the window are priced; perfect sites saturate
    the 16-bit sums and pass)."""
    n = int(os.environ.get("TNT_FILTER_PROPERTY_N", "40000")) // 8
    name, kw = LONG_OLIGO_SETS[k]
    res = check_option_set(oracle, name, kw, n, 200 + k, lo=65, hi=128)
    assert res["rejected_by_filter"] > n // 100


def test_filter_is_off_where_its_premises_fail(oracle):
    rng = np.random.default_rng(5)
    oligos, targets, rows = make_windows(rng, 200)
    codes = np.concatenate([synth.encode(t) for t in targets])
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    reqs = tc.make_reqs([r[:6] + (9.0e-7,) for r in rows])
    for kw in (dict(min_primer_tm=0.0, min_probe_tm=0.0), dict(min_primer_tm=-5.0, min_probe_tm=0.0, max_primer_dg=0.0, max_probe_dg=3.0),
               dict(use_dinkelbach=1, min_primer_tm=40.0, min_probe_tm=40.0)):
        fb = oracle.filter_bound(abi.options_default(**kw), oligos, codes, offsets, reqs)
        assert (fb[:, 1] == np.iinfo(np.int32).min).all(), kw
    # an inosine or an IUPAC code in the target only loosens the bound; a code the window loader skips takes the window out
    # of the filter's reach
    opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=40.0)
    oligos, targets, rows = make_windows(rng, 200, degen=0.0)
    codes = np.concatenate([synth.encode(t) for t in targets])
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    reqs = tc.make_reqs([r[:6] + (9.0e-7,) for r in rows])
    plain = oracle.filter_bound(opt, oligos, codes, offsets, reqs)
    codes2 = codes.copy()
    codes2[int(offsets[3]) + 12] = abi.DB_CODES.index("I")
    codes2[int(offsets[4]) + 12] = abi.DB_CODES.index("N")
    codes2[int(offsets[5]) + 12] = 16  # DB_GAP (seq.h:34): skipped by the window loader
    fb = oracle.filter_bound(opt, oligos, codes2, offsets, reqs)
    assert tuple(fb[5]) == (np.iinfo(np.int32).max, np.iinfo(np.int32).min)
    assert fb[3, 0] >= plain[3, 0] and fb[3, 1] == plain[3, 1]
    assert fb[4, 0] >= plain[4, 0] and fb[4, 1] == plain[4, 1]
    assert (np.delete(fb, [3, 4, 5], axis=0) == np.delete(plain, [3, 4, 5], axis=0)).all()
