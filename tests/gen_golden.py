#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE (oracle/_ref/libtntref.so, built
from the sources under /root/reference by oracle/Makefile).  Only runs in the build container.

  g2_binds.json.gz    G2: (oligo, target window, strand, Ct) cases through the reference's NucCruc:
                      perfect sites, 1-4 substitutions, indels (bulges / internal loops), end fraying,
                      inosine in the oligo, every IUPAC target code, windows truncated at sequence ends,
                      low-complexity repeats (multiple maximal cells / path splits), random seeds.
                      tm/dH/dS/dG as float32 bit patterns, anchors, counts, ranges, alignment text.
  g3_structure.json.gz  G3: hairpin / homodimer / heterodimer Tm of assay oligos (tntblast_local.cpp:656-686): random
                      oligos, designed stem-loops (incl. the special tri/tetra-loops), palindromes, repeats.
  g6_fasta.json.gz    G6: FASTA ingest (sequence_data_fastx.cpp:13-79,187-382): a text exercising every reader rule (CRLF, blank
                      lines, lower case, U, IUPAC, unknown characters, '*', '-', '>' inside deflines and mid-line) and the
                      codes / deflines the reference's sequence_data reads from it.
  g7_cli.json.gz      G7: end to end -- assay file + FASTA database through the reference BINARY (oracle/_ref/tntblast,
                      OMP_NUM_THREADS=1) for PCR, TaqMan, degenerate, padlock + probe and option-variant cases; the
                      expected "-m 0" output files, byte for byte.
  g4_seeds.json.gz    G4: seed lists, both strands, incl. N-runs / IUPAC targets (pins SURVEY Q1/Q2).
  g5_search_cfgN.json.gz  G5: end-to-end hit sets of miniature copies of BASELINE configs 1-5 and of the
                      README amplicon case (README.md:130-203).
  (g1_tables.json is written by oracle/gen_params.py.)
Fixtures hold inputs and expected outputs only.
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import tntcheck as tc  # noqa: E402
from thermonucleotideblast_amd import abi, synth  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def dump(name, obj):
    with open(os.path.join(GOLD, name), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:  # byte-stable
        f.write(json.dumps(obj, separators=(",", ":")).encode())
    print(name, os.path.getsize(os.path.join(GOLD, name)), "bytes")


def opt_dict(o):
    return {n: getattr(o, n) for n, _ in abi.tnt_options._fields_}


def bind_cases(rng):
    """Hand-built targets: each target carries one site derived from an oligo + flanks."""
    oligos = []
    for L in (18, 20, 22, 25, 28, 33, 40):
        for _ in range(3):
            oligos.append(synth._rand_oligo(rng, L, L))
    oligos += ["TTCCCCTTTGGAGGCATC", "GGAGGATCCAACAGCAAGG", "CGATGGGCTTTCAGGACAGGTGT",      # README assay
               "AAAAAAAAAAAAAAAAAAAA", "ATATATATATATATATATAT", "GCGCGCGCGCGCGCGCGC", "ACACACACACACACACACAC",  # low complexity
               "GGGGGGGGGGTTTTTTTTTT", "ACGTACGTACGTACGTACGT",
               "ACGTIACGTTGCAIGGATCCA", "TTGIIACGATCGATCGGATC"]                           # inosine (Q2)
    targets, reqs = [], []

    def add(site, o, strand, qloc=0, left=None, right=None):
        left = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(0, 12)))) if left is None else left
        right = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(0, 12)))) if right is None else right
        t = left + site + right
        targets.append(t)
        tloc = max(0, len(left) + qloc)
        reqs.append((len(targets) - 1, o, strand, qloc, tloc, int(rng.integers(0, 2)), float(rng.choice([9e-7, 2.5e-7, 4.5e-7]))))

    for oi, o in enumerate(oligos):
        plain = o.replace("I", "ACGT"[int(rng.integers(0, 4))])
        for rep in range(14):
            site = list(plain)
            kind = rep % 7
            if kind == 1:
                for _ in range(int(rng.integers(1, 5))):
                    i = int(rng.integers(0, len(site)))
                    site[i] = "ACGT"[("ACGT".index(site[i]) + int(rng.integers(1, 4))) % 4]
            elif kind == 2:   # deletion(s) in the target -> bulge in the query
                i = int(rng.integers(3, len(site) - 3))
                del site[i:i + int(rng.integers(1, 4))]
            elif kind == 3:   # insertion(s) in the target
                i = int(rng.integers(3, len(site) - 3))
                site[i:i] = list("".join("ACGT"[j] for j in rng.integers(0, 4, int(rng.integers(1, 4)))))
            elif kind == 4:   # internal loop: mismatches next to an indel
                i = int(rng.integers(4, len(site) - 5))
                site[i] = "ACGT"[("ACGT".index(site[i]) + 1) % 4]
                site[i + 1] = "ACGT"[("ACGT".index(site[i + 1]) + 2) % 4]
                if rng.random() < 0.5:
                    del site[i + 2]
            elif kind == 5:   # IUPAC codes in the target
                for _ in range(int(rng.integers(1, 6))):
                    i = int(rng.integers(0, len(site)))
                    site[i] = "MRSVWYHKDBNI"[int(rng.integers(0, 12))]
            elif kind == 6:   # frayed ends / G.T tandems
                site[0] = "ACGT"[("ACGT".index(site[0]) + 1) % 4]
                site[-1] = "ACGT"[("ACGT".index(site[-1]) + 1) % 4]
            site = "".join(site)
            qloc = int(rng.integers(0, 6))
            if rep % 2 == 0:
                add(site, oi, abi.MINUS, qloc)                    # oligo == plus strand text
            else:
                add(synth.revcomp(site), oi, abi.PLUS, qloc)
        # windows clipped by the ends of the sequence
        add(plain[3:], oi, abi.MINUS, 0, left="", right="AC")
        add(plain[:-3], oi, abi.MINUS, 0, left="GT", right="")
        add(synth.revcomp(plain)[2:], oi, abi.PLUS, 0, left="", right="")
        # unknown / gap codes inside the window (Q6)
        add(plain[:6] + "X" + plain[6:], oi, abi.MINUS, 0)
    return oligos, targets, reqs


def structure_oligos(rng):
    """Oligos for G3: random, README, palindromes / repeats, designed hairpins with 3..7-nt loops."""
    oligos = [synth._rand_oligo(rng, 15, 40) for _ in range(150)]
    oligos += ["GGAGGATCCAACAGCAAGG", "TTCCCCTTTGGAGGCATC", "CGATGGGCTTTCAGGACAGGTGT", "GCGCGCGCGCGCGCGC", "ATATATATATATATAT",
               "AAAAAAAATTTTTTTT", "GGGGGAAAACCCCC", "ACGTIACGTTGCAIGGATCCA", "GCGAAGC", "GCGAAAGC", "ACGT", "ACGTA",
               "CGCGAATTCGCG", "GGGCGAAGCCC", "GGGCGAAAGCCC", "CCCTTTTGGG"]
    for _ in range(100):
        stem = synth._rand_oligo(rng, 4, 9)
        loop = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(3, 8))))
        head = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(0, 6))))
        tail = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(0, 5))))
        oligos.append(head + stem + loop + synth.revcomp(stem) + tail)
    return oligos


def g7_case(cli_compare, name, cfg, nt, na, seed, flags):
    """One end-to-end case through the reference binary: inputs + every output file + the closing stdout lines."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        cli_compare.write_case(d, cfg, nt, na, seed, fastq=(name == "fastq"), name=name)
        out = os.path.join(d, "ref.out")
        r = subprocess.run([cli_compare.REF, "-i", os.path.join(d, "assays.txt"), "-d", os.path.join(d, "db.fna"), "-o", out] + flags,
                           env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        files = {f[len("ref.out"):]: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f.startswith("ref.out")}
        case = {"name": name, "flags": flags, "assays": open(os.path.join(d, "assays.txt")).read(),
                "fasta": open(os.path.join(d, "db.fna")).read(), "files": files, "closing": cli_compare.closing_lines(r.stdout)}
    print("G7", name, {k: len(v) for k, v in files.items()}, "bytes,", sum(v.count("name = ") for v in files.values()), "records,",
          len(case["closing"]), "closing stdout lines")
    return case


def add_g7_cases(names):
    """`python tests/gen_golden.py g7 <case> ...`: (re)generates only the named G7 cases, the others stay as they are."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import cli_compare
    import golden_util as gu
    have = gu.load("g7_cli.json.gz")["cases"]
    for name, cfg, nt, na, seed, flags in cli_compare.CASES:
        if name in names:
            have = [c for c in have if c["name"] != name] + [g7_case(cli_compare, name, cfg, nt, na, seed, flags)]
    dump("g7_cli.json.gz", {"cases": have})


def gen_g8():
    """G8: the reference's per-target BindCache hands the FIRST evaluation of an (oligo string, window) to every later
    caller, whatever its strand concentration and thresholds (tntblast.h:247-270, bind_oligo.cpp:600-612,750-803).  An
    assay set in which one oligo is a primer (-e 50) of one assay and the TaqMan probe (-E 25) of another: the reference's
    hits (probe Tm at the primer's Ct, records missing where the first cascade stopped early) next to the inputs."""
    ref = tc.RefImpl()
    rng = np.random.default_rng(8)
    X, Y, F2, R2 = (synth._rand_oligo(rng, n, n) for n in (22, 20, 20, 20))
    assays = [synth.Assay("pcr", X, Y, None), synth.Assay("taqman", F2, R2, X)]
    opt = abi.options_default(min_primer_tm=50.0, min_probe_tm=25.0)
    targets = []
    for t in range(80):
        site = synth._mutate(rng, X, 4)
        amp = list("ACGT"[i] for i in rng.integers(0, 4, 300))
        amp[:len(F2)] = F2
        amp[40:40 + len(X)] = site
        amp[150:150 + len(Y)] = synth.revcomp(synth._mutate(rng, Y, 1))
        amp[300 - len(R2):] = synth.revcomp(R2)
        seq = "".join("ACGT"[i] for i in rng.integers(0, 4, 50)) + "".join(amp) + "".join("ACGT"[i] for i in rng.integers(0, 4, 50))
        targets.append(seq if t % 2 else synth.revcomp(seq))
    codes = np.concatenate([synth.encode(t) for t in targets])
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    hits = ref.search(opt, synth.expand_assays(assays), codes, offsets, threads=1)
    dump("g8_bindcache.json.gz", {"assays": [[a.name, a.forward, a.reverse, a.probe] for a in assays], "options": opt_dict(opt),
                                  "targets": targets, "hits": [list(k) for k in tc.sorted_hit_keys(hits)]})
    print("G8: %d reference hits" % len(hits))


def add_g5_inputs():
    """`python tests/gen_golden.py g5-inputs`: stores the inputs of the G5 cases (targets, assay rows) inside their fixtures
    -- they used to be regenerated from numpy seeds by the tests -- after checking that they still reproduce the stored hits."""
    import golden_util as gu
    ref = tc.RefImpl()
    for cfg in (1, 2, 3, 4, 5):
        name = "g5_search_cfg%d.json.gz" % cfg
        fx = gu.load(name)
        wl = synth.make_workload(cfg, n_targets=fx["n_targets"], n_assays=fx["n_assays"])
        hits = [list(k) for k in tc.sorted_hit_keys(ref.search(wl.options, wl.expanded, wl.codes, wl.offsets))]
        assert hits == fx["hits"], "config %d: the regenerated inputs no longer give the stored hits" % cfg
        fx["assays"] = [[a.name, a.forward, a.reverse, a.probe] for a in wl.assays]
        fx["targets"] = [synth.decode(wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])]) for t in range(wl.n_targets)]
        dump(name, fx)
        print("G5 cfg", cfg, "inputs stored:", len(fx["targets"]), "targets,", len(fx["assays"]), "assays")


def gen_g9():
    """G9: --dinkelbach (NucCruc::approximate_tm_heterodimer's iteration, nuc_cruc.cpp:2399-2432).  Every third request of
    G2 (its oligos, targets and requests stay in g2_binds.json.gz) through the reference with use_dinkelbach, two option
    sets, the same record fields as G2; plus hairpin / homodimer / heterodimer Tm of the G3 oligos."""
    import golden_util as gu
    ref = tc.RefImpl()
    fx = gu.load("g2_binds.json.gz")
    codes, offsets = gu.pack_targets(fx["targets"])
    idx = list(range(0, len(fx["requests"]), 3))
    reqs = tc.make_reqs([tuple(fx["requests"][i]) for i in idx])
    cases = []
    so = gu.load("g3_structure.json.gz")["oligos"]
    for name, opt in (("dinkelbach", abi.options_default(use_dinkelbach=1, min_primer_tm=40.0, min_probe_tm=40.0)),
                      ("dinkelbach_salt", abi.options_default(use_dinkelbach=1, salt=0.2, target_t=320.0, min_primer_tm=30.0, min_probe_tm=30.0,
                                                              primer_clamp=2, max_mismatch=4))):
        out = ref.bind_batch(opt, fx["oligos"], codes, offsets, reqs)
        recs = [[b.status, b.target_start, b.target_stop, tc.f32_bits(b.tm), tc.f32_bits(b.dH), tc.f32_bits(b.dS), tc.f32_bits(b.dg),
                 b.anchor5, b.anchor3, b.num_mm, b.num_gap, b.poly_degen, b.loc5, b.loc3, b.q_first, b.q_last, b.t_first, b.t_last,
                 b.valid, b.dp_max, b.n_max, b.align.decode()] for b in out]
        st = [list(ref.structure(opt, o, so[(i * 7 + 3) % len(so)], 9e-7, 4e-7)) for i, o in enumerate(so)]
        cases.append({"name": name, "options": opt_dict(opt), "expected": recs, "structure": st})
        print("G9", name, len(recs), "binds,", sum(1 for r in recs if r[0] == 0), "bound")
    dump("g9_dinkelbach.json.gz", {"g2_request_indices": idx, "cases": cases, "structure_partner": "oligos[(i * 7 + 3) % n]",
                                   "ct_a": 9e-7, "ct_b": 4e-7})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        return gen_g9()
    if len(sys.argv) > 1 and sys.argv[1] == "g5-inputs":
        return add_g5_inputs()
    if len(sys.argv) > 2 and sys.argv[1] == "g7":
        return add_g7_cases(sys.argv[2:])
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        return gen_g8()
    ref = tc.RefImpl()
    rng = np.random.default_rng(20260227)

    # ---- G3
    so = structure_oligos(np.random.default_rng(3))
    cases = []
    for name, opt in (("default", abi.options_default()), ("dangle", abi.options_default(allow_dangle_5=1, allow_dangle_3=1)),
                      ("salt", abi.options_default(salt=0.2, target_t=320.0))):
        exp = [list(ref.structure(opt, o, so[(i * 7 + 3) % len(so)], 9e-7, 4e-7)) for i, o in enumerate(so)]
        cases.append({"name": name, "options": opt_dict(opt), "expected": exp})
    dump("g3_structure.json.gz", {"oligos": so, "partner": "oligos[(i * 7 + 3) % n]", "ct_a": 9e-7, "ct_b": 4e-7,
                                  "fields": ["hairpin_tm", "homodimer_tm", "heterodimer_tm"], "cases": cases})
    print("G3: %d oligos, %d non-zero Tm at defaults" % (len(so), sum(1 for r in cases[0]["expected"] for v in r if v)))

    # ---- G2
    oligos, targets, rows = bind_cases(rng)
    # plus ~9k random seeds from a synthetic workload (mostly spurious windows)
    wl = synth.make_workload(3, n_targets=420, n_assays=10)
    wo = sorted({o for e in wl.expanded for o in (e.forward, e.reverse, e.probe) if o})
    seeds = ref.seeds(wl.options, wo, wl.codes, wl.offsets)
    base_t = len(targets)
    for t in range(wl.n_targets):
        targets.append(synth.decode(wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])]))
    base_o = len(oligos)
    oligos += wo
    for s in seeds[::2][:9500]:
        rows.append((base_t + s[0], base_o + s[1], s[2], s[3], s[4], 0, 9e-7))
    codes = np.concatenate([synth.encode(t) for t in targets])
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    fixtures = []
    for name, opt in (("default", abi.options_default(min_primer_tm=40.0, min_probe_tm=40.0)),
                      ("clamps", abi.options_default(min_primer_tm=30.0, min_probe_tm=35.0, primer_clamp=3, probe_clamp_5=2,
                                                     probe_clamp_3=2, max_mismatch=3, max_gap=1, max_poly_degen=1,
                                                     max_primer_dg=-5.0, max_probe_dg=-6.0))):
        reqs = tc.make_reqs(rows)
        out = ref.bind_batch(opt, oligos, codes, offsets, reqs)
        recs = []
        for b in out:
            recs.append([b.status, b.target_start, b.target_stop, tc.f32_bits(b.tm), tc.f32_bits(b.dH), tc.f32_bits(b.dS),
                         tc.f32_bits(b.dg), b.anchor5, b.anchor3, b.num_mm, b.num_gap, b.poly_degen, b.loc5, b.loc3,
                         b.q_first, b.q_last, b.t_first, b.t_last, b.valid, b.dp_max, b.n_max, b.align.decode()])
        fixtures.append({"name": name, "options": opt_dict(opt), "expected": recs})
    dump("g2_binds.json.gz", {"oligos": oligos, "targets": targets, "requests": [list(r) for r in rows],
                              "fields": ["status", "target_start", "target_stop", "tm", "dH", "dS", "dg", "anchor5", "anchor3",
                                         "num_mm", "num_gap", "poly_degen", "loc5", "loc3", "q_first", "q_last", "t_first",
                                         "t_last", "valid", "dp_max", "n_max", "align"],
                              "cases": fixtures})
    st = [r[0] for r in fixtures[0]["expected"]]
    print("G2: %d cases, %d bound at defaults, %d with >1 maximal cell" % (
        len(st), st.count(0), sum(1 for r in fixtures[0]["expected"] if r[20] > 1)))

    # ---- G6
    text = synth.quirky_fasta(np.random.default_rng(6), n_records=60)
    codes, offsets, names = ref.parse_fasta(text)
    dump("g6_fasta.json.gz", {"text_latin1": text.decode("latin-1"), "deflines": names,
                              "targets": [synth.decode(codes[int(offsets[t]):int(offsets[t + 1])]) for t in range(len(names))]})
    print("G6: %d records, %d bases" % (len(names), int(offsets[-1])))

    # ---- G7
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import cli_compare
    cases = []
    for name, cfg, nt, na, seed, flags in cli_compare.CASES:
        if name not in cli_compare.LONG_RECORDS:
            nt, na = min(nt, 120), min(na, 12)
        with tempfile.TemporaryDirectory() as d:
            if name == "long_target":
                continue  # covered by scripts/cli_compare.py on the GPU box (the fragment cases hold long records)
            cli_compare.write_case(d, cfg, nt, na, seed, fastq=(name == "fastq"), name=name)
            out = os.path.join(d, "ref.out")
            r = subprocess.run([cli_compare.REF, "-i", os.path.join(d, "assays.txt"), "-d", os.path.join(d, "db.fna"), "-o", out] + flags,
                               env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            files = {f[len("ref.out"):]: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f.startswith("ref.out")}
            cases.append({"name": name, "flags": flags, "assays": open(os.path.join(d, "assays.txt")).read(),
                          "fasta": open(os.path.join(d, "db.fna")).read(), "files": files,
                          "closing": cli_compare.closing_lines(r.stdout)})
            print("G7", name, {k: len(v) for k, v in files.items()}, "bytes,", sum(v.count("name = ") for v in files.values()), "records,",
                  len(cases[-1]["closing"]), "closing stdout lines")
    dump("g7_cli.json.gz", {"cases": cases})

    # ---- G4
    wl = synth.make_workload(5, n_targets=100, n_assays=12)
    codes = wl.codes.copy()
    codes[int(wl.offsets[3]) + 100:int(wl.offsets[3]) + 140] = 15   # N-run
    wo = sorted({o for e in wl.expanded for o in (e.forward, e.reverse, e.probe) if o}) + ["ACGTIACGTTGCAIGGATCCA", "AAAAAAAAAAAAAAAA"]
    out = {}
    for k in (7, 5, 8):
        opt = abi.options_default(hash_word_size=k)
        n_t = 100 if k == 7 else 12
        out[str(k)] = ref.seeds(opt, wo, codes[:int(wl.offsets[n_t])], wl.offsets[:n_t + 1])
    dump("g4_seeds.json.gz", {"oligos": wo, "targets": [synth.decode(codes[int(wl.offsets[t]):int(wl.offsets[t + 1])]) for t in range(100)],
                              "seeds": out})

    # ---- G5
    cases = {1: (1000, None), 2: (1000, 96), 3: (1000, 96), 4: (600, 80), 5: (800, 80)}
    for cfg, (nt, na) in cases.items():
        wl = synth.make_workload(cfg, n_targets=nt, n_assays=na)
        hits = ref.search(wl.options, wl.expanded, wl.codes, wl.offsets)
        dump("g5_search_cfg%d.json.gz" % cfg, {"config": cfg, "n_targets": nt, "n_assays": na, "options": opt_dict(wl.options),
                                               "assays": [[a.name, a.forward, a.reverse, a.probe] for a in wl.assays],
                                               "targets": [synth.decode(wl.codes[int(wl.offsets[t]):int(wl.offsets[t + 1])]) for t in range(wl.n_targets)],
                                               "n_hits": len(hits), "hits": [list(k) for k in tc.sorted_hit_keys(hits)]})
        print("G5 cfg", cfg, "hits", len(hits))
    # README example (README.md:130-203): the printed 147-nt amplicon as target
    amp = ("TTCCCCTTTGGAGGCATCCAAGCGATGGGCTTTCAGGACAGGTGTACCTCCCAAGAATGTTGAGTATACAGAAGGGGAGGAAGCCAAAACATGCTACAATATAAGTG"
           "TAACGGATCCCTCTGGAAAATCCTTGCTGTTGGATCCTCC")
    t = "ACGTACGT" + amp + "ACGTACGT"
    assays = [synth.Assay("gibb-marburg", "TTCCCCTTTGGAGGCATC", "GGAGGATCCAACAGCAAGG", "CGATGGGCTTTCAGGACAGGTGT")]
    ex = synth.expand_assays(assays)
    opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=45.0)
    hits = ref.search(opt, ex, synth.encode(t), np.array([0, len(t)], dtype=np.uint64))
    dump("g5_readme.json.gz", {"target": t, "assay": [assays[0].forward, assays[0].reverse, assays[0].probe],
                               "options": opt_dict(opt), "hits": [list(k) for k in tc.sorted_hit_keys(hits)]})
    print("README hits", len(hits))


if __name__ == "__main__":
    main()
