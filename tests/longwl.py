"""Workloads with oligos of 65..128 bases (long hybridisation probes, padlock arms, one long primer): planted sites with
substitutions and small indels on random targets.  Used by the CPU test that pins the oracle against the reference for
such oligos and by the GPU test that holds the HIP path against the oracle."""
import numpy as np

from thermonucleotideblast_amd import abi, synth


def _edit(rng, s, n_sub, n_indel):
    s = list(synth._mutate(rng, s, n_sub))
    for _ in range(int(rng.integers(0, n_indel + 1))):
        k = int(rng.integers(5, len(s) - 5))
        if rng.random() < 0.5:
            del s[k]
        else:
            s.insert(k, "ACGT"[int(rng.integers(0, 4))])
    return "".join(s)


def make(kind, seed, n_targets=60, tl=700):
    """kind: 'probe' (hybridisation probes 65..128), 'padlock' (arms 65..100), 'pcr' (primer pairs, one primer 65..90)."""
    rng = np.random.default_rng([4242, seed])
    codes, offsets = synth._targets(rng, n_targets, tl)
    opt = abi.options_default(min_primer_tm=45.0, min_probe_tm=45.0)
    assays = []
    for a in range(6):
        if kind == "probe":
            P = synth._rand_oligo(rng, 65, 128)
            if a == 0:
                P = synth._rand_oligo(rng, 128, 128)
            assays.append(synth.Assay("long%d" % a, None, None, P))
            sites = [P]
        elif kind == "padlock":
            opt.assay_format = abi.ASSAY_PADLOCK
            F, R = synth._rand_oligo(rng, 65, 100), synth._rand_oligo(rng, 20, 100)
            assays.append(synth.Assay("long%d" % a, F, R, None))
            sites = [F + R]  # the two arms side by side (padlock_search.cpp:133-137), either strand below
        else:
            F, R = synth._rand_oligo(rng, 65, 90), synth._rand_oligo(rng, 18, 25)
            if a % 2:
                F, R = R, F
            assays.append(synth.Assay("long%d" % a, F, R, None))
            amp = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(20, 120))))
            sites = [F + amp + synth.revcomp(R)]
        for _ in range(int(rng.integers(6, 12))):
            t = int(rng.integers(0, n_targets))
            s = _edit(rng, sites[0], 5, 2)
            if rng.random() < 0.5:
                s = synth.revcomp(s)
            if len(s) + 2 >= tl:
                continue
            pos = int(rng.integers(1, tl - len(s) - 1))
            synth._put(codes, int(offsets[t]), pos, s)
    # a few IUPAC letters
    pos = rng.integers(0, n_targets * tl, 40)
    codes[pos] = rng.integers(5, 16, 40).astype(np.uint8)
    wl = synth.Workload("long-%s" % kind, opt, assays, codes, offsets)
    wl.expanded = synth.expand_assays(assays)
    return wl


def make_beyond_128(seed, lengths=(150, 220, 400, 1000), n_targets=24, tl=3000):
    """Probe-only assays with hybridisation probes of more than 128 bases (up to the reference's 1 024): planted sites with
    substitutions and small indels on either strand, a few IUPAC letters; `-A PROBE`-style rows (probe column only)."""
    rng = np.random.default_rng([777, seed])
    codes, offsets = synth._targets(rng, n_targets, tl)
    opt = abi.options_default(min_primer_tm=45.0, min_probe_tm=50.0)
    assays = []
    for a, ln in enumerate(lengths):
        P = synth._rand_oligo(rng, ln, ln)
        assays.append(synth.Assay("probe%d" % ln, None, None, P))
        for _ in range(int(rng.integers(3, 7))):
            t = int(rng.integers(0, n_targets))
            s = _edit(rng, P, max(3, ln // 25), 2)
            if rng.random() < 0.5:
                s = synth.revcomp(s)
            pos = int(rng.integers(1, tl - len(s) - 1))
            synth._put(codes, int(offsets[t]), pos, s)
    pos = rng.integers(0, n_targets * tl, 30)
    codes[pos] = rng.integers(5, 16, 30).astype(np.uint8)
    wl = synth.Workload("beyond-128", opt, assays, codes, offsets)
    wl.expanded = synth.expand_assays(assays)
    return wl


def make_primers_beyond_128(kind, seed, n_targets=10, tl=2600, every=2):
    """kind: 'pcr' (primer pairs, one or both primers of 129..520 bases), 'taqman' (a probe of more than 128 bases between
    ordinary primers, or a long primer next to an ordinary probe), 'padlock' (arms of more than 128 bases).  Planted sites with
    substitutions and small indels on either strand, a few IUPAC letters.  The reference takes oligos of up to 1 024 bases in
    every role (nuc_cruc.h:80)."""
    rng = np.random.default_rng([991, seed, {"pcr": 0, "taqman": 1, "padlock": 2}[kind]])
    codes, offsets = synth._targets(rng, n_targets, tl)
    opt = abi.options_default(min_primer_tm=45.0, min_probe_tm=45.0)
    if kind == "padlock":
        opt.assay_format = abi.ASSAY_PADLOCK
    assays = []
    shapes = {"pcr": [(129, 22), (300, 24), (20, 260), (150, 135), (520, 21)],
              "taqman": [(21, 23, 160), (140, 22, 30), (24, 20, 300), (200, 180, 150)],
              "padlock": [(129, 40), (30, 200), (150, 150), (260, 25)]}[kind]
    shapes = shapes[seed % every::every]  # (a long oligo costs the reference a second per seed: a few shapes per case)
    for a, shape in enumerate(shapes):
        if kind == "pcr":
            F, R = synth._rand_oligo(rng, shape[0], shape[0]), synth._rand_oligo(rng, shape[1], shape[1])
            assays.append(synth.Assay("long%d" % a, F, R, None))
            amp = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(20, 150))))
            site = F + amp + synth.revcomp(R)
        elif kind == "taqman":
            F, R, P = (synth._rand_oligo(rng, n, n) for n in shape)
            assays.append(synth.Assay("long%d" % a, F, R, P))
            g1 = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(5, 60))))
            g2 = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(5, 60))))
            site = F + g1 + (P if rng.random() < 0.5 else synth.revcomp(P)) + g2 + synth.revcomp(R)
        else:
            F, R = synth._rand_oligo(rng, shape[0], shape[0]), synth._rand_oligo(rng, shape[1], shape[1])
            assays.append(synth.Assay("long%d" % a, F, R, None))
            site = F + R
        for _ in range(int(rng.integers(3, 6))):
            t = int(rng.integers(0, n_targets))
            s = _edit(rng, site, max(3, len(site) // 40), 2)
            if rng.random() < 0.5:
                s = synth.revcomp(s)
            if len(s) + 2 >= tl:
                continue
            pos = int(rng.integers(1, tl - len(s) - 1))
            synth._put(codes, int(offsets[t]), pos, s)
    pos = rng.integers(0, n_targets * tl, 30)
    codes[pos] = rng.integers(5, 16, 30).astype(np.uint8)
    wl = synth.Workload("beyond-128-%s" % kind, opt, assays, codes, offsets)
    wl.expanded = synth.expand_assays(assays)
    return wl
