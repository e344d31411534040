"""Synthetic workloads for BASELINE.json's five configurations.

RNG = numpy PCG64 (`default_rng(seed)`); targets are i.i.d. uniform ACGT, oligos uniform ACGT
with 40-60 % GC, amplicons / ligation sites / probe sites are planted in a fraction of the
targets with 0-3 substitutions per site.  `scale` shrinks the number of targets (the 1/1000
copies are the parity fixtures).  Pure numpy; no native code.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "I": "I", "M": "K", "R": "Y", "S": "S", "V": "B",
         "W": "W", "Y": "R", "H": "D", "K": "M", "D": "H", "B": "V", "N": "N"}
_IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "I": "I", "M": "AC", "R": "GA", "S": "GC", "V": "GCA",
          "W": "AT", "Y": "TC", "H": "ACT", "K": "GT", "D": "GAT", "B": "GTC", "N": "ATGC"}
_CODE = {c: i for i, c in enumerate(abi.DB_CODES)}


def revcomp(s: str) -> str:
    return "".join(_COMP[c] for c in reversed(s))


def encode(s: str) -> np.ndarray:
    """ASCII -> 1-byte codes of seq.h:148-189 (U -> T, unknown letters -> 17)."""
    out = np.full(len(s), abi.DB_UNKNOWN, dtype=np.uint8)
    for i, c in enumerate(s.upper()):
        if c == "U":
            c = "T"
        if c == "-":
            out[i] = abi.DB_GAP
        elif c in _CODE:
            out[i] = _CODE[c]
    return out


def decode(codes: np.ndarray) -> str:
    lut = abi.DB_CODES + "-?"
    return "".join(lut[int(c)] for c in codes)


def expand_oligo(s: str) -> List[str]:
    """expand_nucleic_acid, degenerate_na.cpp:69-173: odometer, FIRST base varies fastest."""
    alts = [_IUPAC[c] for c in s.upper()]
    n = 1
    for a in alts:
        n *= len(a)
    out = []
    idx = [0] * len(alts)
    for _ in range(n):
        out.append("".join(a[i] for a, i in zip(alts, idx)))
        for j in range(len(alts)):
            idx[j] += 1
            if idx[j] == len(alts[j]):
                idx[j] = 0
            else:
                break
    return out


def degeneracy(s: str) -> int:
    d = 1
    for c in s.upper():
        d *= len(_IUPAC[c])
    return d


@dataclass
class Assay:
    """One row of the assay file (input.cpp:43-168): name + 1..3 oligos, 5'->3'."""
    name: str
    forward: Optional[str] = None
    reverse: Optional[str] = None
    probe: Optional[str] = None


@dataclass
class ExpandedAssay:
    id: int
    degen_id: int
    forward: Optional[str]
    reverse: Optional[str]
    probe: Optional[str]
    forward_degen: int = 1
    reverse_degen: int = 1
    probe_degen: int = 1


def expand_assays(assays: List[Assay], rescale_ct: bool = True) -> List[ExpandedAssay]:
    """expand_degenerate_signatures, tntblast_util.cpp:579-713."""
    out = []
    did = 0
    for aid, a in enumerate(assays):
        primers = []
        if a.forward and a.reverse:
            primers = [(f, r) for f in expand_oligo(a.forward) for r in expand_oligo(a.reverse)]
        probes = expand_oligo(a.probe) if a.probe else []
        df = degeneracy(a.forward) if (rescale_ct and a.forward) else 1
        dr = degeneracy(a.reverse) if (rescale_ct and a.reverse) else 1
        dp = degeneracy(a.probe) if (rescale_ct and a.probe) else 1
        if not primers:
            for p in probes:
                out.append(ExpandedAssay(aid, did, None, None, p, 1, 1, dp))
                did += 1
        else:
            for f, r in primers:
                if not probes:
                    out.append(ExpandedAssay(aid, did, f, r, None, df, dr, 1))
                    did += 1
                else:
                    for p in probes:
                        out.append(ExpandedAssay(aid, did, f, r, p, df, dr, dp))
                        did += 1
    return out


@dataclass
class Workload:
    name: str
    options: abi.tnt_options
    assays: List[Assay]
    codes: np.ndarray    # uint8, concatenated 1-byte codes
    offsets: np.ndarray  # uint64, n_targets + 1
    expanded: List[ExpandedAssay] = field(default_factory=list)

    @property
    def n_targets(self):
        return len(self.offsets) - 1

    @property
    def n_bases(self):
        return int(self.offsets[-1])

    @property
    def n_oligos(self):
        return sum((1 if e.forward else 0) + (1 if e.reverse else 0) + (1 if e.probe else 0) for e in self.expanded)

    @property
    def evals(self):
        """metric numerator: expanded oligos x target bases."""
        return self.n_oligos * self.n_bases


def _rand_oligo(rng, lo, hi):
    while True:
        n = int(rng.integers(lo, hi + 1))
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, n))
        gc = (s.count("G") + s.count("C")) / n
        if 0.4 <= gc <= 0.6:
            return s


def _mutate(rng, s, nmax):
    s = list(s)
    for _ in range(int(rng.integers(0, nmax + 1))):
        i = int(rng.integers(0, len(s)))
        s[i] = "ACGT"[(("ACGT".index(s[i]) if s[i] in "ACGT" else 0) + int(rng.integers(1, 4))) % 4]
    return "".join(s)


def _first_expansion(s):
    return "".join(_IUPAC[c][0] for c in s)


def _degenerate(rng, s):
    s = list(s)
    for _ in range(int(rng.integers(1, 4))):
        i = int(rng.integers(1, len(s) - 1))
        s[i] = "RYKMSWN"[int(rng.integers(0, 7))]
    return "".join(s)


def _targets(rng, n, length):
    codes = rng.integers(0, 4, size=n * length, dtype=np.uint8)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(length))
    return codes, offsets


def _put(codes, base, pos, s):
    codes[base + pos: base + pos + len(s)] = encode(s)


def make_workload(config: int, scale: float = 1.0, n_targets: Optional[int] = None,
                  n_assays: Optional[int] = None, seed: Optional[int] = None, shard: int = 0) -> Workload:
    """config 1..5 = BASELINE.json `configs`; scale multiplies the number of targets.

    The assay set depends on (config, seed) only; `shard` selects an independent set of targets for
    the same assays (one shard per GPU in the weak-scaling runs)."""
    base_seed = config if seed is None else seed
    rng = np.random.default_rng([base_seed, 7])            # oligos
    trng = np.random.default_rng([base_seed, 1000 + shard])  # targets and planted sites
    opt = abi.options_default(min_primer_tm=40.0, min_probe_tm=40.0)

    if config == 1:    # 1 primer pair x 1k x 1kb
        nt, tl, na, plant = 1000, 1000, 1, 0.10
        kind = "pcr"
    elif config == 2:  # 96 primer pairs x 1M x 1kb
        nt, tl, na, plant = 1_000_000, 1000, 96, 0.0002
        kind = "pcr"
    elif config == 3:  # 96 TaqMan triplets x 1M x 1kb
        nt, tl, na, plant = 1_000_000, 1000, 96, 0.0002
        kind = "taqman"
    elif config == 4:  # 1k assays incl. degenerate oligos x 10M x 2kb (sharded over 8 GPUs)
        nt, tl, na, plant = 10_000_000, 2000, 1000, 0.00002
        kind = "pcr_degen"
    elif config == 5:  # padlock + probe mix x 5M x 1kb with degenerate target bases
        nt, tl, na, plant = 5_000_000, 1000, 1000, 0.00004
        kind = "padlock_probe"
        opt.assay_format = abi.ASSAY_PADLOCK
    else:
        raise ValueError("config must be 1..5")

    nt = int(n_targets) if n_targets is not None else max(8, int(round(nt * scale)))
    na = int(n_assays) if n_assays is not None else na
    # keep the expected number of planted sites per assay useful on shrunken copies
    plant = max(plant, min(0.5, 4.0 / nt))

    codes, offsets = _targets(trng, nt, tl)
    assays = []

    for a in range(na):
        name = "assay%d" % a
        if kind in ("pcr", "taqman", "pcr_degen"):
            F, R = _rand_oligo(rng, 18 if config != 1 else 20, 25 if config != 1 else 20), \
                _rand_oligo(rng, 18 if config != 1 else 20, 25 if config != 1 else 20)
            P = _rand_oligo(rng, 22, 28) if kind == "taqman" else None
            Fw, Rw = F, R
            if kind == "pcr_degen" and rng.random() < 0.2:
                if rng.random() < 0.5:
                    Fw = _degenerate(rng, F)
                    F = _first_expansion(Fw)
                else:
                    Rw = _degenerate(rng, R)
                    R = _first_expansion(Rw)
            assays.append(Assay(name, Fw, Rw, P))
            rng_save, rng = rng, trng
            n_plant = rng.binomial(nt, plant)
            for t in rng.integers(0, nt, n_plant):
                amp = int(rng.integers(80, 401))
                amp = max(amp, len(F) + len(R) + (len(P) + 4 if P else 0) + 2)
                if amp >= tl - 2:
                    continue
                pos = int(rng.integers(1, tl - amp - 1))
                base = int(offsets[t])
                flip = rng.random() < 0.5  # amplicon in either orientation
                f_site, r_site = _mutate(rng, F, 3), revcomp(_mutate(rng, R, 3))
                seg = list("ACGT"[i] for i in rng.integers(0, 4, amp))
                seg[:len(F)] = f_site
                seg[amp - len(R):] = r_site
                if P:
                    pp = int(rng.integers(len(F) + 1, amp - len(R) - len(P)))
                    ps = _mutate(rng, P, 2)
                    seg[pp:pp + len(P)] = ps if rng.random() < 0.5 else revcomp(ps)
                seg = "".join(seg)
                _put(codes, base, pos, revcomp(seg) if flip else seg)
            rng = rng_save
        else:  # padlock_probe
            if a % 2 == 0:
                # padlock: 5' probe (forward oligo, downstream) and 3' probe (reverse oligo, upstream)
                # hybridise adjacently: target carries revcomp(reverse + forward)... see padlock_search.cpp:133-137
                F, R = _rand_oligo(rng, 20, 25), _rand_oligo(rng, 20, 25)
                assays.append(Assay(name, F, R, None))
                n_plant = trng.binomial(nt, plant)
                for t in trng.integers(0, nt, n_plant):
                    site = _mutate(trng, F, 2) + _mutate(trng, R, 2)
                    pos = int(trng.integers(1, tl - len(site) - 1))
                    _put(codes, int(offsets[t]), pos, site if trng.random() < 0.5 else revcomp(site))
            else:
                P = _rand_oligo(rng, 25, 40)
                assays.append(Assay(name, None, None, P))
                n_plant = trng.binomial(nt, plant)
                for t in trng.integers(0, nt, n_plant):
                    site = _mutate(trng, P, 3)
                    pos = int(trng.integers(1, tl - len(site) - 1))
                    _put(codes, int(offsets[t]), pos, site if trng.random() < 0.5 else revcomp(site))

    if config == 5:
        # on average 5 IUPAC bases per target and an N-run of length 10 in 1 % of the targets
        n_deg = trng.poisson(5.0 * nt)
        pos = trng.integers(0, nt * tl, n_deg)
        codes[pos] = trng.integers(5, 16, n_deg).astype(np.uint8)
        for t in np.nonzero(trng.random(nt) < 0.01)[0]:
            p = int(trng.integers(0, tl - 10))
            codes[int(offsets[t]) + p: int(offsets[t]) + p + 10] = abi.DB_CODES.index("N")

    wl = Workload("config%d" % config, opt, assays, codes, offsets)
    wl.expanded = expand_assays(assays)
    return wl


def to_fasta(codes, offsets, width=70, first=0, name="seq"):
    """FASTA text (bytes) of a workload's targets: `>name<i>` deflines, `width` bases per line."""
    alphabet = np.frombuffer(b"ACGTIMRSVWYHKDBN-X", dtype=np.uint8)
    out = []
    for t in range(len(offsets) - 1):
        seq = alphabet[np.asarray(codes[int(offsets[t]):int(offsets[t + 1])], dtype=np.uint8)].tobytes()
        out.append(b">%s%d synthetic target\n" % (name.encode(), first + t))
        out.extend(seq[i:i + width] + b"\n" for i in range(0, len(seq), width))
        if not seq:
            out.append(b"\n")
    return b"".join(out)


def quirky_fasta(rng, n_records=40):
    """FASTA text exercising every rule of the reference's reader (sequence_data_fastx.cpp:13-79,187-382):
    CRLF line ends, blank lines, lower case, U, every IUPAC code, unknown characters, '*' and '-' inside the
    sequence, white space inside lines, '>' inside a defline, a '>' in the middle of a sequence line (opens a
    record), leading blanks in the defline, an empty sequence, text before the first record, no final newline."""
    letters = b"ACGTacgtUuIiMRSVWYHKDBNmrsvwyhkdbn"
    recs = [b"this line precedes the first record and is ignored\nACGT\n"]
    for r in range(n_records):
        eol = b"\r\n" if r % 3 == 1 else b"\n"
        d = b">" + (b"  \t" if r % 5 == 2 else b"") + b"rec%d some description > with a second bracket" % r + eol
        n = int(rng.integers(0, 400)) if r % 7 else 0
        seq = bytes(rng.choice(np.frombuffer(letters, dtype=np.uint8), size=n))
        body = bytearray()
        i = 0
        while i < len(seq):
            w = int(rng.integers(1, 80))
            line = bytearray(seq[i:i + w])
            if r % 4 == 0 and len(line) > 3:
                line.insert(2, ord("*"))
                line.insert(1, ord("-"))
                line.insert(3, ord(" "))
            if r % 6 == 3 and len(line) > 2:
                line[1] = ord("X")
                line.append(ord("?"))
            body += bytes(line) + eol
            if r % 9 == 4:
                body += eol
            i += w
        if r % 11 == 5:
            body += b"ACGT>midline record" + eol + b"GGCC" + eol
        recs.append(d + bytes(body))
    recs.append(b">last record without a final newline\nACGTNNNNACGT")
    return b"".join(recs)


def quirky_fastq(rng, n_records=40):
    """FASTQ text exercising the rules of the reference's reader (sequence_data_fastx.cpp:81-186,384-583): quality
    lines that start with '@' or '+', '+' lines repeating the name, blanks before the name, lower case / IUPAC /
    unknown characters, '*', '-' and blanks inside the sequence, an empty sequence, a CRLF record (read as an empty
    sequence: only one end-of-line character is skipped), blank lines between records, no final newline."""
    letters = b"ACGTacgtUuNnRYKMSWBDHVIi"
    out = bytearray()
    for r in range(n_records):
        n = int(rng.integers(1, 200)) if r % 7 else 0
        seq = bytearray(rng.choice(np.frombuffer(letters, dtype=np.uint8), size=n))
        if r % 4 == 0 and len(seq) > 4:
            seq.insert(2, ord("*"))
            seq.insert(1, ord("-"))
            seq.insert(3, ord(" "))
        if r % 6 == 3 and len(seq) > 2:
            seq[1] = ord("X")
            seq.append(ord("?"))
        qual = bytearray(rng.integers(35, 74, size=len(seq)).astype(np.uint8))
        if len(qual):
            if r % 3 == 1:
                qual[0] = ord("@")
            elif r % 3 == 2:
                qual[0] = ord("+")
        eol = b"\r\n" if r % 9 == 8 else b"\n"
        name = b"read%d length=%d" % (r, len(seq))
        out += b"@" + (b" \t" if r % 5 == 2 else b"") + name + eol + bytes(seq) + eol
        out += b"+" + (name if r % 2 else b"") + eol + bytes(qual) + eol
        if r % 10 == 9:
            out += b"\n"
    out += b"@last read without a final newline\nACGTNNNNACGT\n+\nIIIIIIIIIIII"
    return bytes(out)
