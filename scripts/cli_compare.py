#!/usr/bin/env python3
"""End-to-end check of the command-line front end on the GPU box: the same assay file + FASTA database through
the reference binary (oracle/_ref/tntblast, OMP_NUM_THREADS=1) and through tntblast-mi355x; the two output
files must be byte-identical.   python scripts/cli_compare.py [workdir]
With a second argument `stub`: oracle/_ref/tntblast_stub -- the reference's own program with the work loop of its OpenMP
region replaced by the binding of INTEGRATION.md (oracle/build_stub.py) -- takes the front end's place, on the cases that
need neither masking nor fragmentation (both are the caller's business at the C ABI)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import longwl  # noqa: E402
from thermonucleotideblast_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "tntblast")
CLI = os.path.join(ROOT, "thermonucleotideblast_amd", "tntblast-mi355x")
STUB = os.path.join(ROOT, "oracle", "_ref", "tntblast_stub")
STUB_SKIPS = ("mask_primers_k", "mask_probe_K", "replace_primers_r", "mask_all_krK_m1", "fragment_default", "fragment_small")

CASES = [  # (name, config, targets, assays, seed, extra flags)
    ("pcr", 2, 300, 24, 3, ["-e", "40"]),
    ("taqman", 3, 300, 24, 4, ["-e", "40", "-E", "40"]),
    ("degenerate", 4, 120, 20, 5, ["-e", "40"]),
    ("padlock_probe", 5, 200, 24, 6, ["-A", "PADLOCK", "-E", "40"]),
    ("pcr_options", 3, 250, 20, 7, ["-e", "35", "-E", "38", "--primer-clamp", "2", "--probe-clamp5", "1", "-s", "0.1", "-t", "5e-7",
                                    "--single-primer-pcr", "F", "-l", "600", "-L", "T"]),
    ("no_align", 2, 200, 16, 8, ["-e", "45", "-a", "F", "-M", "F", "--max-mismatch", "3"]),
    # one 150-kb record next to short ones: searched in overlapping windows here, unfragmented by the reference
    ("long_target", 3, 200, 16, 9, ["-e", "40", "-E", "40"]),
    # the other output formats and post-processing options (every file the run writes is compared, and the closing
    # statistics / assay summary on stdout)
    ("fasta_m1", 3, 300, 24, 10, ["-e", "40", "-E", "40", "-m", "1", "-L", "T"]),
    ("network_m2", 2, 300, 24, 11, ["-e", "40", "-m", "2"]),
    ("inverse_target_m3", 2, 300, 24, 12, ["-e", "40", "-m", "3"]),
    ("inverse_query_m4", 2, 300, 24, 13, ["-e", "58", "-m", "4"]),
    ("per_query_n", 3, 300, 12, 14, ["-e", "40", "-E", "40", "-n", "T"]),
    ("best_match", 2, 300, 24, 15, ["-e", "30", "--best-match", "-S", "T"]),
    ("summary_degenerate", 4, 120, 20, 16, ["-e", "40", "-S", "T"]),
    # assay front end: every primer / probe combination of the input assays; primer pairs turned into single probes
    ("plex", 3, 150, 5, 17, ["-e", "40", "-E", "40", "--plex", "T", "-S", "T"]),
    ("plex_padlock", 5, 150, 6, 18, ["-A", "PADLOCK", "-E", "40", "--plex", "T"]),
    ("force_probe", 3, 150, 8, 19, ["-A", "PROBE", "-E", "45"]),
    # the same search on a FASTQ database
    ("fastq", 3, 200, 16, 20, ["-e", "40", "-E", "40"]),
    # primer / probe masking of the amplicon text (mask_binding_sites, tntblast_util.cpp:97-576)
    ("mask_primers_k", 3, 200, 16, 23, ["-e", "40", "-E", "40", "-k", "T"]),
    ("mask_probe_K", 3, 200, 16, 24, ["-e", "40", "-E", "40", "-K", "T"]),
    ("replace_primers_r", 2, 200, 16, 25, ["-e", "40", "-r", "T"]),
    ("mask_all_krK_m1", 3, 200, 16, 26, ["-e", "38", "-E", "38", "-k", "T", "-r", "T", "-K", "T", "-m", "1"]),
    # Tm by Dinkelbach iteration (binds and the hairpin / dimer values of every record)
    ("dinkelbach", 3, 200, 16, 27, ["-e", "40", "-E", "40", "--dinkelbach", "T"]),
    # targets longer than --max-target-len are searched in overlapping fragments (tntblast_local.cpp:282-289,438-468):
    # matches cut off by a fragment's edge are dropped, matches found by two fragments are made unique
    # (uniquify_results).  One 620-kb record at the default threshold (500 000); one 64-kb record cut into six.
    ("fragment_default", 2, 660, 12, 21, ["-e", "40"]),
    ("fragment_small", 3, 104, 12, 22, ["-e", "40", "-E", "40", "--max-target-len", "12000", "-l", "500"]),
    # oligos of 65..128 bases (tests/longwl.py; the "config" names its kind): hybridisation probes, padlock arms, a long primer
    ("long_probes", "probe", 60, 6, 31, ["-E", "45"]),
    ("long_padlock_arms", "padlock", 60, 6, 33, ["-A", "PADLOCK", "-E", "45"]),
    ("long_primer", "pcr", 60, 6, 34, ["-e", "45"]),
    # hybridisation probes of more than 128 bases (tests/longwl.py make_beyond_128): the wavefront-per-window kernel, seeds in
    # segments, alignment texts far beyond the 448-byte fields of a hit
    ("probes_beyond_128", "beyond128", 10, 3, 36, ["-A", "PROBE", "-E", "50"]),
    # PCR primers, TaqMan probes and padlock arms of more than 128 bases (tests/longwl.py make_primers_beyond_128): segments of one
    # oligo's word list feed one list per (target, assay)
    ("primers_beyond_128", "pcr>128", 10, 3, 1, ["-e", "45"]),
    ("taqman_beyond_128", "taqman>128", 10, 3, 2, ["-e", "45", "-E", "45"]),
    ("padlock_arms_beyond_128", "padlock>128", 10, 3, 4, ["-A", "PADLOCK", "-E", "45"]),
    # row a6: the probe of every TaqMan assay is also the forward primer of a PCR assay listed next to it -- one oligo string under
    # two strand concentrations / threshold sets: the reference's per-target BindCache decides (the library replays it)
    ("shared_oligos", 3, 250, 10, 35, ["-e", "45", "-E", "30", "--primer-clamp", "2", "--probe-clamp5", "1"]),
]

LONG_RECORDS = {"long_target": 150, "fragment_default": 620, "fragment_small": 64}
MAX_TARGET_LEN = {"fragment_default": 500000, "fragment_small": 12000}


def closing_lines(stdout):
    """stdout from the match count to the end, without the timing line (the progress display before it differs)"""
    lines = stdout.split("\n")
    for i, l in enumerate(lines):
        if "target sequence matches" in l or "inverse target matches" in l or l.startswith("*** Assay Summary"):
            return [x for x in lines[i:] if x and not x.startswith("Search completed")]
    return []


def write_case(d, cfg, nt, na, seed, long_target=False, fastq=False, name=None):
    if cfg == "beyond128":
        wl = longwl.make_beyond_128(seed, lengths=(150, 320, 700), n_targets=nt, tl=2200)
    elif isinstance(cfg, str) and cfg.endswith(">128"):
        wl = longwl.make_primers_beyond_128(cfg[:-4], seed, n_targets=nt)
    else:
        wl = longwl.make(cfg, seed, n_targets=nt) if isinstance(cfg, str) else synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
    offsets = wl.offsets
    if long_target or name in LONG_RECORDS:  # the first records become one
        k = LONG_RECORDS.get(name, 150)
        offsets = np.concatenate([[0], wl.offsets[k:]]).astype(np.uint64)
    if name in MAX_TARGET_LEN:
        # copies of planted sites around the first cut of the long record: one amplicon across the cut, one inside the
        # stretch both fragments read, one at the very start of the second fragment
        nb = int(offsets[1])
        approx = len(b">seq0 synthetic target\n") + nb + (nb + 69) // 70  # bytes of the record, as the reference counts
        n = 2
        while approx > n * MAX_TARGET_LEN[name]:
            n += 1
        cut = approx // n + (1 if approx % n else 0)  # last base of the first fragment's own part
        rng = np.random.default_rng(seed)
        codes = wl.codes.copy()
        for a, at in zip(wl.assays[:3], (cut - 120, cut + 300, cut + 1)):
            amp = 260
            seg = list("ACGT"[i] for i in rng.integers(0, 4, amp))
            seg[:len(a.forward)] = a.forward
            seg[amp - len(a.reverse):] = synth.revcomp(a.reverse)
            if a.probe:
                seg[100:100 + len(a.probe)] = a.probe
            codes[at:at + amp] = synth.encode("".join(seg))
        wl.codes = codes
    with open(os.path.join(d, "db.fna"), "wb") as f:
        if fastq:  # one sequence line per record; the quality line starts with '@' or '+' now and then
            for t in range(len(offsets) - 1):
                seq = synth.decode(wl.codes[int(offsets[t]):int(offsets[t + 1])]).encode()
                q = bytearray(b"I" * len(seq))
                if q and t % 3:
                    q[0] = ord("@") if t % 3 == 1 else ord("+")
                f.write(b"@seq%d synthetic read\n" % t + seq + b"\n+\n" + bytes(q) + b"\n")
        else:
            f.write(synth.to_fasta(wl.codes, offsets))
    assays = list(wl.assays)
    if name == "shared_oligos":
        extra = [synth.Assay("swap%d" % i, a.probe, a.reverse, None) for i, a in enumerate(wl.assays)]
        assays = [x for i, pair in enumerate(zip(wl.assays, extra)) for x in (pair if i % 2 == 0 else pair[::-1])]
    with open(os.path.join(d, "assays.txt"), "w") as f:
        f.write("# name forward reverse [probe]   (tntblast assay file)\n")
        for a in assays:
            cols = [a.name] + [x for x in (a.forward, a.reverse, a.probe) if x]
            f.write("\t".join(cols) + "\n")


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "cli")
    stub = len(sys.argv) > 2 and sys.argv[2] == "stub"
    program = STUB if stub else CLI
    bad = 0
    for name, cfg, nt, na, seed, flags in CASES:
        if stub and name in STUB_SKIPS:
            continue
        d = os.path.join(work, name)
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(d):
            if f.startswith("ref.out") or f.startswith("hip.out"):
                os.remove(os.path.join(d, f))
        write_case(d, cfg, nt, na, seed, fastq=(name == "fastq"), name=name)
        common = ["-i", os.path.join(d, "assays.txt"), "-d", os.path.join(d, "db.fna")] + flags
        env = dict(os.environ, OMP_NUM_THREADS="1")
        r1 = subprocess.run([REF] + common + ["-o", os.path.join(d, "ref.out")], env=env, capture_output=True, text=True)
        r2 = subprocess.run([program] + common + ["-o", os.path.join(d, "hip.out")], env=env if stub else None, capture_output=True, text=True)
        if r1.returncode or r2.returncode:
            print(name, "FAILED to run: ref rc", r1.returncode, "hip rc", r2.returncode, r1.stderr[-300:], r2.stderr[-300:])
            bad += 1
            continue
        files_ref = sorted(f[len("ref.out"):] for f in os.listdir(d) if f.startswith("ref.out"))
        files_hip = sorted(f[len("hip.out"):] for f in os.listdir(d) if f.startswith("hip.out"))
        same = files_ref == files_hip and bool(files_ref)
        n_bytes = n_rec = 0
        first = None
        for suffix in files_ref if same else []:
            a = open(os.path.join(d, "ref.out" + suffix), "rb").read()
            b = open(os.path.join(d, "hip.out" + suffix), "rb").read()
            n_bytes += len(a)
            n_rec += a.count(b"name = ")
            if a != b and first is None:
                same = False
                la, lb = a.decode("latin-1").split("\n"), b.decode("latin-1").split("\n")
                for i, (x, y) in enumerate(zip(la, lb)):
                    if x != y:
                        first = "  first difference in out%s at line %d:\n    ref: %s\n    hip: %s" % (suffix, i + 1, x, y)
                        break
                else:
                    first = "  out%s: %d vs %d lines" % (suffix, len(la), len(lb))
        c1, c2 = closing_lines(r1.stdout), closing_lines(r2.stdout)
        if c1 != c2:
            same = False
            first = first or "  closing stdout differs:\n    ref: %s\n    hip: %s" % (c1[:6], c2[:6])
        print("%-18s %s  %d file(s) %s, %d bytes, %d records, %d closing stdout lines" % (
            name, "identical" if same else "DIFFERENT", len(files_ref), files_ref, n_bytes, n_rec, len(c1)))
        if not same:
            bad += 1
            print(first or "  file sets differ: %s vs %s" % (files_ref, files_hip))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
