#!/usr/bin/env python3
"""End-to-end check of the command-line front end on the GPU box: the same assay file + FASTA database through
the reference binary (oracle/_ref/tntblast, OMP_NUM_THREADS=1) and through tntblast-mi355x; the two output
files must be byte-identical.   python scripts/cli_compare.py [workdir]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermonucleotideblast_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "tntblast")
CLI = os.path.join(ROOT, "thermonucleotideblast_amd", "tntblast-mi355x")

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
]


def closing_lines(stdout):
    """stdout from the match count to the end, without the timing line (the progress display before it differs)"""
    lines = stdout.split("\n")
    for i, l in enumerate(lines):
        if "target sequence matches" in l or "inverse target matches" in l or l.startswith("*** Assay Summary"):
            return [x for x in lines[i:] if x and not x.startswith("Search completed")]
    return []


def write_case(d, cfg, nt, na, seed, long_target=False, fastq=False):
    wl = synth.make_workload(cfg, n_targets=nt, n_assays=na, seed=seed)
    offsets = wl.offsets
    if long_target:  # the first 150 records become one
        offsets = np.concatenate([[0], wl.offsets[150:]]).astype(np.uint64)
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
    with open(os.path.join(d, "assays.txt"), "w") as f:
        f.write("# name forward reverse [probe]   (tntblast assay file)\n")
        for a in wl.assays:
            cols = [a.name] + [x for x in (a.forward, a.reverse, a.probe) if x]
            f.write("\t".join(cols) + "\n")


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "cli")
    bad = 0
    for name, cfg, nt, na, seed, flags in CASES:
        d = os.path.join(work, name)
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(d):
            if f.startswith("ref.out") or f.startswith("hip.out"):
                os.remove(os.path.join(d, f))
        write_case(d, cfg, nt, na, seed, long_target=(name == "long_target"), fastq=(name == "fastq"))
        common = ["-i", os.path.join(d, "assays.txt"), "-d", os.path.join(d, "db.fna")] + flags
        env = dict(os.environ, OMP_NUM_THREADS="1")
        r1 = subprocess.run([REF] + common + ["-o", os.path.join(d, "ref.out")], env=env, capture_output=True, text=True)
        r2 = subprocess.run([CLI] + common + ["-o", os.path.join(d, "hip.out")], capture_output=True, text=True)
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
