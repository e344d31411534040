"""CPU, build container only: the assay front end of tntblast-mi355x (parser, --plex expansion, degenerate expansion) against
the reference binary (oracle/_ref/tntblast), through --dump-query.  The front end prints the query list before it
creates the device context, so this needs no GPU."""
import os
import random
import subprocess

import pytest

pytestmark = pytest.mark.ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "tntblast")
CLI = os.path.join(ROOT, "thermonucleotideblast_amd", "tntblast-mi355x")


def _queries(stdout):
    return [l for l in stdout.split("\n") if "\t" in l and not l.startswith("\t")]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_plex_and_degenerate_expansion_match_the_reference(tmp_path, seed):
    if not (os.path.exists(REF) and os.path.exists(CLI)):
        pytest.skip("needs oracle/_ref/tntblast and the built front end")
    rnd = random.Random(seed)
    (tmp_path / "db.fna").write_text(">s1\n" + "".join(rnd.choice("ACGT") for _ in range(200)) + "\n")
    compared = 0
    for case in range(10):
        n = rnd.randint(2, 14)
        pool = ["".join(rnd.choice("ACGT") for _ in range(rnd.randint(18, 22))) for _ in range(max(3, n))]
        if rnd.random() < 0.4:
            pool[0] = pool[0][:5] + rnd.choice("RYKMSWN") + pool[0][6:]
        fmt = rnd.choice(["PCR", "PCR", "PADLOCK"])
        plex = rnd.random() < 0.8
        lines = []
        for i in range(n):
            cols = ["as%d" % i, rnd.choice(pool), rnd.choice(pool)]
            if fmt == "PCR" and rnd.random() < 0.4:
                cols.append(rnd.choice(pool))
            if fmt == "PCR" and rnd.random() < 0.1:
                cols = cols[:2]
            lines.append(" ".join(cols))
        (tmp_path / "assays.txt").write_text("# name forward reverse [probe]\n" + "\n".join(lines) + "\n")
        args = ["-i", str(tmp_path / "assays.txt"), "-d", str(tmp_path / "db.fna"), "-e", "70", "-E", "70", "--plex", "T" if plex else "F",
                "--dump-query", "T", "-A", fmt, "-o", str(tmp_path / "o.txt")]
        r1 = subprocess.run([REF] + args, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
        r2 = subprocess.run([CLI] + args, capture_output=True, text=True)
        a, b = _queries(r1.stdout), _queries(r2.stdout)
        assert a == b, (case, fmt, plex, lines)
        compared += len(a)
    assert compared > 50
