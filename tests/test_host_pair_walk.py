"""The bounded walks of pair_group's pairing loops (round 4: amplicon() stops at the first reverse primer whose loc5 puts the amplicon
beyond max_len, padlock() walks the stretch of the downstream list a ligation can come from) report the pairs of the reference's walk
over every pair, in its order: tests/native/host_pair_walk_check.hip compiled both ways with hipcc (host code only runs; no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bounded_pairing_walks_report_the_same_pairs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = os.path.join(ROOT, "tests", "native", "host_pair_walk_check.hip")
    outs = []
    for name, flags in (("bounded", []), ("full", ["-DTNT_EXP_FULL_WALK"])):
        exe = str(tmp_path / ("host_pair_walk_" + name))
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-pass-failed", "-pthread"] + flags + ["-o", exe, src],
                       check=True, cwd=ROOT, capture_output=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
        outs.append(r.stdout.splitlines())
    assert len(outs[0]) == 240 and outs[0] == outs[1]
    hits = [int(line.split(": ")[1].split(" hits")[0]) for line in outs[0]]
    assert sum(1 for h in hits if h > 20) > 60 and all("err 0" in line for line in outs[0])
    assert sum(1 for line, h in zip(outs[0], hits) if " kind 1 " in line and h > 0) > 20  # padlock / MIPS lists too
