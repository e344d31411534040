"""The host replay of std::list::sort used by the long-target path (cache-sized blocks, identity merges skipped) gives the
order of the plain breadth-first merge tree: compiled from tests/native/host_list_sort_check.hip with hipcc (host code only runs;
no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blocked_list_sort_equals_the_breadth_first_merge_tree(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / "host_list_sort_check")
    src = os.path.join(ROOT, "tests", "native", "host_list_sort_check.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-pass-failed", "-o", exe, src], check=True, cwd=ROOT, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3000 cases, 0 differ" in r.stdout
