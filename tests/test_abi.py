"""CPU: the C-ABI library cross-compiles for gfx950, loads, and exports every symbol the header declares.
No compute call is made (there is no GPU here and no CPU fallback in the product)."""
import ctypes as C
import os
import re

import pytest

import tntcheck as tc
from thermonucleotideblast_amd import abi, engine

ROOT = tc.ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tnt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tnt_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol(hip_lib):
    syms = header_symbols()
    assert syms, "no symbols parsed from the header"
    assert sorted(engine.EXPORTED_SYMBOLS) == syms
    for s in syms:
        assert hasattr(hip_lib, s), s
    assert hip_lib.tnt_abi_version() == 7


def test_struct_layouts_match_header(hip_lib):
    # tnt_options_default fills the ctypes mirror with the reference defaults (options.h:85-170)
    o = abi.tnt_options()
    hip_lib.tnt_options_default(C.byref(o))
    d = abi.options_default()
    for n, _ in abi.tnt_options._fields_:
        assert getattr(o, n) == getattr(d, n), n
    assert C.sizeof(abi.tnt_hit) == 30 * 4 + 3 * abi.TNT_ALIGN_TEXT
    assert C.sizeof(abi.tnt_bind) == 21 * 4 + abi.TNT_ALIGN_TEXT
    assert C.sizeof(abi.tnt_bind_req) == 28 and C.sizeof(abi.tnt_seed) == 20


def test_no_gpu_means_loud_failure(hip_lib):
    """Without a HIP device context creation must fail with TNT_E_DEVICE, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    o = abi.options_default()
    rc = hip_lib.tnt_ctx_create(C.byref(o), None, 0, 0, C.byref(h))
    assert rc == -2
    assert b"no CPU fallback" in hip_lib.tnt_last_error(None)


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "thermonucleotideblast_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                for needle in ("tnt_oracle", "libtntoracle", "libtntref", "tntcheck"):
                    assert needle not in text, (f, needle)
                assert not re.search(r"#include\s*[<\"].*oracle/", text), f


def test_cli_front_end_builds_and_validates_arguments(hip_lib):
    """tntblast-mi355x: built next to the library; argument errors are reported before any GPU call."""
    import subprocess
    import __graft_entry__ as g
    assert os.path.exists(g.CLI)
    r = subprocess.run([g.CLI], capture_output=True, text=True)
    assert r.returncode != 0 and "tntblast-mi355x" in r.stderr
    r = subprocess.run([g.CLI, "-i", "x", "-d", "y", "-m", "7"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unknown output format" in r.stderr
    r = subprocess.run([g.CLI, "-i", "x", "-d", "y", "-k", "T", "-K", "T", "-r", "T"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to open x" in r.stderr  # masking flags are accepted (G7 holds their output)
    r = subprocess.run([g.CLI, "-i", "x", "-d", "y", "--dinkelbach", "T"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to open x" in r.stderr  # accepted
    r = subprocess.run([g.CLI, "-i", "x", "-d", "y", "-D", "z"], capture_output=True, text=True)
    assert r.returncode != 0 and "but not both" in r.stderr  # options.cpp:535
    r = subprocess.run([g.CLI, "-i", "x", "-D", "z"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to open x" in r.stderr  # -D alone names the database like -d
    r = subprocess.run([g.CLI, "-i", "x", "-d", "y", "--max-target-len", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "max-target-len" in r.stderr
    r = subprocess.run([g.CLI, "-i", "x", "-d", "y", "--query-seg", "sometimes"], capture_output=True, text=True)
    assert r.returncode != 0 and "query-seg" in r.stderr
    r = subprocess.run([g.CLI, "-i", "x", "-d", "y", "--query-seg", "adaptive", "--max-target-len", "1000"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to open x" in r.stderr  # both accepted: the run gets as far as the assay file
    r = subprocess.run([g.CLI, "-i", "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "dbase" in r.stderr


def test_hits_unpack_rejects_corrupt_blobs(hip_lib):
    """tnt_hits_unpack is host code: a truncated or corrupt blob from a peer rank is refused before its count is
    handed out (a count the blob cannot hold would make the caller allocate for it)."""
    import struct
    n = C.c_size_t(123)
    blob = struct.pack("<II", 4_000_000_000, 16) + b"\0" * 8          # count far beyond what 16 bytes hold
    assert hip_lib.tnt_hits_unpack(blob, len(blob), None, 0, C.byref(n)) == -1 and n.value == 0
    blob = struct.pack("<II", 0, 4)                                    # total below the header size
    assert hip_lib.tnt_hits_unpack(blob, len(blob), None, 0, C.byref(n)) == -1
    blob = struct.pack("<II", 0, 64)                                   # total beyond the bytes handed over
    assert hip_lib.tnt_hits_unpack(blob, len(blob), None, 0, C.byref(n)) == -1
    blob = struct.pack("<II", 0, 8)                                    # the empty blob
    assert hip_lib.tnt_hits_unpack(blob, len(blob), None, 0, C.byref(n)) == 0 and n.value == 0
