"""Test helpers: uniform Python access to the three implementations of the hot path.

  RefImpl     oracle/_ref/libtntref.so   the reference's own objects behind oracle/ref_harness.cpp
  OracleImpl  oracle/libtntoracle.so     this repository's CPU restatement (oracle/tnt_oracle.cpp)
  (the HIP product is reached through thermonucleotideblast_amd.Engine)

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from thermonucleotideblast_amd import abi  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libtntref.so")
ORACLE_SO = os.path.join(ORACLE_DIR, "libtntoracle.so")


def build_oracle():
    """Compile the checkers (building the checker is not using it)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle", "ref"], check=True)


def have_ref():
    return os.path.exists(REF_SO)


def c_assays(expanded):
    arr = (abi.tnt_assay * len(expanded))()
    keep = []
    for i, e in enumerate(expanded):
        arr[i].id, arr[i].degen_id = e.id, e.degen_id
        for nm in ("forward", "reverse", "probe"):
            v = getattr(e, nm)
            b = v.encode() if v else None
            keep.append(b)
            setattr(arr[i], nm, b)
        arr[i].forward_degen, arr[i].reverse_degen, arr[i].probe_degen = e.forward_degen, e.reverse_degen, e.probe_degen
    arr._keep = keep
    return arr


def c_strings(strs):
    arr = (C.c_char_p * len(strs))()
    arr[:] = [s.encode() for s in strs]
    return arr


def as_u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def as_u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


def struct_to_dict(s):
    d = {}
    for n, t in s._fields_:
        v = getattr(s, n)
        if isinstance(v, bytes):
            v = v.decode()
        d[n] = v
    return d


def f32_bits(x):
    return int(np.float32(x).view(np.uint32))


def hit_key(h, with_text=True, text_of=None):
    """Canonical, order-independent, bit-exact key of one hit (floats as uint32 bit patterns).  text_of: reads an alignment
    field (Engine.long_text: an alignment of an oligo beyond 64 bases that does not fit the field is kept by the context);
    texts are compared up to the field's size, where the CPU checkers cut theirs."""
    k = []
    for n, t in abi.tnt_hit._fields_:
        v = getattr(h, n) if not isinstance(h, dict) else h[n]
        if t is C.c_float:
            k.append(f32_bits(v))
        elif n.endswith("_align"):
            if with_text:
                if text_of is not None:
                    k.append(text_of(v)[:abi.TNT_ALIGN_TEXT - 1])
                else:
                    k.append(v.decode() if isinstance(v, bytes) else v)
        else:
            k.append(int(v))
    return tuple(k)


def bind_key(b):
    k = []
    for n, t in abi.tnt_bind._fields_:
        v = getattr(b, n)
        if t is C.c_float:
            k.append(f32_bits(v))
        elif n == "align":
            k.append(v.decode())
        else:
            k.append(int(v))
    return tuple(k)


def bind_compare_fields(a, b, require_text=True):
    """Field-wise diff of two tnt_bind records under the parity rules: everything is compared when
    an alignment exists; without one only status/window/tm/dH/dS/dg/valid/dp_max/n_max count."""
    diffs = []
    names = ["status", "target_start", "target_stop", "tm", "dH", "dS", "dg", "valid", "dp_max", "n_max"]
    if a.valid and b.valid:
        names += ["anchor5", "anchor3", "num_mm", "num_gap", "poly_degen", "loc5", "loc3",
                  "q_first", "q_last", "t_first", "t_last"]
        if require_text:
            names.append("align")
    for n in names:
        va, vb = getattr(a, n), getattr(b, n)
        if isinstance(va, float):
            if f32_bits(va) != f32_bits(vb):
                diffs.append((n, va, vb))
        elif va != vb:
            diffs.append((n, va, vb))
    return diffs


class _CpuImpl:
    prefix = None
    path = None

    def __init__(self):
        self.lib = C.CDLL(self.path)
        p = self.prefix
        self._tables = getattr(self.lib, p + "tables")
        self._seeds = getattr(self.lib, p + "seeds")
        self._bind = getattr(self.lib, p + "bind_batch")
        self._search = getattr(self.lib, p + "search")
        self._free = getattr(self.lib, p + "free")
        self._free.argtypes = [C.c_void_p]
        self._err = getattr(self.lib, p + "last_error")
        self._err.restype = C.c_char_p
        self._seeds.argtypes = [C.POINTER(abi.tnt_options), C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint8),
                                C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32,
                                C.POINTER(C.POINTER(abi.tnt_seed)), C.POINTER(C.c_size_t)]
        self._bind.argtypes = [C.POINTER(abi.tnt_options), C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint8),
                               C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32,
                               C.POINTER(abi.tnt_bind_req), C.c_size_t, C.POINTER(abi.tnt_bind)]
        self._search.argtypes = [C.POINTER(abi.tnt_options), C.POINTER(abi.tnt_assay), C.c_size_t,
                                 C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32,
                                 C.POINTER(C.POINTER(abi.tnt_hit)), C.POINTER(C.c_size_t), C.c_int]

    def error(self):
        return self._err().decode()

    def tables(self, opt):
        dg = (C.c_int32 * 2401)()
        H = (C.c_float * 2401)()
        S = (C.c_float * 2401)()
        loop = (C.c_float * 513)()
        bulge = (C.c_float * 513)()
        hair = (C.c_float * 513)()
        sc = (C.c_float * 8)()
        rc = self._tables(C.byref(opt), dg, H, S, loop, bulge, hair, sc)
        assert rc == 0, self.error()
        f = lambda a, n=None: np.frombuffer(a, dtype=np.float32)[:n].copy().view(np.uint32)  # noqa: E731
        return {"delta_g": np.frombuffer(dg, dtype=np.int32).copy(), "H": f(H), "S": f(S),
                "loop_S": f(loop, 512), "bulge_S": f(bulge, 512), "scalars": f(sc)}

    def seeds(self, opt, oligos, codes, offsets, first=0):
        codes, pc = as_u8(codes)
        offsets, po = as_u64(offsets)
        out = C.POINTER(abi.tnt_seed)()
        n = C.c_size_t()
        rc = self._seeds(C.byref(opt), c_strings(oligos), len(oligos), pc, po, len(offsets) - 1, first,
                         C.byref(out), C.byref(n))
        assert rc == 0, self.error()
        res = [(out[i].seq_index, out[i].oligo, out[i].strand, out[i].query_loc, out[i].target_loc) for i in range(n.value)]
        self._free(out)
        return res

    def bind_batch(self, opt, oligos, codes, offsets, reqs, first=0):
        codes, pc = as_u8(codes)
        offsets, po = as_u64(offsets)
        out = (abi.tnt_bind * len(reqs))()
        rc = self._bind(C.byref(opt), c_strings(oligos), len(oligos), pc, po, len(offsets) - 1, first,
                        reqs, len(reqs), out)
        assert rc == 0, self.error()
        return out

    def filter_bound(self, opt, oligos, codes, offsets, reqs, first=0, tier=1):
        """Checker of the bind pre-filter (oracle only): (n, 2) int32 array of [bound, threshold] per request.
        tier=2: the second tier's bound (terminal stacks charged as they are) against the same threshold."""
        if tier != 1:
            self.lib.tntor_set_filter_tier(int(tier))
            try:
                return self.filter_bound(opt, oligos, codes, offsets, reqs, first)
            finally:
                self.lib.tntor_set_filter_tier(1)
        codes, pc = as_u8(codes)
        offsets, po = as_u64(offsets)
        out = np.zeros((len(reqs), 2), dtype=np.int32)
        f = self.lib.tntor_filter_bound
        f.argtypes = [C.POINTER(abi.tnt_options), C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
                      C.c_size_t, C.c_uint32, C.POINTER(abi.tnt_bind_req), C.c_size_t, C.POINTER(C.c_int32)]
        rc = f(C.byref(opt), c_strings(oligos), len(oligos), pc, po, len(offsets) - 1, first, reqs, len(reqs),
               out.ctypes.data_as(C.POINTER(C.c_int32)))
        assert rc == 0, self.error()
        return out

    def search(self, opt, expanded, codes, offsets, first=0, threads=0, bind_cache=True):
        """bind_cache=False (oracle only): every use of an oligo is evaluated on its own instead of through the reference's
        per-target BindCache (a6) -- the semantics of TNT_MI355X_UNCACHED_SHARED_OLIGOS."""
        if not bind_cache:
            self.lib.tntor_set_bind_cache(0)
            try:
                return self.search(opt, expanded, codes, offsets, first, threads)
            finally:
                self.lib.tntor_set_bind_cache(1)
        codes, pc = as_u8(codes)
        offsets, po = as_u64(offsets)
        arr = c_assays(expanded)
        out = C.POINTER(abi.tnt_hit)()
        n = C.c_size_t()
        rc = self._search(C.byref(opt), arr, len(expanded), pc, po, len(offsets) - 1, first, C.byref(out),
                          C.byref(n), threads)
        assert rc == 0, self.error()
        hits = (abi.tnt_hit * n.value)()
        if n.value:
            C.memmove(hits, out, C.sizeof(abi.tnt_hit) * n.value)
        self._free(out)
        return hits


    def first_cull(self, opt, expanded, codes, offsets, first=0):
        """Rows (seq_index, assay, role, strand, query_loc, target_loc, minus_first) of the seeds that survive amplicon()'s
        early exits and the first cull (PCR assays)."""
        codes, pc = as_u8(codes)
        offsets, po = as_u64(offsets)
        arr = c_assays(expanded)
        f = getattr(self.lib, self.prefix + "first_cull")
        f.argtypes = [C.POINTER(abi.tnt_options), C.POINTER(abi.tnt_assay), C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
                      C.c_size_t, C.c_uint32, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_size_t)]
        out = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        rc = f(C.byref(opt), arr, len(expanded), pc, po, len(offsets) - 1, first, C.byref(out), C.byref(n))
        assert rc == 0, self.error()
        rows = np.ctypeslib.as_array(out, shape=(max(1, n.value), 7))[:n.value].copy()
        self._free(out)
        return rows

    def structure(self, opt, a, b=None, ct_a=9.0e-7, ct_b=9.0e-7):
        """(hairpin, homodimer, heterodimer) Tm of tntblast_local.cpp:656-686 as float32 bit patterns."""
        h, d, x = C.c_float(), C.c_float(), C.c_float()
        f = getattr(self.lib, self.prefix + "oligo_structure")
        rc = f(C.byref(opt), a.encode(), C.c_float(ct_a), C.byref(h), C.byref(d))
        assert rc == 0, self.error()
        if b:
            g = getattr(self.lib, self.prefix + "heterodimer")
            rc = g(C.byref(opt), a.encode(), b.encode(), C.c_float(ct_a), C.c_float(ct_b), C.byref(x))
            assert rc == 0, self.error()
        return f32_bits(h.value), f32_bits(d.value), (f32_bits(x.value) if b else None)


    def _fasta_out(self, rc, codes, offsets, n, defs):
        assert rc == 0, self.error()
        nt = n.value
        off = np.ctypeslib.as_array(offsets, shape=(nt + 1,)).copy()
        cod = np.ctypeslib.as_array(codes, shape=(max(1, int(off[-1])),)).copy()[:int(off[-1])]
        names = C.string_at(defs).decode("latin-1").split("\n")[:nt]
        for p in (codes, offsets, defs):
            self._free(p)
        return cod, off, names

    def parse_fasta(self, text):
        """FASTA bytes -> (codes, offsets, deflines).  The reference reads files, so RefImpl goes through a temp file."""
        codes, offsets, n, defs = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)(), C.c_size_t(), C.c_void_p()
        if self.prefix == "tntor_":
            f = self.lib.tntor_parse_fasta
            f.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint64)),
                          C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
            rc = f(text, len(text), C.byref(codes), C.byref(offsets), C.byref(n), C.byref(defs))
        else:
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".fna", delete=False) as tmp:
                tmp.write(text)
            try:
                return self.read_fasta(tmp.name)
            finally:
                os.unlink(tmp.name)
        return self._fasta_out(rc, codes, offsets, n, defs)

    def read_fasta(self, path):
        codes, offsets, n, defs = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)(), C.c_size_t(), C.c_void_p()
        assert self.prefix == "tntref_"
        f = self.lib.tntref_read_fasta
        f.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t),
                      C.POINTER(C.c_void_p)]
        rc = f(os.fsencode(path), C.byref(codes), C.byref(offsets), C.byref(n), C.byref(defs))
        return self._fasta_out(rc, codes, offsets, n, defs)


class RefImpl(_CpuImpl):
    prefix = "tntref_"
    path = REF_SO


class OracleImpl(_CpuImpl):
    prefix = "tntor_"
    path = ORACLE_SO


def make_reqs(rows):
    """rows: iterable of (seq_index, oligo, strand, query_loc, target_loc, role, ct)."""
    rows = list(rows)
    arr = (abi.tnt_bind_req * len(rows))()
    for i, r in enumerate(rows):
        (arr[i].seq_index, arr[i].oligo, arr[i].strand, arr[i].query_loc, arr[i].target_loc, arr[i].role, arr[i].ct) = r
    return arr


def sorted_hit_keys(hits, with_text=True, text_of=None):
    return sorted(hit_key(h, with_text, text_of) for h in hits)
