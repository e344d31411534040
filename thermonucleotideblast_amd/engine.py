"""Python host mirror of the hot path's C ABI (include/tnt_mi355x.h) over libtnt_mi355x.so.

The reference's entry points for this path are the per-(target, assay) C++ functions
`amplicon()`, `padlock()`, `hybrid()` (tntblast.h:409-472) called from the OpenMP loop body of
tntblast_local.cpp:388-793.  `Engine.search()` is the batch-granular replacement of that loop
body: the whole (expanded) assay set against a device-resident batch of targets, returning the
`hybrid_sig` payload of every hit.  Errors surface as `TntError` carrying the library's message
(the reference throws `const char*`, throw.h:17-18).

There is no CPU fallback: if the HIP library is missing or no GPU is visible this raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# TNT_MI355X_LIB points at an alternative build of the same library (kernel experiments)
LIB_PATH = os.environ.get("TNT_MI355X_LIB") or os.path.join(_HERE, "libtnt_mi355x.so")

_lib = None


class TntError(RuntimeError):
    pass


def unpack_hits(blob):
    """Compact hit records (bytes-like, tnt_search_hits_compact's layout) -> ctypes array of tnt_hit (host only)."""
    lib = load_library()
    raw = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
    n = C.c_size_t()
    if lib.tnt_hits_unpack(raw, len(blob), None, 0, C.byref(n)) != 0:
        raise TntError("malformed hit blob")
    out = (abi.tnt_hit * n.value)()
    if lib.tnt_hits_unpack(raw, len(blob), out, n.value, C.byref(n)) != 0:
        raise TntError("malformed hit blob")
    return out


def load_library():
    """Load libtnt_mi355x.so (built in-tree by __graft_entry__.build()).  Never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TntError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64.so.7; whichever copy is loaded first serves the whole
    # process.  Load torch's first (when torch is installed) so torch tensors, RCCL and these kernels always
    # share ONE HIP runtime regardless of import order.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.tnt_abi_version.restype = C.c_int
    lib.tnt_ctx_warnings.argtypes = [C.c_void_p]
    lib.tnt_ctx_warnings.restype = C.c_char_p
    lib.tnt_long_pair_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.tnt_long_pair_stats.restype = C.c_int
    lib.tnt_options_default.argtypes = [P(abi.tnt_options)]
    lib.tnt_last_error.restype = C.c_char_p
    lib.tnt_last_error.argtypes = [C.c_void_p]
    lib.tnt_ctx_create.argtypes = [P(abi.tnt_options), P(abi.tnt_assay), C.c_size_t, C.c_int, P(C.c_void_p)]
    lib.tnt_ctx_destroy.argtypes = [C.c_void_p]
    lib.tnt_ctx_num_oligos.argtypes = [C.c_void_p]
    lib.tnt_ctx_oligo.argtypes = [C.c_void_p, C.c_int]
    lib.tnt_ctx_oligo.restype = C.c_char_p
    lib.tnt_ctx_delta_g.argtypes = [C.c_void_p, P(C.c_int32)]
    lib.tnt_db_create.argtypes = [C.c_void_p, P(C.c_uint8), P(C.c_uint64), C.c_size_t, C.c_uint32, P(C.c_void_p)]
    lib.tnt_db_destroy.argtypes = [C.c_void_p]
    lib.tnt_db_num_targets.argtypes = [C.c_void_p]
    lib.tnt_db_num_targets.restype = C.c_size_t
    lib.tnt_db_num_bases.argtypes = [C.c_void_p]
    lib.tnt_db_num_bases.restype = C.c_uint64
    lib.tnt_search.argtypes = [C.c_void_p, C.c_void_p, P(P(abi.tnt_hit)), P(C.c_size_t)]
    lib.tnt_search_enqueue.argtypes = [C.c_void_p, C.c_void_p]
    lib.tnt_search_collect.argtypes = [C.c_void_p, P(P(abi.tnt_hit)), P(C.c_size_t)]
    lib.tnt_search_hits_device.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_size_t)]
    lib.tnt_search_hits_compact.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_size_t), P(C.c_size_t)]
    lib.tnt_hits_unpack.argtypes = [C.c_void_p, C.c_size_t, P(abi.tnt_hit), C.c_size_t, P(C.c_size_t)]
    lib.tnt_last_timing.argtypes = [C.c_void_p, P(abi.tnt_timing)]
    lib.tnt_seeds.argtypes = [C.c_void_p, C.c_void_p, P(P(abi.tnt_seed)), P(C.c_size_t)]
    lib.tnt_bind_batch.argtypes = [C.c_void_p, C.c_void_p, P(abi.tnt_bind_req), C.c_size_t, P(abi.tnt_bind)]
    lib.tnt_bind_batch_routed.argtypes = [C.c_void_p, C.c_void_p, P(abi.tnt_bind_req), C.c_size_t, P(abi.tnt_bind), C.c_int]
    lib.tnt_filter_bound.argtypes = [C.c_void_p, C.c_void_p, P(abi.tnt_bind_req), C.c_size_t, P(C.c_int32)]
    lib.tnt_filter_bound_tier2.argtypes = [C.c_void_p, C.c_void_p, P(abi.tnt_bind_req), C.c_size_t, P(C.c_int32)]
    lib.tnt_hit_sequence.argtypes = [C.c_void_p, C.c_void_p, P(abi.tnt_hit), C.c_char_p, C.c_size_t]
    lib.tnt_db_create_from_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, P(C.c_void_p)]
    lib.tnt_db_load_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, P(C.c_void_p)]
    lib.tnt_fasta_open.argtypes = [C.c_char_p, P(C.c_void_p)]
    lib.tnt_fasta_next.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, P(C.c_void_p)]
    lib.tnt_fasta_close.argtypes = [C.c_void_p]
    lib.tnt_fasta_close.restype = None
    lib.tnt_db_defline.argtypes = [C.c_void_p, C.c_size_t, P(C.c_char_p), P(C.c_size_t)]
    lib.tnt_db_target_info.argtypes = [C.c_void_p, C.c_size_t, P(C.c_uint32), P(C.c_uint64)]
    lib.tnt_db_target_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, P(C.c_uint8), C.c_size_t]
    lib.tnt_db_target_codes.restype = C.c_int64
    lib.tnt_db_ingest_timing.argtypes = [C.c_void_p, P(abi.tnt_ingest_timing)]
    lib.tnt_oligo_structure.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_float, C.c_float, P(C.c_float), P(C.c_float), P(C.c_float)]
    lib.tnt_db_skip_targets.argtypes = [C.c_void_p, C.c_void_p, P(C.c_uint8), C.c_size_t]
    lib.tnt_long_text.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.tnt_seed_scan_only.argtypes = [C.c_void_p, C.c_void_p, C.c_int, P(C.c_float), P(C.c_uint64)]
    lib.tnt_search_collect_compact.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_size_t), P(C.c_size_t)]
    lib.tnt_device_count.restype = C.c_int
    lib.tnt_gang_create.argtypes = [P(C.c_void_p), C.c_int, P(C.c_void_p)]
    lib.tnt_gang_destroy.argtypes = [C.c_void_p]
    lib.tnt_gang_destroy.restype = None
    lib.tnt_gang_transport.argtypes = [C.c_void_p]
    lib.tnt_gang_transport.restype = C.c_char_p
    lib.tnt_gang_last_error.argtypes = [C.c_void_p]
    lib.tnt_gang_last_error.restype = C.c_char_p
    lib.tnt_gang_gather.argtypes = [C.c_void_p, P(C.c_uint8), P(C.c_void_p), P(C.c_size_t), P(C.c_size_t)]
    if lib.tnt_abi_version() != abi.TNT_ABI_VERSION:
        raise TntError("ABI version mismatch")
    _lib = lib
    return lib


# every symbol include/tnt_mi355x.h declares (checked by the CPU test-suite)
EXPORTED_SYMBOLS = [
    "tnt_options_default", "tnt_abi_version", "tnt_ctx_create", "tnt_ctx_destroy", "tnt_last_error", "tnt_ctx_warnings", "tnt_long_pair_stats", "tnt_filter_bound_tier2",
    "tnt_ctx_num_oligos", "tnt_ctx_oligo", "tnt_ctx_delta_g", "tnt_db_create", "tnt_db_destroy",
    "tnt_db_skip_targets", "tnt_db_num_targets", "tnt_db_num_bases", "tnt_search", "tnt_search_enqueue", "tnt_search_collect", "tnt_search_hits_device", "tnt_search_hits_compact", "tnt_hits_unpack",
    "tnt_last_timing", "tnt_seeds", "tnt_first_cull", "tnt_bind_batch", "tnt_bind_batch_routed", "tnt_filter_bound", "tnt_hit_sequence", "tnt_oligo_structure",
    "tnt_db_create_from_fasta", "tnt_db_load_fasta", "tnt_db_defline", "tnt_db_target_info", "tnt_db_target_codes", "tnt_db_ingest_timing",
    "tnt_fasta_open", "tnt_fasta_next", "tnt_fasta_close",
    "tnt_long_text", "tnt_seed_scan_only", "tnt_search_collect_compact", "tnt_device_count", "tnt_gang_create", "tnt_gang_destroy", "tnt_gang_transport", "tnt_gang_last_error", "tnt_gang_gather",
]


def _c_assays(expanded):
    arr = (abi.tnt_assay * max(1, len(expanded)))()
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


class TargetBatch:
    """Device-resident packed targets (replaces read_bio_seq + DNAHash::hash, tntblast_local.cpp:510,534)."""

    def __init__(self, engine, codes=None, offsets=None, first_seq_index=0, fasta_text=None, fasta_path=None):
        self.engine = engine
        if fasta_text is not None or fasta_path is not None:
            # FASTA ingest, parsed on the GPU (replaces sequence_data_fastx.cpp:13-79,187-382)
            self._h = C.c_void_p()
            if fasta_path is not None:
                rc = engine.lib.tnt_db_load_fasta(engine._h, os.fsencode(fasta_path), first_seq_index, C.byref(self._h))
            else:
                raw = fasta_text if isinstance(fasta_text, (bytes, bytearray)) else fasta_text.encode()
                self._keep = raw
                rc = engine.lib.tnt_db_create_from_fasta(engine._h, raw, len(raw), first_seq_index, C.byref(self._h))
            engine._check(rc)
            self.n_targets = int(engine.lib.tnt_db_num_targets(self._h))
            self.n_bases = int(engine.lib.tnt_db_num_bases(self._h))
            return
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if offsets.ndim != 1 or len(offsets) < 1 or int(offsets[-1]) > codes.size:
            raise TntError("bad offsets")
        self._h = C.c_void_p()
        rc = engine.lib.tnt_db_create(engine._h, codes.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1,
                                      first_seq_index, C.byref(self._h))
        engine._check(rc)
        self.n_targets = len(offsets) - 1
        self.n_bases = int(offsets[-1])

    def skip_targets(self, mask):
        """Leave the targets with mask[i] != 0 out of every later search (tnt_db_skip_targets)."""
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        self.engine._check(self.engine.lib.tnt_db_skip_targets(self.engine._h, self._h, m.ctypes.data_as(C.POINTER(C.c_uint8)), m.size))

    def defline(self, i):
        s, n = C.c_char_p(), C.c_size_t()
        self.engine._check(self.engine.lib.tnt_db_defline(self._h, i, C.byref(s), C.byref(n)))
        return C.string_at(s, n.value).decode() if n.value else ""

    def target_codes(self, i, cap=1 << 24):
        """1-byte codes (seq.h:12-35) of target i, rebuilt on the device from the packed rows."""
        buf = (C.c_uint8 * cap)()
        n = self.engine.lib.tnt_db_target_codes(self.engine._h, self._h, i, buf, cap)
        if n < 0:
            self.engine._check(int(n))
        return np.frombuffer(buf, dtype=np.uint8, count=int(n)).copy()

    def ingest_timing(self):
        t = abi.tnt_ingest_timing()
        self.engine._check(self.engine.lib.tnt_db_ingest_timing(self._h, C.byref(t)))
        return t

    def close(self):
        if self._h:
            self.engine.lib.tnt_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastaReader:
    """Batches of whole FASTA records as device-resident targets; the next batch is read while the current one is
    searched (tnt_fasta_open / tnt_fasta_next)."""

    def __init__(self, engine, path, max_bytes=1 << 30):
        self.engine, self.max_bytes = engine, max_bytes
        self._h = C.c_void_p()
        if engine.lib.tnt_fasta_open(os.fsencode(path), C.byref(self._h)) != 0:
            raise TntError("Unable to open fasta sequence file for reading")

    def __iter__(self):
        return self

    def __next__(self):
        h = C.c_void_p()
        self.engine._check(self.engine.lib.tnt_fasta_next(self._h, self.engine._h, self.max_bytes, C.byref(h)))
        if not h:
            raise StopIteration
        b = TargetBatch.__new__(TargetBatch)
        b.engine, b._h = self.engine, h
        b.n_targets = int(self.engine.lib.tnt_db_num_targets(h))
        b.n_bases = int(self.engine.lib.tnt_db_num_bases(h))
        return b

    def close(self):
        if self._h:
            self.engine.lib.tnt_fasta_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One context per GPU: options + expanded assay set + thermodynamic tables on `device`."""

    def __init__(self, options, expanded_assays, device=0):
        self.lib = load_library()
        self.options = options
        self.expanded = list(expanded_assays)
        self._h = C.c_void_p()
        arr = _c_assays(self.expanded)
        rc = self.lib.tnt_ctx_create(C.byref(options), arr, len(self.expanded), device, C.byref(self._h))
        if rc != 0:
            raise TntError("tnt_ctx_create: %s (code %d)" % (self.lib.tnt_last_error(None).decode(), rc))
        self.oligos = [self.lib.tnt_ctx_oligo(self._h, i).decode() for i in range(self.lib.tnt_ctx_num_oligos(self._h))]

    def _check(self, rc):
        if rc != 0:
            raise TntError("%s (code %d)" % (self.lib.tnt_last_error(self._h).decode(), rc))

    def close(self):
        if self._h:
            self.lib.tnt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pack(self, codes, offsets, first_seq_index=0):
        return TargetBatch(self, codes, offsets, first_seq_index)

    def pack_fasta(self, text, first_seq_index=0):
        """FASTA text (bytes or str) -> device-resident targets, parsed on the GPU."""
        return TargetBatch(self, first_seq_index=first_seq_index, fasta_text=text)

    def load_fasta(self, path, first_seq_index=0):
        """FASTA file (plain or gzip) -> device-resident targets."""
        return TargetBatch(self, first_seq_index=first_seq_index, fasta_path=path)

    def delta_g(self):
        out = (C.c_int32 * 2401)()
        self._check(self.lib.tnt_ctx_delta_g(self._h, out))
        return np.frombuffer(out, dtype=np.int32).copy()

    def search(self, batch):
        """seeds -> cull -> bind -> pair for every (target, assay); returns a ctypes array of tnt_hit."""
        hits = C.POINTER(abi.tnt_hit)()
        n = C.c_size_t()
        self._check(self.lib.tnt_search(self._h, batch._h, C.byref(hits), C.byref(n)))
        out = (abi.tnt_hit * n.value)()
        if n.value:
            C.memmove(out, hits, C.sizeof(abi.tnt_hit) * n.value)
        return out

    def enqueue(self, batch):
        self._check(self.lib.tnt_search_enqueue(self._h, batch._h))

    def collect(self, copy=True):
        """Hit records of the enqueued search.  copy=False returns a view of the library-owned (pinned) buffer
        that stays valid until the next search on this context."""
        hits = C.POINTER(abi.tnt_hit)()
        n = C.c_size_t()
        self._check(self.lib.tnt_search_collect(self._h, C.byref(hits), C.byref(n)))
        if not n.value:
            return (abi.tnt_hit * 0)()
        view = (abi.tnt_hit * n.value).from_address(C.addressof(hits.contents))
        if not copy:
            return view
        out = (abi.tnt_hit * n.value)()
        C.memmove(out, view, C.sizeof(abi.tnt_hit) * n.value)
        return out

    def long_text(self, field):
        """The text of an alignment field (bytes of tnt_hit::*_align / tnt_bind::align): the field itself, or the text the context
        keeps for an alignment that does not fit TNT_ALIGN_TEXT (tnt_long_text)."""
        raw = bytes(field)
        if not raw.startswith(b"\x7fLT"):
            return raw.split(b"\0", 1)[0].decode("latin-1")
        buf = C.create_string_buffer(8192)
        n = self.lib.tnt_long_text(self._h, raw, buf, len(buf))
        if n < 0:
            raise TntError("tnt_long_text failed (%d)" % n)
        return buf.raw[:n].decode("latin-1")

    def scan_only(self, batch, repeats=5):
        """(ms per pass, raw candidates per pass) of the seed stage's scan on its own (tnt_seed_scan_only)."""
        ms, n = C.c_float(), C.c_uint64()
        self._check(self.lib.tnt_seed_scan_only(self._h, batch._h, repeats, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def collect_device(self):
        """Collects the enqueued search and leaves the hit records on the device (tnt_search_collect with hits == NULL);
        returns their number."""
        n = C.c_size_t()
        self._check(self.lib.tnt_search_collect(self._h, None, C.byref(n)))
        return n.value

    def collect_compact(self):
        """Collects the enqueued search; the hits reach the host in compact form (tnt_search_collect_compact).  Returns
        (host address of the blob -- page-locked memory of the context, valid until its next search --, bytes, hits)."""
        ptr, nb, n = C.c_void_p(), C.c_size_t(), C.c_size_t()
        self._check(self.lib.tnt_search_collect_compact(self._h, C.byref(ptr), C.byref(nb), C.byref(n)))
        return (ptr.value or 0), nb.value, n.value

    def unpack(self, address, n_bytes, n_hits):
        """tnt_hits_unpack: the records of a compact blob in host memory."""
        out = (abi.tnt_hit * max(1, n_hits))()
        n = C.c_size_t()
        rc = self.lib.tnt_hits_unpack(C.c_void_p(address), n_bytes, out, n_hits, C.byref(n))
        if rc != 0 or n.value != n_hits:
            raise TntError("tnt_hits_unpack failed (%d)" % rc)
        return (abi.tnt_hit * n_hits).from_buffer(out) if n_hits else (abi.tnt_hit * 0)()

    def hits_device(self):
        """(device pointer, count) of the last collected search's hit records in device memory (those of targets seeded in
        windows, paired on host threads, are copied behind the pair kernel's own): see tnt_search_hits_device."""
        ptr = C.c_void_p()
        n = C.c_size_t()
        rc = self.lib.tnt_search_hits_device(self._h, C.byref(ptr), C.byref(n))
        if rc == -4:  # TNT_E_UNSUPPORTED
            return None
        self._check(rc)
        return (ptr.value or 0), n.value

    def hits_compact(self):
        """(device pointer, bytes, count) of the last collected search's hits in compact form (tnt_search_hits_compact)."""
        ptr, nb, n = C.c_void_p(), C.c_size_t(), C.c_size_t()
        rc = self.lib.tnt_search_hits_compact(self._h, C.byref(ptr), C.byref(nb), C.byref(n))
        if rc == -4:  # TNT_E_UNSUPPORTED
            return None
        self._check(rc)
        return (ptr.value or 0), nb.value, n.value

    def warnings(self):
        """tnt_ctx_warnings: "" or what the caller should know about the options (the Q11 regime)."""
        return (self.lib.tnt_ctx_warnings(self._h) or b"").decode()

    def long_pair_stats(self):
        """tnt_long_pair_stats: the lists of windowed targets in the last search -- (lists on the device, PCR lists handed back
        to the host threads, lists of the host threads from the start, hits the device reported)."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.tnt_long_pair_stats(self._h, out))
        return tuple(int(x) for x in out)

    def timing(self):
        t = abi.tnt_timing()
        self._check(self.lib.tnt_last_timing(self._h, C.byref(t)))
        return t

    def seeds(self, batch):
        s = C.POINTER(abi.tnt_seed)()
        n = C.c_size_t()
        self._check(self.lib.tnt_seeds(self._h, batch._h, C.byref(s), C.byref(n)))
        return [(s[i].seq_index, self.oligos[s[i].oligo], s[i].strand, s[i].query_loc, s[i].target_loc) for i in range(n.value)]

    def first_cull(self, batch):
        """(n, 7) int32 rows (seq_index, assay, role, strand, query_loc, target_loc, minus_first): the seeds of the PCR assays
        that survive amplicon()'s early exits and the first cull (tnt_first_cull)."""
        import numpy as np
        rows = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        self._check(self.lib.tnt_first_cull(self._h, batch._h, C.byref(rows), C.byref(n)))
        if not n.value:
            return np.zeros((0, 7), dtype=np.int32)
        return np.ctypeslib.as_array(rows, shape=(n.value, 7)).copy()

    def bind_batch(self, batch, reqs, route=0):
        """route 0: the unit-level kernel instance (full record); 1: the throughput instances a search runs, queue split
        by oligo length; 2: the longest general throughput instance; 3: like 1 with every bucket bound by the
        wavefront-per-window kernel; 4: the unit-level instance of that kernel (tnt_bind_batch_routed)."""
        out = (abi.tnt_bind * len(reqs))()
        self._check(self.lib.tnt_bind_batch_routed(self._h, batch._h, reqs, len(reqs), out, route))
        return out

    def filter_bound(self, batch, reqs):
        """(n, 2) int32 array [bound, threshold] of the bind stage's pre-filter for every request (tnt_filter_bound)."""
        out = np.zeros((len(reqs), 2), dtype=np.int32)
        self._check(self.lib.tnt_filter_bound(self._h, batch._h, reqs, len(reqs), out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def filter_bound_tier2(self, batch, reqs):
        """(n, 2) int32 array [credit of the maximal-score paths, threshold] of the pre-filter's second tier (tnt_filter_bound_tier2)."""
        out = np.zeros((len(reqs), 2), dtype=np.int32)
        self._check(self.lib.tnt_filter_bound_tier2(self._h, batch._h, reqs, len(reqs), out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def oligo_structure(self, a, b=None, ct_a=9.0e-7, ct_b=9.0e-7):
        """(hairpin Tm, homodimer Tm, heterodimer Tm or None) as printed with every hit (tntblast_local.cpp:656-686)."""
        h, d, x = C.c_float(), C.c_float(), C.c_float()
        self._check(self.lib.tnt_oligo_structure(self._h, a.encode(), b.encode() if b else None, ct_a, ct_b,
                                                 C.byref(h), C.byref(d), C.byref(x)))
        return h.value, d.value, (x.value if b else None)

    def hit_sequence(self, batch, hit):
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.tnt_hit_sequence(self._h, batch._h, C.byref(hit), buf, len(buf))
        if n < 0:
            raise TntError("tnt_hit_sequence failed (%d)" % n)
        return buf.value.decode()


class Gang:
    """Several contexts of one process, one per device (or several on one device: host transport), and the gather of
    their hits to the host (tnt_gang_*): what tntblast-mi355x --gpus N runs on."""

    def __init__(self, engines):
        self.lib = load_library()
        self.engines = list(engines)
        arr = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
        self._h = C.c_void_p()
        rc = self.lib.tnt_gang_create(arr, len(self.engines), C.byref(self._h))
        if rc != 0:
            raise TntError((self.lib.tnt_gang_last_error(None) or b"").decode() or "tnt_gang_create failed (%d)" % rc)

    @property
    def transport(self):
        return self.lib.tnt_gang_transport(self._h).decode()

    def gather(self, has=None):
        """Per rank the hit records (unpacked) of its last collected search; None for ranks left out."""
        n = len(self.engines)
        blobs, nb, nh = (C.c_void_p * n)(), (C.c_size_t * n)(), (C.c_size_t * n)()
        mask = None if has is None else (C.c_uint8 * n)(*[1 if x else 0 for x in has])
        rc = self.lib.tnt_gang_gather(self._h, mask, blobs, nb, nh)
        if rc != 0:
            raise TntError(self.lib.tnt_gang_last_error(self._h).decode())
        return [None if (has is not None and not has[r]) else self.engines[r].unpack(blobs[r] or 0, nb[r], nh[r]) for r in range(n)]

    def close(self):
        if self._h:
            self.lib.tnt_gang_destroy(self._h)
            self._h = C.c_void_p()
