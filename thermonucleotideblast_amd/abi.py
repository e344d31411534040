"""ctypes mirror of include/tnt_mi355x.h (struct layouts + constants).

Pure declarations: importing this module loads no native code.
"""
import ctypes as C

TNT_ABI_VERSION = 7
TNT_MAX_OLIGO = 1024
TNT_FLANK = 4
TNT_ALIGN_TEXT = 448

ASSAY_PCR, ASSAY_PROBE, ASSAY_PADLOCK, ASSAY_MIPS = 0, 1, 2, 3
STRAND_PLUS, STRAND_MINUS, STRAND_BOTH = 1, 2, 3
PLUS, MINUS = 0, 1

# seq.h:12-35
DB_CODES = "ACGTIMRSVWYHKDBN"
DB_GAP, DB_UNKNOWN = 16, 17


class tnt_options(C.Structure):
    _fields_ = [
        ("min_primer_tm", C.c_float), ("max_primer_tm", C.c_float),
        ("min_primer_dg", C.c_float), ("max_primer_dg", C.c_float),
        ("min_probe_tm", C.c_float), ("max_probe_tm", C.c_float),
        ("min_probe_dg", C.c_float), ("max_probe_dg", C.c_float),
        ("salt", C.c_float), ("primer_strand", C.c_float), ("probe_strand", C.c_float),
        ("asymmetric_strand_ratio", C.c_float), ("target_t", C.c_float),
        ("primer_clamp", C.c_int32), ("min_max_primer_clamp", C.c_int32),
        ("probe_clamp_5", C.c_int32), ("probe_clamp_3", C.c_int32),
        ("max_gap", C.c_int32), ("max_mismatch", C.c_int32), ("max_poly_degen", C.c_int32),
        ("max_len", C.c_int32), ("assay_format", C.c_int32), ("target_strand", C.c_int32),
        ("hash_word_size", C.c_int32), ("single_primer_pcr", C.c_int32),
        ("allow_dangle_5", C.c_int32), ("allow_dangle_3", C.c_int32), ("use_dinkelbach", C.c_int32),
    ]


def options_default(**kw):
    """Defaults of options.h:85-170 / tntblast.h:19-76."""
    o = tnt_options()
    o.min_primer_tm, o.max_primer_tm = 0.0, 9999.0
    o.min_primer_dg, o.max_primer_dg = -9999.0, 0.0
    o.min_probe_tm, o.max_probe_tm = 0.0, 9999.0
    o.min_probe_dg, o.max_probe_dg = -9999.0, 0.0
    o.salt = 50.0e-3
    o.primer_strand = 9.0e-7
    o.probe_strand = 2.5e-7
    o.asymmetric_strand_ratio = 1.0
    o.target_t = 310.15
    o.primer_clamp = 0
    o.min_max_primer_clamp = -1
    o.probe_clamp_5 = o.probe_clamp_3 = 0
    o.max_gap = o.max_mismatch = 999
    o.max_poly_degen = 3
    o.max_len = 2000
    o.assay_format = ASSAY_PCR
    o.target_strand = STRAND_BOTH
    o.hash_word_size = 7
    o.single_primer_pcr = 1
    o.allow_dangle_5 = o.allow_dangle_3 = 0
    o.use_dinkelbach = 0
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class tnt_assay(C.Structure):
    _fields_ = [
        ("id", C.c_int32), ("degen_id", C.c_int32),
        ("forward", C.c_char_p), ("reverse", C.c_char_p), ("probe", C.c_char_p),
        ("forward_degen", C.c_int32), ("reverse_degen", C.c_int32), ("probe_degen", C.c_int32),
    ]


class tnt_hit(C.Structure):
    _fields_ = [
        ("assay_id", C.c_int32), ("degen_id", C.c_int32), ("seq_index", C.c_int32),
        ("amplicon_start", C.c_int32), ("amplicon_stop", C.c_int32),
        ("probe_start", C.c_int32), ("probe_stop", C.c_int32),
        ("forward_tm", C.c_float), ("reverse_tm", C.c_float), ("probe_tm", C.c_float),
        ("forward_dH", C.c_float), ("forward_dS", C.c_float),
        ("reverse_dH", C.c_float), ("reverse_dS", C.c_float),
        ("probe_dH", C.c_float), ("probe_dS", C.c_float),
        ("primer_strand", C.c_int32), ("probe_strand", C.c_int32),
        ("forward_primer_clamp", C.c_int32), ("reverse_primer_clamp", C.c_int32),
        ("forward_mm", C.c_int32), ("forward_gap", C.c_int32),
        ("reverse_mm", C.c_int32), ("reverse_gap", C.c_int32),
        ("probe_mm", C.c_int32), ("probe_gap", C.c_int32),
        ("forward_oligo", C.c_int32), ("reverse_oligo", C.c_int32),
        ("has_primers", C.c_int32), ("has_probe", C.c_int32),
        ("forward_align", C.c_char * TNT_ALIGN_TEXT),
        ("reverse_align", C.c_char * TNT_ALIGN_TEXT),
        ("probe_align", C.c_char * TNT_ALIGN_TEXT),
    ]


class tnt_seed(C.Structure):
    _fields_ = [("seq_index", C.c_int32), ("oligo", C.c_int32), ("strand", C.c_int32),
                ("query_loc", C.c_int32), ("target_loc", C.c_int32)]


class tnt_bind(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("target_start", C.c_int32), ("target_stop", C.c_int32),
        ("tm", C.c_float), ("dH", C.c_float), ("dS", C.c_float), ("dg", C.c_float),
        ("anchor5", C.c_int32), ("anchor3", C.c_int32), ("num_mm", C.c_int32),
        ("num_gap", C.c_int32), ("poly_degen", C.c_int32),
        ("loc5", C.c_int32), ("loc3", C.c_int32),
        ("q_first", C.c_int32), ("q_last", C.c_int32), ("t_first", C.c_int32), ("t_last", C.c_int32),
        ("valid", C.c_int32), ("dp_max", C.c_int32), ("n_max", C.c_int32),
        ("align", C.c_char * TNT_ALIGN_TEXT),
    ]


class tnt_bind_req(C.Structure):
    _fields_ = [("seq_index", C.c_int32), ("oligo", C.c_int32), ("strand", C.c_int32),
                ("query_loc", C.c_int32), ("target_loc", C.c_int32), ("role", C.c_int32),
                ("ct", C.c_float)]


class tnt_timing(C.Structure):
    _fields_ = [("seed_ms", C.c_float), ("bind_ms", C.c_float), ("pair_ms", C.c_float),
                ("total_ms", C.c_float), ("n_seeds", C.c_uint64), ("n_binds", C.c_uint64),
                ("n_bound", C.c_uint64), ("n_hits", C.c_uint64), ("n_filtered", C.c_uint64), ("n_cells", C.c_uint64)]


class tnt_ingest_timing(C.Structure):
    _fields_ = [("read_ms", C.c_float), ("index_ms", C.c_float), ("h2d_ms", C.c_float), ("count_ms", C.c_float),
                ("layout_ms", C.c_float), ("pack_ms", C.c_float), ("text_bytes", C.c_uint64), ("bases", C.c_uint64),
                ("exceptions", C.c_uint64)]


HIT_SCALAR_FIELDS = [n for n, t in tnt_hit._fields_ if not n.endswith("_align")]
HIT_FLOAT_FIELDS = [n for n, t in tnt_hit._fields_ if t is C.c_float]
BIND_FLOAT_FIELDS = ["tm", "dH", "dS", "dg"]
