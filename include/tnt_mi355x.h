/*
 * tnt_mi355x.h -- C ABI of the MI355X-native ThermonucleotideBLAST hot path.
 *
 * The reference (jgans/thermonucleotideBLAST v2.77) has no plugin/FFI seam; its
 * per-target inner loop is the body of the `while(true)` work loop inside
 * `#pragma omp parallel` in tntblast_local.cpp:388-793, which calls the three
 * C++ entry points declared in tntblast.h:409-472 (amplicon / padlock / hybrid)
 * once per (target, assay).  This header is the batch-granular replacement for
 * that loop body: a whole assay set against a device-resident batch of targets.
 *
 * Everything here is plain C: pointers, sizes, PODs.  No C++/torch types cross
 * the boundary.  All functions return 0 on success and a negative TNT_E_* code
 * on failure; tnt_last_error() returns the message (the reference throws
 * `const char*` literals, throw.h:17-18, caught in tntblast_local.cpp:1377-1391).
 *
 * The same structs (and the same function signatures with the prefixes
 * `tntor_` / `tntref_`) are implemented by the CPU restatement under oracle/
 * and by the harness that links the reference's own objects (oracle/_ref), so
 * parity tests can diff the three implementations field by field.
 */
#ifndef TNT_MI355X_H
#define TNT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNT_ABI_VERSION 7

/* Longest oligo (the reference's MAX_SEQUENCE_LENGTH, nuc_cruc.h:80).  Up to 64 bases the rows of the alignment live
 * in registers, one window per lane; 65..128 run through an instance that keeps them in scratch memory; 129..1024 are
 * bound by the wavefront-per-window kernel (query rows across the lanes), in every role (a context with a primer, TaqMan
 * probe or padlock arm of that length searches every target by the route of the targets that are seeded in windows).   */
#define TNT_MAX_OLIGO 1024
/* NUM_FLANK_BASE, tntblast.h:76 */
#define TNT_FLANK 4
/* window = oligo + 2*flank (bind_oligo.cpp:139-142) */
#define TNT_MAX_WINDOW (TNT_MAX_OLIGO + 2 * TNT_FLANK)
/* alignment text: 3 lines of "5' " + <=(L+W) columns + " 3'" (nuc_cruc_output.cpp:8-213).  The fixed fields hold
 * every alignment of a 64-mer.  A longer text is kept by the context (until its next search / bind call) and the
 * field holds a reference to it: read every alignment field through tnt_long_text().                              */
#define TNT_ALIGN_TEXT 448

/* error codes */
#define TNT_OK 0
#define TNT_E_ARG (-1)      /* bad argument */
#define TNT_E_DEVICE (-2)   /* no GPU / HIP failure: there is NO CPU fallback */
#define TNT_E_LIMIT (-3)    /* oligo longer than TNT_MAX_OLIGO, overflow of a device list */
#define TNT_E_UNSUPPORTED (-4)

/* assay formats, numerically equal to hybrid_sig.h:19 */
enum { TNT_ASSAY_PCR = 0, TNT_ASSAY_PROBE = 1, TNT_ASSAY_PADLOCK = 2, TNT_ASSAY_MIPS = 3 };
/* target strands, numerically equal to seq.h:38-42 */
enum { TNT_STRAND_PLUS = 1, TNT_STRAND_MINUS = 2, TNT_STRAND_BOTH = 3 };
/* hybrid_sig::PLUS / MINUS, hybrid_sig.h:117 */
enum { TNT_PLUS = 0, TNT_MINUS = 1 };

/* 1-byte target base codes, seq.h:12-35 (input format of the path) */
enum {
	TNT_DB_A = 0, TNT_DB_C, TNT_DB_G, TNT_DB_T, TNT_DB_I, TNT_DB_M, TNT_DB_R, TNT_DB_S, TNT_DB_V,
	TNT_DB_W, TNT_DB_Y, TNT_DB_H, TNT_DB_K, TNT_DB_D, TNT_DB_B, TNT_DB_N, TNT_DB_GAP, TNT_DB_UNKNOWN
};

/* The Options fields the hot path consumes (options.h:33-76, defaults options.h:85-170,
 * tntblast.h:19-76); used at tntblast_local.cpp:232-234,363-367,566-624.               */
typedef struct tnt_options {
	float min_primer_tm, max_primer_tm;     /* -e / -x */
	float min_primer_dg, max_primer_dg;     /* -z / -g */
	float min_probe_tm, max_probe_tm;       /* -E / -X */
	float min_probe_dg, max_probe_dg;       /* -Z / -G */
	float salt;                             /* -s  [Na+] (M) */
	float primer_strand;                    /* -t  Ct primers */
	float probe_strand;                     /* -T  Ct probe */
	float asymmetric_strand_ratio;          /* -y  forward = ratio * primer_strand */
	float target_t;                         /* --temperature (K) */
	int32_t primer_clamp;                   /* --primer-clamp (3') */
	int32_t min_max_primer_clamp;           /* --min-max-primer-clamp, -1 = off */
	int32_t probe_clamp_5, probe_clamp_3;   /* --probe-clamp5/3 */
	int32_t max_gap, max_mismatch, max_poly_degen;
	int32_t max_len;                        /* -l */
	int32_t assay_format;                   /* -A */
	int32_t target_strand;                  /* --target-strand */
	int32_t hash_word_size;                 /* -W (3..8) */
	int32_t single_primer_pcr;              /* --single-primer-pcr */
	int32_t allow_dangle_5, allow_dangle_3; /* --dangle5/3 */
	int32_t use_dinkelbach;                 /* --dinkelbach */
} tnt_options;

/* One expanded (non-degenerate) assay: expand_degenerate_signatures, tntblast_util.cpp:579-713.
 * Absent oligos are NULL.  *_degen are the Ct divisors (amplicon_search.cpp:85-87).       */
typedef struct tnt_assay {
	int32_t id;
	int32_t degen_id;
	const char *forward;
	const char *reverse;
	const char *probe;
	int32_t forward_degen, reverse_degen, probe_degen;
} tnt_assay;

/* Result payload of one hit = hybrid_sig (hybrid_sig.h:121-164) without the host-side
 * string-table indices; alignment strings are carried inline.                          */
typedef struct tnt_hit {
	int32_t assay_id, degen_id, seq_index;
	int32_t amplicon_start, amplicon_stop;  /* amplicon_range */
	int32_t probe_start, probe_stop;        /* probe_range */
	float forward_tm, reverse_tm, probe_tm;
	float forward_dH, forward_dS, reverse_dH, reverse_dS, probe_dH, probe_dS;
	int32_t primer_strand, probe_strand;    /* TNT_PLUS / TNT_MINUS */
	int32_t forward_primer_clamp, reverse_primer_clamp;
	int32_t forward_mm, forward_gap, reverse_mm, reverse_gap, probe_mm, probe_gap;
	/* which oligo of the assay sits in the forward / reverse slot (0 = forward oligo,
	 * 1 = reverse oligo): single-primer amplicons swap identities, amplicon_search.cpp:453-463 */
	int32_t forward_oligo, reverse_oligo;
	int32_t has_primers, has_probe;
	char forward_align[TNT_ALIGN_TEXT];
	char reverse_align[TNT_ALIGN_TEXT];
	char probe_align[TNT_ALIGN_TEXT];
} tnt_hit;

/* One seed after the per-(oligo,strand) diagonal de-duplication of
 * match_oligo_to_{minus,plus}_strand (bind_oligo.cpp:84-122).                      */
typedef struct tnt_seed {
	int32_t seq_index;
	int32_t oligo;      /* index into the context's unique-oligo table */
	int32_t strand;     /* TNT_MINUS: DNAHash::find, TNT_PLUS: find_complement */
	int32_t query_loc;  /* DNAHash_iterator::offset() = index in the word list (seq_hash.h:214) */
	int32_t target_loc; /* k-mer start in the target */
} tnt_seed;

/* Outcome of one oligo/window binding = what bind_oligo_to_{minus,plus}_strand derives from
 * NucCruc for one seed (bind_oligo.cpp:502-748).                                            */
typedef struct tnt_bind {
	int32_t status;     /* 0 bound; 1 tm, 2 dg, 3 anchor5, 4 anchor3, 5 mismatch, 6 gap, 7 poly-degen rejected */
	int32_t target_start, target_stop; /* window [start, stop) */
	float tm, dH, dS, dg;
	int32_t anchor5, anchor3, num_mm, num_gap, poly_degen;
	int32_t loc5, loc3;
	int32_t q_first, q_last, t_first, t_last; /* alignment_range, nuc_cruc_anchor.cpp:379-396 */
	int32_t valid;      /* curr_align.valid */
	int32_t dp_max;     /* align_dimer return value */
	int32_t n_max;      /* number of maximal cells */
	char align[TNT_ALIGN_TEXT];
} tnt_bind;

/* explicit bind request (unit-level entry to the bind kernel) */
typedef struct tnt_bind_req {
	int32_t seq_index;
	int32_t oligo;      /* unique-oligo index */
	int32_t strand;     /* TNT_MINUS / TNT_PLUS */
	int32_t query_loc, target_loc;
	int32_t role;       /* 0 primer thresholds, 1 probe thresholds */
	float ct;           /* strand concentration */
} tnt_bind_req;

typedef struct tnt_ctx tnt_ctx; /* one per GPU, single caller */
typedef struct tnt_db tnt_db;   /* device-resident packed targets */

void tnt_options_default(tnt_options *opt);
int tnt_abi_version(void);

/* Context = options + assay set + thermodynamic tables resident on `device`. */
int tnt_ctx_create(const tnt_options *opt, const tnt_assay *assays, size_t n_assays, int device, tnt_ctx **out);
void tnt_ctx_destroy(tnt_ctx *ctx);
const char *tnt_last_error(const tnt_ctx *ctx); /* ctx may be NULL: last creation error */
/* Warnings about the context's options, "" when there are none (ABI v6).  One today: thresholds in the regime of SURVEY
 * Q11 -- a class of oligos whose thresholds accept Tm = 0 and dG = 0.  For a window without any valid alignment the reference
 * evaluates the first / last match its NucCruc object kept from the previous valid window (alignment::clear() leaves them,
 * nuc_cruc.h:360-371,386-387) and may report it as a hit with Tm = 0 (bind_oligo.cpp:600-624); which ones depends on the order
 * its threads met the windows.  This library reports such windows as not bound: under these options it can list fewer hits. */
const char *tnt_ctx_warnings(const tnt_ctx *ctx);
/* Where the lists of the targets that are seeded in windows (records beyond one seed work unit: genomes, chromosomes) were
 * sequenced and paired during the context's last search (ABI v7) -- per (target, assay) list the bind / cull sequence and the
 * pairing loops of amplicon() (amplicon_search.cpp:131-674), or the list sorts and pairing of hybrid() / padlock()
 * (probe_search.cpp:67-230, padlock_search.cpp:62-361): out[0] = lists done on the device, out[1] = PCR lists the device handed
 * back to the host threads because sort_by_oligo_loc (amplicon_search.cpp:12-26) is not a strict weak order on their entries,
 * out[2] = lists that were the host threads' from the start (assays under the BindCache replay, the segments of probes beyond
 * 128 bases, contexts with primers beyond 128 bases), out[3] = hits the device reported.  All zero when the search met no such
 * target. */
int tnt_long_pair_stats(const tnt_ctx *ctx, uint64_t out[4]);

/* unique-oligo table of the context (assay oligos de-duplicated by string) */
int tnt_ctx_num_oligos(const tnt_ctx *ctx);
const char *tnt_ctx_oligo(const tnt_ctx *ctx, int i);

/* NucCruc::update_dp_param result (nuc_cruc.cpp:340-487): 2401 int32 */
int tnt_ctx_delta_g(const tnt_ctx *ctx, int32_t *out2401);

/* Pack a batch of targets onto the GPU.  `codes` = concatenated 1-byte codes (seq.h:12-35),
 * `offsets[n_targets+1]` = start of each target.  Replaces read_bio_seq + DNAHash::hash
 * (tntblast_local.cpp:510,534).  `first_seq_index` = global index of target 0.            */
int tnt_db_create(tnt_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, size_t n_targets,
                  uint32_t first_seq_index, tnt_db **out);
void tnt_db_destroy(tnt_db *db);
/* Leave targets out of every later search of this database (skip[i] != 0; targets from n on are searched): the reference
 * never searches a record longer than --max-target-len whole, only its fragments (tntblast_local.cpp:282-289,438-468),
 * so a front end marks such records here and searches their fragments as a database of their own.               */
int tnt_db_skip_targets(tnt_ctx *ctx, tnt_db *db, const uint8_t *skip, size_t n);
size_t tnt_db_num_targets(const tnt_db *db);
uint64_t tnt_db_num_bases(const tnt_db *db);

/* Target ingest: FASTA text -> device-resident packed targets, parsed on the GPU.  Replaces
 * sequence_data::load_fasta + read_bio_seq_fasta_slow (sequence_data_fastx.cpp:13-79,187-382) for a whole file:
 * a record starts at the first '>' of a line; the defline runs to the first '\n' or '\r' (leading white space
 * dropped); white space, '*' and '-' are dropped from the sequence, the rest goes through ascii_to_hash_base
 * (seq.h:148-189: case-insensitive, U -> T, IUPAC kept, anything else DB_UNKNOWN).  Target i of the database is
 * record i of the text (seq_index = first_seq_index + i).  `text` is borrowed for the call.
 * Text whose first non-space character is '@' is FASTQ (file_type, annotation.cpp:296-374) and read like load_fastq +
 * read_bio_seq_fastq_slow do (sequence_data_fastx.cpp:81-186,384-583): '@' at the start of a line opens a record unless
 * the previous line was the '+' line; the sequence is the single line after the defline.                  */
int tnt_db_create_from_fasta(tnt_ctx *ctx, const char *text, size_t n_bytes, uint32_t first_seq_index, tnt_db **out);
/* Same from a file; gzip-compressed files are inflated with zlib like the reference does (gzopen/gzread). */
int tnt_db_load_fasta(tnt_ctx *ctx, const char *path, uint32_t first_seq_index, tnt_db **out);
/* Streaming form for databases larger than one wants resident: batches of whole records, about `max_bytes` of text
 * each (at least one record), read and inflated by a background thread while the caller searches the previous batch.
 * seq_index keeps counting across batches.  *out == NULL at the end of the file.                              */
typedef struct tnt_fasta_reader tnt_fasta_reader;
int tnt_fasta_open(const char *path, tnt_fasta_reader **out);
int tnt_fasta_next(tnt_fasta_reader *reader, tnt_ctx *ctx, size_t max_bytes, tnt_db **out);
void tnt_fasta_close(tnt_fasta_reader *reader);
/* Defline of target i (databases made from FASTA text; empty otherwise).  Owned by the database. */
int tnt_db_defline(const tnt_db *db, size_t i, const char **defline, size_t *len);
/* Length of target i in bases and the size of its record in the file (defline, line ends and all): the "approximate
 * length" the reference cuts long targets by (sequence_data_fastx.cpp:63-76, tntblast_local.cpp:282-289); equal to the
 * number of bases for databases packed from codes.                                                                     */
int tnt_db_target_info(const tnt_db *db, size_t i, uint32_t *n_bases, uint64_t *record_bytes);
/* 1-byte codes (seq.h:12-35) of target i, rebuilt from the packed rows; returns the length or a TNT_E_* code. */
int64_t tnt_db_target_codes(tnt_ctx *ctx, tnt_db *db, size_t i, uint8_t *codes, size_t cap);
/* Where the time of the last ingest went (ms) and what it moved. */
typedef struct tnt_ingest_timing {
	float read_ms;    /* file read / inflate (tnt_db_load_fasta only) */
	float index_ms;   /* host: record boundaries and deflines */
	float h2d_ms;     /* text -> device */
	float count_ms;   /* device: bases and exceptions per record */
	float layout_ms;  /* host: row offsets (device -> host lengths, prefix sums, host -> device) */
	float pack_ms;    /* device: 2-bit rows + exception lists */
	uint64_t text_bytes, bases, exceptions;
} tnt_ingest_timing;
int tnt_db_ingest_timing(const tnt_db *db, tnt_ingest_timing *out);

/* The hot path: seeds -> cull -> bind -> pair for every (target, assay).  Hits are owned by
 * the context until the next tnt_search / tnt_ctx_destroy.                                  */
int tnt_search(tnt_ctx *ctx, tnt_db *db, const tnt_hit **hits, size_t *n_hits);

/* Asynchronous form used by bench.py: enqueue on the context's stream, no host sync.  tnt_search_collect with
 * hits == NULL leaves the records on the device (tnt_search_hits_device / _compact, tnt_gang_gather take them from
 * there) instead of copying sizeof(tnt_hit) = 1464 bytes per hit to the host.                                       */
int tnt_search_enqueue(tnt_ctx *ctx, tnt_db *db);
int tnt_search_collect(tnt_ctx *ctx, const tnt_hit **hits, size_t *n_hits);
/* Collect, and hand the hits to the host in the compact form of tnt_search_hits_compact (about 290 bytes per hit;
 * tnt_hits_unpack is the inverse): the blob sits in page-locked memory of the context until its next search.      */
int tnt_search_collect_compact(tnt_ctx *ctx, const void **host_blob, size_t *n_bytes, size_t *n_hits);

/* The hits of the last collected search where the pair kernel left them, in device memory (same records, same order as
 * the host copy): what a multi-GPU driver hands to RCCL without a round trip through the host (the reference's master
 * receives packed hit lists from its MPI workers, tntblast_master.cpp:811-828).  Valid until the next search on this
 * context.  (The hits of targets that are seeded in windows are paired on host threads and copied behind the device's own.) */
int tnt_search_hits_device(tnt_ctx *ctx, const void **device_hits, size_t *n_hits);

/* The same hits in compact form, packed on the device: what the gather of a multi-GPU run sends (about 350 bytes per hit
 * instead of sizeof(tnt_hit) = 1464; the reference's workers send packed hit lists too, tntblast_master.cpp:811-828).
 *   blob   = { uint32 n_hits, uint32 n_bytes } record*
 *   record = the 120 bytes of tnt_hit before its texts | uint16 length[3] | uint16 0 | the three texts | pad to 4 bytes
 * Valid until the next search on this context.  tnt_hits_unpack() is the host-side inverse (no GPU involved): with
 * out == NULL it only reports the count.                                                                           */
int tnt_search_hits_compact(tnt_ctx *ctx, const void **device_blob, size_t *n_bytes, size_t *n_hits);
int tnt_hits_unpack(const void *blob, size_t n_bytes, tnt_hit *out, size_t cap, size_t *n_hits);

/* ---- several GPUs in one process (ABI v5).  The reference's master deals targets to its MPI workers and receives their
 * packed hit lists (tntblast_master.cpp:438-461,811-828; tntblast_worker.cpp:173-186,602).  Here: one context per
 * device (tnt_ctx_create(..., device, ...)), one caller thread per context while they search, targets dealt out by the
 * caller (they are independent; seq_index is global through first_seq_index), and tnt_gang_gather to bring the hits of
 * the searches they collected last to the host: the compact blobs travel to the first context's device over RCCL
 * (ncclSend / ncclRecv in one group, an all-reduce of the hit counts as the check) and cross to the host in one copy;
 * where RCCL cannot serve (absent, or two contexts on one device) every device copies its own blob ("host" transport,
 * same bytes).  TNT_MI355X_GATHER=host|rccl forces one.                                                              */
int tnt_device_count(void); /* visible HIP devices (0: none) */
typedef struct tnt_gang tnt_gang;
int tnt_gang_create(tnt_ctx *const *ctx, int n, tnt_gang **out);
void tnt_gang_destroy(tnt_gang *gang);
const char *tnt_gang_transport(const tnt_gang *gang);  /* "rccl", or "host" / "host (<why not rccl>)" */
const char *tnt_gang_last_error(const tnt_gang *gang); /* gang may be NULL: last creation error */
/* For every rank r (has == NULL or has[r] != 0) whose context holds a collected search: blobs[r] -> its compact hit blob
 * in host memory owned by the gang (valid until the next gather), n_bytes[r], n_hits[r]; other ranks get NULL / 0.    */
int tnt_gang_gather(tnt_gang *gang, const uint8_t *has, const void **blobs, size_t *n_bytes, size_t *n_hits);

/* per-stage device timings of the last search, in ms (HIP events on the context stream) */
typedef struct tnt_timing {
	float seed_ms, bind_ms, pair_ms, total_ms;
	uint64_t n_seeds, n_binds, n_bound, n_hits;
	uint64_t n_filtered; /* of n_binds: windows the pre-filter rejected before the full bind kernel (their Tm cannot reach the minimum) */
	uint64_t n_cells;    /* DP cells of the n_binds windows: sum of oligo length x window width (align_dimer fills three states per cell,
	                        nuc_cruc.cpp:492-696): the algorithmic work of the bind stage (ABI v6) */
} tnt_timing;
int tnt_last_timing(tnt_ctx *ctx, tnt_timing *out);

/* Unit-level entries (parity tests of SURVEY 8a rows a1-a3 and a5-a14). */
int tnt_seeds(tnt_ctx *ctx, tnt_db *db, const tnt_seed **seeds, size_t *n_seeds);
/* Unit-level view of row a4: the seeds of every (target, PCR assay) that survive amplicon()'s early exits and the first
 * cull_oligo_match (amplicon_search.cpp:96-131, :679-765) = the work items the bind stage is handed.  Rows of seven ints:
 * seq_index, assay (position in the context's assay list), role (0 forward, 1 reverse, 2 probe), strand, query_loc,
 * target_loc, and 1 when the minus strand is bound first (:131).  The array belongs to the context (valid until the next
 * call).  Targets that are seeded in windows: their lists are assembled and culled whole on the device (same rows). */
int tnt_first_cull(tnt_ctx *ctx, tnt_db *db, const int32_t **rows, size_t *n_rows);
int tnt_bind_batch(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, tnt_bind *out);
/* Measurement entry: the scan of the seed stage on its own (DNAHash::hash + the word look-ups of find / find_complement,
 * seq_hash.h:524-642,749-779, with nothing behind them): every k-mer of every whole-target work unit is looked up in the
 * inverted query-word table of each assay chunk and the candidates are counted.  `repeats` timed passes behind one
 * warm-up; *ms_per_pass = HIP-event time of one pass over the database (all chunks), *n_candidates = raw seeds of one pass
 * (before the diagonal de-duplication).  Streams 0.25 bytes per target base and chunk: the stage SURVEY 8(d) prices against
 * the HBM roofline.                                                                                                      */
int tnt_seed_scan_only(tnt_ctx *ctx, tnt_db *db, int repeats, float *ms_per_pass, uint64_t *n_candidates);
/* The same requests through the kernels tnt_search() runs (bind_oligo.cpp:502-748 for each): TNT_ROUTE_BUCKETS splits
 * the queue by oligo length like a search phase does, so every request is bound by the throughput instance of its
 * length (exact / plain ladder; the general ladder for windows with degenerate target bases); TNT_ROUTE_MERGED binds
 * the whole queue with the longest general instance (what a search does with short queues).  These instances keep a
 * status for every request and the record + alignment text of bound ones only: status is always set; tm, dH, dS, dg,
 * anchors, num_mm, num_gap, loc5, loc3, align are set when status == 0; the other fields are -1 (0 for the window).
 * TNT_ROUTE_WAVE is TNT_ROUTE_BUCKETS with every bucket bound by the wavefront-per-window form of the kernel (one wavefront
 * per request, query rows across the lanes, trace words in LDS); TNT_ROUTE_WAVE_UNIT is the unit-level form of that
 * kernel: every field set, like TNT_ROUTE_UNIT. */
enum { TNT_ROUTE_UNIT = 0, TNT_ROUTE_BUCKETS = 1, TNT_ROUTE_MERGED = 2, TNT_ROUTE_WAVE = 3, TNT_ROUTE_WAVE_UNIT = 4 };
int tnt_bind_batch_routed(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, tnt_bind *out, int route);

/* Unit-level entry to the bind stage's pre-filter (tnt_filter.hip.h): for every request score[2k] = an upper bound, in
 * 1e-3 kcal/mol (saturating at 32 767), of -(dH - Tk dS) + constant over all alignments of the window at Tk = minimum Tm
 * of the request's threshold set, score[2k+1] = the value it must reach for the window to go on to the full bind kernel
 * (INT32_MIN: the filter is off for this threshold set).  IUPAC codes in the window count as "any of A C G T"; a window
 * with inosine or a skipped code gives INT32_MAX / INT32_MIN (never rejected); oligos beyond 64 bases have no filter
 * instance: -1 / -1.  bind_oligo_to_*_strand rejects what stays below on Tm (bind_oligo.cpp:600-606).           */
int tnt_filter_bound(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, int32_t *score);
/* The pre-filter's SECOND tier for the same requests (ABI v6; tnt_sbound.hip.h): a trace-free fill of align_dimer's recurrence
 * (nuc_cruc.cpp:492-696) whose states carry, next to the score, the credit the first tier's argument grants the path: score[2 i]
 * = the credit of the maximal-score paths in 1e-3 kcal/mol (INT32_MAX: above 20.48 kcal/mol, or a window this tier does not
 * serve -- IUPAC codes in the window, oligos beyond 40 bases), score[2 i + 1] = the first tier's threshold (INT32_MIN: not
 * served / class never rejected).  A window is rejected iff score[2 i] < score[2 i + 1]. */
int tnt_filter_bound_tier2(tnt_ctx *ctx, tnt_db *db, const tnt_bind_req *req, size_t n, int32_t *score);

/* Oligo self-structure printed with every hit (tntblast_local.cpp:656-686): hairpin and homodimer Tm of oligo
 * `a` at strand(ct_a, ct_a), and, when `b` is given, the heterodimer Tm of query a / target b at
 * strand(ct_a, ct_b) (NucCruc::approximate_tm_hairpin / _homodimer / _heterodimer, nuc_cruc.cpp:2397-2616).
 * Depends on the oligos and options only; computed on the GPU by the unit-level bind kernel instance.   */
int tnt_oligo_structure(tnt_ctx *ctx, const char *a, const char *b, float ct_a, float ct_b, float *hairpin_a,
                        float *homodimer_a, float *heterodimer_ab);

/* The text of an alignment field (tnt_hit::*_align, tnt_bind::align): the field itself, or -- for an alignment of an oligo
 * beyond 64 bases that does not fit TNT_ALIGN_TEXT -- the full text the context keeps since its last search / bind call
 * (the field then holds a 17-character reference that starts with 0x7F).  Returns the length written (without NUL), or < 0. */
int tnt_long_text(tnt_ctx *ctx, const char *field, char *buf, size_t cap);

/* amplicon / probe text of a hit, copied from the packed targets (amplicon_search.cpp:508-537,
 * probe_search.cpp:138-152): returns length written (without NUL) or <0.                      */
int tnt_hit_sequence(tnt_ctx *ctx, tnt_db *db, const tnt_hit *hit, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TNT_MI355X_H */
