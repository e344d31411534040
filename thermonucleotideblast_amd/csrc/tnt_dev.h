// tnt_dev.h -- structures shared by the host side (tnt_host.hip) and the gfx950 kernels
// (tnt_kernels.hip).  Product code: must not include anything from oracle/.
#ifndef TNT_DEV_H
#define TNT_DEV_H

#include <stdint.h>

namespace tnt {

// BASE::nucleic_acid, nuc_cruc.h:181-188
enum : int { bA = 0, bC, bG, bT, bI, bE, bGAP, bM, bR, bS, bV, bW, bY, bH, bK, bD, bB, bN };
constexpr int NB = 7;    // NUM_BASE
constexpr int NBP = 49;  // NUM_BASE_PAIR
constexpr int P_AT = bA * NB + bT, P_TA = bT * NB + bA, P_CG = bC * NB + bG, P_GC = bG * NB + bC;
constexpr int P_GT = bG * NB + bT, P_TG = bT * NB + bG, P_EE = bE * NB + bE, P__ = bGAP * NB + bGAP;

constexpr int MAX_OLIGO = 1024;           // TNT_MAX_OLIGO = the reference's MAX_SEQUENCE_LENGTH (nuc_cruc.h:80): stride of the oligo tables
constexpr int MAX_LANE_OLIGO = 128;       // longest oligo the lane-per-window kernels (bind_kernel, filter_kernel) hold; beyond: wbind_kernel
constexpr int FLANK = 4;                  // NUM_FLANK_BASE, tntblast.h:76
constexpr int MAX_WINDOW = MAX_LANE_OLIGO + 2 * FLANK;
constexpr int MAX_COLS = MAX_LANE_OLIGO + MAX_WINDOW + 4; // longest traceback path + dangles
constexpr int ALIGN_TEXT = 448;           // TNT_ALIGN_TEXT

// trace word of one DP cell: one byte, see the bind kernel.  Predecessor states follow nuc_cruc.h:62-64
// (im1_jm1 = came from M, im1_j = from I_target, i_jm1 = from I_query).  Every "came from" bit is stored
// inverted and taken against the state clamped at zero (bit = candidate < max(state, 0): one subtract + one
// sign-bit shift), so a negative state shows as ALL of its bits set -- which a maximum of the candidates
// cannot produce -- and needs no bit of its own.
constexpr int TW_M_LE0 = 1 << 0;      // M <= 0 (with TW_M_NEG not matched: M == 0)
constexpr int TW_IT_GT = 1 << 1;      // I_target: extend < insert  (trace = from M only)
constexpr int TW_IT_LT = 1 << 2;      // I_target: insert < extend  (trace = from I_target only)
constexpr int TW_IQ_GT = 1 << 3;      // I_query:  extend < insert
constexpr int TW_IQ_LT = 1 << 4;      // I_query:  insert < extend
constexpr int TW_M_NOT_IQ = 1 << 5;   // clear: M_trace & i_jm1
constexpr int TW_M_NOT_IT = 1 << 6;   // clear: M_trace & im1_j
constexpr int TW_M_NOT_M = 1 << 7;    // clear: M_trace & im1_jm1
// state < 0  <=>  (tw & TW_x_NEG) == TW_x_NEG
constexpr int TW_IT_NEG = TW_IT_GT | TW_IT_LT;
constexpr int TW_IQ_NEG = TW_IQ_GT | TW_IQ_LT;
constexpr int TW_M_NEG = TW_M_NOT_IQ | TW_M_NOT_IT | TW_M_NOT_M;

// Thermodynamic tables resident in HBM (19 KB floats + 9.6 KB ints); delta_g is staged into LDS.
struct Tables {
	int32_t dg[NBP * NBP];  // update_dp_param, nuc_cruc.cpp:340-487
	float H[NBP * NBP];     // param_H (== loop terminal H, nuc_cruc_santa_lucia.cpp:591-595)
	float S[NBP * NBP];
	uint8_t hs_map[NBP * NBP + 3]; // H[i], S[i] == hs_H[hs_map[i]], hs_S[hs_map[i]]: the pairs are few, the LDS copy is small
	float hs_H[256], hs_S[256];
	float loop_S[513];
	float bulge_S[513];
	float hairpin_S[513];
	uint32_t special_key[136]; // special hairpin loops: length << 24 | 3 bits per base (A C G T), 5' closing base first
	float special_H[136], special_S[136];
	int32_t n_special;
	float sym_S;            // param_symmetry_S (homodimers)
	uint8_t bbp[18 * 18];   // best_base_pair LUT, nuc_cruc.cpp:14-213 (row = first argument)
	uint8_t pad0[4];
	uint32_t rt[18];        // rt[x] nibble q = resolve_degenerate(x | q) for partner q in A,C,G,T,I,E,GAP
	int32_t term[NBP];      // class constants of delta_g (see the bind kernel): term[c] = g((x,GAP), c) for real pairs c,
	                        // term[(GAP,x)] = g((GAP,x),(y,GAP)) (forbidden)
	int32_t dg_const[4];    // BULGE, g(__,(x,GAP)), g(__, real pair), g((GAP,x),(y,GAP))
	// --dinkelbach re-derives delta_g at another temperature per window and pass: every entry as the triple it was made
	// from, dg = max(int((dk_H - T * dk_S) * 10000), dk_F)  (update_dp_param, nuc_cruc.cpp:357-486; dk_S holds S + the
	// entry's salt term, dk_F = 0 for the clamped classes and INT32_MIN otherwise)
	float dk_H[NBP * NBP], dk_S[NBP * NBP];
	int32_t dk_F[NBP * NBP];
	int32_t term_src[NBP];  // term[c] == dg[term_src[c]], dg_const[k] == dg[const_src[k]]
	int32_t const_src[4];
	float init_H, init_S, at_H, at_S, salt, asym_S, bulge_at_S;
	float temperature;      // target_T
	float log_na;           // logf([Na+]) computed on the host (no device logf: bit-exactness)
	int32_t dangle5, dangle3;
	int32_t word_size;      // hash word length k
};

// threshold cascade of bind_oligo.cpp:600-714
struct Thresh {
	float min_tm, max_tm, min_dg, max_dg;
	int32_t clamp5, clamp3, max_gap, max_mm, max_poly;
	int32_t pad[3];
};

// one (assay, role) = one oligo with its strand concentration and threshold set
struct Slot {
	int32_t oligo;   // unique-oligo index
	float log_ct;    // logf(strand concentration), host computed
	int32_t thresh;  // index into Thresh table
	int32_t assay;   // expanded assay index (or -1)
	int32_t bucket;  // length bucket = which bind kernel instance handles this oligo ...
	int32_t bucket_general; // ... and the one for windows that hold a degenerate target base
	int32_t qloc_bias; // added to the 7-bit word index of a work item: the seeds of an oligo of more than 134 bases come from
	                   // segments of 128 query words each (a segment is a slot of its own), and unit-level requests name any word
	int32_t pad;
};

// Instances of the bind kernel (rows kept in registers): every oligo length up to 40 has its own (probes of 33 to 40 bases
// since round 4: config 5), so the oligos of one launch all have exactly LMAX rows; coarser above.
// Exact instances (every oligo of the launch has exactly LMAX rows: no per-row length test; plain windows only) ...
#define TNT_EXACT_LADDER(X) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40)
// ... and a ladder (L <= LMAX) in two flavours: plain windows (A/C/G/T targets) for the other lengths, and
// general ones for windows with degenerate target bases and for short queues that are bound in one launch
#define TNT_LMAX_LADDER(X) X(16) X(20) X(24) X(28) X(32) X(36) X(40) X(44) X(48) X(52) X(56) X(60) X(64)
// oligos of 65 to 128 bases (long hybridisation probes, MIPS arms): the same two flavours, rows mostly in scratch memory;
// they meet the pre-filter (128-row instances of the F / D kinds) when the context's calibration says it pays
#define TNT_LONG_LADDER(X) X(128)
// instances of the pre-filter kernel (tnt_filter.hip.h): the row counts of both ladders above
#define TNT_FILTER_LADDER(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(44) X(48) X(52) X(56) X(60) X(64)
constexpr int MAX_BUCKETS = 48;
constexpr int BUCKET_EXACT = 0x100, BUCKET_PLAIN = 0x200; // flags on the instance id kept by the host (lmax | flags)
constexpr int BUCKET_WAVE = 0x400; // the wavefront-per-window kernel serves the bucket: the low byte holds the query rows per lane (4, 8, 16: oligos of up to 256, 512, 1 024 bases)
constexpr int bucket_rows(int id) { return (id & BUCKET_WAVE) ? 64 * (id & 0xFF) : (id & 0xFF); }

// Workgroup size of the bind kernel instances.  One lane owns one DP; a lane needs 8 * LMAX + 48 bytes of LDS
// (window, query, alignment columns, best alignment, trace prefetch) next to the 24 KB of tables every workgroup
// holds once.  Up to 22 rows the instance fits 168 VGPRs and one 768-thread workgroup fits a CU's LDS: three
// wavefronts per SIMD, the third one fills the issue slots the other two leave while they wait in their serial
// tails (+4 % on those buckets).  Longer oligos need the registers of a two-wave budget (two 256-thread workgroups
// per CU); forcing them into 168 VGPRs spills and was measured slower.
// Workgroup shape of a bind instance: 768 threads (three wavefronts per SIMD, one workgroup per CU) where the fill fits
// 168 VGPRs without spilling -- up to 22 rows for any instance, up to 28 for the plain ones -- else 256 threads.
constexpr bool bind_wide(int lmax, bool plain) { return lmax <= 22 || (plain && lmax <= 28); }
constexpr int bind_threads_for(int lmax, bool plain) { return bind_wide(lmax, plain) ? 768 : 256; }
constexpr int bind_blocks_per_cu(int lmax, bool plain) { return (bind_wide(lmax, plain) || lmax > 64) ? 2 : 4; } // persistent grid: 2x the resident workgroups
constexpr int BUCKET_STRIDE = 32; // words between two bucket counters (one 128-byte line each)

// bind work item (one seed that survived the first cull)
struct Item {
	uint32_t target;  // target index inside the packed batch
	uint32_t tloc;    // seed target_loc
	uint32_t info;    // bits 0-6 query_loc (word index), bit 7 strand (1 = minus), bit 8 phase B, bits 9-31 Slot index
	uint32_t group;   // physical index of its Group (marked alive when the item binds)
};
constexpr uint32_t ITEM_MINUS = 1u << 7;
constexpr uint32_t ITEM_PHASE_B = 1u << 8; // bound only if a phase-A item of its group bound (second strand / probe)
constexpr int ITEM_SLOT_SHIFT = 9;

constexpr int LTEXT_REF_CHARS = 17; // a text field that refers to the long-text arena: 0x7F 'L' 'T', 8 hex digits of offset, 6 of length

// result of one bind; pool record only exists for bound items
struct BoundRec {
	int32_t loc5, loc3;
	float tm, dH, dS;
	int32_t a5, a3, mm, gap;
	int32_t text_len;   // strlen(text)
	int32_t pad[2];
	char text[ALIGN_TEXT];
};

// full per-item record (unit-level API tnt_bind_batch), mirrors tnt_bind
struct BindFull {
	int32_t status, target_start, target_stop;
	float tm, dH, dS, dg;
	int32_t a5, a3, mm, gap, poly;
	int32_t loc5, loc3;
	int32_t q_first, q_last, t_first, t_last;
	int32_t valid, dp_max, n_max;
	char text[ALIGN_TEXT];
};

// per (target, assay) group of items emitted by the seed kernel in first-cull order
struct Group {
	uint32_t target;  // seed work unit (a target or one window of it): the pair kernel looks the target up through it
	uint32_t assay;   // expanded assay index | GROUP_* flags
	uint32_t first;   // first item
	uint32_t count;   // | GROUP_ALIVE once one of its items bound
};
constexpr uint32_t GROUP_ALIVE = 1u << 31;
constexpr uint32_t GROUP_MINUS_FIRST = 1u << 31; // bind the minus strand first (amplicon_search.cpp:131)
constexpr uint32_t GROUP_ASSAY_MASK = (1u << 24) - 1u;

// packed targets in HBM
struct DbView {
	const uint32_t *packed;     // 2 bits/base (code & 3), 16 bases per word, each target starts on a 16-byte boundary
	const uint64_t *word_off;   // [n+1] first word of each target
	const uint32_t *len;        // [n] bases
	const uint32_t *exc_off;    // [n+1] range in the exception lists (codes > 3)
	const uint32_t *exc_pos;    // position inside the target, ascending
	const uint8_t *exc_code;    // seq.h code (4..17)
	uint32_t n_targets;
	uint32_t first_seq_index;
};

// mirrors tnt_hit (include/tnt_mi355x.h); written by the pairing kernels
struct Hit {
	int32_t assay_id, degen_id, seq_index;
	int32_t amplicon_start, amplicon_stop, probe_start, probe_stop;
	float forward_tm, reverse_tm, probe_tm;
	float forward_dH, forward_dS, reverse_dH, reverse_dS, probe_dH, probe_dS;
	int32_t primer_strand, probe_strand;
	int32_t forward_primer_clamp, reverse_primer_clamp;
	int32_t forward_mm, forward_gap, reverse_mm, reverse_gap, probe_mm, probe_gap;
	int32_t forward_oligo, reverse_oligo;
	int32_t has_primers, has_probe;
	char forward_align[ALIGN_TEXT];
	char reverse_align[ALIGN_TEXT];
	char probe_align[ALIGN_TEXT];
};

// expanded assay as the kernels see it
struct AssayDev {
	int32_t id, degen_id;
	int32_t slot_f, slot_r, slot_p; // Slot indices (or -1)
	int32_t kind;                   // 0 PCR (amplicon), 1 padlock/MIPS, 2 probe (hybrid)
	int32_t has_probe;
	int32_t cache_replay;           // bit 0: one of its oligos is used under another strand concentration / threshold set elsewhere in the
	                                // context: the assay is paired by the replay of the reference's BindCache (tnt_pair.hip.h, row a6);
	                                // bit 1: a probe of more than 128 bases seeded in segments (this assay is one of them): its hits carry
	                                // the seed's diagonal in amplicon_start until the segments' hits are merged (tnt_host.hip)
};

// seed-table entry: one query word of one (assay, role, strand)
//   bits 0-6 word index, bit 7 strand (1 = minus / DNAHash::find, 0 = plus / find_complement),
//   bits 8-9 role (0 F, 1 R, 2 P), bit 10 probe / padlock assay (hash-overload lists),
//   bit 11 the PCR assay has a probe, bit 12 plain A/C/G/T oligo (word index == base offset),
//   bits 13-31 expanded assay index
constexpr uint32_t WT_MINUS = 1u << 7, WT_NONPCR = 1u << 10, WT_HASPROBE = 1u << 11, WT_PLAIN = 1u << 12;
constexpr int WT_ASSAY_SHIFT = 13;

// Sharded append-only outputs of the seed kernel: N_SHARD regions per array, one counter per region
// (SHARD_STRIDE words apart so the counters sit on different cache lines).
constexpr uint32_t N_SHARD = 32, SHARD_STRIDE = 32;

// linear index -> physical index of a sharded array: `pre` = exclusive prefix sums of the shard counts
// (N_SHARD + 1 entries), `region` = capacity of one shard
#if defined(__HIPCC__)
// wave-synchronous LDS hand-off between phases of the same wavefront
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Inclusive prefix sum over the 64 lanes of a wavefront in the VALU's data-parallel-primitive forms: four row_shr steps scan
// every row of 16 lanes, row_bcast:15 carries a row's total into the next row, row_bcast:31 the first half's into the second
// (gfx9 controls; lanes without a source take 0).  Six v_add_u32_dpp -- the shuffle form is six LDS crossbar reads
// (ds_bpermute_b32) each followed by a compare, a select and an add.
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x)
{
	x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x111, 0xF, 0xF, false)); // row_shr:1
	x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x112, 0xF, 0xF, false)); // row_shr:2
	x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x114, 0xF, 0xF, false)); // row_shr:4
	x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x118, 0xF, 0xF, false)); // row_shr:8
	x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1 and 3
	x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x143, 0xC, 0xF, false)); // row_bcast:31 into rows 2 and 3
	return x;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t x) { return uint32_t(__builtin_amdgcn_readlane(int(x), 63)); }
__device__ __forceinline__ uint32_t shard_locate(const uint32_t *pre, uint32_t lin, uint32_t region)
{
	uint32_t r = 0;
#pragma unroll
	for (uint32_t step = N_SHARD / 2; step > 0; step >>= 1)
		if (pre[r + step] <= lin) r += step;
	return r * region + (lin - pre[r]);
}
__device__ __forceinline__ void shard_prefix(const uint32_t *counts, uint32_t *pre, int tid)
{
	// called by the first wavefront of a workgroup; pre[N_SHARD] = total
	if (tid == 0) {
		uint32_t run = 0;
		for (uint32_t r = 0; r < N_SHARD; r++) {
			pre[r] = run;
			run += counts[r * SHARD_STRIDE];
		}
		pre[N_SHARD] = run;
	}
}
#endif

// status word set by kernels
constexpr uint32_t ERR_SEED_OVERFLOW = 1, ERR_ITEM_OVERFLOW = 2, ERR_POOL_OVERFLOW = 4, ERR_HIT_OVERFLOW = 8,
                   ERR_GROUP_OVERFLOW = 16, ERR_LIST_OVERFLOW = 32, ERR_STACK_OVERFLOW = 64, ERR_TIE_OVERFLOW = 128,
                   ERR_FILTER_UNSOUND = 256, // audit mode: the pre-filter would have rejected a window the bind kernel accepts
                   ERR_TEXT_OVERFLOW = 512;  // the alignment text of an oligo longer than 64 bases does not fit TNT_ALIGN_TEXT

} // namespace tnt
#endif
