// tnt_bind.hip.h -- gfx950 bind kernel: one wavefront LANE per (oligo, target-window) pair.
//
// Replaces, for one seed, everything bind_oligo_to_{minus,plus}_strand derives from NucCruc
// (bind_oligo.cpp:502-748 / :1202-1446): window load, align_dimer (nuc_cruc.cpp:492-696),
// tm_dimer / enumerate_dimer_alignments / trace_back (nuc_cruc.cpp:973-1170,1409-1618,2517-2540),
// evaluate_alignment (nuc_cruc.cpp:1620-2299), anchors / mismatch / gap / poly-degen counts
// (nuc_cruc_anchor.cpp, nuc_cruc.h:389-483), the threshold cascade and, for bound oligos,
// the alignment text (nuc_cruc_output.cpp:74-205).
//
// Layout choices:
//  * DP is swept column-major; the previous column's (M, I_query, I_target) of every query row
//    lives in VGPRs (static unroll over LMAX rows), so the recurrence needs no LDS traffic.
//  * delta_g (9.6 KB) and the best_base_pair LUT are staged in LDS once per workgroup; the
//    target window of every lane sits in LDS as bytes [column][lane].
//  * one 8-bit trace word per cell goes to an HBM scratch slab laid out [cell][lane] so a
//    wavefront writes 64 contiguous bytes per cell; the traceback reads it back (L2 resident).
//  * float scoring uses no FMA contraction (-ffp-contract=off), IEEE division, and host-computed
//    logarithms, so Tm/dH/dS are bit-identical to the reference's float32 arithmetic.
#ifndef TNT_BIND_HIP_H
#define TNT_BIND_HIP_H

#include <hip/hip_runtime.h>
#include <cstddef>
#include <stdint.h>
#include <type_traits>
#include "tnt_dev.h"

namespace tnt {

struct BindArgs {
	const Tables *tab;
	DbView db;
	const uint8_t *olig_base;  // [n][MAX_OLIGO] BASE codes, 5'->3'
	const uint32_t *olig_q4;   // [n][MAX_OLIGO/4] reversed (row i-1 = query[L-i]), one byte per row = 4 * base code
	const int32_t *olig_len;
	const Slot *slots;
	const Thresh *thresh;
	const Item *items;
	uint32_t n_items;
	const uint32_t *n_items_dev; // when set, the item count is read from the device counter (n_items caps it)
	const uint32_t *index;       // optional work list: item indices to bind (one length bucket of a phase queue)
	const uint32_t *index_base_dev; // when set, the work list starts at index[*index_base_dev]
	uint32_t index_stride;       // words between two entries of the work list (0 / 1: plain indices; 4: the seed kernel's 16-byte queue entries)
	int32_t *result;           // per item: pool index (>= 0) or -status
	BoundRec *pool;
	uint32_t pool_cap;
	uint32_t *pool_count;
	BindFull *full;            // optional (unit-level API): full record per item
	uint8_t *trace;            // scratch, per block slot
	uint8_t *best;             // scratch for the best alignment's columns (instances that do not keep them in LDS)
	uint32_t *stack;
	uint16_t *ties;            // [TIE_CAP][lane] further cells with the maximal score
	uint32_t *err;
	Group *groups;             // a group is put on the active list when its first item binds
	uint32_t *active;
	uint32_t *n_active;
	uint32_t *debug;           // experimental builds only (TNT_EXP_STATS)
	const uint8_t *audit;      // optional (audit mode of the pre-filter): 1 = the filter would have rejected this item
	// alignment texts that do not fit ALIGN_TEXT (oligos beyond 64 bases): written here, the record's field holds a reference
	char *ltext;
	uint32_t ltext_cap;
	uint32_t *ltext_count;     // bytes handed out
	// Round 5 (opt-in, TNT_MI355X_DEFER=1: measured, no gain -- DESIGN.md 4.2): the throughput instances of up to 64 rows can
	// leave what lies behind the Tm / dG tests of the cascade -- anchors, counts, the other thresholds, footprint, record and
	// alignment text -- to bind_finish_kernel: one window in five gets that far, so inside the bind kernel that code runs with
	// a fifth of the lanes.  A window that passes the two tests parks its state here.  Null (the default): everything inside
	// the bind kernel.
	struct PendingRec *pending;
	uint32_t pending_cap;
	uint32_t *n_pending;
};

// state of a window behind the Tm / dG tests (bind_oligo.cpp:600-624): what bind_finish needs of it
constexpr int PENDING_WIN = 64 + 2 * FLANK; // window bases of an oligo of up to 64 bases
constexpr int PENDING_COLS = 64 + PENDING_WIN + 4 + 4; // alignment columns (rounded up to dwords)
struct PendingRec {
	uint32_t idx, group;
	int32_t start, stop;
	float tm, dH, dS;
	uint32_t slot;
	int16_t n_cols, fm_q, fm_t, lm_q, lm_t, pad0;
	uint8_t L, W, minus, pad1;
	uint8_t win[PENDING_WIN];   // TARGET(k)
	uint8_t cols[PENDING_COLS]; // columns k0 .. k1 - 1 of the best alignment
};
static_assert(sizeof(PendingRec) % 4 == 0 && offsetof(PendingRec, win) % 4 == 0 && offsetof(PendingRec, cols) % 4 == 0, "records are moved as dwords");

// reference to a text in the long-text arena, as a text field holds it: 0x7F 'L' 'T' + offset (8 hex digits) + length (6 hex digits)

constexpr int TIE_CAP = 256;      // maximal cells of one DP beyond which the kernel reports ERR_TIE_OVERFLOW
constexpr unsigned long long WC_MASK = // watson_and_crick[], nuc_cruc.cpp:229-238
	(1ull << P_AT) | (1ull << P_TA) | (1ull << P_CG) | (1ull << P_GC) | (1ull << (bA * NB + bI)) | (1ull << (bI * NB + bA)) |
	(1ull << (bT * NB + bI)) | (1ull << (bI * NB + bT)) | (1ull << (bG * NB + bI)) | (1ull << (bI * NB + bG)) |
	(1ull << (bC * NB + bI)) | (1ull << (bI * NB + bC)) | (1ull << (bI * NB + bI));

__device__ __forceinline__ bool is_wc(int pair) { return (WC_MASK >> pair) & 1ull; }
__device__ __forceinline__ bool is_virtual(int b) { return b == bE || b == bGAP; }
__device__ __forceinline__ bool non_virtual_pair(int p) { return (p % NB < bE) && (p / NB < bE); }
__device__ __forceinline__ bool pair_has_gap(int p) { return (p % NB == bGAP) || (p / NB >= bGAP); }

// is_complemetary_base, nuc_cruc_anchor.cpp:8-139 (bit0 A, bit1 T, bit2 G, bit3 C); BASE codes 0..17.  The masks are
// 4-bit fields of compile-time words (the serial tail asks per alignment column: no memory behind it).
constexpr unsigned long long pack_nibbles(const int (&v)[18], int from)
{
	unsigned long long w = 0;
	for (int i = 0; i < 16 && from + i < 18; i++) w |= (unsigned long long)(v[from + i] & 15) << (4 * i);
	return w;
}
constexpr int QUERY_MASK[18] = {1, 8, 4, 2, 15, 0, 0, 1 | 8, 4 | 1, 4 | 8, 4 | 8 | 1, 1 | 2, 2 | 8, 1 | 8 | 2, 4 | 2, 4 | 1 | 2, 4 | 2 | 8, 15};
// the target base is mapped to its complement first
constexpr int TARGET_MASK[18] = {2, 4, 8, 1, 15, 0, 0, 2 | 4, 8 | 2, 8 | 4, 8 | 4 | 2, 2 | 1, 1 | 4, 2 | 4 | 1, 8 | 1, 8 | 2 | 1, 8 | 1 | 4, 15};
constexpr unsigned long long QM_LO = pack_nibbles(QUERY_MASK, 0), QM_HI = pack_nibbles(QUERY_MASK, 16);
constexpr unsigned long long TM_LO = pack_nibbles(TARGET_MASK, 0), TM_HI = pack_nibbles(TARGET_MASK, 16);
__device__ __forceinline__ uint32_t nibble(unsigned long long lo, unsigned long long hi, int i)
{
	return uint32_t((i < 16 ? lo : hi) >> ((i & 15) * 4)) & 15u;
}
__device__ __forceinline__ bool is_complementary(int q, int t) { return (nibble(QM_LO, QM_HI, q) & nibble(TM_LO, TM_HI, t)) != 0; }
// the letter NucCruc prints for a BASE code (nuc_cruc_output.cpp): "ACGTI$-MRSVWYHKDBN"
__device__ __forceinline__ char base_letter(int b)
{
	constexpr unsigned long long L0 = 0x4D2D244954474341ull;         // "ACGTI$-M", first letter in the low byte
	constexpr unsigned long long L1 = 0x444B485957565352ull;         // "RSVWYHKD"
	constexpr unsigned long long L2 = 0x4E42ull;                     // "BN"
	const unsigned long long w = b < 8 ? L0 : (b < 16 ? L1 : L2);
	return char(uint32_t(w >> ((b & 7) * 8)) & 255u);
}

// seq.h code (0..15) -> BASE code of the base itself (bind_oligo.cpp:1227-1276) ...
__device__ __forceinline__ int db_same(int c) { return c < 5 ? c : c + 2; }
// ... and of its complement (bind_oligo.cpp:524-573): A C G T I M R S V W Y H K D B N -> T G C A I K Y S B W R D M H V N
constexpr unsigned long long pack_bytes(int a, int b, int c, int d, int e, int f, int g, int h)
{
	return (unsigned long long)a | ((unsigned long long)b << 8) | ((unsigned long long)c << 16) | ((unsigned long long)d << 24) |
	       ((unsigned long long)e << 32) | ((unsigned long long)f << 40) | ((unsigned long long)g << 48) | ((unsigned long long)h << 56);
}
constexpr unsigned long long DB_COMP_LO = pack_bytes(bT, bG, bC, bA, bI, bK, bY, bS), DB_COMP_HI = pack_bytes(bB, bW, bR, bD, bM, bH, bV, bN);
__device__ __forceinline__ int db_comp(int c) { return int(uint32_t((c < 8 ? DB_COMP_LO : DB_COMP_HI) >> ((c & 7) * 8)) & 255u); }

// scoring scalars and the short ends of the loop / bulge entropy tables, staged in LDS for the serial tail
struct TailConst {
	float init_H, init_S, sym_S, at_H, at_S, salt, asym_S, bulge_at_S, temperature, log_na;
	int32_t dangle5, dangle3;
	float loop_S[64], bulge_S[64];
};

// per-lane evaluation context for the serial tail
struct LaneCtx {
	const Tables *tab;
	const uint8_t *lut;       // LDS best_base_pair LUT
	const uint8_t *hs_map;    // LDS: pair-stack index -> entry of hs_val
	const float2 *hs_val;     // LDS: the distinct (param_H, param_S) pairs
	const TailConst *k;       // LDS
	const uint8_t *win;       // LDS window base of this lane: win[k * BIND_THREADS]
	const uint8_t *query;     // global oligo bases 5'->3'
	uint8_t *cols;            // alignment columns of this lane (LDS), stride BIND_THREADS
	int L, W;
	float log_ct;
};

__device__ __forceinline__ int lut_pair(const uint8_t *lut, int x, int y)
{
	return ((x | y) < NB) ? x * NB + y : int(lut[x * 18 + y]);
}

// one alignment column in a byte: query side (A C G T I $ -: 3 bits) | target side (18 codes: 5 bits) << 3
__device__ __forceinline__ int col_q(uint8_t c) { return c & 7; }
__device__ __forceinline__ int col_t(uint8_t c) { return c >> 3; }
__device__ __forceinline__ uint8_t mk_col(int q, int t) { return uint8_t(q | (t << 3)); }

struct Score {
	float dH, dS, tm;
	bool ok;
};

// LDS byte addresses as plain integers: a table base folded into a per-column register leaves one (byte-select)
// add per read in the fill loop
typedef __attribute__((address_space(3))) const int lds_cint;
__device__ __forceinline__ int lds_addr(const void *p)
{
	return int(uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)p)));
}
#ifdef TNT_EXP_NOLDS
__device__ __forceinline__ int lds_load(int addr)
{
	int v = addr;
	asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));
	return v;
}
#else
__device__ __forceinline__ int lds_load(int addr) { return *(lds_cint *)(uintptr_t(uint32_t(addr))); }
#endif

// tw = 2 * tw + (x == y): v_cmp_eq + v_addc_co
__device__ __forceinline__ uint32_t push_eq(uint32_t tw, int x, int y)
{
	asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(tw) : "v"(x), "v"(y) : "vcc");
	return tw;
}

// x >> 31 as a full-rate shift (kept opaque: the compiler would turn it into a compare + select)
__device__ __forceinline__ uint32_t sign_bit(int x)
{
	uint32_t r;
	asm("v_lshrrev_b32 %0, 31, %1" : "=v"(r) : "v"(x));
	return r;
}

struct AlignInfo {
	int k0, k1;               // column range inside the scratch buffer
	int fm_q, fm_t, lm_q, lm_t;
};

// NucCruc::mode, nuc_cruc.h:494
enum : int { MODE_HOMO_DIMER = 0, MODE_HETERO_DIMER = 1, MODE_HAIRPIN = 2 };

// evaluate_alignment, nuc_cruc.cpp:1620-2299.  Columns k0..k1-1 of c.cols.  In HAIRPIN mode the sums start
// from the loop terms computed by the caller (evaluate_hairpin_alignment, :2301-2394).
template <int BIND_THREADS> // lane stride of the per-lane LDS / scratch slices
__device__ Score evaluate_alignment(const LaneCtx &c, int k0, int k1, int mode = MODE_HETERO_DIMER, float dH0 = 0.0f, float dS0 = 0.0f)
{
	const Tables &T = *c.tab;
	const TailConst &K = *c.k;
	// param_H / param_S through the LDS copy: 152 distinct (H, S) pairs behind a byte map
#define HS_ADD(IDX)                                              \
	do {                                                         \
		const float2 v_ = c.hs_val[c.hs_map[(IDX)]];             \
		dH += v_.x;                                              \
		dS += v_.y;                                              \
	} while (0)
#define HS_SUB(IDX)                                              \
	do {                                                         \
		const float2 v_ = c.hs_val[c.hs_map[(IDX)]];             \
		dH -= v_.x;                                              \
		dS -= v_.y;                                              \
	} while (0)
	const uint8_t *lut = c.lut;
#define COLQ(k) col_q(c.cols[(k) * BIND_THREADS])
#define COLT(k) col_t(c.cols[(k) * BIND_THREADS])
	int terminal = P__, lastlast = P__, last = P__, curp = P__;
	float dH = dH0, dS = dS0;
	if (mode != MODE_HAIRPIN) { // :1627-1633
		dH = K.init_H;
		dS = K.init_S + (mode == MODE_HOMO_DIMER ? K.sym_S : 0.0f);
	}
	unsigned nqg = 0, ntg = 0, nmm = 0, num_base = 0;
	bool terminal_5 = false;
	const int n = k1 - k0;

	{
		const int q0 = COLQ(k0), t0 = COLT(k0);
		curp = lut_pair(lut, q0, t0);
		if (is_wc(curp)) {
			terminal_5 = true;
			if (curp == P_AT || curp == P_TA) {
				dH += K.at_H;
				dS += K.at_S;
			}
		}
		num_base += is_virtual(q0) ? 0 : 1;
		num_base += is_virtual(t0) ? 0 : 1;
	}

	for (int k = 1; k < n; k++) {
		const uint8_t cw = c.cols[(k0 + k) * BIND_THREADS];
		const int qk = col_q(cw), tk = col_t(cw);
		lastlast = last;
		last = curp;
		curp = lut_pair(lut, qk, tk);
		const bool align_start = (k == 1), align_stop = (k == n - 1);
		const bool in_loop = (qk == bGAP) || (tk == bGAP) || (!is_wc(last) && !is_wc(curp));

		if (!in_loop) {
			if (align_start && !is_wc(last) && non_virtual_pair(last)) { // frayed start, :1726-1745
				int tmp = lut_pair(lut, last / NB, bE);
				HS_ADD(tmp * NBP + curp);
				tmp = lut_pair(lut, bE, last % NB);
				HS_ADD(tmp * NBP + curp);
			} else if (align_stop && !is_wc(curp) && non_virtual_pair(curp)) { // frayed stop, :1747-1763
				int tmp = lut_pair(lut, qk, bE);
				HS_ADD(last * NBP + tmp);
				tmp = lut_pair(lut, bE, tk);
				HS_ADD(last * NBP + tmp);
			} else {
				HS_ADD(last * NBP + curp);
			}
			num_base += is_virtual(qk) ? 0 : 1;
			num_base += is_virtual(tk) ? 0 : 1;
		}

		if (is_wc(curp) || curp == P_EE) { // :1794
			terminal = curp;
			if (!terminal_5) {
				terminal_5 = true;
				if (curp == P_AT || curp == P_TA) {
					dH += K.at_H;
					dS += K.at_S;
				}
			}
			const unsigned max_gap = nqg > ntg ? nqg : ntg;
			if (nmm > 1 || (max_gap > 0 && nmm == 1)) { // internal loop, :1824
				const unsigned gap_diff = (nqg > ntg) ? nqg - ntg : ntg - nqg;
				const unsigned loop_size = nmm * 2 + gap_diff;
				if (loop_size == 2 && (last == P_GT || last == P_TG) && (lastlast == P_GT || lastlast == P_TG)) {
					HS_ADD(lastlast * NBP + last);
					num_base += 2;
				} else {
					dS += loop_size < 64u ? K.loop_S[loop_size] : T.loop_S[loop_size];
					dS += float(gap_diff) * K.asym_S;
					int rhs_q = k - 1, rhs_t = k - 1;
					HS_SUB(last * NBP + curp);
					if (!pair_has_gap(last)) { // right terminal mismatch, :1923-1935
						HS_ADD(last * NBP + curp);
					} else { // :1936-1992
						int mm = P__;
						if (last / NB == bGAP) {
							while (true) {
								const int qq = COLQ(k0 + rhs_q);
								if (!is_virtual(qq)) {
									mm = lut_pair(lut, qq, last % NB);
									break;
								}
								if (rhs_q == 0) break;
								rhs_q--;
							}
						} else {
							while (true) {
								const int tt = COLT(k0 + rhs_t);
								if (!is_virtual(tt)) {
									mm = lut_pair(lut, last / NB, tt);
									break;
								}
								if (rhs_t == 0) break;
								rhs_t--;
							}
						}
						HS_ADD(mm * NBP + curp);
					}
					// left terminal mismatch, :2000-2077
					int lhs_q = k - 1, lhs_t = k - 1;
					while (true) {
						const int pm = lut_pair(lut, COLQ(k0 + lhs_q), COLT(k0 + lhs_t));
						if (is_wc(pm)) {
							lhs_q++, lhs_t++;
							if (COLQ(k0 + lhs_q) != bGAP && COLT(k0 + lhs_t) != bGAP) {
								const int mm = lut_pair(lut, COLQ(k0 + lhs_q), COLT(k0 + lhs_t));
								HS_SUB(pm * NBP + mm);
							} else {
								num_base += 2;
								while (COLQ(k0 + lhs_q) == bGAP) lhs_q++;
								while (COLT(k0 + lhs_t) == bGAP) lhs_t++;
							}
							const int mm = lut_pair(lut, COLQ(k0 + lhs_q), COLT(k0 + lhs_t));
							HS_ADD(pm * NBP + mm);
							break;
						}
						if (lhs_q == 0) break;
						lhs_q--, lhs_t--;
					}
					if (rhs_q != lhs_q) num_base++;
					if (rhs_t != lhs_t) num_base++;
				}
			} else if (nqg || ntg) { // bulge, :2105-2185
				const unsigned bulge = nqg > ntg ? nqg : ntg;
				if (bulge == 1) {
					HS_ADD(lastlast * NBP + curp);
				}
				dS += bulge < 64u ? K.bulge_S[bulge] : T.bulge_S[bulge];
				if (bulge != 1 && (qk == bA || qk == bT)) dS += K.bulge_at_S;
				if (bulge != 1) { // has_AT_initiation, :2888-2905
					int i = k;
					do {
						i--;
					} while (i != 0 && (COLQ(k0 + i) == bGAP || COLT(k0 + i) == bGAP));
					const int p = lut_pair(lut, COLQ(k0 + i), COLT(k0 + i));
					if (p == P_AT || p == P_TA) dS += K.bulge_at_S;
				}
			}
			nqg = ntg = nmm = 0;
		} else {
			nmm += (!is_virtual(qk) && !is_virtual(tk)) ? 1 : 0;
		}
		nqg += (qk == bGAP) ? 1 : 0;
		ntg += (tk == bGAP) ? 1 : 0;
	}
#undef COLQ
#undef COLT
	if (terminal == P_AT || terminal == P_TA) {
		dH += K.at_H;
		dS += K.at_S;
	}
	Score s;
	s.dH = dH;
	s.dS = dS;
	s.tm = 0.0f;
	s.ok = false;
	if (dH >= 0.0f) return s; // :2253
	dS += K.salt * (0.5f * float(num_base) - 1.0f) * K.log_na; // :2270
	const float tm = (mode == MODE_HAIRPIN) ? dH / dS - 273.15f // :2286-2289
	                                        : dH / (1.9872e-3f * c.log_ct + dS) - 273.15f; // :2291
	s.dS = dS;
	s.tm = tm > 0.0f ? tm : 0.0f;
	s.ok = true;
	return s;
}

// evaluate_hairpin_alignment, nuc_cruc.cpp:2301-2394: loop entropy, special tri/tetra-loop bonus
// (find_loop_index, :2619-2869 = exact match of the 5 / 6 bases starting at the 5' closing base),
// closing-AT penalty or terminal mismatch, then the stem through evaluate_alignment(HAIRPIN).
template <int BIND_THREADS>
__device__ Score evaluate_hairpin(const LaneCtx &c, const Tables *tab, const AlignInfo &al, const uint8_t *gquery, const uint8_t *lds_query,
                                  int L)
{
	const Tables &T = *tab;
#define HQ(K) (gquery ? int(gquery[(K)]) : int(lds_query[(L - 1 - (K)) * BIND_THREADS]))
	const int last_3 = al.fm_q, last_5 = al.fm_t;
	const unsigned loop_len = unsigned(last_3 - last_5 - 1);
	float dH = 0.0f, dS = 0.0f;
	dS += T.hairpin_S[loop_len < 513u ? loop_len : 512u];
	const int last_pair = lut_pair(c.lut, HQ(last_5), HQ(last_3));
	if (loop_len == 3 || loop_len == 4) {
		const int n = int(loop_len) + 2;
		uint32_t key = uint32_t(n) << 24; // 3 bits per base, A C G T; anything else cannot match
		bool plain = true;
		for (int k = 0; k < n; k++) {
			const int b = HQ(last_5 + k);
			plain &= b <= bT;
			key |= uint32_t(b & 7) << (3 * k);
		}
		if (plain)
			for (int i = 0; i < T.n_special; i++)
				if (T.special_key[i] == key) {
					dH += T.special_H[i];
					dS += T.special_S[i];
					break;
				}
	}
	if (loop_len == 3) {
		if (last_pair == P_AT || last_pair == P_TA) dS += T.bulge_at_S;
	} else {
		const int curp = lut_pair(c.lut, HQ(last_5 + 1), HQ(last_3 - 1));
		const float2 v_ = c.hs_val[c.hs_map[last_pair * NBP + curp]]; // param_hairpin_terminal == param (nuc_cruc_santa_lucia.cpp:594-603)
		dH += v_.x;
		dS += v_.y;
	}
#undef HQ
	return evaluate_alignment<BIND_THREADS>(c, al.k0, al.k1, MODE_HAIRPIN, dH, dS);
}

// The trace words of one DP as the serial tail reads them.  Lane-per-window kernels keep them in an HBM slab laid out
// [column][row][lane]: the path mostly runs down a diagonal, so PF cells of the current diagonal are fetched at once (one
// memory latency instead of one per step) and parked in LDS.
template <int LMAX, int STRIDE, int PF> struct SlabTrace {
	const uint8_t *trace; // this lane's first trace word
	uint8_t *pf;          // this lane's PF-entry slice of the LDS prefetch buffer
	int pf_i, pf_j;       // first cell of the diagonal run held in pf (none: -1)
	__device__ __forceinline__ void reset() { pf_i = pf_j = -1; }
	__device__ __forceinline__ int get(int li, int lj)
	{
		int d = pf_i - li;
		if (pf_i < 0 || d < 0 || d >= PF || pf_j - lj != d) {
			uint8_t v[PF];
#pragma unroll
			for (int k = 0; k < PF; k++) v[k] = (li - k >= 1 && lj - k >= 1) ? trace[(size_t(lj - k - 1) * LMAX + size_t(li - k - 1)) * STRIDE] : uint8_t(0);
#pragma unroll
			for (int k = 0; k < PF; k++) pf[k * STRIDE] = v[k];
			pf_i = li, pf_j = lj;
			d = 0;
		}
		return int(pf[d * STRIDE]);
	}
};

// The serial tail of one (oligo, window): everything behind the DP fill -- tm_dimer's loop over the maximal cells,
// The back of the serial tail: what bind_oligo_to_*_strand does with a window once Tm and dG passed (bind_oligo.cpp:626-748) --
// anchors and counts (nuc_cruc_anchor.cpp, nuc_cruc.h:389-483), the other thresholds, alignment_range, the bound record and the
// alignment text (operator<<, nuc_cruc_output.cpp:74-205).  Called at the end of bind_tail (unit-level instance, oligos beyond 64
// bases, --dinkelbach) or, for the windows the throughput instances parked, by bind_finish_kernel.
//   win / cols / best: per-lane slices with stride STRIDE; qry (QRY_LDS) the oligo in row order, else `query` 5'->3' in memory.
#define QUERY(K) (QRY_LDS ? int(qry[(L - 1 - (K)) * STRIDE]) : int(query[(K)]))
#define TARGET(K) int(win[(K) * STRIDE])
template <bool FULL, bool QRY_LDS, bool LONG_TEXT, int STRIDE>
__device__ __forceinline__ void bind_finish(const BindArgs &a, const uint8_t *win, const uint8_t *qry, const uint8_t *query, uint8_t *cols, const uint8_t *best, AlignInfo bi,
                                            const bool best_valid, int status, const Thresh &th, const float tm, const float best_dH, const float best_dS, const float dg,
                                            const uint32_t idx, const Item &it, const int L, const int W, const int start, const int stop, const bool minus, const int gmax, const int nmax)
{
	int a5 = 0, a3 = 0, mm = 0, gap = 0, poly = 0, loc5 = 0, loc3 = 0;
	// The columns of the best alignment sit in `best` -- HBM for the instances whose LDS is taken by other things (every plain
	// instance of 23 rows and more) -- and the counts below and the alignment text read them one at a time, each read waiting for
	// the one before it (a rolled loop; the text's stores in between may alias them as far as the compiler knows): some hundred
	// memory round trips per record, 100 000 of the 509 000 cycles a wavefront spent on 64 windows (TNT_EXP_STATS: thresholds
	// 37 000, output 84 000).  `cols` (LDS) is free once the enumeration is through: the columns move there eight loads at a time.
	if (best_valid && (FULL || status == 0)) {
		for (int k = bi.k0; k < bi.k1; k += 8) {
			uint8_t v[8];
#pragma unroll
			for (int u = 0; u < 8; u++) v[u] = k + u < bi.k1 ? best[(k + u) * STRIDE] : uint8_t(0);
#pragma unroll
			for (int u = 0; u < 8; u++)
				if (k + u < bi.k1) cols[(k + u) * STRIDE] = v[u];
		}
	}
	if (best_valid && (FULL || status == 0)) {
		const int front_q = col_q(cols[bi.k0 * STRIDE]), front_t = col_t(cols[bi.k0 * STRIDE]);
		const int back_q = col_q(cols[(bi.k1 - 1) * STRIDE]), back_t = col_t(cols[(bi.k1 - 1) * STRIDE]);
		{ // anchor5_query, nuc_cruc_anchor.cpp:143-192
			int qi = 0, ti = bi.fm_q + bi.fm_t;
			bool go = !(front_t == bE);
			if (front_q == bE) ti--;
			if (ti >= W) go = false;
			while (go) {
				if (qi >= L || ti < 0) break;
				if (!is_complementary(QUERY(qi), TARGET(ti))) break;
				a5++, qi++, ti--;
			}
		}
		{ // anchor3_query, nuc_cruc_anchor.cpp:249-298
			int qi = L - 1, ti = (bi.lm_q + bi.lm_t + 1) - L;
			bool go = !(back_t == bE);
			if (back_q == bE) ti++;
			if (ti >= W || ti < 0) go = false;
			while (go) {
				if (qi < 0 || ti >= W) break;
				if (!is_complementary(QUERY(qi), TARGET(ti))) break;
				a3++, qi--, ti++;
			}
		}
		{ // num_mismatch_by_query / num_gap / max_contiguous_target_degen, nuc_cruc.h:389-483
			int aligned = 0, run = 0;
			for (int k = bi.k0; k < bi.k1; k++) {
				const uint8_t cw = cols[k * STRIDE];
				const int q = col_q(cw), t = col_t(cw);
				if (!is_virtual(q)) {
					if (!is_virtual(t) && !is_complementary(q, t)) mm++;
					aligned++;
				}
				gap += (q == bGAP) + (t == bGAP);
				if (t >= bM && t <= bN) {
					run++;
					if (run > poly) poly = run;
				} else {
					run = 0;
				}
			}
			mm += L - aligned;
		}
		if (!status) {
			if (a5 < th.clamp5) status = 3;
			else if (a3 < th.clamp3) status = 4;
			else if (mm > th.max_mm) status = 5;
			else if (gap > th.max_gap) status = 6;
			else if (poly > th.max_poly) status = 7;
		}
		// alignment_range (nuc_cruc_anchor.cpp:379-396) -> plus-strand footprint
		const int qf = bi.fm_q, ql = bi.lm_q, tf = bi.lm_t, tl = bi.fm_t;
		loc5 = loc3 = start;
		if (minus) { // bind_oligo.cpp:721-731
			loc5 += (stop - start - 1) - tl;
			loc3 += (stop - start - 1) - tf;
			loc5 -= qf;
			loc3 += (L - 1) - ql;
		} else { // bind_oligo.cpp:1424-1434
			loc5 += tf;
			loc3 += tl;
			loc3 += qf;
			loc5 -= (L - 1) - ql;
		}
	} else if (!best_valid && !status) {
		status = 8; // SURVEY Q11 fence: no alignment is never a bind
	}

	// ------------------------------------------------------------------ output
	char *text = nullptr;
	int32_t res = -status;
	BoundRec *rec = nullptr;
	if (!FULL && status == 0 && a.audit && a.audit[idx]) atomicOr(a.err, ERR_FILTER_UNSOUND);
	if (status == 0 && !FULL) {
		const uint32_t slotp = atomicAdd(a.pool_count, 1u);
		if (slotp < a.pool_cap) {
			rec = a.pool + slotp;
			rec->loc5 = loc5, rec->loc3 = loc3;
			rec->tm = tm, rec->dH = best_dH, rec->dS = best_dS;
			rec->a5 = a5, rec->a3 = a3, rec->mm = mm, rec->gap = gap;
			text = rec->text;
			res = int32_t(slotp);
			if (a.groups && !(atomicOr(&a.groups[it.group].count, GROUP_ALIVE) & GROUP_ALIVE))
				a.active[atomicAdd(a.n_active, 1u)] = it.group;
		} else {
			atomicOr(a.err, ERR_POOL_OVERFLOW);
			res = -100;
		}
	}
	if (FULL) {
		BindFull &f = a.full[idx];
		f.status = status, f.target_start = start, f.target_stop = stop;
		f.tm = tm, f.dH = best_dH, f.dS = best_dS, f.dg = dg;
		f.a5 = a5, f.a3 = a3, f.mm = mm, f.gap = gap, f.poly = poly;
		f.loc5 = loc5, f.loc3 = loc3;
		f.q_first = bi.fm_q, f.q_last = bi.lm_q, f.t_first = bi.lm_t, f.t_last = bi.fm_t;
		f.valid = best_valid ? 1 : 0;
		f.dp_max = gmax;
		f.n_max = nmax;
		f.text[0] = 0;
		if (best_valid) text = f.text;
	} else {
		a.result[idx] = res;
	}

	if (text) { // operator<<(ostream&, NucCruc), nuc_cruc_output.cpp:74-205
		int prefix = bi.fm_q < (W - 1 - bi.fm_t) ? bi.fm_q : (W - 1 - bi.fm_t);
		if (prefix < 0) prefix = 0;
		int suffix = (L - 1 - bi.lm_q) < bi.lm_t ? (L - 1 - bi.lm_q) : bi.lm_t;
		if (suffix < 0) suffix = 0;
		// three lines of prefix + columns + suffix and 18 characters of decoration: the field holds every alignment of a
		// 64-mer; one of a longer oligo that does not fit goes to the context's long-text arena and the field gets a reference
		// to it (tnt_long_text reads through it).  Never a cut text: without room in the arena the call fails.
		// (written a dword at a time: every lane writes into a record of its own, so a byte store is a memory request)
		int o = 0;
		uint32_t acc = 0;
		static_assert(offsetof(BoundRec, text) % 4 == 0 && offsetof(BindFull, text) % 4 == 0 && sizeof(BoundRec) % 4 == 0 && sizeof(BindFull) % 4 == 0 && ALIGN_TEXT % 4 == 0,
		              "text fields on 4-byte boundaries");
		uint32_t *text32 = reinterpret_cast<uint32_t *>(text);
		const int text_chars = 3 * (prefix + (bi.k1 - bi.k0) + suffix) + 18;
		uint32_t lref = 0xFFFFFFFFu; // where the text sits in the arena
		if (LONG_TEXT && text_chars >= ALIGN_TEXT) {
			const uint32_t need = (uint32_t(text_chars) + 4u) & ~3u;
			const uint32_t at = a.ltext ? atomicAdd(a.ltext_count, need) : 0xFFFFFFFFu;
			if (a.ltext && at <= a.ltext_cap && need <= a.ltext_cap - at) {
				lref = at;
				text32 = reinterpret_cast<uint32_t *>(a.ltext + at);
			} else {
				atomicOr(a.err, ERR_TEXT_OVERFLOW);
				prefix = suffix = 0, bi.k1 = bi.k0;
			}
		}
		auto put = [&](char ch) __attribute__((always_inline)) {
			acc |= uint32_t(uint8_t(ch)) << ((o & 3) * 8);
			o++;
			if ((o & 3) == 0) text32[(o >> 2) - 1] = acc, acc = 0;
		};
		put('5'), put('\''), put(' ');
		for (int i = 0; i < prefix; i++) put(base_letter(QUERY(bi.fm_q - prefix + i)));
		for (int k = bi.k0; k < bi.k1; k++) put(base_letter(col_q(cols[k * STRIDE])));
		for (int i = 0; i < suffix; i++) put(base_letter(QUERY(bi.lm_q + 1 + i)));
		put(' '), put('3'), put('\''), put('\n');
		put(' '), put(' '), put(' ');
		for (int i = 0; i < prefix; i++) put(is_complementary(QUERY(bi.fm_q - prefix + i), TARGET(bi.fm_t + prefix - i)) ? ':' : ' ');
		for (int k = bi.k0; k < bi.k1; k++) {
			const uint8_t cw = cols[k * STRIDE];
			put(is_complementary(col_t(cw), col_q(cw)) ? '|' : ' ');
		}
		for (int i = 0; i < suffix; i++) put(is_complementary(QUERY(bi.lm_q + 1 + i), TARGET(bi.lm_t - i - 1)) ? ':' : ' ');
		put('\n');
		put('3'), put('\''), put(' ');
		for (int i = prefix; i > 0; i--) put(base_letter(TARGET(bi.fm_t + i)));
		for (int k = bi.k0; k < bi.k1; k++) put(base_letter(col_t(cols[k * STRIDE])));
		for (int i = 1; i <= suffix; i++) put(base_letter(TARGET(bi.lm_t - i)));
		put(' '), put('5'), put('\'');
		text32[o >> 2] = acc; // the rest and the terminating zero (acc holds zeros above the characters written)
		if (rec) rec->text_len = o; // (the text's own length also where the field only refers to it: sort_by_bound_match compares it)
		if (lref != 0xFFFFFFFFu) { // the field: 0x7F L T, offset, length -- no zero byte inside, one behind
			uint32_t *f32 = reinterpret_cast<uint32_t *>(text);
			auto hex = [](uint32_t v, int d) __attribute__((always_inline)) { return uint32_t("0123456789abcdef"[(v >> (4 * d)) & 15u]); };
			const uint32_t len = uint32_t(o);
			f32[0] = 0x7Fu | (uint32_t('L') << 8) | (uint32_t('T') << 16) | (hex(lref, 7) << 24);
			f32[1] = hex(lref, 6) | (hex(lref, 5) << 8) | (hex(lref, 4) << 16) | (hex(lref, 3) << 24);
			f32[2] = hex(lref, 2) | (hex(lref, 1) << 8) | (hex(lref, 0) << 16) | (hex(len, 5) << 24);
			f32[3] = hex(len, 4) | (hex(len, 3) << 8) | (hex(len, 2) << 16) | (hex(len, 1) << 24);
			f32[4] = hex(len, 0);
		}
	}
}
#undef QUERY
#undef TARGET

// enumerate_dimer_alignments + trace_back, evaluate_alignment, the threshold cascade, anchors and counts, footprint, record
// and alignment text.  Shared by the lane-per-window kernel (bind_kernel, below: STRIDE = workgroup size, per-lane slices of
// LDS / scratch arrays) and the wavefront-per-window kernel (tnt_wbind.hip.h: STRIDE = 1, one lane runs it).
//   FULL / DINK: as for bind_kernel.  QRY_LDS: `qry` holds the oligo in row order (row i-1 = query[L-i]).
//   LONG_TEXT: the alignment text may not fit ALIGN_TEXT (oligos beyond 64 bases).
//   A maximal cell is (i << CODE_SHIFT) | j; cells are numbered li * wstride + lj for the branch-cursor stack, whose entries
//   are  key (cell * 4 + state, 24 bits) | trace mask << 24 | cursor << 28.
// Returns true when a Dinkelbach pass has to follow (Tcur updated).
//   DEFER: the instance may park a window behind the Tm / dG tests for bind_finish_kernel (BindArgs::pending).
template <bool FULL, bool DINK, bool QRY_LDS, bool LONG_TEXT, int STRIDE, int CODE_SHIFT, typename TIE_T, class TRACE, bool DEFER = false>
__device__ __forceinline__ bool bind_tail(const BindArgs &a, TRACE &tr, const uint8_t *s_lut, const uint8_t *s_hs_map, const float2 *s_hs_val, const TailConst &s_k,
                                          const uint8_t *win, const uint8_t *qry, uint8_t *cols, uint8_t *best, uint32_t *stack, TIE_T *ties, int ncols, int wstride,
                                          uint32_t idx, const Item &it, const Slot &slot, int L, int W, int start, int stop, bool minus, int xmode, int gmax, int nmax,
                                          int max_ij, float &Tcur, float &dk_q, long long st_fill0 = 0)
{
	LaneCtx c;
	c.tab = a.tab;
	c.lut = s_lut;
	c.hs_map = s_hs_map;
	c.hs_val = s_hs_val;
	c.k = &s_k;
	c.win = win;
	c.query = a.olig_base + size_t(slot.oligo) * MAX_OLIGO;
	c.cols = cols;
	c.L = L;
	c.W = W;
	c.log_ct = slot.log_ct;

#ifdef TNT_EXP_STATS
	long long st_trace = 0, st_trim = 0, st_eval = 0;
	const long long st_tail0 = __builtin_readcyclecounter();
#endif
	bool best_valid = false;
	float best_dH = 0.0f, best_dS = 0.0f, best_tm = 0.0f;
	AlignInfo bi = {0, 0, 0, 0, 0, 0};

#define QUERY(K) (QRY_LDS ? int(qry[(L - 1 - (K)) * STRIDE]) : int(c.query[(K)]))
#define TARGET(K) int(win[(K) * STRIDE])

	// tm_dimer: every maximal cell in row-major order (nuc_cruc.cpp:2533-2537).  Cell codes (i << 8 | j)
	// sort row-major numerically; the first maximal cell lives in max_ij, later ties in ties[1..nmax-1].
	if (nmax > TIE_CAP) {
		atomicOr(a.err, ERR_TIE_OVERFLOW);
		nmax = TIE_CAP;
	}
	if (nmax > 1) {
		ties[0] = TIE_T(max_ij);
		for (int x = 1; x < nmax; x++) { // insertion sort, rare path
			const TIE_T v = ties[x * STRIDE];
			int y = x - 1;
			while (y >= 0 && ties[y * STRIDE] > v) {
				ties[(y + 1) * STRIDE] = ties[y * STRIDE];
				y--;
			}
			ties[(y + 1) * STRIDE] = v;
		}
	}
	for (int mc = 0; mc < nmax; mc++) {
		const int code = (nmax == 1) ? max_ij : int(ties[mc * STRIDE]);
		const int ci = code >> CODE_SHIFT, cj = code & ((1 << CODE_SHIFT) - 1);

		// ---------------- enumerate_dimer_alignments, nuc_cruc.cpp:973-1170
		bool first_time = true;
		int sp = 0;
		int zero_count = -1;
		unsigned trace_count = 0;
		float best_dg = best_dH - Tcur * best_dS;
		while (true) {
			if (!first_time && sp == 0 && zero_count <= 0) break;
			if (16u < trace_count) break;
			trace_count++;
			first_time = false;

#ifdef TNT_EXP_STATS
			const long long st_a = __builtin_readcyclecounter();
#endif
			// ---------------- trace_back, nuc_cruc.cpp:1409-1618
			AlignInfo al;
			al.k0 = 1;
			int n = 0; // columns written at cols[1 + n]
			int li = ci, lj = cj;
			al.fm_q = L - li;
			al.fm_t = lj - 1;
			al.lm_q = 0;
			al.lm_t = 0;
			int truncate_at_zero = 0;
			bool count_zeros = false;
			if (zero_count < 0) {
				zero_count = 0;
				count_zeros = true;
			} else {
				truncate_at_zero = zero_count--;
			}
			int key = -1;
			int mask = 1;
			tr.reset();
			while (true) {
				bool valid = true;
				int local;
				if (((mask & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1)) > 1) { // PATH_SPLIT
					int f = 0;
					for (; f < sp; f++)
						if (int(stack[f * STRIDE] & 0xFFFFFFu) == key) break;
					if (f == sp) {
						const int curb = (mask & 1) ? 1 : ((mask & 2) ? 2 : 4);
						if (sp < ncols) {
							stack[sp * STRIDE] = uint32_t(key) | (uint32_t(mask) << 24) | (uint32_t(curb) << 28);
							sp++;
						} else {
							atomicOr(a.err, ERR_STACK_OVERFLOW);
						}
						local = curb;
					} else {
						local = int(stack[f * STRIDE] >> 28);
					}
				} else {
					local = mask;
				}
				// border cells hold M = I_query = I_target = -1 (nuc_cruc.h:531-536)
				const bool inside = (li >= 1 && lj >= 1);
				// The path mostly runs down a diagonal: PF cells of the current diagonal are fetched at once (one
				// memory latency instead of one per step) and refilled when the walk leaves the run.
				int tw = TW_M_NEG | TW_IQ_NEG | TW_IT_NEG; // border cell
				if (inside) tw = tr.get(li, lj);
				const int cidx = li * wstride + lj;
				if (local == 1) { // query_target
					if (li > L || lj < 1 || li < 1 /* Q7 fence */) {
						valid = false;
					} else {
						if ((tw & TW_M_NEG) == TW_M_NEG) valid = false;
						else if (tw & TW_M_LE0) {
							if (count_zeros) zero_count++;
							else if (--truncate_at_zero == 0) valid = false;
						}
						cols[(1 + n) * STRIDE] = mk_col(QUERY(L - li), TARGET(lj - 1));
						n++;
						al.lm_q = L - li;
						al.lm_t = lj - 1;
						mask = ((tw & TW_M_NOT_M) ? 0 : 1) | ((tw & TW_M_NOT_IT) ? 0 : 2) | ((tw & TW_M_NOT_IQ) ? 0 : 4);
						key = cidx * 4 + 0;
						li--, lj--;
					}
				} else if (local == 4) { // gap_target: GAP in the query strand, state I_query
					if (lj < 1) {
						valid = false;
					} else {
						if ((tw & TW_IQ_NEG) == TW_IQ_NEG) valid = false;
						cols[(1 + n) * STRIDE] = mk_col(bGAP, TARGET(lj - 1));
						n++;
						al.lm_q = L - li + 1;
						al.lm_t = lj - 1;
						mask = ((tw & TW_IQ_LT) ? 0 : 1) | ((tw & TW_IQ_GT) ? 0 : 4); // insert >= extend: from M; extend >= insert: from I_query
						key = cidx * 4 + 1;
						lj--;
					}
				} else { // query_gap: GAP in the target strand, state I_target
					if (li > L || li < 1 /* Q7 fence */) {
						valid = false;
					} else {
						if ((tw & TW_IT_NEG) == TW_IT_NEG) valid = false;
						cols[(1 + n) * STRIDE] = mk_col(QUERY(L - li), bGAP);
						n++;
						al.lm_q = L - li;
						al.lm_t = lj;
						mask = ((tw & TW_IT_LT) ? 0 : 1) | ((tw & TW_IT_GT) ? 0 : 2);
						key = cidx * 4 + 2;
						li--;
					}
				}
				if (!valid) break;
			}
			al.k1 = 1 + n;
#ifdef TNT_EXP_STATS
			const long long st_b = __builtin_readcyclecounter();
			st_trace += st_b - st_a;
#endif

			// trim frayed ends, :1022-1054
			while (al.k1 > al.k0) {
				const uint8_t cw = cols[(al.k1 - 1) * STRIDE];
				if (is_wc(lut_pair(s_lut, col_q(cw), col_t(cw)))) break;
				if (!is_virtual(col_q(cw))) --al.lm_q;
				if (!is_virtual(col_t(cw))) ++al.lm_t;
				al.k1--;
			}
			while (al.k1 > al.k0) {
				const uint8_t cw = cols[al.k0 * STRIDE];
				if (is_wc(lut_pair(s_lut, col_q(cw), col_t(cw)))) break;
				if (!is_virtual(col_q(cw))) ++al.fm_q;
				if (!is_virtual(col_t(cw))) --al.fm_t;
				al.k0++;
			}
			// advance the branch odometer, :1058-1066
			if (zero_count == 0 && sp > 0) {
				while (sp > 0) {
					uint32_t e = stack[(sp - 1) * STRIDE];
					const int m = int((e >> 24) & 15u);
					int cb = int(e >> 28);
					bool more = false;
					while ((cb = cb << 1) < 8)
						if (cb & m) {
							more = true;
							break;
						}
					if (more) {
						stack[(sp - 1) * STRIDE] = (e & 0xFFFFFFFu) | (uint32_t(cb) << 28);
						break;
					}
					sp--;
				}
				zero_count = -1;
			}
#define CONSIDER(SCORE)                                                                              \
	do {                                                                                             \
	const Score s_ = (SCORE);                                                                    \
	if (s_.ok) {                                                                                 \
		const float local_dg = s_.dH - Tcur * s_.dS;                                               \
		if (!best_valid || local_dg < best_dg) { /* :1160-1167 */                                \
			best_valid = true;                                                                   \
			best_dg = local_dg;                                                                  \
			best_dH = s_.dH, best_dS = s_.dS, best_tm = s_.tm;                                   \
			bi = al;                                                                             \
			for (int k = al.k0; k < al.k1; k++) best[k * STRIDE] = cols[k * STRIDE]; \
		}                                                                                        \
	}                                                                                            \
	} while (0)

			if (FULL && xmode == 3) {
				// ---------------- enumerate_hairpin_alignments, nuc_cruc.cpp:1264-1405
				if (al.k1 - al.k0 >= 3) CONSIDER(evaluate_hairpin<STRIDE>(c, a.tab, al, QRY_LDS ? nullptr : c.query, qry, L));
				// the open end of the stem always gets its dangling / frayed bases (:1307-1324)
				if (al.lm_t != 0 || al.lm_q != L - 1) {
					int qb, tb;
					if (al.lm_t == 0) tb = bE;
					else tb = QUERY(--al.lm_t);
					if (al.lm_q == L - 1) qb = bE;
					else qb = QUERY(++al.lm_q);
					cols[al.k1 * STRIDE] = mk_col(qb, tb);
					al.k1++;
				}
				const int align_size = al.k1 - al.k0;
				if (align_size < 3) continue;
				CONSIDER(evaluate_hairpin<STRIDE>(c, a.tab, al, QRY_LDS ? nullptr : c.query, qry, L));
				if (align_size <= 3) continue;
				// an A-T closing pair carries a penalty: try without it (:1356-1405)
				const int lp = lut_pair(s_lut, QUERY(al.fm_t), QUERY(al.fm_q));
				if (lp == P_GC || lp == P_CG) continue;
				al.fm_q++;
				al.fm_t--;
				al.k0++;
				CONSIDER(evaluate_hairpin<STRIDE>(c, a.tab, al, QRY_LDS ? nullptr : c.query, qry, L));
				continue;
			}

			// dangling ends, :1088-1137 (off by default, tntblast.h:55-56)
			if (s_k.dangle5 && (al.fm_q != 0 || al.fm_t != W - 1)) {
				int qb, tb;
				if (al.fm_q == 0) qb = bE;
				else qb = QUERY(--al.fm_q);
				if (al.fm_t == W - 1) tb = bE;
				else tb = TARGET(++al.fm_t);
				if (al.k1 == al.k0) al.k0 = al.k1 = 1; // empty alignment: restart inside the buffer
				al.k0--;
				cols[al.k0 * STRIDE] = mk_col(qb, tb);
			}
			if (s_k.dangle3 && (al.lm_q != L - 1 || al.lm_t != 0)) {
				int qb, tb;
				if (al.lm_q == L - 1) qb = bE;
				else qb = QUERY(++al.lm_q);
				if (al.lm_t == 0) tb = bE;
				else tb = TARGET(--al.lm_t);
				cols[al.k1 * STRIDE] = mk_col(qb, tb);
				al.k1++;
			}
			if (al.k1 - al.k0 < 3) continue; // :1148
#ifdef TNT_EXP_STATS
			const long long st_c = __builtin_readcyclecounter();
			st_trim += st_c - st_b;
#endif

#ifndef TNT_EXP_NOEVAL
			CONSIDER(evaluate_alignment<STRIDE>(c, al.k0, al.k1, (FULL && xmode == 2) ? MODE_HOMO_DIMER : MODE_HETERO_DIMER));
#endif
#ifdef TNT_EXP_STATS
			st_eval += __builtin_readcyclecounter() - st_c;
#endif
#undef CONSIDER
		}
	}

	if (DINK) { // the while condition of the Dinkelbach loop, nuc_cruc.cpp:2414-2432
		const float last_q = dk_q;
		dk_q = best_dH - Tcur * best_dS; // delta_G() at this pass's temperature (0 when nothing aligned: curr_align.clear())
		if (dk_q < 0.0f && dk_q > last_q) {
			Tcur = 273.15f + best_tm; // temperature(NC_ZERO_C + local_tm)
			return true;           // fill again at that temperature
		}
	}
#ifdef TNT_EXP_STATS
	const long long st_thr0 = __builtin_readcyclecounter();
#endif
	// ------------------------------------------------------------------ thresholds, bind_oligo.cpp:600-714
	const Thresh th = a.thresh[slot.thresh];
	const float tm = best_tm;
	const float dg = best_dH - s_k.temperature * best_dS; // delta_G(), nuc_cruc.h:1217
	int status = 0;
	if (tm < th.min_tm || tm > th.max_tm) status = 1;
	else if (dg < th.min_dg || dg > th.max_dg) status = 2;

#ifdef TNT_EXP_NOOUT
	a.result[idx] = -1 - (status & 1);
	return false;
#endif
	if (DEFER && a.pending) {
		// What is left runs for the windows that passed both tests only (one in five on config 2): their state goes to the queue of
		// bind_finish_kernel, lanes of a wavefront share one atomic; the others have their status.
		const bool park = status == 0 && best_valid;
		if (!park) {
			a.result[idx] = -(status ? status : 8); // (no valid alignment under thresholds that accept Tm = 0: the Q11 fence)
		} else {
			const unsigned long long m = __builtin_amdgcn_ballot_w64(true); // the lanes that park
			const int me = int(__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)));
			uint32_t at = 0;
			if (me == 0) at = atomicAdd(a.n_pending, uint32_t(__builtin_popcountll(m)));
			at = uint32_t(__shfl(int(at), __builtin_ctzll(m))) + uint32_t(me);
			if (at >= a.pending_cap) {
				atomicOr(a.err, ERR_POOL_OVERFLOW);
				a.result[idx] = -100;
			} else {
				PendingRec &p = a.pending[at];
				p.idx = idx, p.group = it.group, p.start = start, p.stop = stop;
				p.tm = tm, p.dH = best_dH, p.dS = best_dS;
				p.slot = it.info >> ITEM_SLOT_SHIFT;
				p.n_cols = int16_t(bi.k1 - bi.k0), p.fm_q = int16_t(bi.fm_q), p.fm_t = int16_t(bi.fm_t), p.lm_q = int16_t(bi.lm_q), p.lm_t = int16_t(bi.lm_t), p.pad0 = 0;
				p.L = uint8_t(L), p.W = uint8_t(W), p.minus = minus ? 1 : 0, p.pad1 = 0;
				uint32_t *pw = reinterpret_cast<uint32_t *>(p.win), *pc = reinterpret_cast<uint32_t *>(p.cols);
				for (int k = 0; k < W; k += 4) {
					uint32_t v = 0;
#pragma unroll
					for (int u = 0; u < 4; u++) v |= (k + u < W ? uint32_t(win[(k + u) * STRIDE]) : 0u) << (8 * u);
					pw[k >> 2] = v;
				}
				const int nc = bi.k1 - bi.k0;
				for (int k = 0; k < nc; k += 4) {
					uint32_t v = 0;
#pragma unroll
					for (int u = 0; u < 4; u++) v |= (k + u < nc ? uint32_t(best[(bi.k0 + k + u) * STRIDE]) : 0u) << (8 * u);
					pc[k >> 2] = v;
				}
			}
		}
	} else {
		bind_finish<FULL, QRY_LDS, LONG_TEXT, STRIDE>(a, win, qry, c.query, cols, best, bi, best_valid, status, th, tm, best_dH, best_dS, dg, idx, it, L, W, start, stop, minus, gmax, nmax);
	}
#ifdef TNT_EXP_STATS
	{ // wave-level cycle breakdown of the tail (s_memtime ticks / 64, lane 0 of every wave; waves re-converge here)
		const long long st_end = __builtin_readcyclecounter();
		if ((threadIdx.x & 63) == 0) {
			atomicAdd(a.debug + 0, uint32_t((st_tail0 - st_fill0) >> 6));
			atomicAdd(a.debug + 1, uint32_t(st_trace >> 6));
			atomicAdd(a.debug + 2, uint32_t(st_trim >> 6));
			atomicAdd(a.debug + 3, uint32_t(st_eval >> 6));
			atomicAdd(a.debug + 4, uint32_t((st_thr0 - st_tail0 - st_trace - st_trim - st_eval) >> 6));
			atomicAdd(a.debug + 5, uint32_t((st_end - st_thr0) >> 6)); // thresholds + output
			atomicAdd(a.debug + 7, 1u);
		}
	}
#endif
#undef QUERY
#undef TARGET
	return false;
}

// The bind kernel proper.  LMAX = rows kept in registers (oligos of this launch have L <= LMAX).
// EXACT: every item of the launch has L == LMAX (the host buckets the queues by length).
// PLAIN: no window of the launch holds a degenerate (IUPAC) target base (the bucket kernel routes windows that
// do to a general instance), so rt(t, q) = t and the table address splits into column and row constants.
// DINK: --dinkelbach (NucCruc::approximate_tm_*, nuc_cruc.cpp:2399-2432,2459-2498,2548-2593): the alignment is computed
// at 0 C, then again and again at the Tm the last pass gave while its free energy there is negative and still rising;
// every pass has its own delta_g (update_dp_param at that temperature), which a lane derives on the fly from the
// (H, S, floor) triples of the entries (Tables::dk_*).  General instances only.
template <int LMAX, bool FULL, bool EXACT = false, bool PLAIN = false, bool DINK = false>
__global__ __launch_bounds__(bind_threads_for(LMAX, PLAIN), (bind_threads_for(LMAX, PLAIN) > 256 ? 1 : ((LMAX <= 40 && !DINK) ? 2 : 1))) void bind_kernel(BindArgs a)
{
	static_assert(!(FULL && (EXACT || PLAIN)), "the unit-level instance takes any length and any target base");
	static_assert(!(DINK && (EXACT || PLAIN)), "the Dinkelbach passes run on the general instances");
	constexpr int BIND_THREADS = bind_threads_for(LMAX, PLAIN);
	constexpr int WMAX = LMAX + 2 * FLANK;
	constexpr int NCOLS = LMAX + WMAX + 4;

	// delta_g planes.  General instances: plane 0 delta_g(prev, cur), plane 1 term(cur), index ((x'*7 + x)*49 + q'*7 + q),
	// and term[] by pair.  Plain instances only meet x, x' in ACGT (x' = GAP in the first column): a compact index
	// ((x'*4 + x)*49 + q'*7 + q) with GAP stored as x' = 4, a third plane term(x', q) and a fourth with the first
	// column's special case folded in, all under the same index: the four reads of a cell share one address
	// (immediate offsets reach the planes).
	constexpr int DG_PLANE = PLAIN ? 16 * 64 : 38 * 64; // ints per plane, a multiple of 64 for ds_read2st64_b32
	constexpr int DG_PLANES = PLAIN ? 4 : 2;
	__shared__ int32_t s_dgp[DG_PLANES * DG_PLANE];
	__shared__ int32_t s_term[PLAIN ? 1 : NBP];
	__shared__ int32_t s_const[4];
	// DINK: the same planes as triples (value = max(int((H - T * S) * 10000), F)); [plane][entry], then term[49], const[4]
	constexpr int DK_N = DINK ? 2 * DG_PLANE + NBP + 4 : 1;
	__shared__ float s_dkH[DK_N], s_dkS[DK_N];
	__shared__ int32_t s_dkF[DK_N];
	__shared__ uint32_t s_rt[18];
	__shared__ uint8_t s_lut[18 * 18 + 4];
	__shared__ uint8_t s_win[WMAX * BIND_THREADS];
	constexpr bool QRY_LDS = LMAX <= 32;             // larger instances keep the LDS budget for two workgroups per CU
	__shared__ uint8_t s_qry[QRY_LDS ? LMAX * BIND_THREADS : 1]; // oligo bases of every lane, row order (row i-1 = query[L-i])
	__shared__ uint8_t s_cols[NCOLS * BIND_THREADS]; // alignment columns of the path being scored
	constexpr bool BEST_LDS = BIND_THREADS > 256 ? LMAX <= 22 : LMAX <= 28;             // columns of the best path so far: LDS while two workgroups per CU still fit
	__shared__ uint8_t s_best[BEST_LDS ? NCOLS * BIND_THREADS : 1];
	__shared__ uint8_t s_hs_map[NBP * NBP + 3];
	__shared__ float2 s_hs_val[256];
	__shared__ TailConst s_k;
	constexpr int PF = 8;                            // trace words fetched ahead along the traceback's diagonal
	__shared__ uint8_t s_pf[PF * BIND_THREADS];

	const int tid = threadIdx.x;
	// A workgroup without work leaves before it stages the tables: the grid is sized for full queues, the queues behind the
	// pre-filter are short, and the launches of a phase share the CUs with each other (side streams) -- an idle workgroup
	// that fills 40 KB of LDS tables first keeps a CU from the launch next to it.
	{
		uint32_t n_all = a.n_items;
		if (a.n_items_dev) {
			const uint32_t n_dev = *a.n_items_dev;
			n_all = n_dev < n_all ? n_dev : n_all;
		}
		if (uint32_t(blockIdx.x) * uint32_t(BIND_THREADS) >= n_all) return;
	}
	if (PLAIN) {
		for (int at = tid; at < 5 * 4 * NBP; at += BIND_THREADS) {
			const int xpc = at / (4 * NBP), x = (at / NBP) & 3, qp = (at % NBP) / NB, q = at % NB;
			const int xp = xpc == 4 ? bGAP : xpc;
			const int cpair = x * NB + q;
			s_dgp[at] = a.tab->dg[(xp * NB + qp) * NBP + cpair];
			s_dgp[DG_PLANE + at] = a.tab->term[cpair];
			s_dgp[2 * DG_PLANE + at] = a.tab->term[xp * NB + q];
			// what the I_query -> M candidate subtracts: term(cur), but FORB_M in the first column (x' = GAP)
			s_dgp[3 * DG_PLANE + at] = xpc == 4 ? a.tab->dg_const[2] : a.tab->term[cpair];
		}
	} else {
		for (int i = tid; i < NBP * NBP; i += BIND_THREADS) {
			const int p = i / NBP, cpair = i % NBP; // prev pair (x', q'), current pair (x, q)
			const int at = ((p / NB) * NB + cpair / NB) * NBP + (p % NB) * NB + cpair % NB;
			s_dgp[at] = a.tab->dg[i];
			s_dgp[DG_PLANE + at] = a.tab->term[cpair];
		}
		if (tid < NBP) s_term[tid] = a.tab->term[tid];
		if (DINK) {
			auto triple = [&](int at, int src) {
				s_dkH[DINK ? at : 0] = src >= 0 ? a.tab->dk_H[src] : 0.0f;
				s_dkS[DINK ? at : 0] = src >= 0 ? a.tab->dk_S[src] : 0.0f;
				s_dkF[DINK ? at : 0] = src >= 0 ? a.tab->dk_F[src] : 0;
			};
			for (int i = tid; i < NBP * NBP; i += BIND_THREADS) {
				const int p = i / NBP, cpair = i % NBP;
				const int at = ((p / NB) * NB + cpair / NB) * NBP + (p % NB) * NB + cpair % NB;
				triple(at, i);
				triple(DG_PLANE + at, a.tab->term_src[cpair]);
			}
			if (tid < NBP) triple(2 * DG_PLANE + tid, a.tab->term_src[tid]);
			if (tid < 4) triple(2 * DG_PLANE + NBP + tid, a.tab->const_src[tid]);
		}
	}
	if (tid < 4) s_const[tid] = a.tab->dg_const[tid];
	for (int i = tid; i < 18 * 18; i += BIND_THREADS) s_lut[i] = a.tab->bbp[i];
	if (tid < 18) s_rt[tid] = a.tab->rt[tid];
	for (int i = tid; i < NBP * NBP; i += BIND_THREADS) s_hs_map[i] = a.tab->hs_map[i];
	for (int i = tid; i < 256; i += BIND_THREADS) s_hs_val[i] = make_float2(a.tab->hs_H[i], a.tab->hs_S[i]);
	if (tid < 64) s_k.loop_S[tid] = a.tab->loop_S[tid], s_k.bulge_S[tid] = a.tab->bulge_S[tid];
	if (tid == 0) {
		const Tables &t = *a.tab;
		s_k.init_H = t.init_H, s_k.init_S = t.init_S, s_k.sym_S = t.sym_S, s_k.at_H = t.at_H, s_k.at_S = t.at_S;
		s_k.salt = t.salt, s_k.asym_S = t.asym_S, s_k.bulge_at_S = t.bulge_at_S, s_k.temperature = t.temperature, s_k.log_na = t.log_na;
		s_k.dangle5 = t.dangle5, s_k.dangle3 = t.dangle3;
	}
	__syncthreads();

	uint8_t *trace = a.trace + size_t(blockIdx.x) * (size_t(LMAX) * WMAX * BIND_THREADS) + tid;
	uint8_t *cols = s_cols + tid;
	uint8_t *qry = s_qry + tid;
	uint8_t *best = BEST_LDS ? s_best + tid : a.best + size_t(blockIdx.x) * (size_t(NCOLS) * BIND_THREADS) + tid;
	uint32_t *stack = a.stack + size_t(blockIdx.x) * (size_t(NCOLS) * BIND_THREADS) + tid;
	uint16_t *ties = a.ties + size_t(blockIdx.x) * (size_t(TIE_CAP) * BIND_THREADS) + tid;
	uint8_t *win = s_win + tid;

	const uint32_t *index = a.index;
	if (index && a.index_base_dev) index += *a.index_base_dev;
	uint32_t n_items = a.n_items;
	if (a.n_items_dev) {
		const uint32_t n_dev = *a.n_items_dev;
		n_items = n_dev < a.n_items ? n_dev : a.n_items;
	}
	for (uint32_t base = blockIdx.x * BIND_THREADS; base < n_items; base += gridDim.x * BIND_THREADS) {
		uint32_t idx = 0xFFFFFFFFu; // (a queue filled by the seed kernel has slots without an item: QUEUE_HOLE, tnt_seed2.hip.h)
		if (base + tid < n_items) idx = index ? index[size_t(base + tid) * (a.index_stride ? a.index_stride : 1u)] : base + tid;
		const bool active = idx != 0xFFFFFFFFu;
		if (!active) idx = 0;

		// ------------------------------------------------------------------ item decode
		Item it = active ? a.items[idx] : Item{0, 0, 0, 0};
		const Slot slot = a.slots[it.info >> ITEM_SLOT_SHIFT];
		const int L = active ? a.olig_len[slot.oligo] : 0;
		const bool minus = (it.info & ITEM_MINUS) != 0;
		const int qloc = int(it.info & 127u) + slot.qloc_bias;
		const uint32_t tlen = (FULL && (!active || (it.target >> 29))) ? 0u : a.db.len[it.target]; // no target set behind self-structure requests
		// bind_oligo.cpp:502-505
		int start = int(it.tloc) - (qloc + FLANK);
		if (start < 0) start = 0;
		int stop = start + L + 2 * FLANK;
		if (stop > int(tlen)) stop = int(tlen);
		int W = 0;
		// self-structure requests (FULL instance only): bits 29-30 of `target` select the mode, the target is an oligo
		const int xmode = FULL ? int((it.target >> 29) & 3u) : 0; // 0 window of a packed target, 1 heterodimer, 2 homodimer, 3 hairpin
		if (FULL && active && xmode) {
			const int xo = (xmode == 1) ? int(it.target & 0xFFFFFFu) : slot.oligo; // the oligo that plays the target
			W = a.olig_len[xo];
			start = 0, stop = W;
			for (int k = 0; k < W; k++) win[k * BIND_THREADS] = a.olig_base[size_t(xo) * MAX_OLIGO + k]; // set_target / set_duplex
		}

		// ------------------------------------------------------------------ window load (Q6: codes > 15 are skipped)
		if (active && !(FULL && xmode)) {
			const uint32_t *pk = a.db.packed + a.db.word_off[it.target];
			const uint32_t e0 = a.db.exc_off[it.target], e1 = a.db.exc_off[it.target + 1];
			uint32_t e = e0;
			int nskip = 0;
			if (e1 > e0) {
				// first exception at or after `start`
				uint32_t lo = e0, hi = e1;
				while (lo < hi) {
					const uint32_t mid = (lo + hi) >> 1;
					if (a.db.exc_pos[mid] < uint32_t(start)) lo = mid + 1;
					else hi = mid;
				}
				e = lo;
				for (uint32_t x = e; x < e1 && a.db.exc_pos[x] < uint32_t(stop); x++)
					if (a.db.exc_code[x] > 15) nskip++;
			}
			W = (stop - start) - nskip;
			int r = 0;
			for (int pos = start; pos < stop; pos++) {
				int code = int((pk[pos >> 4] >> ((pos & 15) * 2)) & 3u);
				if (e < e1 && a.db.exc_pos[e] == uint32_t(pos)) {
					code = a.db.exc_code[e];
					e++;
				}
				if (code > 15) continue;
				if (minus) win[(W - 1 - r) * BIND_THREADS] = uint8_t(db_comp(code));
				else win[r * BIND_THREADS] = uint8_t(db_same(code));
				r++;
			}
		}

#ifdef TNT_EXP_STATS
		const long long st_fill0 = __builtin_readcyclecounter();
#endif
		// ------------------------------------------------------------------ DP fill (align_dimer, nuc_cruc.cpp:492-696)
		// Column-major sweep, one column per iteration, all LMAX rows unrolled and branch-free: rows
		// beyond L and columns beyond W compute throw-away values (their trace words land in scratch
		// cells nobody reads) and are masked out of the running maximum only.
		//
		// delta_g lookups.  Of the seven table reads per cell of the reference (nuc_cruc.cpp:536-646) only
		// the stack read g(bp(pt,pq), bp(t,q)) needs the full 49x49 table.  update_dp_param (:379-486) gives
		// every other combination the same value for all members of a class, which the host verifies when it
		// builds the tables (tnt_host.hip, fill_tables):
		//   g((x,GAP) or (GAP,x), c) = g(c, (x,GAP) or (GAP,x)) = term(c)       c a real pair: AT-like / GC / mismatch
		//   g((x,GAP),(y,GAP)) = g((GAP,x),(GAP,y)) = BULGE;   g(__, c) = FORB_M;   g((GAP,x),(y,GAP)) = FORB_XY
		//   g(__, (x,GAP) or (GAP,x)) = C0
		// LDS holds int2 {g(prev,cur), term(cur)} per (prev,cur) (byte address prev*392 + cur*8) and term[] per
		// pair (term[(GAP,x)] = FORB_XY so the first column needs no special case).
		//   rt(x, q) = resolve_degenerate(x | q) from the 4-bit-per-partner word s_rt[x]
		uint32_t qpk[(LMAX + 3) / 4];
#pragma unroll
		for (int w = 0; w < (LMAX + 3) / 4; w++) qpk[w] = active ? a.olig_q4[slot.oligo * (MAX_OLIGO / 4) + w] : 0u;
		if (QRY_LDS) {
#pragma unroll
			for (int i = 0; i < LMAX; i++) qry[i * BIND_THREADS] = uint8_t((qpk[i >> 2] >> ((i & 3) * 8 + 2)) & 7u);
		}
		// per-row part of the table address: 4 * (7 * q' + q), q' = base of the row above (GAP above row 1)
		uint32_t Rpk[(LMAX + 3) / 4];
		{
			uint32_t qprev4 = 4u * bGAP;
#pragma unroll
			for (int i = 0; i < LMAX; i++) {
				const uint32_t q4 = (qpk[i >> 2] >> ((i & 3) * 8)) & 255u;
				if ((i & 3) == 0) Rpk[i >> 2] = 0;
				Rpk[i >> 2] |= (7u * qprev4 + q4) << ((i & 3) * 8);
				qprev4 = q4;
			}
		}

		// uniform trip count over the wavefront keeps the unrolled body convergent
		int Wwave = W;
		for (int o = 32; o > 0; o >>= 1) {
			const int other = __shfl_xor(Wwave, o);
			Wwave = other > Wwave ? other : Wwave;
		}
		// temperature of this pass: the target temperature, or the Dinkelbach sequence 0 C, Tm_1, Tm_2, ...
		float Tcur = DINK ? 273.15f : s_k.temperature;
		float dk_q = -999999.9f;
		auto dkv = [&](int at) __attribute__((always_inline)) { // delta_g entry `at` of the triple planes at Tcur
			const int v = int((s_dkH[DINK ? at : 0] - Tcur * s_dkS[DINK ? at : 0]) * 10000.0f);
			const int f = s_dkF[DINK ? at : 0];
			return v > f ? v : f;
		};
	dinkelbach_pass:
		int oM[LMAX], oQ[LMAX], oT[LMAX];
#pragma unroll
		for (int i = 0; i < LMAX; i++) oM[i] = FULL ? -1 : 0, oQ[i] = oT[i] = 0;

		int gmax = -1, nmax = 0, max_ij = 0;
		uint32_t prtw = s_rt[bGAP];
		int pt = bGAP; // target base of the previous column
		const char *sdg = reinterpret_cast<const char *>(s_dgp);
		const char *sterm = reinterpret_cast<const char *>(s_term);
		const int c_bulge = DINK ? dkv(2 * DG_PLANE + NBP + 0) : s_const[0], c_zero = DINK ? dkv(2 * DG_PLANE + NBP + 1) : s_const[1],
		          c_forb_m = DINK ? dkv(2 * DG_PLANE + NBP + 2) : s_const[2], c_forb_xy = DINK ? dkv(2 * DG_PLANE + NBP + 3) : s_const[3];
		uint8_t *tcol = trace;
#pragma unroll 1
		for (int j = 1; j <= Wwave; j++) {
			const bool colok = j <= W;
			const bool firstcol = j == 1; // wave-uniform
			const int t = colok ? int(win[(j - 1) * BIND_THREADS]) : bGAP;
			const uint32_t rtw = s_rt[t];
			const int v5 = firstcol ? c_zero : c_bulge;          // g(bp(pt,GAP), bp(t,GAP))
			int bM = FULL ? -1 : 0, bT = 0; // border cells hold -1 (nuc_cruc.h:531-536), kept clamped like the gap states
			int Lcol = colok ? L : 0;         // rows that count in this column
			if (FULL && xmode == 3) Lcol = (L - 3 - j) > 0 ? (L - 3 - j) : 0; // align_hairpin's triangle, nuc_cruc.cpp:794-810
			int colmax = INT32_MIN, held = INT32_MIN;
			int term_up = 0;                                     // term(cur) of the row above
			// keep the per-row constants from being hoisted out of the column loop (register pressure)
#pragma unroll
			for (int w = 0; w < (LMAX + 3) / 4; w++) {
				if (PLAIN) asm volatile("" : "+v"(Rpk[w]));
				else asm volatile("" : "+v"(qpk[w]), "+v"(Rpk[w]));
			}
			// Table index (s_dgp): ((x' * 7 + x) * 49 + q' * 7 + q with prev pair (x', q') and current pair (x, q),
			// x = rt(t, q).  Without degenerate target bases (x = t, x' = pt) the address splits into a per-column
			// and a per-row constant, and the compact plain layout puts term(x', q) under the same index: one add
			// per cell serves all three reads.
			{
				int colbase = PLAIN ? lds_addr(s_dgp) + ((pt > 3 ? 4 : pt) * 4 + (t & 3)) * (4 * NBP) : 0; // (t & 3: columns beyond W)
				int xp = int((prtw >> 24) & 15u); // rt(pt, GAP): row 0 of the previous column
				// reads of row k (their addresses depend on the bases only)
				auto issue = [&](const int k, int &ox, int &oy, int &ot, int &o2) __attribute__((always_inline)) {
					const int q4 = int((qpk[(k - 1) >> 2] >> (((k - 1) & 3) * 8)) & 255u), r4 = int((Rpk[(k - 1) >> 2] >> (((k - 1) & 3) * 8)) & 255u);
					if (PLAIN) {
						const int ea = colbase + r4;
						ox = lds_load(ea), oy = lds_load(ea + 4 * DG_PLANE), ot = lds_load(ea + 8 * DG_PLANE), o2 = lds_load(ea + 12 * DG_PLANE);
					} else {
						const int x = int(__builtin_amdgcn_ubfe(rtw, uint32_t(q4), 4u));
						const int ea = xp * (4 * 343) + x * (4 * 49) + r4; // xp = rt(pt, q of the row above)
						xp = int(__builtin_amdgcn_ubfe(prtw, uint32_t(q4), 4u));
						const int ta = xp * 28 + q4;
						if (DINK) {
							ox = dkv(ea >> 2), oy = dkv(DG_PLANE + (ea >> 2)), ot = dkv(2 * DG_PLANE + (ta >> 2));
						} else {
							ox = *reinterpret_cast<const int *>(sdg + ea);
							oy = *reinterpret_cast<const int *>(sdg + ea + 4 * DG_PLANE);
							ot = *reinterpret_cast<const int *>(sterm + ta);
						}
						o2 = firstcol ? c_forb_m : oy;
					}
				};
				// The reads run two rows ahead: row i also forms the diagonal candidates of row i + 1 (they need the
				// old states of row i, which die there -- the new states can take their registers, no copies at the
				// loop's back edge) and so needs row i + 1's table values.
				int nx_x, nx_y, nx_t, nx_2, n2_x = 0, n2_y = 0, n2_t = 0, n2_2 = 0;
				issue(1, nx_x, nx_y, nx_t, nx_2);
				if (LMAX >= 2) issue(2, n2_x, n2_y, n2_t, n2_2);
				int dg1 = 0 - nx_x;                             // row 1: the border row above holds max(0, -1) = 0, 0, 0
				int dg2 = 0 - nx_2;
				int dg3 = 0 - c_forb_m;
#pragma unroll
				for (int i = 1; i <= LMAX; i++) {
					const int termv = nx_y, term_p4 = nx_t;
					nx_x = n2_x, nx_y = n2_y, nx_t = n2_t, nx_2 = n2_2;
					// the reads stay in row order: without this fence the scheduler hoists every row's (independent)
					// reads to the top of the column and the results do not fit the register budget
					if (PLAIN && i > 1) asm volatile("" : "+v"(colbase) : "v"(bM));
					if (i + 2 <= LMAX) issue(i + 2, n2_x, n2_y, n2_t, n2_2);
					// oQ / oT are stored clamped at zero; so is oM in the throughput instances (the unit-level instance
					// keeps M itself: it reports dp_max / n_max also when no cell is positive)
					const int cM = oM[i - 1], cQ = oQ[i - 1], cT = oT[i - 1];
					const int cMz = FULL ? (cM > 0 ? cM : 0) : cM;
					int dn1 = 0, dn2 = 0, dn3 = 0;
					if (i < LMAX) dn1 = cMz - nx_x, dn2 = cQ - nx_2, dn3 = cT - nx_y;
					int iqi = cMz - term_p4;
					int iqe = cQ - v5;
					// the old states of this row are dead from here on
					asm volatile("" : "+v"(dn1), "+v"(dn2), "+v"(dn3), "+v"(iqi), "+v"(iqe), "+v"(dg1));

					// reference if-chain (nuc_cruc.cpp:552-597) == max3 with "equal to the maximum" trace bits.
					// The trace word is built as a shift register, one v_alignbit (sign bit) per flag; equalities are
					// taken as "not less than the maximum" (a full-rate subtract feeds the sign).  Layout: TW_* in tnt_dev.h.
					int M = dg1 > dg2 ? dg1 : dg2;
					M = M > dg3 ? M : dg3;
					const int Iq = iqi > iqe ? iqi : iqe;
					const int iti = (FULL ? (bM > 0 ? bM : 0) : bM) - (i == 1 ? c_forb_xy : term_up);
					const int ite = bT - (i == 1 ? c_zero : c_bulge);
					const int It = iti > ite ? iti : ite;

					// Every flag is taken against the value clamped at zero (which the next column needs anyway): a
					// negative state sets ALL "not from" flags of that state, a combination a maximum cannot produce,
					// so "state < 0" costs no bit of its own and the word fits one byte.
					const int Mz = M > 0 ? M : 0, Iqz = Iq > 0 ? Iq : 0, Itz = It > 0 ? It : 0;
					uint32_t tw;
					tw = sign_bit(dg1 - Mz);                                     // set: M did not come from M        (im1_jm1)
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(dg3 - Mz), 31);  // set: M did not come from I_target (im1_j)
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(dg2 - Mz), 31);  // set: M did not come from I_query  (i_jm1)
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(iqi - Iqz), 31); // insert < extend: I_query only from I_query
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(iqe - Iqz), 31); // extend < insert: I_query only from M
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(iti - Itz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(ite - Itz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(Mz - 1), 31);    // M <= 0
#ifdef TNT_EXP_NOSTORE
					asm volatile("" ::"v"(tw));
#else
					tcol[(i - 1) * BIND_THREADS] = uint8_t(tw);
#endif

					// column maximum over the valid rows; the running maximum is updated once per column below
					if (EXACT) {
						if (i & 1) {
							if (i == LMAX) colmax = colmax > M ? colmax : M;
							else held = M;
						} else {
							const int m2 = held > M ? held : M; // one v_max3 per two rows
							colmax = colmax > m2 ? colmax : m2;
						}
					} else {
						const int Mv = (i <= Lcol) ? M : INT32_MIN;
						colmax = colmax > Mv ? colmax : Mv;
					}

					const int Mc = FULL ? M : Mz;
					dg1 = dn1, dg2 = dn2, dg3 = dn3;
					bM = Mc, bT = Itz;
					oM[i - 1] = Mc, oQ[i - 1] = Iqz, oT[i - 1] = bT;
					term_up = termv;
				}
			}
			if (EXACT && !colok) colmax = INT32_MIN;
			// Running maximum, nuc_cruc.cpp:670-691.  The reference visits cells row-major; here the visiting
			// order is column-major, rows ascending.  The SET of cells holding the final maximum does not depend
			// on the order; it is collected as (i << 8 | j) codes and sorted row-major before the traceback.
			// (throughput instances compare the clamped copies: only positive maxima are collected, a window
			// without a positive cell has no alignment of three columns and is rejected either way)
			if (colmax >= gmax && (FULL || colmax > 0)) {
#pragma unroll
				for (int i = 1; i <= LMAX; i++) {
					if ((EXACT || i <= Lcol) && oM[i - 1] == colmax) {
						const int code = (i << 8) | j;
						if (colmax > gmax) {
							gmax = colmax;
							nmax = 1;
							max_ij = code;
						} else {
							if (nmax == 0) max_ij = code;
							else if (nmax < TIE_CAP) ties[nmax * BIND_THREADS] = uint16_t(code);
							nmax++;
						}
					}
				}
			}
			prtw = rtw;
			pt = t;
			tcol += LMAX * BIND_THREADS;
		}

		// ------------------------------------------------------------------ serial tail
		if (!active) continue;
#ifdef TNT_EXP_NOTAIL
		a.result[idx] = -(gmax & 1) - 1 - (nmax & 1) - (max_ij & 1);
		continue;
#endif

		{
			SlabTrace<LMAX, BIND_THREADS, PF> tr{trace, s_pf + tid, -1, -1};
			if (bind_tail<FULL, DINK, QRY_LDS, (LMAX > 64), BIND_THREADS, 8, uint16_t, SlabTrace<LMAX, BIND_THREADS, PF>, (!FULL && !DINK && LMAX <= 64)>(a, tr, s_lut, s_hs_map, s_hs_val, s_k, win, qry, cols, best, stack, ties, NCOLS, WMAX + 1, idx, it,
			                                                                           slot, L, W, start, stop, minus, xmode, gmax, nmax, max_ij, Tcur, dk_q
#ifdef TNT_EXP_STATS
			                                                                           , st_fill0
#endif
			                                                                           ))
				goto dinkelbach_pass;
		}
	}
}

// The windows the throughput instances parked behind the Tm / dG tests (BindArgs::pending), one lane each: anchors, counts,
// the other thresholds, footprint, bound record, alignment text -- bind_finish, with every lane at work.  One launch per bind
// phase, whatever the oligo lengths (a record carries its own).  Compiled once, in tnt_host.hip's translation unit.
constexpr int FIN_THREADS = 256;
#ifdef TNT_FINISH_KERNEL
__global__ __launch_bounds__(FIN_THREADS) void bind_finish_kernel(BindArgs a)
{
	__shared__ uint8_t s_win[PENDING_WIN * FIN_THREADS];   // [k][thread]: the layout bind_finish reads
	__shared__ uint8_t s_cols[PENDING_COLS * FIN_THREADS];
	const int tid = threadIdx.x;
	uint32_t n = *a.n_pending;
	n = n < a.pending_cap ? n : a.pending_cap;
	for (uint32_t base = blockIdx.x * FIN_THREADS; base < n; base += gridDim.x * FIN_THREADS) {
		const uint32_t i = base + uint32_t(tid);
		if (i >= n) continue; // (no barrier below: every lane works on slices of its own)
		const PendingRec &p = a.pending[i];
		const int L = p.L, W = p.W, nc = p.n_cols;
		const uint32_t *pw = reinterpret_cast<const uint32_t *>(p.win), *pc = reinterpret_cast<const uint32_t *>(p.cols);
		for (int k = 0; k < W; k += 4) {
			const uint32_t v = pw[k >> 2];
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (k + u < PENDING_WIN) s_win[(k + u) * FIN_THREADS + tid] = uint8_t(v >> (8 * u));
		}
		for (int k = 0; k < nc; k += 4) {
			const uint32_t v = pc[k >> 2];
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (k + u < PENDING_COLS) s_cols[(k + u) * FIN_THREADS + tid] = uint8_t(v >> (8 * u));
		}
		const Slot slot = a.slots[p.slot];
		const Thresh th = a.thresh[slot.thresh];
		Item it = Item{0, 0, p.slot << ITEM_SLOT_SHIFT, p.group};
		const AlignInfo bi = {0, nc, p.fm_q, p.fm_t, p.lm_q, p.lm_t};
		const float dg = p.dH - a.tab->temperature * p.dS; // delta_G(), nuc_cruc.h:1217
		bind_finish<false, false, false, FIN_THREADS>(a, s_win + tid, nullptr, a.olig_base + size_t(slot.oligo) * MAX_OLIGO, s_cols + tid, s_cols + tid, bi, true, 0, th, p.tm, p.dH,
		                                              p.dS, dg, p.idx, it, L, W, p.start, p.stop, p.minus != 0, 0, 0);
	}
}
#endif

// host-side launchers of the throughput instances, defined in tnt_bind_inst.hip
#define X(N) void launch_bind_##N(const BindArgs &a, int blocks, hipStream_t stream);
TNT_LMAX_LADDER(X)
TNT_LONG_LADDER(X)
#undef X
#define X(N) void launch_bind_plain_##N(const BindArgs &a, int blocks, hipStream_t stream);
TNT_LMAX_LADDER(X)
TNT_LONG_LADDER(X)
#undef X
#define X(N) void launch_bind_exact_##N(const BindArgs &a, int blocks, hipStream_t stream);
TNT_EXACT_LADDER(X)
#undef X
// the wavefront-per-window form (tnt_wbind.hip.h, instances in tnt_wbind_inst.hip): N = query rows per lane, oligos of up
// to 64 * N bases; `full` = the unit-level form (any request kind, dp_max / n_max reported)
#define TNT_WBIND_LADDER(X) X(1) X(2) X(4) X(8) X(16)
#define X(N) void launch_wbind_##N(const BindArgs &a, int blocks, hipStream_t stream); void launch_wbind_full_##N(const BindArgs &a, int blocks, hipStream_t stream); \
	void launch_wbind_dink_##N(const BindArgs &a, int blocks, bool full, hipStream_t stream);
TNT_WBIND_LADDER(X)
#undef X
constexpr int wb_waves(int R) { return R == 1 ? 4 : (R == 2 ? 2 : 1); }    // wavefronts (= windows in flight) per workgroup
constexpr bool wb_trace_lds(int R) { return R <= 2; }                      // the trace words of a window fit LDS (beyond: a slab per wavefront in HBM)
constexpr size_t wb_slab_bytes(int R) { return wb_trace_lds(R) ? 0 : size_t(64 * R) * size_t(64 * R + 2 * FLANK); }
constexpr int wb_rows_per_lane(int rows) { return rows <= 64 ? 1 : (rows <= 128 ? 2 : (rows <= 256 ? 4 : (rows <= 512 ? 8 : 16))); }
constexpr int wb_blocks_per_cu(int R) { return R == 1 ? 3 : 2; } // resident workgroups (LDS: 54 KB at R = 1, 67 KB at R = 2, 30 to 44 KB beyond: the trace words in HBM)

} // namespace tnt
#endif
