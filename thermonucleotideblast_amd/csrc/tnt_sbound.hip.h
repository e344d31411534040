// tnt_sbound.hip.h -- gfx950 second tier of the bind stage's pre-filter (round 5): a bound over the alignments align_dimer
// can actually PRODUCE, from a trace-free fill of its own recurrence.
//
// The first tier (tnt_filter.hip.h / tnt_ufilter.hip.h) bounds the stability of ANY alignment of a window.  Four of five
// windows it lets through still fail the Tm threshold: its slack lies in alignments evaluate_alignment prices well and the DP
// never chooses (a mismatch costs nothing once the pair next to it may be left unpaired inside a free symmetric loop --
// DESIGN.md 4.2c).  NucCruc does not score any alignment, it scores the MAXIMAL-SCORE paths of align_dimer
// (nuc_cruc.cpp:492-696: tm_dimer walks back from every maximal cell, nuc_cruc.cpp:2533-2537, trace_back follows the
// predecessors that achieved each cell's maximum, ties included).  So:
//
//   credit(A) := what the first tier's argument grants an alignment along ITS OWN path -- per diagonal step the stack of the two
//                columns when one of them is Watson-Crick and the stack is favourable; at an aligned column next to a gap run
//                the stack with its diagonally adjacent cell (the terminal stack of the loop, nuc_cruc.cpp:1923-2077) or the
//                stack across a single-base bulge; every loop / bulge penalty dropped.  Phi(A) - const <= credit(A).
//   bound     := max { credit(A) : A a maximal-score path of the fill }  >=  Phi(the alignment NucCruc scores) - const.
//
// Both live in ONE integer per DP state: V = score * 2^11 + credit, score in the fill's own units (1e-4 kcal/mol, exact),
// credit in units of 0.02 kcal/mol rounded up.  The recurrence is align_dimer's with every table entry replaced by
// (-delta_g) * 2^11 + credit: a maximum over predecessors picks the higher score and, among equal scores, the higher credit --
// exactly the set of predecessors trace_back would follow.  A state below zero restarts at zero as in the fill (max(0, .)).
// The window's maximum V gives the maximal score and the largest credit any maximal-score path ends with; the window goes on
// to the bind kernel iff that credit reaches the first tier's threshold.  Credits only grow along a path and a step adds less
// than 2^10 units, so bit 10 of the credit field is a guard: once it is set in any state the window passes (20.48 kcal/mol is
// above every threshold this tier is enabled for) and what the sums do afterwards is ignored.
//
// No trace words, no traceback, no scoring: 13 VALU instructions and three LDS reads at immediate offsets per cell against the
// fill's 36 and nothing behind them; the first tier hands its survivors over group by group, so a wavefront holds ONE oligo's
// table (sbound_kernel: plain windows -- A/C/G/T targets -- of oligos of up to 40 bases in the regrouped buckets).  Windows with an
// IUPAC code or an inosine take sbound_degen_kernel (the general buckets of phase A: a column's code resolved against the query
// base of every row, one table for all oligos); everything else goes from the first tier straight to the bind kernel.
// tests: tests/test_filter_bound.py (soundness of the oracle-side restatement, tier 3), tests/test_gpu_parity.py
// (kernels == restatement; searches with the tier on / off / in audit mode).
#ifndef TNT_SBOUND_HIP_H
#define TNT_SBOUND_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tnt_dev.h"
#include "tnt_filter.hip.h"

namespace tnt {

constexpr int SB_SHIFT = 11;                 // V = score << SB_SHIFT | credit
constexpr int SB_CREDIT_MASK = (1 << SB_SHIFT) - 1;
constexpr int SB_GUARD = 1 << (SB_SHIFT - 1); // credit >= 20.48 kcal/mol: the window passes
constexpr int SB_UNITS = 50;                 // credit units per kcal/mol (0.02 kcal/mol each)
constexpr int SB_X = 5;                      // target base of a column: A C G T, 4 = none (before the first / behind the last column)
constexpr int SB_ENTRIES = SB_X * SB_X * NBP; // index ((x' * 5 + x) * 49 + q' * 7 + q): diagonal predecessor (x', q'), cell (x, q)
constexpr int SB_MAX_ROWS = 40;              // instances (score < 2^20 units for oligos of up to 40 bases: the host checks the table)
constexpr int SB_NEG = -(1 << 30);           // a forbidden transition
constexpr int SB_THREADS = 256, SB_WAVES = SB_THREADS / 64;
constexpr int SB_MAX_BINS = 1024;            // groups of one launch (their block counts are prefix-summed in LDS)
constexpr uint32_t SB_BLOCK = 64;            // survivors of one group a wavefront takes at a time
constexpr uint32_t SB_CHUNK = 1;             // blocks a wavefront grabs per atomic (a bucket has a few thousand blocks for the device's three thousand wavefronts: taken one at a time)
constexpr uint32_t SB_HOLE = 0xFFFFFFFFu;
// column index of (x', x): the sixteen pairs of real bases first, so that the lanes of a wavefront -- which differ in nothing but
// their column -- read sixteen different bank groups (the pitch of a column is an odd number of 16-byte entries)
constexpr int SB_COLS = SB_X * SB_X;
__host__ __device__ constexpr int sb_col(int xp, int x) { return (xp < 4 && x < 4) ? xp * 4 + x : (xp == 4 ? 16 + x : 21 + xp); }
// Per-oligo table of a wavefront, two arrays: T12[column][row] = {diagonal step, gap state -> aligned column} (8 bytes, one
// ds_read_b64: banks modulo 64 in pairs, column pitch an odd number of pairs) and T3[column][row] = {aligned column -> gap state}
// (4 bytes, ds_read_b32: banks modulo 32, column pitch twice an odd number of dwords): the sixteen columns of real base pairs
// fall on sixteen different bank pairs / even banks, whatever the row.
constexpr int sb_pitch12(int lmax) { return ((lmax + 1) | 1) * 8; }                          // bytes
constexpr int sb_pitch3(int lmax) { return (((lmax + 1 + 1) / 2) | 1) * 2 * 4; }             // bytes
constexpr int sb_waves_per_simd(int lmax) { return lmax <= 25 ? 4 : (lmax <= 37 ? 3 : 2); } // (registers: three per row; LDS: four per-oligo tables per workgroup)

// One 16-byte entry per (x', x, q', q), read with ONE ds_read_b96: [0] the diagonal step, [1] into an aligned column from a gap
// state, [2] from an aligned column into a gap state (read at the NEXT row's / the NEXT column's entry), [3] unused.
struct SBoundTables {
	int32_t entry[FILT_CLASSES][SB_ENTRIES][4];
	int32_t bulge, zero, forb_m, forb_xy;    // the fill's class constants (Tables::dg_const), times 2^11
	int32_t enabled[FILT_CLASSES];           // the class's bound is taken by this tier too (thresholds below the guard, scores that fit)
};

struct SBoundArgs {
	const SBoundTables *tab;
	const FilterTables *ftab;     // thresholds of the first tier (theta0, tk, ct_scale, enabled)
	DbView db;
	const uint32_t *olig_q4;      // [n][MAX_OLIGO / 4] reversed oligo, one byte per row = 4 * base code
	const int32_t *olig_len;
	const Slot *slots;
	const Item *items;
	// The first tier's survivors BY GROUP (one oligo and threshold class each): the survivors of group g are
	// index[bin_end[g] - bin_count[g] ...], bin_fill[g] of them (ufilter_kernel wrote them to the front of the group's run of the
	// regrouped queue; bin_count null: bin_end holds the first index itself).
	const uint32_t *index;
	const uint32_t *bin_end, *bin_count, *bin_fill;
	uint32_t n_bins;
	uint32_t *next_work;          // work counter of the launch (zeroed per batch)
	uint32_t *survivors;          // out: the windows that go on to the bind kernel (one dense list)
	uint32_t *n_survivors;
	int32_t *result;              // rejected windows: -1 (Tm below the minimum, bind_oligo.cpp:600-606)
	uint32_t *n_rejected;         // statistics
	uint8_t *audit;               // optional: audit mode marks what this tier would reject and lets everything through
	int32_t *score_out;           // optional (unit-level entry): [2 * idx] credit in 1e-3 kcal/mol (INT32_MAX: guard), [2 * idx + 1] threshold
	int32_t *dp_out;              // optional (unit-level entry): [idx] the maximal score of the fill
};

// One wavefront = 64 survivors of ONE oligo (as in ufilter_kernel): its table sits in LDS as T[column (x', x)][row], an entry per
// row -- the read of a cell is an immediate offset on the column's base register, and the lanes differ in the column only.
// EXACT: every oligo of the launch has exactly LMAX bases (the exact-length buckets; the instance of 16 rows serves shorter ones).
template <int LMAX, bool EXACT> __global__ __launch_bounds__(SB_THREADS, sb_waves_per_simd(LMAX)) void sbound_kernel(SBoundArgs a)
{
	static_assert(LMAX <= SB_MAX_ROWS, "instances of up to 40 rows");
	constexpr int WMAX = LMAX + 2 * FLANK;
	constexpr int NW = (WMAX + 15) / 16;
	constexpr int PITCH12 = sb_pitch12(LMAX), PITCH3 = sb_pitch3(LMAX); // bytes
	static_assert(PITCH12 >= (LMAX + 1) * 8 && PITCH3 >= (LMAX + 1) * 4 && ((PITCH12 / 8) & 1) && ((PITCH3 / 8) & 1), "column pitches");
	typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
	typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
	__shared__ __attribute__((aligned(16))) int32_t s_tab[SB_WAVES][SB_COLS * (PITCH12 + PITCH3) / 4];
	__shared__ uint32_t s_pref[SB_MAX_BINS + 1];  // blocks of the groups before group g
	__shared__ uint32_t s_part[SB_THREADS];
	__shared__ uint32_t s_keep[SB_WAVES][128];
	__shared__ uint32_t s_wcount[SB_WAVES], s_wbase;
	typedef __attribute__((address_space(3))) const int32_t lds_ci32;
	typedef __attribute__((address_space(3))) const volatile i32x2 lds_ci32x2; // (volatile: not merged into ds_read2_b64, which banks modulo 32)

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// ---- the launch's work: blocks of 64 survivors, group after group
	{
		const uint32_t n_bins = a.n_bins < uint32_t(SB_MAX_BINS) ? a.n_bins : uint32_t(SB_MAX_BINS);
		constexpr uint32_t PER = SB_MAX_BINS / SB_THREADS;
		uint32_t mine[PER], sum = 0;
#pragma unroll
		for (uint32_t k = 0; k < PER; k++) {
			const uint32_t g = uint32_t(tid) * PER + k;
			mine[k] = g < n_bins ? (a.bin_fill[g] + SB_BLOCK - 1) / SB_BLOCK : 0u;
			sum += mine[k];
		}
		s_part[tid] = sum;
		__syncthreads();
		for (int d = 1; d < SB_THREADS; d <<= 1) { // inclusive scan
			const uint32_t v = tid >= d ? s_part[tid - d] : 0u;
			__syncthreads();
			s_part[tid] += v;
			__syncthreads();
		}
		uint32_t run = s_part[tid] - sum;
#pragma unroll
		for (uint32_t k = 0; k < PER; k++) {
			s_pref[uint32_t(tid) * PER + k] = run;
			run += mine[k];
		}
		if (tid == SB_THREADS - 1) s_pref[SB_MAX_BINS] = run;
		__syncthreads();
	}
	const uint32_t n_work = s_pref[SB_MAX_BINS];
	const int c_bulge = a.tab->bulge, c_zero = a.tab->zero, c_forb_m = a.tab->forb_m, c_forb_xy = a.tab->forb_xy;
	uint32_t n_rej = 0, n_keep = 0;
	int cur_oligo = -1, cur_cls = -1;
	int32_t *tabw = s_tab[wave];
	const int tab_base = int(uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)tabw)));
	const int tab3_base = tab_base + SB_COLS * PITCH12;
	auto rd2 = [&](int addr) __attribute__((always_inline)) {
		const i32x2 v = *(lds_ci32x2 *)(uintptr_t(uint32_t(addr)));
		return v;
	};
	auto rd = [&](int addr) __attribute__((always_inline)) { return *(lds_ci32 *)(uintptr_t(uint32_t(addr))); };

	for (;;) {
		uint32_t w0 = 0;
		if (lane == 0) w0 = atomicAdd(a.next_work, SB_CHUNK);
		w0 = uint32_t(__builtin_amdgcn_readfirstlane(int(w0)));
		if (w0 >= n_work) break;
		const uint32_t w1 = w0 + SB_CHUNK < n_work ? w0 + SB_CHUNK : n_work;
		// the group of block w0: the last one whose first block is not behind it
		uint32_t g = 0;
		{
			uint32_t lo = 0, hi = a.n_bins < uint32_t(SB_MAX_BINS) ? a.n_bins : uint32_t(SB_MAX_BINS);
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_pref[mid] <= w0) lo = mid;
				else hi = mid;
			}
			g = lo;
		}
		for (uint32_t w = w0; w < w1; w++) {
			while (s_pref[g + 1] <= w) g++; // (groups without survivors have no block)
			const uint32_t fill = a.bin_fill[g], first = a.bin_end[g] - (a.bin_count ? a.bin_count[g] : 0u);
			const uint32_t at = (w - s_pref[g]) * SB_BLOCK + uint32_t(lane);
			const uint32_t idx = at < fill ? a.index[first + at] : SB_HOLE;
			const bool active = idx != SB_HOLE;
			const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
			if (am == 0) continue;
			const Item it = active ? a.items[idx] : Item{0, 0, 0, 0};
			// the block's oligo and threshold class: those of its first survivor (a group holds one (oligo, class))
			const int lead = __builtin_ctzll(am);
			const uint32_t lead_slot = uint32_t(__builtin_amdgcn_readlane(int(it.info), lead)) >> ITEM_SLOT_SHIFT;
			const Slot ls = a.slots[lead_slot];
			const int oligo = __builtin_amdgcn_readfirstlane(ls.oligo), cls = __builtin_amdgcn_readfirstlane(filter_class(ls.thresh));
			const int L = EXACT ? LMAX : __builtin_amdgcn_readfirstlane(a.olig_len[oligo]);
			if (oligo != cur_oligo || cls != cur_cls) {
				// T[column][i] = entry[cls][(x' * 5 + x) * 49 + 7 * q' + q] for i = 0 .. LMAX: q the base of row i + 1 (none behind the
				// oligo: the row behind the last one is read for the credit of a gap opened in the last row), q' the base of the row above
				// (none above the first)
				filt_wave_sync(); // (the lanes' reads of the last block's table are through)
				const i32x4 *src = reinterpret_cast<const i32x4 *>(&a.tab->entry[cls][0][0]);
				const uint32_t *q4 = a.olig_q4 + size_t(oligo) * (MAX_OLIGO / 4);
#pragma unroll 4
				for (int e0 = 0; e0 < SB_COLS * (LMAX + 1); e0 += 64) { // (unrolled by four: the reads of four rounds are in flight together)
					const int e = e0 + lane;
					if (e >= SB_COLS * (LMAX + 1)) break;
					const int xx = e / (LMAX + 1), i = e - xx * (LMAX + 1);
					const int xp = xx / SB_X, x = xx - xp * SB_X;
					uint32_t q = bGAP, qp = bGAP;
					if (i < L) q = (q4[i >> 2] >> ((i & 3) * 8 + 2)) & 7u;
					if (i >= 1 && i - 1 < L) qp = (q4[(i - 1) >> 2] >> (((i - 1) & 3) * 8 + 2)) & 7u;
					const i32x4 v = src[xx * NBP + int(7u * qp + q)];
					const int cc = sb_col(xp, x);
					*reinterpret_cast<i32x2 *>(reinterpret_cast<char *>(tabw) + cc * PITCH12 + i * 8) = i32x2{v.x, v.y};
					*reinterpret_cast<int32_t *>(reinterpret_cast<char *>(tabw) + SB_COLS * PITCH12 + cc * PITCH3 + i * 4) = v.z;
				}
				cur_oligo = oligo, cur_cls = cls;
				filt_wave_sync();
			}
			const Slot slot = a.slots[active ? it.info >> ITEM_SLOT_SHIFT : lead_slot];
			const bool minus = (it.info & ITEM_MINUS) != 0;
			// the window, bind_oligo.cpp:502-505 (plain windows only: no exception inside)
			int start = int(it.tloc) - (int(it.info & 127u) + slot.qloc_bias + FLANK);
			if (start < 0) start = 0;
			int stop = start + L + 2 * FLANK;
			const int tlen = active ? int(a.db.len[it.target]) : 0;
			if (stop > tlen) stop = tlen;
			const int W = active ? stop - start : 0;
			// ... in column order, 2 bits per column (ufilter_kernel's form): column j = target[j - 1] as NucCruc holds it
			uint32_t ov[NW];
			{
				const uint32_t *pk = a.db.packed + (active ? a.db.word_off[it.target] : 0ull) + uint32_t(start >> 4);
				uint32_t raw[NW + 1];
#pragma unroll
				for (int k = 0; k <= NW; k++) raw[k] = active ? pk[k] : 0u;
				const uint32_t sh = uint32_t(start & 15) * 2u;
#pragma unroll
				for (int k = 0; k < NW; k++) ov[k] = __builtin_amdgcn_alignbit(raw[k + 1], raw[k], sh);
				if (__builtin_amdgcn_ballot_w64(minus) != 0) {
					uint32_t rv[NW + 1];
#pragma unroll
					for (int k = 0; k < NW; k++) {
						const uint32_t r = __builtin_bitreverse32(ov[NW - 1 - k]);
						rv[k] = ~(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
					}
					rv[NW] = 0;
					const uint32_t slack = uint32_t(16 * NW - W) * 2u;
#pragma unroll
					for (int step = NW - 1; step >= 1; step--) {
						const bool drop = (slack >> 5) >= uint32_t(step);
#pragma unroll
						for (int k = 0; k < NW; k++) rv[k] = drop ? rv[k + 1] : rv[k];
					}
					const uint32_t bits = slack & 31u;
#pragma unroll
					for (int k = 0; k < NW; k++) {
						const uint32_t v = __builtin_amdgcn_alignbit(rv[k + 1], rv[k], bits);
						ov[k] = minus ? v : ov[k];
					}
				}
			}

			int Wwave = W;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const int other = __shfl_xor(Wwave, o);
				Wwave = other > Wwave ? other : Wwave;
			}
			Wwave = __builtin_amdgcn_readfirstlane(Wwave);

			// states of the previous column, clamped at zero (what the next column reads; the border holds -1: zero)
			int oM[LMAX], oQ[LMAX], oT[LMAX];
#pragma unroll
			for (int i = 0; i < LMAX; i++) oM[i] = oQ[i] = oT[i] = 0;
			int vmax = 0, sticky = 0;
			auto base_of = [&](int j) __attribute__((always_inline)) { // target base of column j (1-based), SB_X - 1 outside the window
				uint32_t wd = ov[0];
#pragma unroll
				for (int k = 1; k < NW; k++) wd = ((j - 1) >> 4) == k ? ov[k] : wd;
				const int b = int((wd >> (((j - 1) & 15) * 2)) & 3u);
				return (j >= 1 && j <= W) ? b : SB_X - 1;
			};
			auto col_of = [&](int xp, int x) __attribute__((always_inline)) { // sb_col, with selects
				const int inner = xp * 4 + x, edge = xp == 4 ? 16 + x : 21 + xp;
				return (xp < 4 && x < 4) ? inner : edge;
			};
			int tp = SB_X - 1, t = base_of(1);
#pragma unroll 1
			for (int j = 1; j <= Wwave; j++) {
				const int tn = base_of(j + 1);
				const bool colok = j <= W;
				const bool firstcol = j == 1;
				const int cA = col_of(tp, t);                          // (x', x) = (previous column, this column)
				const int colA = tab_base + cA * PITCH12, colA3 = tab3_base + cA * PITCH3;
				const int colN3 = tab3_base + col_of(t, tn) * PITCH3;  // (this column, next column): the credit of a gap opened in the target strand
				const int v5 = firstcol ? c_zero : c_bulge;
				const int q_first = firstcol ? SB_NEG / 2 : 0;         // (no aligned column before the first: nothing closes a gap state there)
				int uM = 0, uT = 0;         // (i - 1, j), new values
				int colmax = 0, colst = 0;
				// the entries of rows i and i + 1 are in flight while row i - 1 is computed
				i32x2 e0 = rd2(colA), e1 = rd2(colA + 8);
				int z1 = rd(colA3 + 4); // (the NEXT row's entry: what row 1 pays to open a gap in the query strand)
				int g0 = rd(colN3);
				// Row i also forms the diagonal candidates of row i + 1 (the bind kernel's arrangement): they need the old states of
				// row i, which die there, so the new states take their registers and nothing is copied at the loop's back edge.
				int dg1 = e0.x, dg2 = e0.y + q_first, dg3 = c_forb_m; // row 1: the border row above holds 0, 0, 0
#pragma unroll
				for (int i = 1; i <= LMAX; i++) {
					const i32x2 e2 = i + 1 <= LMAX ? rd2(colA + 8 * (i + 1 <= LMAX ? i + 1 : LMAX)) : e1;
					const int z2 = i + 1 <= LMAX ? rd(colA3 + 4 * (i + 1 <= LMAX ? i + 1 : LMAX)) : 0;
					const int g1 = i < LMAX ? rd(colN3 + 4 * (i < LMAX ? i : 0)) : 0;
					const int cM = oM[i - 1], cQ = oQ[i - 1], cT = oT[i - 1]; // (i, j - 1)
					// align_dimer, nuc_cruc.cpp:536-666, every table value = (-delta_g) << 11 | credit
					int dn1 = 0, dn2 = 0, dn3 = 0;
					if (i < LMAX) dn1 = cM + e1.x, dn2 = cQ + e1.y + q_first, dn3 = cT + e1.y; // e1: the NEXT row's entry, (q_i, q_i+1)
					int qi = cM + z1, qe = cQ - v5;
					asm volatile("" : "+v"(dn1), "+v"(dn2), "+v"(dn3), "+v"(qi), "+v"(qe), "+v"(dg1)); // the old states of this row are dead from here on
					int M = dg1 > dg2 ? dg1 : dg2;
					M = M > dg3 ? M : dg3;
					const int Iq = qi > qe ? qi : qe;
					const int ti = uM + (i == 1 ? c_forb_xy : g0), te = uT - (i == 1 ? c_zero : c_bulge);
					const int It = ti > te ? ti : te;
					const int Mz = M > 0 ? M : 0, Iqz = Iq > 0 ? Iq : 0, Itz = It > 0 ? It : 0;
					if (EXACT) {
						colmax = Mz > colmax ? Mz : colmax;
						colst |= Mz | Iqz | Itz;
					} else {
						const bool valid = i <= L; // (wave-uniform)
						colmax = (valid && Mz > colmax) ? Mz : colmax;
						colst |= valid ? (Mz | Iqz | Itz) : 0;
					}
					uM = Mz, uT = Itz;
					oM[i - 1] = Mz, oQ[i - 1] = Iqz, oT[i - 1] = Itz;
					dg1 = dn1, dg2 = dn2, dg3 = dn3;
					e0 = e1, e1 = e2, z1 = z2, g0 = g1;
				}
				vmax = (colok && colmax > vmax) ? colmax : vmax;
				sticky |= colok ? colst : 0;
				tp = t, t = tn;
			}
			// the verdict: the credit of the best maximal-score path against the first tier's threshold (same constants, same units)
			const int credit = (vmax & SB_CREDIT_MASK) * (FILT_UNITS / SB_UNITS); // 1e-3 kcal/mol
			const bool guard = (sticky & SB_GUARD) != 0;
			bool pass = true;
			int theta = INT32_MIN;
			if (a.tab->enabled[cls] && a.ftab->enabled[cls]) {
				const float th = a.ftab->theta0[cls] - a.ftab->ct_scale[cls] * (a.ftab->tk[cls] * (1.9872e-3f * slot.log_ct));
				theta = int(floorf(th * float(FILT_UNITS))) - 1;
				pass = guard || credit >= theta;
			}
			if (a.score_out && active) a.score_out[2 * idx] = guard ? INT32_MAX : credit, a.score_out[2 * idx + 1] = theta;
			if (a.dp_out && active) a.dp_out[idx] = vmax >> SB_SHIFT;
			if (a.audit) {
				if (active && !pass) a.audit[idx] = 1;
				pass = true;
			}
			if (active && !pass && a.result) a.result[idx] = -1;
			if (a.survivors) { // as in the first tier: survivors leave a per-wavefront LDS buffer 64 at a time
				const unsigned long long m = __builtin_amdgcn_ballot_w64(active && pass);
				n_rej += uint32_t(__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && !pass)));
				if (active && pass) s_keep[wave][n_keep + uint32_t(__builtin_popcountll(m & ((1ull << lane) - 1ull)))] = idx;
				n_keep += uint32_t(__builtin_popcountll(m));
				if (n_keep >= 64u) {
					filt_wave_sync();
					uint32_t at2 = 0;
					if (lane == 0) at2 = atomicAdd(a.n_survivors, 64u);
					at2 = uint32_t(__builtin_amdgcn_readfirstlane(int(at2)));
					a.survivors[at2 + lane] = s_keep[wave][lane];
					const uint32_t rest = n_keep - 64u;
					const uint32_t moved = uint32_t(lane) < rest ? s_keep[wave][64 + lane] : 0u;
					filt_wave_sync();
					if (uint32_t(lane) < rest) s_keep[wave][lane] = moved;
					n_keep = rest;
				}
				filt_wave_sync();
			}
		}
	}
	if (a.survivors) {
		if (lane == 0) s_wcount[wave] = n_keep;
		__syncthreads();
		if (tid == 0) {
			uint32_t run = 0;
			for (int w = 0; w < SB_WAVES; w++) {
				const uint32_t c = s_wcount[w];
				s_wcount[w] = run;
				run += c;
			}
			s_wbase = run ? atomicAdd(a.n_survivors, run) : 0u;
		}
		__syncthreads();
		if (uint32_t(lane) < n_keep) a.survivors[s_wbase + s_wcount[wave] + lane] = s_keep[wave][lane];
	}
	if (a.n_rejected && lane == 0 && n_rej) atomicAdd(a.n_rejected, n_rej);
}

// ---------------------------------------------------------------------------------------------------------------------
// The instance for the GENERAL buckets: windows that hold an IUPAC code or an inosine (the exception lists of the target).
// align_dimer resolves the target base of a cell against the query base it is paired with (nuc_cruc.cpp:14-213; against "none"
// where the partner is a gap), so the column part of a transition's table index changes from row to row: one table for all
// oligos, indexed by the RESOLVED bases (A C G T I none), the address formed per cell.  One lane per survivor of the first
// tier's general instance (a dense list); rows beyond the oligo are masked (the buckets of the general ladder hold several
// lengths).  Windows with a code the window loader skips (gap, unknown: the window shrinks, Q6) pass.
constexpr int SBD_X = 6;                                  // A C G T I none
constexpr int SBD_ENTRIES = SBD_X * SBD_X * NBP;
constexpr int SBD_THREADS = 256;
struct SBoundTablesD {
	int32_t entry[FILT_CLASSES][SBD_ENTRIES][4];
};
struct SBoundDegenArgs {
	const SBoundTables *tab;      // constants, enabled[]
	const SBoundTablesD *dtab;
	const FilterTables *ftab;
	const uint32_t *rt;           // Tables::rt: rt[x] nibble q = the base x resolves to against partner q
	DbView db;
	const uint32_t *olig_q4;
	const int32_t *olig_len;
	const Slot *slots;
	const Item *items;
	const uint32_t *index;        // the first tier's survivors, from index[*index_base_dev]
	const uint32_t *index_base_dev;
	const uint32_t *n_items_dev;
	uint32_t n_items;
	uint32_t *survivors;
	uint32_t *n_survivors;
	int32_t *result;
	uint32_t *n_rejected;
	uint8_t *audit;
	int32_t *score_out;
};

template <int LMAX> __global__ __launch_bounds__(SBD_THREADS, 2) void sbound_degen_kernel(SBoundDegenArgs a)
{
	static_assert(LMAX <= SB_MAX_ROWS, "instances of up to 40 rows");
	constexpr int THREADS = SBD_THREADS;
	constexpr int WMAX = LMAX + 2 * FLANK;
	constexpr int NQ = (LMAX + 2 + 3) / 4;   // 4 * base code of rows 0 .. LMAX + 1, four to a register
	constexpr int CLS_BYTES = SBD_ENTRIES * 16;
	typedef int32_t i32x3 __attribute__((ext_vector_type(3)));
	__shared__ __attribute__((aligned(16))) int32_t s_tab[FILT_CLASSES * SBD_ENTRIES * 4];
	__shared__ uint8_t s_win[WMAX * THREADS];           // the window's base codes, column j of lane l at (j - 1) * THREADS + l
	__shared__ uint32_t s_rt[18];
	__shared__ uint32_t s_keep[THREADS / 64][128];
	__shared__ uint32_t s_wcount[THREADS / 64], s_wbase;
	typedef __attribute__((address_space(3))) const int32_t lds_ci32;
	typedef __attribute__((address_space(3))) const i32x3 lds_ci32x3;

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t n_items = a.n_items;
	if (a.n_items_dev) {
		const uint32_t n_dev = *a.n_items_dev;
		n_items = n_dev < n_items ? n_dev : n_items;
	}
	if (uint32_t(blockIdx.x) * uint32_t(THREADS) >= n_items) return; // (nothing to do: leave before staging 56 KB of tables)
	for (int i = tid; i < FILT_CLASSES * SBD_ENTRIES * 4; i += THREADS) s_tab[i] = (&a.dtab->entry[0][0][0])[i];
	if (tid < 18) { // the resolution words as table coordinates: nibble q = min(base, 5) (5: none)
		const uint32_t w = a.rt[tid];
		uint32_t v = 0;
		for (int q = 0; q < 8; q++) {
			const uint32_t x = (w >> (4 * q)) & 15u;
			v |= (x < 5u ? x : 5u) << (4 * q);
		}
		s_rt[tid] = v;
	}
	__syncthreads();
	const int c_bulge = a.tab->bulge, c_zero = a.tab->zero, c_forb_m = a.tab->forb_m, c_forb_xy = a.tab->forb_xy;
	const uint32_t *index = a.index + (a.index_base_dev ? *a.index_base_dev : 0u);
	uint32_t *surv = a.survivors ? a.survivors + (a.index_base_dev ? *a.index_base_dev : 0u) : nullptr;
	uint32_t n_rej = 0, n_keep = 0;
	uint8_t *win = s_win + tid;
	auto rd3 = [&](int addr) __attribute__((always_inline)) { return *(lds_ci32x3 *)(uintptr_t(uint32_t(addr))); };
	auto rd = [&](int addr) __attribute__((always_inline)) { return *(lds_ci32 *)(uintptr_t(uint32_t(addr))); };

	for (uint32_t base = blockIdx.x * THREADS; base < n_items; base += gridDim.x * THREADS) {
		const uint32_t idx = base + tid < n_items ? index[base + tid] : 0xFFFFFFFFu;
		const bool active = idx != 0xFFFFFFFFu;
		const Item it = active ? a.items[idx] : Item{0, 0, 0, 0};
		const Slot slot = a.slots[it.info >> ITEM_SLOT_SHIFT];
		const int L = active ? a.olig_len[slot.oligo] : 0;
		const bool minus = (it.info & ITEM_MINUS) != 0;
		// the window, bind_oligo.cpp:502-505, with the codes of its exception list; a skipped code: not served
		int start = int(it.tloc) - (int(it.info & 127u) + slot.qloc_bias + FLANK);
		if (start < 0) start = 0;
		int stop = start + L + 2 * FLANK;
		const int tlen = active ? int(a.db.len[it.target]) : 0;
		if (stop > tlen) stop = tlen;
		const int W = active ? stop - start : 0;
		bool served = active && L <= LMAX;
		if (active) {
			const uint32_t *pk = a.db.packed + a.db.word_off[it.target];
			const uint32_t e0 = a.db.exc_off[it.target], e1 = a.db.exc_off[it.target + 1];
			uint32_t lo = e0, hi = e1;
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (a.db.exc_pos[mid] < uint32_t(start)) lo = mid + 1;
				else hi = mid;
			}
			uint32_t e = lo;
			for (int pos = start; pos < stop; pos++) {
				int code = int((pk[pos >> 4] >> ((pos & 15) * 2)) & 3u);
				if (e < e1 && a.db.exc_pos[e] == uint32_t(pos)) code = a.db.exc_code[e++];
				if (code > 15) served = false, code = 0;
				const int r = pos - start;
				if (minus) win[(W - 1 - r) * THREADS] = uint8_t(db_comp(code));
				else win[r * THREADS] = uint8_t(db_same(code));
			}
		}
		// 4 * base code of rows 0 .. LMAX + 1 (none above the first row and behind the oligo), and the rows' part of the table index
		uint32_t q4p[NQ];
#pragma unroll
		for (int k = 0; k < NQ; k++) q4p[k] = 0;
#pragma unroll
		for (int i = 0; i <= LMAX + 1; i++) {
			uint32_t q4 = 4u * bGAP;
			if (i >= 1 && i <= LMAX) {
				const uint32_t q4w = active ? a.olig_q4[slot.oligo * (MAX_OLIGO / 4) + ((i - 1) >> 2)] : 0u;
				q4 = (q4w >> (((i - 1) & 3) * 8)) & 255u;
				if (i > L) q4 = 4u * bGAP;
			}
			q4p[i >> 2] |= q4 << ((i & 3) * 8);
		}
		auto q4_of = [&](int i) __attribute__((always_inline)) { return (q4p[i >> 2] >> ((i & 3) * 8)) & 255u; }; // 4 * base code of row i
		const int cls = filter_class(slot.thresh);
		const int tab_base = int(uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)s_tab))) + cls * CLS_BYTES;
		// the rows' part of an entry's address, with the table's base: tab_base + 16 * (7 q_i-1 + q_i) for i = 1 .. LMAX + 1, two to a register
		constexpr int NRT = (LMAX + 2 + 1) / 2;
		uint32_t rtab[NRT];
#pragma unroll
		for (int k = 0; k < NRT; k++) rtab[k] = 0;
#pragma unroll
		for (int i = 1; i <= LMAX + 1; i++) rtab[i >> 1] |= (uint32_t(tab_base) + (7u * q4_of(i - 1) + q4_of(i)) * 4u) << ((i & 1) * 16);
		auto rtab_of = [&](int i) __attribute__((always_inline)) { return (rtab[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu; };
		// the base a column's code resolves to against the query base of row i, as a table coordinate (5: none; s_rt holds them clamped)
		auto rx = [&](uint32_t rtw, uint32_t q4) __attribute__((always_inline)) { return __builtin_amdgcn_ubfe(rtw, q4, 4u); };

		int Wwave = W;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const int other = __shfl_xor(Wwave, o);
			Wwave = other > Wwave ? other : Wwave;
		}
		Wwave = __builtin_amdgcn_readfirstlane(Wwave);

		int oM[LMAX], oQ[LMAX], oT[LMAX];
#pragma unroll
		for (int i = 0; i < LMAX; i++) oM[i] = oQ[i] = oT[i] = 0;
		int vmax = 0, sticky = 0;
		auto code_of = [&](int j) __attribute__((always_inline)) { return (j >= 1 && j <= W) ? int(win[(j - 1) * THREADS]) : int(bGAP); };
		uint32_t prtw = s_rt[bGAP], rtw = s_rt[code_of(1)];
#pragma unroll 1
		for (int j = 1; j <= Wwave; j++) {
			const uint32_t nrtw = s_rt[code_of(j + 1)];
			const bool colok = j <= W;
			const bool firstcol = j == 1;
			const int v5 = firstcol ? c_zero : c_bulge;
			const int q_first = firstcol ? SB_NEG / 2 : 0;
#pragma unroll
			for (int k = 0; k < NQ; k++) asm volatile("" : "+v"(q4p[k])); // (kept in registers, not re-derived per column)
#pragma unroll
			for (int k = 0; k < NRT; k++) asm volatile("" : "+v"(rtab[k]));
			// entry of row i: (x', x) = (previous column against row i - 1, this column against row i), rows (q_i-1, q_i);
			// the gap-opening credit of row i: (this column against row i - 1, next column against row i)
			auto entry_at = [&](int i) __attribute__((always_inline)) {
				const uint32_t col = rx(prtw, q4_of(i - 1)) * uint32_t(SBD_X) + rx(rtw, q4_of(i));
				return int(col * uint32_t(NBP * 16) + rtab_of(i));
			};
			auto gap_at = [&](int i) __attribute__((always_inline)) {
				const uint32_t col = rx(rtw, q4_of(i - 1)) * uint32_t(SBD_X) + rx(nrtw, q4_of(i));
				return int(col * uint32_t(NBP * 16) + rtab_of(i)) + 8;
			};
			int uM = 0, uT = 0;
			int colmax = 0, colst = 0;
			i32x3 e0 = rd3(entry_at(1)), e1 = rd3(entry_at(2));
			int g0 = rd(gap_at(1));
			int dg1 = e0.x, dg2 = e0.y + q_first, dg3 = c_forb_m; // row 1: the border row above holds 0, 0, 0
#pragma unroll
			for (int i = 1; i <= LMAX; i++) {
				const i32x3 e2 = i + 1 <= LMAX ? rd3(entry_at(i + 2 <= LMAX + 1 ? i + 2 : LMAX + 1)) : e1;
				const int g1 = i < LMAX ? rd(gap_at(i + 1 <= LMAX ? i + 1 : 1)) : 0;
				const int cM = oM[i - 1], cQ = oQ[i - 1], cT = oT[i - 1]; // (i, j - 1)
				int dn1 = 0, dn2 = 0, dn3 = 0;
				if (i < LMAX) dn1 = cM + e1.x, dn2 = cQ + e1.y + q_first, dn3 = cT + e1.y; // e1: the NEXT row's entry
				int qi = cM + e1.z, qe = cQ - v5;
				asm volatile("" : "+v"(dn1), "+v"(dn2), "+v"(dn3), "+v"(qi), "+v"(qe), "+v"(dg1));
				int M = dg1 > dg2 ? dg1 : dg2;
				M = M > dg3 ? M : dg3;
				const int Iq = qi > qe ? qi : qe;
				const int ti = uM + (i == 1 ? c_forb_xy : g0), te = uT - (i == 1 ? c_zero : c_bulge);
				const int It = ti > te ? ti : te;
				const int Mz = M > 0 ? M : 0, Iqz = Iq > 0 ? Iq : 0, Itz = It > 0 ? It : 0;
				const bool valid = i <= L;
				colmax = (valid && Mz > colmax) ? Mz : colmax;
				colst |= valid ? (Mz | Iqz | Itz) : 0;
				uM = Mz, uT = Itz;
				oM[i - 1] = Mz, oQ[i - 1] = Iqz, oT[i - 1] = Itz;
				dg1 = dn1, dg2 = dn2, dg3 = dn3;
				e0 = e1, e1 = e2, g0 = g1;
			}
			vmax = (colok && colmax > vmax) ? colmax : vmax;
			sticky |= colok ? colst : 0;
			prtw = rtw, rtw = nrtw;
		}
		const int credit = (vmax & SB_CREDIT_MASK) * (FILT_UNITS / SB_UNITS); // 1e-3 kcal/mol
		const bool guard = (sticky & SB_GUARD) != 0;
		bool pass = true;
		int theta = INT32_MIN;
		if (served && a.tab->enabled[cls] && a.ftab->enabled[cls]) {
			const float th = a.ftab->theta0[cls] - a.ftab->ct_scale[cls] * (a.ftab->tk[cls] * (1.9872e-3f * slot.log_ct));
			theta = int(floorf(th * float(FILT_UNITS))) - 1;
			pass = guard || credit >= theta;
		}
		if (a.score_out && active) a.score_out[2 * idx] = (guard || !served) ? INT32_MAX : credit, a.score_out[2 * idx + 1] = theta;
		if (a.audit) {
			if (active && !pass) a.audit[idx] = 1;
			pass = true;
		}
		if (active && !pass && a.result) a.result[idx] = -1;
		if (surv) {
			const unsigned long long m = __builtin_amdgcn_ballot_w64(active && pass);
			n_rej += uint32_t(__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && !pass)));
			if (active && pass) s_keep[wave][n_keep + uint32_t(__builtin_popcountll(m & ((1ull << lane) - 1ull)))] = idx;
			n_keep += uint32_t(__builtin_popcountll(m));
			if (n_keep >= 64u) {
				filt_wave_sync();
				uint32_t at = 0;
				if (lane == 0) at = atomicAdd(a.n_survivors, 64u);
				at = uint32_t(__builtin_amdgcn_readfirstlane(int(at)));
				surv[at + lane] = s_keep[wave][lane];
				const uint32_t rest = n_keep - 64u;
				const uint32_t moved = uint32_t(lane) < rest ? s_keep[wave][64 + lane] : 0u;
				filt_wave_sync();
				if (uint32_t(lane) < rest) s_keep[wave][lane] = moved;
				n_keep = rest;
			}
			filt_wave_sync();
		}
	}
	if (surv) {
		if (lane == 0) s_wcount[wave] = n_keep;
		__syncthreads();
		if (tid == 0) {
			uint32_t run = 0;
			for (int w = 0; w < THREADS / 64; w++) {
				const uint32_t c = s_wcount[w];
				s_wcount[w] = run;
				run += c;
			}
			s_wbase = run ? atomicAdd(a.n_survivors, run) : 0u;
		}
		__syncthreads();
		if (uint32_t(lane) < n_keep) surv[s_wbase + s_wcount[wave] + lane] = s_keep[wave][lane];
	}
	if (a.n_rejected && lane == 0 && n_rej) atomicAdd(a.n_rejected, n_rej);
}

// host-side launchers, defined in tnt_bind_inst.hip (TNT_SBOUND_LADDER: the plain instances of up to 40 rows)
#define TNT_SBOUND_LADDER(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40)
#define X(N) void launch_sbound_##N(const SBoundArgs &a, int blocks, hipStream_t stream);
TNT_SBOUND_LADDER(X)
#undef X
// ... and of the general buckets (ladder instances of up to 40 rows)
#define TNT_SBOUND_DEGEN_LADDER(X) X(16) X(20) X(24) X(28) X(32) X(36) X(40)
#define X(N) void launch_sbound_degen_##N(const SBoundDegenArgs &a, int blocks, hipStream_t stream);
TNT_SBOUND_DEGEN_LADDER(X)
#undef X

} // namespace tnt
#endif
