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
// No trace words, no traceback, no scoring: 17 instructions and two LDS reads per cell against the fill's 36 and nothing behind them.  Plain
// windows (A/C/G/T targets) of oligos of up to 40 bases; everything else goes from the first tier straight to the bind kernel.
// tests: tests/test_filter_bound.py (soundness of the oracle-side restatement, tier 3), tests/test_gpu_parity.py
// (kernel == restatement; searches with the tier on / off / in audit mode).
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
// Workgroup size per instance: the tables take 39 KB of LDS per workgroup, the states 3 registers per row.  Up to 25 rows fit 128
// registers (4 wavefronts per SIMD: two workgroups of 8); beyond that 3 or 2 per SIMD, reached with workgroups of 4.
constexpr int sb_threads(int lmax) { return lmax <= 25 ? 512 : 256; }
constexpr int sb_waves_per_simd(int lmax) { return lmax <= 25 ? 4 : (lmax <= 37 ? 3 : 2); }

// One 16-byte entry per (x', x, q', q), read with ONE ds_read_b128: [0] the diagonal step, [1] into an aligned column from a gap
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
	const uint32_t *index;        // the first tier's survivors (item indices), from index[*index_base_dev]
	const uint32_t *index_base_dev;
	const uint32_t *n_items_dev;
	uint32_t n_items;             // cap on the device count
	uint32_t *survivors;          // out: the windows that go on to the bind kernel, from survivors[*index_base_dev]
	uint32_t *n_survivors;
	int32_t *result;              // rejected windows: -1 (Tm below the minimum, bind_oligo.cpp:600-606)
	uint32_t *n_rejected;         // statistics
	uint8_t *audit;               // optional: audit mode marks what this tier would reject and lets everything through
	int32_t *score_out;           // optional (unit-level entry): [2 * idx] credit in 1e-3 kcal/mol (INT32_MAX: guard), [2 * idx + 1] threshold
	int32_t *dp_out;              // optional (unit-level entry): [idx] the maximal score of the fill
};

// EXACT: every oligo of the launch has exactly LMAX bases (the exact-length buckets; the instance of 16 rows serves shorter ones).
template <int LMAX, bool EXACT> __global__ __launch_bounds__(sb_threads(LMAX), sb_waves_per_simd(LMAX)) void sbound_kernel(SBoundArgs a)
{
	static_assert(LMAX <= SB_MAX_ROWS, "instances of up to 40 rows");
	constexpr int THREADS = sb_threads(LMAX);
	constexpr int WMAX = LMAX + 2 * FLANK;
	constexpr int NW = (WMAX + 15) / 16;
	constexpr int NR = (LMAX + 1 + 3) / 4;   // row constants, four to a register
	constexpr int CLS_BYTES = SB_ENTRIES * 16;
	typedef int32_t i32x4 __attribute__((ext_vector_type(3))); // (three of the entry's four words: a dead fourth register would be reused before the read lands)
	__shared__ __attribute__((aligned(16))) int32_t s_tab[FILT_CLASSES * SB_ENTRIES * 4];
	__shared__ uint32_t s_keep[THREADS / 64][128];
	__shared__ uint32_t s_wcount[THREADS / 64], s_wbase;
	typedef __attribute__((address_space(3))) const int32_t lds_ci32;
	typedef __attribute__((address_space(3))) const i32x4 lds_ci32x4;

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t n_items = a.n_items;
	if (a.n_items_dev) {
		const uint32_t n_dev = *a.n_items_dev;
		n_items = n_dev < n_items ? n_dev : n_items;
	}
	if (uint32_t(blockIdx.x) * uint32_t(THREADS) >= n_items) return; // (nothing to do: leave before staging 39 KB of tables)
	for (int i = tid; i < FILT_CLASSES * SB_ENTRIES * 4; i += THREADS) s_tab[i] = (&a.tab->entry[0][0][0])[i];
	__syncthreads();
	const int c_bulge = a.tab->bulge, c_zero = a.tab->zero, c_forb_m = a.tab->forb_m, c_forb_xy = a.tab->forb_xy;
	const uint32_t *index = a.index + (a.index_base_dev ? *a.index_base_dev : 0u);
	uint32_t *surv = a.survivors ? a.survivors + (a.index_base_dev ? *a.index_base_dev : 0u) : nullptr;
	uint32_t n_rej = 0, n_keep = 0;

	for (uint32_t base = blockIdx.x * THREADS; base < n_items; base += gridDim.x * THREADS) {
		const uint32_t idx = base + tid < n_items ? index[base + tid] : 0xFFFFFFFFu;
		const bool active = idx != 0xFFFFFFFFu;
		const Item it = active ? a.items[idx] : Item{0, 0, 0, 0};
		const Slot slot = a.slots[it.info >> ITEM_SLOT_SHIFT];
		const int L = active ? (EXACT ? LMAX : a.olig_len[slot.oligo]) : 0;
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
		// row constants 7 * q' + q of rows 1 .. LMAX + 1, a byte each (q' the base of the row above, none above the first; rows
		// behind the oligo hold "none": the row behind the last one is read for the credit of a gap opened in the last row)
		uint32_t Rp[NR];
		{
#pragma unroll
			for (int k = 0; k < NR; k++) Rp[k] = 0;
			uint32_t qprev = bGAP;
#pragma unroll
			for (int i = 0; i <= LMAX; i++) {
				uint32_t q = bGAP;
				if (i < LMAX) {
					const uint32_t q4w = active ? a.olig_q4[slot.oligo * (MAX_OLIGO / 4) + (i >> 2)] : 0u;
					q = (q4w >> ((i & 3) * 8 + 2)) & 7u;
					if (!EXACT && i >= L) q = bGAP;
				}
				Rp[i >> 2] |= (7u * qprev + q) << ((i & 3) * 8);
				qprev = q;
			}
		}
		const int cls = filter_class(slot.thresh);
		const int tab_base = int(uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)s_tab))) + cls * CLS_BYTES;
		auto rd4 = [&](int addr) __attribute__((always_inline)) { return *(lds_ci32x4 *)(uintptr_t(uint32_t(addr))); };
		auto rd = [&](int addr) __attribute__((always_inline)) { return *(lds_ci32 *)(uintptr_t(uint32_t(addr))); };
		auto rowc = [&](int i) __attribute__((always_inline)) { return int((Rp[i >> 2] >> ((i & 3) * 8)) & 0xFFu) * 16; }; // byte offset of row i + 1's entry

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
			uint32_t w = ov[0];
#pragma unroll
			for (int k = 1; k < NW; k++) w = ((j - 1) >> 4) == k ? ov[k] : w;
			const int b = int((w >> (((j - 1) & 15) * 2)) & 3u);
			return (j >= 1 && j <= W) ? b : SB_X - 1;
		};
		int tp = SB_X - 1, t = base_of(1);
#pragma unroll 1
		for (int j = 1; j <= Wwave; j++) {
			const int tn = base_of(j + 1);
			const bool colok = j <= W;
			const bool firstcol = j == 1;
			const int colA = tab_base + (tp * SB_X + t) * (NBP * 16);     // (x', x) = (previous column, this column)
			const int colN = tab_base + (t * SB_X + tn) * (NBP * 16) + 8; // (this column, next column): the credit of a gap opened in the target strand
			const int v5 = firstcol ? c_zero : c_bulge;
			const int q_first = firstcol ? SB_NEG / 2 : 0;                // (no aligned column before the first: nothing closes a gap state there)
#pragma unroll
			for (int k = 0; k < NR; k++) asm volatile("" : "+v"(Rp[k])); // (kept in registers, not re-derived per column)
			int uM = 0, uT = 0;         // (i - 1, j), new values
			int colmax = 0, colst = 0;
			// the entries of rows i and i + 1 are in flight while row i - 1 is computed
			i32x4 e0 = rd4(colA + rowc(0)), e1 = rd4(colA + rowc(1));
			int g0 = rd(colN + rowc(0));
			// Row i also forms the diagonal candidates of row i + 1 (the bind kernel's arrangement): they need the old states of
			// row i, which die there, so the new states take their registers and nothing is copied at the loop's back edge.
			int dg1 = e0.x, dg2 = e0.y + q_first, dg3 = c_forb_m; // row 1: the border row above holds 0, 0, 0
#pragma unroll
			for (int i = 1; i <= LMAX; i++) {
				const i32x4 e2 = i + 1 <= LMAX ? rd4(colA + rowc(i + 1 <= LMAX ? i + 1 : LMAX)) : e1;
				const int g1 = i < LMAX ? rd(colN + rowc(i < LMAX ? i : 0)) : 0;
				const int cM = oM[i - 1], cQ = oQ[i - 1], cT = oT[i - 1]; // (i, j - 1)
				// align_dimer, nuc_cruc.cpp:536-666, every table value = (-delta_g) << 11 | credit
				int dn1 = 0, dn2 = 0, dn3 = 0;
				if (i < LMAX) dn1 = cM + e1.x, dn2 = cQ + e1.y + q_first, dn3 = cT + e1.y; // e1: the NEXT row's entry, (q_i, q_i+1)
				int qi = cM + e1.z, qe = cQ - v5;
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
					const bool valid = i <= L;
					colmax = (valid && Mz > colmax) ? Mz : colmax;
					colst |= valid ? (Mz | Iqz | Itz) : 0;
				}
				uM = Mz, uT = Itz;
				oM[i - 1] = Mz, oQ[i - 1] = Iqz, oT[i - 1] = Itz;
				dg1 = dn1, dg2 = dn2, dg3 = dn3;
				e0 = e1, e1 = e2, g0 = g1;
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
		if (surv) { // as in the first tier: survivors leave a per-wavefront LDS buffer 64 at a time
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

} // namespace tnt
#endif
