// tnt_filter.hip.h -- gfx950 pre-filter of the bind stage: a cheap, SOUND upper bound of the stability any alignment
// of an (oligo, window) pair can reach, used to reject windows that cannot pass the minimum-Tm threshold before the
// full bind kernel (DP with traces, traceback, enumeration, scoring: tnt_bind.hip.h) looks at them.
//
// Why it is sound (DESIGN.md section 4.2a has the derivation term by term against nuc_cruc.cpp:1620-2299):
// bind_oligo_to_*_strand accepts a window only if Tm >= min_tm (bind_oligo.cpp:600-606).  With dH < 0 required
// (nuc_cruc.cpp:2253) and Tm = dH / (R ln Ct + dS) - 273.15 (:2291) that means  -(dH - Tk dS) >= -Tk R ln Ct  at
// Tk = min_tm + 273.15.  evaluate_alignment builds dH, dS from
//   * a constant (initiation) and at most two terminal-AT terms,
//   * one nearest-neighbour stack per column that is not inside a loop (plus its share of the salt term),
//   * per loop / bulge: an entropy penalty, and terminal stacks that are the stacks of the DIAGONALLY adjacent cells.
// So  Phi = -(dH - Tk dS) + const  is bounded by the best path through a small automaton over the cells of the window:
//   X(i,j)  best helix value ending in cell (i,j): the stacks along a diagonal, each counted only when favourable
//           (a negative stack is dropped: that also covers every loop that returns to its diagonal, for which the
//           reference charges as little as nothing: loop_S[2] = 0)
//   M2(i,j) best value inside a loop / bulge that has left its diagonal; it is entered from a helix cell at the price of
//           the cheapest bulge / asymmetric loop and left through a diagonal step that earns the adjacent stack, or the
//           best stack across a single-base bulge.
// Everything is integer -- 16-bit, in units of 1e-3 kcal/mol, saturating (a bound that saturates passes) -- with the table
// entries rounded towards "more stable", and the threshold carries a margin for the float32 rounding of the reference's sums.  A window whose bound stays below the threshold is rejected with
// the status the full kernel would have given it (Tm too low); every other window goes on to the full kernel, so the
// filter can only cost time, never change a result.  tests/test_filter_bound.py checks "accepted => bound >= threshold"
// on millions of random windows against the oracle.
#ifndef TNT_FILTER_HIP_H
#define TNT_FILTER_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tnt_dev.h"

namespace tnt {

constexpr int FILT_X = 6;                        // target base of a column: A C G T, 4 = none (before the first column / behind the window),
constexpr int FILT_NONE = 4, FILT_ANY = 5;       // 5 = an IUPAC code (whatever resolve_degenerate, nuc_cruc.cpp:14-201, makes of it is one of A C G T) or inosine
constexpr int FILT_ENTRIES = FILT_X * FILT_X * NBP; // index ((x' * 6 + x) * 49 + q' * 7 + q): previous column / row = (x', q'), current = (x, q)
constexpr int FILT_PLANE = 28 * 64;              // entries per class (>= FILT_ENTRIES)
constexpr int FILT_CLASSES = 2;                  // threshold sets by minimum Tm: primers, probes
constexpr int FILT_UNITS = 1000;                 // table units per kcal/mol
constexpr int FILT_NEG = -16384;                 // "not possible": -16.4 kcal/mol (a price no path that matters can pay) ...
#ifndef TNT_FILT_UNPACK_MAX
#define TNT_FILT_UNPACK_MAX 40
#endif
constexpr int FILT_SAT = 32767;                  // ... and the value sums saturate at: 32.8 kcal/mol, above every threshold

// host-built tables of one context (tnt_host.hip, fill_filter_tables)
struct FilterTables {
	// per class and entry two 16-bit values in one word (one ds_read_b32 per cell): low half the stack, counted when
	// favourable; high half the entry into a loop through this diagonal step, minus the gap price
	uint32_t plane[FILT_CLASSES][FILT_PLANE];
	float tk[FILT_CLASSES];                     // temperature of the bound: min_tm + 273.15 (Tm threshold) or target_t (dG threshold)
	float theta0[FILT_CLASSES];                 // threshold without the strand-concentration term, kcal/mol (margin included)
	float ct_scale[FILT_CLASSES];               // 1: the threshold carries - Tk R ln(Ct) (Tm threshold); 0: it does not (dG threshold)
	int32_t enabled[FILT_CLASSES];              // 0: windows of this class are never rejected; FILT_BY_TM / FILT_BY_DG
};
// Which threshold of the cascade (bind_oligo.cpp:600-624) a class's bound is taken against.  The minimum Tm where there is
// one (> 0 C): a rejected window fails the cascade's FIRST test, so its status is the reference's.  Else the maximum dG,
// where it is negative: the same automaton at the target temperature; acceptance needs dH - T dS <= max_dg, so a window
// whose bound of -(dH - T dS) stays below -max_dg cannot pass -- but it may fail the Tm range first (status 1, not 2), so
// callers that report statuses (tnt_bind_batch_routed) bind such windows instead (FilterArgs2::status_exact).
constexpr int FILT_BY_TM = 1, FILT_BY_DG = 2;

struct FilterArgs2 {
	const FilterTables *tab;
	DbView db;
	const uint32_t *olig_q4;   // [n][MAX_OLIGO/4] reversed oligo, one byte per row = 4 * base code (as the bind kernel reads it)
	const int32_t *olig_len;
	const Slot *slots;
	const Thresh *thresh;
	const Item *items;
	const uint32_t *index;          // work list (one length bucket of a phase queue) ...
	const uint32_t *index_base_dev; // ... starting at index[*index_base_dev]
	const uint32_t *n_items_dev;
	uint32_t n_items;               // cap on the device count
	uint32_t *survivors;            // out: the indices that go on to the full kernel, written from survivors[*index_base_dev]
	uint32_t *n_survivors;          // out: their number (device counter, zeroed by the host)
	int32_t *result;                // rejected windows get their status here (-1: Tm below the minimum)
	uint32_t *n_rejected;           // statistics: windows rejected here (device counter)
	uint8_t *audit;                 // optional: audit mode marks what the filter would reject and lets everything through
	int32_t *score_out;             // optional (unit-level entry): bound and threshold per item, [2 * idx]
	int32_t status_exact;           // reject only where the status is certain (bounds against the minimum Tm)
	// Self-contained queue entries (the round-3 seed kernel writes them for its phase-A items, tnt_seed2.hip.h): item index
	// (0xFFFFFFFF: no item) | first packed word of the window | start & 15, W << 4, L << 12 | Item::info.  With them the
	// kernel reads its queue as a stream and touches neither the items nor the per-target arrays, and a rejected window
	// writes nothing (the seed kernel has set its result to "not bound").
	const uint4 *fat;
};

// bits 0-1 of Thresh index -> class: 0 primer thresholds, 1..3 probe thresholds (tnt_host.hip builds th[0..3])
__device__ __forceinline__ int filter_class(int thresh_index) { return thresh_index == 0 ? 0 : 1; }

// DEGEN: the instance for windows that may hold non-ACGT target codes (the general buckets).  An IUPAC code (M R S V W Y H K
// D B N) is resolved per aligned pair to one of A C G T (resolve_degenerate), so its column takes, entry by entry, the
// maximum over the four bases (the ANY rows of the table): still an upper bound.  Inosine in the target takes the same column
// value (the ANY rows are the maximum over A C G T I).  Codes the window loader skips (gap, unknown: bind_oligo.cpp, Q6) are
// not modelled: such a window passes unfiltered.
// wave-synchronous LDS hand-off inside one wavefront
__device__ __forceinline__ void filt_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LMAX, bool DEGEN> __global__ __launch_bounds__(256) void filter_kernel(FilterArgs2 a)
{
	static_assert(FILT_ENTRIES <= FILT_PLANE, "table size");
	constexpr int THREADS = 256;
	constexpr int WMAX = LMAX + 2 * FLANK;
	__shared__ uint32_t s_tab[FILT_CLASSES * FILT_PLANE];
	typedef short s2 __attribute__((ext_vector_type(2))); // (X, M2) of a cell, or (stack, loop entry) of a table entry: one register

	const int tid = threadIdx.x;
	{ // a workgroup without work leaves before it stages the table (the grid is sized for full queues)
		uint32_t n_all = a.n_items;
		if (a.n_items_dev) {
			const uint32_t n_dev = *a.n_items_dev;
			n_all = n_dev < n_all ? n_dev : n_all;
		}
		if (uint32_t(blockIdx.x) * uint32_t(THREADS) >= n_all) return;
	}
	for (int i = tid; i < FILT_CLASSES * FILT_PLANE; i += THREADS) s_tab[i] = (&a.tab->plane[0][0])[i];
	static_assert(sizeof(a.tab->plane) == sizeof(uint32_t) * FILT_CLASSES * FILT_PLANE, "table size");
	__syncthreads();

	__shared__ uint32_t s_wcount[THREADS / 64], s_wbase;
	__shared__ uint32_t s_keep[THREADS / 64][128]; // survivors of a wavefront on their way to the list
	uint32_t n_rej = 0, n_keep = 0; // windows this wavefront rejected / holds in s_keep (wave-uniform)
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t *index = a.index + (a.index_base_dev ? *a.index_base_dev : 0u);
	uint32_t *surv = a.survivors ? a.survivors + (a.index_base_dev ? *a.index_base_dev : 0u) : nullptr;
	uint32_t n_items = a.n_items;
	if (a.n_items_dev) {
		const uint32_t n_dev = *a.n_items_dev;
		n_items = n_dev < n_items ? n_dev : n_items;
	}

	for (uint32_t base = blockIdx.x * THREADS; base < n_items; base += gridDim.x * THREADS) {
		// (a queue the seed kernel fills directly has slots without an item: 0xFFFFFFFF, QUEUE_HOLE of tnt_seed2.hip.h)
		uint32_t idx;
		Item it = Item{0, 0, 0, 0};
		uint4 fe = make_uint4(0xFFFFFFFFu, 0, 0, 0);
		if (a.fat) {
			if (base + tid < n_items) fe = a.fat[base + tid];
			idx = fe.x;
			it.info = fe.w;
			if (DEGEN && idx != 0xFFFFFFFFu) it = a.items[idx]; // (the exception lists go by target and position)
		} else {
			idx = base + tid < n_items ? index[base + tid] : 0xFFFFFFFFu;
			if (idx != 0xFFFFFFFFu) it = a.items[idx];
		}
		const bool active = idx != 0xFFFFFFFFu;
		const Slot slot = a.slots[it.info >> ITEM_SLOT_SHIFT];
		const bool minus = (it.info & ITEM_MINUS) != 0;
		const int qloc = int(it.info & 127u);
		int L, start, stop, W;
		if (a.fat) {
			L = int((fe.z >> 12) & 0xFFu);
			W = active ? int((fe.z >> 4) & 0xFFu) : 0;
			const int s_abs = int(it.tloc) - (qloc + FLANK);
			start = DEGEN ? (s_abs < 0 ? 0 : s_abs) : int(fe.z & 15u); // (only start & 15 and W matter to the plain instance)
			stop = start + W;
		} else {
			L = active ? a.olig_len[slot.oligo] : 0;
			const uint32_t tlen = active ? a.db.len[it.target] : 0u;
			// window, bind_oligo.cpp:502-505; plain windows only (the bucket pass routes windows with degenerate bases elsewhere)
			start = int(it.tloc) - (qloc + FLANK);
			if (start < 0) start = 0;
			stop = start + L + 2 * FLANK;
			if (stop > int(tlen)) stop = int(tlen);
			W = active ? stop - start : 0;
		}
		// The window stays in registers: the packed words that hold [start, stop), shifted so that base `start` sits in the
		// low bits.  Column j reads base start + j - 1 as it is (plus-strand binding, bind_oligo.cpp:1227-1276) or the
		// complement of base stop - j (minus strand: complement, reversed, :524-573).
		constexpr int NW = (WMAX + 15) / 16; // words of the shifted stream
		uint32_t wv[NW];
		{
			const uint32_t *pk = a.fat ? a.db.packed + fe.y : a.db.packed + (active ? a.db.word_off[it.target] : 0ull) + uint32_t(start >> 4);
			uint32_t raw[NW + 1];
#pragma unroll
			for (int k = 0; k <= NW; k++) raw[k] = active ? pk[k] : 0u; // rows are padded and the array has sixteen words of slack behind its end (NW + 1 <= 10 words are read)
			const uint32_t sh = uint32_t(start & 15) * 2u;
#pragma unroll
			for (int k = 0; k < NW; k++) wv[k] = __builtin_amdgcn_alignbit(raw[k + 1], raw[k], sh);
		}
		// DEGEN: the window's non-ACGT codes from the target's exception list: IUPAC codes mark their position in `am` (bit
		// 2 * (r & 15) of word r >> 4, the layout of `wv`); anything else makes the window pass unfiltered
		uint32_t am[DEGEN ? NW : 1];
		bool unfilterable = false;
		if (DEGEN) {
#pragma unroll
			for (int k = 0; k < NW; k++) am[DEGEN ? k : 0] = 0u;
			if (active) {
				const uint32_t e0 = a.db.exc_off[it.target], e1 = a.db.exc_off[it.target + 1];
				uint32_t lo = e0, hi = e1;
				while (lo < hi) {
					const uint32_t mid = (lo + hi) >> 1;
					if (a.db.exc_pos[mid] < uint32_t(start)) lo = mid + 1;
					else hi = mid;
				}
				for (uint32_t x = lo; x < e1 && a.db.exc_pos[x] < uint32_t(stop); x++) {
					const int code = a.db.exc_code[x];
					if (code > 15) unfilterable = true; // skipped codes (the window shrinks); inosine (4) and the IUPAC codes read as ANY
					const int r = int(a.db.exc_pos[x]) - start;
#pragma unroll
					for (int k = 0; k < NW; k++)
						if ((r >> 4) == k) am[DEGEN ? k : 0] |= 1u << ((r & 15) * 2);
				}
			}
		}
		auto base_at = [&](int r) __attribute__((always_inline)) { // base r of the window, r in [0, W)
			uint32_t w = wv[0];
#pragma unroll
			for (int k = 1; k < NW; k++) w = (r >> 4) == k ? wv[k] : w;
			const int b = int((w >> ((r & 15) * 2)) & 3u);
			if (!DEGEN) return b;
			uint32_t m = am[0];
#pragma unroll
			for (int k = 1; k < NW; k++) m = (r >> 4) == k ? am[DEGEN ? k : 0] : m;
			return ((m >> ((r & 15) * 2)) & 1u) ? -1 : b;
		};
		auto column_base = [&](int j) __attribute__((always_inline)) { // target base of column j in 1..W
			const int b = minus ? base_at(W - j) : base_at(j - 1);
			if (DEGEN && b < 0) return FILT_ANY; // (the complement of an IUPAC code is an IUPAC code)
			return minus ? 3 - b : b;
		};
		// per-row part of the table address, 4 * (7 * q' + q) (one byte, four rows per register): q' = base of the row above
		// (none above row 1), rows behind the oligo are "none" as well (their entries add nothing and open nothing)
		uint32_t Rpk[(LMAX + 3) / 4];
		{
			uint32_t qprev = bGAP;
#pragma unroll
			for (int w = 0; w < (LMAX + 3) / 4; w++) {
				const uint32_t q4w = active ? a.olig_q4[slot.oligo * (MAX_OLIGO / 4) + w] : 0u;
				Rpk[w] = 0;
#pragma unroll
				for (int b = 0; b < 4; b++) {
					const int i = w * 4 + b; // row i + 1
					if (i >= LMAX) continue;
					uint32_t q = (q4w >> (b * 8 + 2)) & 7u;
					if (i >= L) q = bGAP;
					Rpk[w] |= (4u * (7u * qprev + q)) << (8 * b);
					qprev = q;
				}
			}
		}
		static_assert(4 * (NBP - 1) < 256, "row constants are bytes");
		const int cls = filter_class(slot.thresh);
		const int tab_base = int(uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)s_tab))) + cls * (FILT_PLANE * 4);
		typedef __attribute__((address_space(3))) const s2 lds_cs2;

		int Wwave = W;
#ifdef TNT_EXP_FILT_NOSWEEP
		Wwave = 0; // timing experiment: everything but the sweep
#endif
		for (int o = 32; o > 0; o >>= 1) {
			const int other = __shfl_xor(Wwave, o);
			Wwave = other > Wwave ? other : Wwave;
		}
		// The sweep, two columns per trip: the cells of a column live in one register each, (X, M2) as two 16-bit halves,
		// and the two columns of a trip write alternate register sets (no copies at the back edge).  Per cell: one
		// ds_read_b32 and three packed instructions --
		//   a  = entry + (dX, dX)            v_pk_add_i16, saturating    (dX + stack, dX + loop entry)
		//   m  = max(up, old)                v_pk_max_i16                high half: max(M2 above, M2 to the left)
		//   n  = max(a, (m.hi, m.hi))        v_pk_max_i16                (X, M2)
		// (the automaton's M2 also takes the diagonal neighbour's M2: that one is never above the M2 of the cell above, which
		// has it as its left neighbour; round 4 dropped the instruction.  Packed-16 and SDWA forms issue at half the rate of a
		// plain add, scripts/microbench/valu_rate.hip: the table address is a plain add of an unpacked row constant where the
		// registers allow.)
		// The maximum of X over the window needs no instruction per cell: X never decreases along a diagonal (the stack
		// term is >= 0, also through the padding rows and columns), so it is reached in the last row or the last column.
		const s2 EDGE = {0, short(FILT_NEG)}; // a cell outside the matrix: X = 0, M2 = none
		// (instances of more than 64 rows -- oligos of 65 to 128 bases -- update one register set in place: a column reads the
		// cell to its left before it writes its own, so the alternate set is only there to spare the short instances a
		// dependency; 128 rows of it would not fit the register file)
		constexpr bool INPLACE = LMAX > 64;
		s2 PA[LMAX], PB[INPLACE ? 1 : LMAX];
#pragma unroll
		for (int i = 0; i < LMAX; i++) PA[i] = EDGE;
		s2 bestpk = EDGE;
		// short instances keep the row constants one per register (a full-rate add per cell instead of an SDWA byte select)
		constexpr bool UNPACKED = LMAX <= TNT_FILT_UNPACK_MAX;
		int Rrow[UNPACKED ? LMAX : 1];
		if (UNPACKED) {
#pragma unroll
			for (int i = 0; i < LMAX; i++) Rrow[UNPACKED ? i : 0] = int((Rpk[i >> 2] >> ((i & 3) * 8)) & 0xFFu);
		}
		auto row_addr = [&](int i) __attribute__((always_inline)) {
			if (UNPACKED) return Rrow[UNPACKED ? i - 1 : 0];
			return int((Rpk[(i - 1) >> 2] >> (((i - 1) & 3) * 8)) & 0xFFu);
		};
		auto column = [&](const s2 *Pin, s2 *Pout, const int colbase) __attribute__((always_inline)) {
			s2 e[LMAX];
#pragma unroll
#ifdef TNT_EXP_FILT_NOLDS
			for (int i = 1; i <= LMAX; i++) { // timing experiment: the sweep without its table reads
				const int ad = colbase + row_addr(i);
				e[i - 1] = (s2){short(ad & 1023), short(ad >> 4)};
			}
#else
			for (int i = 1; i <= LMAX; i++) e[i - 1] = *(lds_cs2 *)(uintptr_t(uint32_t(colbase + row_addr(i))));
#endif
			s2 D = EDGE, U = EDGE; // (1, j) sees the edge above it and diagonally
#pragma unroll
			for (int i = 1; i <= LMAX; i++) {
				const s2 O = Pin[i - 1]; // (i, j-1)
				const s2 sum = __builtin_elementwise_add_sat(e[i - 1], (s2){D.x, D.x});
				const s2 m = __builtin_elementwise_max(U, O);
				const s2 n = __builtin_elementwise_max(sum, (s2){m.y, m.y});
				// (no maximum with the diagonal cell: its M2 is already inside U's -- the cell above took it in as ITS left
				// neighbour --, and its X cannot exceed sum.x, stacks being >= 0)
				Pout[i - 1] = n;
				D = O, U = n;
			}
			bestpk = __builtin_elementwise_max(bestpk, U); // the last row of this column
		};
		int t = W >= 1 ? column_base(1) : FILT_NONE; // target base of the current column
		int tp = FILT_NONE;                          // no column before the first
#pragma unroll 1
		for (int j = 1; j <= Wwave; j += 2) {
			if (UNPACKED) {
#pragma unroll
				for (int i = 0; i < LMAX; i++) asm volatile("" : "+v"(Rrow[UNPACKED ? i : 0])); // (not re-derived from the packed form inside the loop)
			} else {
#pragma unroll
				for (int w = 0; w < (LMAX + 3) / 4; w++) asm volatile("" : "+v"(Rpk[w])); // keep the row constants packed
			}
			const int t1 = j < W ? column_base(j + 1) : FILT_NONE;     // column j + 1 (none: behind the window, or padding of an odd width)
			const int t2 = j + 1 < W ? column_base(j + 2) : FILT_NONE; // column j + 2
			column(PA, INPLACE ? PA : PB, tab_base + (tp * FILT_X + t) * (NBP * 4));
			column(INPLACE ? PA : PB, PA, tab_base + (t * FILT_X + t1) * (NBP * 4));
			tp = t1, t = t2;
		}
#pragma unroll
		for (int i = 0; i < LMAX; i++) bestpk = __builtin_elementwise_max(bestpk, PA[i]); // the last column
		int best = int(bestpk.x);

		// threshold of this item: theta0 of its class - Tk * R * ln(Ct), in table units, rounded down; a saturated bound passes
		bool pass = true;
		int theta = INT32_MIN;
		if (a.tab->enabled[cls]) {
			const float th = a.tab->theta0[cls] - a.tab->ct_scale[cls] * (a.tab->tk[cls] * (1.9872e-3f * slot.log_ct));
			theta = int(floorf(th * float(FILT_UNITS))) - 1;
			pass = best >= theta || best >= FILT_SAT || (a.status_exact && a.tab->enabled[cls] != FILT_BY_TM);
		}
		if (DEGEN && unfilterable) pass = true, best = INT32_MAX, theta = INT32_MIN;
		if (a.score_out && active) a.score_out[2 * idx] = best, a.score_out[2 * idx + 1] = theta;
		if (a.audit) {
			if (active) a.audit[idx] = pass ? 0 : 1;
			pass = true;
		}
		if (active && !pass && !a.fat) a.result[idx] = -1; // bind_oligo.cpp:600-606: Tm below the minimum (fat entries: set by the seed kernel)
		if (surv) {
			// Survivors collect in a per-wavefront LDS buffer and leave for the list 64 at a time: one global atomic per 64
			// survivors (a hot counter sustains only ~88 atomics per microsecond -- one per wavefront and trip, as a first
			// version had it, cost more than the sweep -- and nothing waits for a round trip every trip).
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
				const uint32_t rest = n_keep - 64u; // < 64
				const uint32_t moved = uint32_t(lane) < rest ? s_keep[wave][64 + lane] : 0u;
				filt_wave_sync();
				if (uint32_t(lane) < rest) s_keep[wave][lane] = moved;
				n_keep = rest;
			}
			filt_wave_sync();
		}
	}
	if (surv) { // what is left in the buffers: one atomic per workgroup
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
	if (a.n_rejected && (tid & 63) == 0 && n_rej) atomicAdd(a.n_rejected, n_rej); // statistics: once per wavefront
}

// host-side launchers, defined in tnt_bind_inst.hip next to the bind kernel instances of the same lengths
#define X(N) void launch_filter_##N(const FilterArgs2 &a, int blocks, hipStream_t stream);
TNT_FILTER_LADDER(X)
TNT_LONG_LADDER(X)
#undef X
#define X(N) void launch_filter_degen_##N(const FilterArgs2 &a, int blocks, hipStream_t stream);
TNT_LMAX_LADDER(X)
TNT_LONG_LADDER(X)
#undef X

} // namespace tnt
#endif
