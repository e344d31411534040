// tnt_wbind.hip.h -- gfx950 bind kernel in the north-star form: one WAVEFRONT per (oligo, target-window) pair.
//
// Same contract as bind_kernel (tnt_bind.hip.h: bind_oligo.cpp:502-748 for one seed; align_dimer nuc_cruc.cpp:492-696,
// trace_back :1409-1618, enumerate_dimer_alignments :973-1170, evaluate_alignment :1620-2299), other mapping:
//  * lane l owns the R consecutive query rows l*R+1 .. l*R+R (R = 1 up to 64 bases, 2 up to 128, ... 16 up to 1024: the
//    reference's MAX_SEQUENCE_LENGTH, nuc_cruc.h:80), the sweep runs along the anti-diagonals: in step s lane l works on
//    column s - l + 1.  What a cell needs of the row above -- (M, I_query, I_target) of (i-1, j) and of (i-1, j-1) and the
//    table term of (i-1, j) -- is what the lane above produced one and two steps earlier: five DPP wave_shr:1 moves per step
//    (states, term, the column's resolve word), no LDS traffic for the recurrence.  Its own row's previous column stays in
//    registers;
//  * delta_g, the SantaLucia tables and the window sit in LDS like in bind_kernel; the one-byte trace words go to LDS too
//    (R <= 2: L x (L + 8) bytes per window, 4.9 KB for a 64-mer, 18 KB for a 128-mer) -- no HBM trace slab -- and to a slab
//    per wavefront beyond (a 1024-mer needs 1 MB);
//  * the running maximum is kept per lane (value, count, first cell); a wave reduction behind the sweep gives the global
//    one.  More than one maximal cell (low-complexity windows) repeats the sweep in a mode that collects them;
//  * the serial tail (bind_tail, shared with bind_kernel) runs on lane 0 out of LDS.
// Lane use of the sweep is L*W / (64 * (L/R + W - 1)): 20 % for a 22-mer, 34 % for a 40-mer, 53 % at 64 and 128 bases.
// DESIGN.md section 4.2b has the A/B against bind_kernel; this kernel is the only one for oligos beyond 128 bases.
#ifndef TNT_WBIND_HIP_H
#define TNT_WBIND_HIP_H

#include <type_traits>
#include "tnt_bind.hip.h"

namespace tnt {

constexpr int WB_CODE_SHIFT = 11;                                          // a maximal cell is (i << 11) | j: i <= 1024, j <= 1032
constexpr int wb_trace_pitch(int R) // bytes between two rows of the trace array: lanes write R rows apart, (R * pitch - 1) / 4 odd spreads them over the banks
{
	int p = 64 * R + 2 * FLANK;
	if (R == 1) while ((p - 1) % 8 != 4) p++;
	return p;
}

// v_mov_b32_dpp wave_shr:1: lane l takes src of lane l - 1, lane 0 keeps `first`
__device__ __forceinline__ int wave_shr1(int first, int src) { return __builtin_amdgcn_update_dpp(first, src, 0x138, 0xf, 0xf, false); }

// trace words as bind_tail reads them: [row][column], in LDS or in the wavefront's slab
struct RowTrace {
	const uint8_t *tr;
	int pitch;
	__device__ __forceinline__ void reset() {}
	__device__ __forceinline__ int get(int li, int lj) const { return int(tr[size_t(li - 1) * size_t(pitch) + size_t(lj - 1)]); }
};

// DINK: --dinkelbach (NucCruc::approximate_tm_heterodimer, nuc_cruc.cpp:2399-2432): the alignment is computed at 0 C, then
// again and again at the Tm the last pass gave while its free energy there is negative and still rising; every pass has its
// own delta_g (update_dp_param at that temperature), so a workgroup is ONE wavefront and rebuilds its LDS planes from the
// (H, S, floor) triples of the entries (Tables::dk_*) in front of every pass.
template <int R, bool FULL, bool DINK = false> __global__ __launch_bounds__(64 * (DINK ? 1 : wb_waves(R))) void wbind_kernel(BindArgs a)
{
	constexpr int WAVES = DINK ? 1 : wb_waves(R);
	constexpr int THREADS = 64 * WAVES;
	constexpr int LMAX = 64 * R;
	constexpr int WMAX = LMAX + 2 * FLANK;
	constexpr int NCOLS = LMAX + WMAX + 4;
	constexpr bool TLDS = wb_trace_lds(R);
	constexpr int PITCH = wb_trace_pitch(R);
	static_assert(LMAX <= MAX_OLIGO, "rows beyond TNT_MAX_OLIGO");

	// tables: the general planes of bind_kernel (any target base): plane 0 delta_g(prev, cur), plane 1 term(cur), index
	// ((x' * 7 + x) * 49 + q' * 7 + q); term[] by pair
	constexpr int DG_PLANE = 38 * 64;
	__shared__ int32_t s_dgp[2 * DG_PLANE];
	__shared__ int32_t s_term[NBP];
	__shared__ int32_t s_const[4];
	__shared__ uint32_t s_rt[18];
	__shared__ uint8_t s_lut[18 * 18 + 4];
	__shared__ uint8_t s_hs_map[NBP * NBP + 3];
	__shared__ float2 s_hs_val[256];
	__shared__ TailConst s_k;
	// per wavefront
	__shared__ uint8_t s_win[WAVES][WMAX + 4];
	__shared__ uint8_t s_qry[WAVES][LMAX + 4];
	__shared__ uint8_t s_cols[WAVES][NCOLS + 4];
	__shared__ uint8_t s_best[WAVES][NCOLS + 4];
	__shared__ uint32_t s_stack[WAVES][NCOLS];
	__shared__ uint32_t s_ties[WAVES][TIE_CAP];
	__shared__ uint32_t s_ntie[WAVES];
	__shared__ uint32_t s_rtc[WAVES][WMAX + 4]; // rt(t, .) of every column of the window (4-bit fields by partner), GAP's behind the last
	__shared__ uint8_t s_trace[WAVES][TLDS ? LMAX * PITCH + 4 : 4];
	__shared__ uint8_t s_dump[WAVES][4]; // where the trace words of cells outside the matrix go

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	uint32_t n_items = a.n_items;
	if (a.n_items_dev) {
		const uint32_t n_dev = *a.n_items_dev;
		n_items = n_dev < n_items ? n_dev : n_items;
	}
	if (uint32_t(blockIdx.x) * uint32_t(WAVES) >= n_items) return; // (before the tables are staged: the grid is sized for full queues)

	// delta_g of entry `src` of Tables::dg at temperature T (update_dp_param, nuc_cruc.cpp:357-486: truncation towards zero, the
	// clamped classes floored at 0)
	auto dg_at = [&](int src, float T) __attribute__((always_inline)) {
		if (src < 0) return 0; // (a class constant without an entry of its own)
		const int v = int((a.tab->dk_H[src] - T * a.tab->dk_S[src]) * 10000.0f);
		const int f = a.tab->dk_F[src];
		return v > f ? v : f;
	};
	auto stage_dg = [&](float T, bool at_target_t) __attribute__((always_inline)) { // the planes of one temperature (the whole workgroup)
		for (int i = tid; i < NBP * NBP; i += THREADS) {
			const int p = i / NBP, cpair = i % NBP; // prev pair (x', q'), current pair (x, q)
			const int at = ((p / NB) * NB + cpair / NB) * NBP + (p % NB) * NB + cpair % NB;
			s_dgp[at] = at_target_t ? a.tab->dg[i] : dg_at(i, T);
			s_dgp[DG_PLANE + at] = at_target_t ? a.tab->term[cpair] : dg_at(a.tab->term_src[cpair], T);
		}
		if (tid < NBP) s_term[tid] = at_target_t ? a.tab->term[tid] : dg_at(a.tab->term_src[tid], T);
		if (tid < 4) s_const[tid] = at_target_t ? a.tab->dg_const[tid] : dg_at(a.tab->const_src[tid], T);
	};
	if (!DINK) stage_dg(0.0f, true);
	for (int i = tid; i < 18 * 18; i += THREADS) s_lut[i] = a.tab->bbp[i];
	if (tid < 18) s_rt[tid] = a.tab->rt[tid];
	for (int i = tid; i < NBP * NBP; i += THREADS) s_hs_map[i] = a.tab->hs_map[i];
	for (int i = tid; i < 256; i += THREADS) s_hs_val[i] = make_float2(a.tab->hs_H[i], a.tab->hs_S[i]);
	if (tid < 64) s_k.loop_S[tid] = a.tab->loop_S[tid], s_k.bulge_S[tid] = a.tab->bulge_S[tid];
	if (tid == 0) {
		const Tables &t = *a.tab;
		s_k.init_H = t.init_H, s_k.init_S = t.init_S, s_k.sym_S = t.sym_S, s_k.at_H = t.at_H, s_k.at_S = t.at_S;
		s_k.salt = t.salt, s_k.asym_S = t.asym_S, s_k.bulge_at_S = t.bulge_at_S, s_k.temperature = t.temperature, s_k.log_na = t.log_na;
		s_k.dangle5 = t.dangle5, s_k.dangle3 = t.dangle3;
	}
	__syncthreads();

	uint8_t *win = s_win[wv], *qry = s_qry[wv], *cols = s_cols[wv], *best = s_best[wv];
	uint32_t *stack = s_stack[wv], *ties = s_ties[wv], *rtc = s_rtc[wv];
	const uint32_t wave_id = uint32_t(blockIdx.x) * WAVES + uint32_t(wv), n_waves = uint32_t(gridDim.x) * WAVES;
	uint8_t *trace = TLDS ? s_trace[wv] : a.trace + size_t(wave_id) * wb_slab_bytes(R);
	uint8_t *dump = s_dump[wv];

	const uint32_t *index = a.index;
	if (index && a.index_base_dev) index += *a.index_base_dev;
	const char *sdg = reinterpret_cast<const char *>(s_dgp);
	const char *sterm = reinterpret_cast<const char *>(s_term);

	for (uint32_t w = wave_id; w < n_items; w += n_waves) { // wave-uniform
		uint32_t idx = index ? index[size_t(w) * (a.index_stride ? a.index_stride : 1u)] : w;
		if (idx == 0xFFFFFFFFu) continue; // (a queue filled by the seed kernel has slots without an item: QUEUE_HOLE)

		// ------------------------------------------------------------------ item decode (as bind_kernel)
		const Item it = a.items[idx];
		const Slot slot = a.slots[it.info >> ITEM_SLOT_SHIFT];
		const int L = a.olig_len[slot.oligo];
		const bool minus = (it.info & ITEM_MINUS) != 0;
		const int qloc = int(it.info & 127u) + slot.qloc_bias;
		const int xmode = FULL ? int((it.target >> 29) & 3u) : 0; // 0 window of a packed target, 1 heterodimer, 2 homodimer, 3 hairpin
		const uint32_t tlen = (FULL && xmode) ? 0u : a.db.len[it.target];
		int start = int(it.tloc) - (qloc + FLANK); // bind_oligo.cpp:502-505
		if (start < 0) start = 0;
		int stop = start + L + 2 * FLANK;
		if (stop > int(tlen)) stop = int(tlen);
		int W = 0;
		wave_sync(); // the tail of the window before is done with the per-wavefront arrays

		// ------------------------------------------------------------------ window -> LDS (Q6: codes > 15 are skipped), lanes side by side
		if (FULL && xmode) {
			const int xo = (xmode == 1) ? int(it.target & 0xFFFFFFu) : slot.oligo; // the oligo that plays the target
			W = a.olig_len[xo];
			start = 0, stop = W;
			for (int k = lane; k < W; k += 64) win[k] = a.olig_base[size_t(xo) * MAX_OLIGO + k]; // set_target / set_duplex
		} else {
			const uint32_t *pk = a.db.packed + a.db.word_off[it.target];
			const uint32_t e0 = a.db.exc_off[it.target], e1 = a.db.exc_off[it.target + 1];
			const int n = stop - start;
			int kept = 0;
			for (int b0 = 0; b0 < n; b0 += 64) {
				const int pos = start + b0 + lane;
				const bool in = b0 + lane < n;
				int code = 0;
				if (in) {
					code = int((pk[pos >> 4] >> ((pos & 15) * 2)) & 3u);
					if (e1 > e0) {
						uint32_t lo = e0, hi = e1;
						while (lo < hi) {
							const uint32_t mid = (lo + hi) >> 1;
							if (a.db.exc_pos[mid] < uint32_t(pos)) lo = mid + 1;
							else hi = mid;
						}
						if (lo < e1 && a.db.exc_pos[lo] == uint32_t(pos)) code = a.db.exc_code[lo];
					}
				}
				const bool keep = in && code <= 15;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
				if (keep) cols[kept + int(__builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)))] = uint8_t(code);
				kept += int(__builtin_popcountll(m));
			}
			W = kept;
			wave_sync();
			// minus strand: the complement, pushed to the front (bind_oligo.cpp:524-573); plus strand: as is (:1227-1276)
			for (int r = lane; r < W; r += 64) {
				const int code = cols[r];
				if (minus) win[W - 1 - r] = uint8_t(db_comp(code));
				else win[r] = uint8_t(db_same(code));
			}
		}
		// the oligo in row order (row i-1 = query[L-i]); rows beyond L hold A
		for (int wq = lane; wq < LMAX / 4; wq += 64) {
			const uint32_t q4w = wq < MAX_OLIGO / 4 ? a.olig_q4[slot.oligo * (MAX_OLIGO / 4) + wq] : 0u;
#pragma unroll
			for (int b = 0; b < 4; b++) qry[wq * 4 + b] = uint8_t((q4w >> (b * 8 + 2)) & 7u);
		}
		if (lane == 0) s_ntie[wv] = 0;
		wave_sync();
		for (int k = lane; k < WMAX + 4; k += 64) rtc[k] = k < W ? s_rt[win[k]] : s_rt[bGAP];
		wave_sync();

		// ------------------------------------------------------------------ per-lane row constants
		int q4[R], r4[R];      // 4 * q of the row; 4 * (7 * q' + q) with q' = the row above (GAP above row 1)
		int q4_above;          // 4 * q of the last row of the lane above
		{
			int qprev = lane == 0 ? bGAP : int(qry[lane * R - 1]);
			q4_above = 4 * qprev;
#pragma unroll
			for (int r = 0; r < R; r++) {
				const int q = int(qry[lane * R + r]);
				q4[r] = 4 * q;
				r4[r] = 4 * (7 * qprev + q);
				qprev = q;
			}
		}
		const bool lane0 = lane == 0;
		const int nl = (L + R - 1) / R;            // lanes that own a row
		const int n_steps = nl + W - 1;            // wave-uniform
		const uint32_t rt_gap = s_rt[bGAP];

		int gmax = -1, nmax = 0, max_ij = 0;
		// temperature of this pass: the target temperature, or the Dinkelbach sequence 0 C, Tm_1, Tm_2, ...
		float Tcur = DINK ? 273.15f : s_k.temperature, dk_q = -999999.9f;
	next_pass:
		if (DINK) { // this pass's delta_g planes (the workgroup is this wavefront)
			stage_dg(Tcur, false);
			if (lane == 0) s_ntie[wv] = 0;
			wave_sync();
			gmax = -1, nmax = 0, max_ij = 0;
		}
		const int c_bulge = s_const[0], c_zero = s_const[1], c_forb_m = s_const[2], c_forb_xy = s_const[3];

		// ------------------------------------------------------------------ the sweep (align_dimer, nuc_cruc.cpp:492-696)
		// COLLECT: the second sweep of a window with several maximal cells appends every cell that holds `gmax` to ties[]
		auto sweep = [&](auto collect_t) __attribute__((always_inline)) {
			constexpr bool COLLECT = decltype(collect_t)::value;
			int oM[R], oQ[R], oT[R]; // this lane's rows at the previous column; oQ / oT clamped at zero, oM too unless FULL
#pragma unroll
			for (int r = 0; r < R; r++) oM[r] = FULL ? -1 : 0, oQ[r] = oT[r] = 0;
			int aM = 0, aQ = 0, aT = 0;                 // (i-1, j-1) of the lane's first row: what came down the lanes one step ago
			int lastM = 0, lastQ = 0, lastT = 0;       // the lane's last row after the step before (clamped), and its term(cur)
			int last_term = c_forb_xy;
			int lmax = FULL ? -1 : 0, lcnt = 0, lcode = 0;
			// The column (its resolve word: rt(t, .) as 4-bit fields) enters at lane 0 and moves down one lane per step.  The
			// table values of a step only hang on the bases, so they are read one step ahead: the reads of step s + 1 are in
			// flight while the cells of step s are computed.
			auto tables = [&](uint32_t rtw, uint32_t prtw, int (&ox)[R], int (&oy)[R], int (&ot)[R]) __attribute__((always_inline)) {
				int xp = int(__builtin_amdgcn_ubfe(prtw, uint32_t(q4_above), 4u)); // rt(pt, q of the row above)
#pragma unroll
				for (int r = 0; r < R; r++) {
					const int x = int(__builtin_amdgcn_ubfe(rtw, uint32_t(q4[r]), 4u)); // rt(t, q)
					const int ea = xp * (4 * 343) + x * (4 * 49) + r4[r];
					xp = int(__builtin_amdgcn_ubfe(prtw, uint32_t(q4[r]), 4u));
					ox[r] = *reinterpret_cast<const int *>(sdg + ea);                  // g(bp(pt,pq), bp(t,q))
					oy[r] = *reinterpret_cast<const int *>(sdg + ea + 4 * DG_PLANE);   // term(bp(t,q))
					ot[r] = *reinterpret_cast<const int *>(sterm + xp * 28 + q4[r]);   // term(bp(pt,q))
				}
			};
			uint32_t rtw = uint32_t(wave_shr1(int(rtc[0]), int(rt_gap))); // step 0: lane 0 meets column 1, the others the border column
			uint32_t rt_ahead = rtc[1];
			int cx[R], cy[R], ct[R];
			tables(rtw, rt_gap, cx, cy, ct);
#pragma unroll 1
			for (int s = 0; s < n_steps; s++) {
				const int j = s - lane + 1;
				const bool colvalid = uint32_t(j - 1) < uint32_t(W);
				const bool firstcol = j == 1;
				const uint32_t rtw_next = uint32_t(wave_shr1(int(rt_ahead), int(rtw)));
				rt_ahead = rtc[s + 2 < WMAX + 3 ? s + 2 : WMAX + 3]; // (steps run to nl + W - 1: behind the window every column reads the GAP word at the array's end)
				int nx[R], ny[R], nt[R];
				tables(rtw_next, rtw, nx, ny, nt);
				// the row above: (i-1, j) as the lane above left it one step ago; the border row for lane 0
				int bM = wave_shr1(0, lastM);
				int bT = wave_shr1(0, lastT);
				const int bQ = wave_shr1(0, lastQ);
				int b_term = wave_shr1(c_forb_xy, last_term);
				const int inM = bM, inQ = bQ, inT = bT;
				const int v5 = firstcol ? c_zero : c_bulge;  // g(bp(pt,GAP), bp(t,GAP))
				int lim = L;
				if (FULL && xmode == 3) lim = L - 3 - j;     // align_hairpin's triangle, nuc_cruc.cpp:794-810
#pragma unroll
				for (int r = 0; r < R; r++) {
					const int i = lane * R + r + 1;
					const int ox = cx[r], oy = cy[r], ot = ct[r];
					const int o2 = firstcol ? c_forb_m : oy;
					const bool row1 = lane0 && r == 0;
					const int cM = oM[r], cQ = oQ[r], cT = oT[r];
					const int cMz = FULL ? (cM > 0 ? cM : 0) : cM;
					const int dg1 = aM - ox;
					const int dg2 = aQ - o2;
					const int dg3 = aT - (row1 ? c_forb_m : oy);
					const int iqi = cMz - ot;
					const int iqe = cQ - v5;
					const int iti = bM - b_term;
					const int ite = bT - (row1 ? c_zero : c_bulge);
					int M = dg1 > dg2 ? dg1 : dg2;
					M = M > dg3 ? M : dg3;
					const int Iq = iqi > iqe ? iqi : iqe;
					const int It = iti > ite ? iti : ite;
					int Mz = M > 0 ? M : 0, Iqz = Iq > 0 ? Iq : 0, Itz = It > 0 ? It : 0;
					// the trace word (TW_* in tnt_dev.h), built like bind_kernel's
					uint32_t tw;
					tw = sign_bit(dg1 - Mz);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(dg3 - Mz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(dg2 - Mz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(iqi - Iqz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(iqe - Iqz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(iti - Itz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(ite - Itz), 31);
					tw = __builtin_amdgcn_alignbit(tw, uint32_t(Mz - 1), 31);
					// cells outside the matrix (the lane has not reached column 1 yet, or is past column W) hold the border's
					// values: zeros after the clamp; their trace words go to a dump slot
					uint8_t *tp = colvalid ? trace + (size_t(i - 1) * PITCH + size_t(j - 1)) : dump;
					*tp = uint8_t(tw);
					if (FULL) M = colvalid ? M : -1;
					Mz = colvalid ? Mz : 0, Iqz = colvalid ? Iqz : 0, Itz = colvalid ? Itz : 0; // (selects: a branch here splits the step's schedule)
					// running maximum of the lane (nuc_cruc.cpp:670-691: all cells that hold the maximum)
					const bool counts = colvalid && i <= lim;
					const int Mv = FULL ? (counts ? M : INT32_MIN) : (counts ? Mz : 0);
					if (COLLECT) {
						if (counts && Mv == gmax) {
							const uint32_t at = atomicAdd(&s_ntie[wv], 1u);
							if (at < uint32_t(TIE_CAP)) ties[at] = uint32_t((i << WB_CODE_SHIFT) | j);
						}
					} else {
						const bool gt = Mv > lmax, eq = Mv == lmax;
						lcode = (gt || (eq && lcnt == 0)) ? ((i << WB_CODE_SHIFT) | j) : lcode;
						lcnt = gt ? 1 : lcnt + (eq ? 1 : 0);
						lmax = gt ? Mv : lmax;
					}
					// this row becomes the row above of the next one; its old states are that row's (i-1, j-1)
					aM = cMz, aQ = cQ, aT = cT;
					oM[r] = FULL ? M : Mz, oQ[r] = Iqz, oT[r] = Itz;
					bM = Mz, bT = Itz, b_term = oy;
				}
				lastM = bM, lastQ = oQ[R - 1], lastT = bT, last_term = b_term;
				aM = inM, aQ = inQ, aT = inT;
				rtw = rtw_next;
#pragma unroll
				for (int r = 0; r < R; r++) cx[r] = nx[r], cy[r] = ny[r], ct[r] = nt[r];
			}
			if (!COLLECT) {
				// global maximum and its cells
				int g = lmax;
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) {
					const int other = __shfl_xor(g, o);
					g = other > g ? other : g;
				}
				const bool mine = lmax == g && lcnt > 0 && (FULL || g > 0);
				int cnt = mine ? lcnt : 0;
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
				const unsigned long long mm = __builtin_amdgcn_ballot_w64(mine);
				gmax = (FULL || g > 0) ? g : -1;
				nmax = cnt;
				max_ij = mm ? __shfl(lcode, int(__builtin_ctzll(mm))) : 0;
			}
		};
		sweep(std::false_type{});
		if (nmax > 1) {
			sweep(std::true_type{});
			wave_sync();
			nmax = int(s_ntie[wv]); // (the same cells, counted again; beyond TIE_CAP the tail raises ERR_TIE_OVERFLOW)
			max_ij = int(ties[0]);
		}
		wave_sync(); // trace words and ties were written by other lanes

		// ------------------------------------------------------------------ serial tail on lane 0
		bool again = false;
		if (lane == 0) {
			RowTrace tr{trace, PITCH};
			again = bind_tail<FULL, DINK, true, (LMAX > 64), 1, WB_CODE_SHIFT, uint32_t>(a, tr, s_lut, s_hs_map, s_hs_val, s_k, win, qry, cols, best, stack, ties, NCOLS, WMAX + 1, idx, it,
			                                                                         slot, L, W, start, stop, minus, xmode, gmax, nmax, max_ij, Tcur, dk_q);
		}
		if (DINK) { // another pass at the temperature lane 0 found?
			again = __builtin_amdgcn_readfirstlane(int(again)) != 0;
			Tcur = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, Tcur)));
			dk_q = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dk_q)));
			if (again) {
				wave_sync();
				goto next_pass;
			}
		}
	}
}

} // namespace tnt
#endif
