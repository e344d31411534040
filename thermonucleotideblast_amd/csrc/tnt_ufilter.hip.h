// tnt_ufilter.hip.h -- gfx950 pre-filter of the bind stage on queues REGROUPED BY OLIGO (round 5).
//
// Same bound, same automaton, same integers as filter_kernel (tnt_filter.hip.h; derivation against
// nuc_cruc.cpp:1620-2299 and bind_oligo.cpp:600-606 in DESIGN.md section 4.2a).  What changes is who sits next to whom:
//
//  * filter_kernel takes its queue as the seed kernel filled it -- the windows of one target back to back, every lane of a
//    wavefront another oligo.  Its one table read per cell is then a scatter over (target dinucleotide x query dinucleotide):
//    PMC (profiles/r04z_pmc_config2_1M_serial.json) counted 3.8 bank-conflict cycles per LDS instruction, 65 % of the
//    LDS-array cycles of the largest kernel family of the step, and the four SIMDs of a CU share that one array.
//  * Here three small kernels (regroup_count / regroup_scan / regroup_scatter: a counting sort by (oligo, threshold class)
//    of the 16-byte queue entries of one length bucket, runs padded to whole blocks of 64 with QUEUE_HOLE) put the windows
//    of ONE oligo into every wavefront.  The query side of the table index is then the same for all 64 lanes, so the
//    wavefront keeps a table of its oligo in LDS, T[k][i] = plane[k * 49 + 7 * q'(i) + q(i)] for row i and target
//    dinucleotide k = x' * 6 + x, and a cell's read is  T + k(lane) * S + i : an immediate offset on a per-column base
//    register (no address arithmetic per cell: the fourth VALU instruction of the old cell is gone), two rows per
//    ds_read_b64, and lanes with different k fall on different bank pairs (S / 2 odd), lanes with the same k broadcast.
//  * The window is put in column order once per window (minus-strand lanes: reversed and complemented), so the target
//    base of column j is a bit field at a wave-uniform position.
//
// The result per window -- bound, threshold, pass / reject -- is the old kernel's, integer by integer
// (tests/test_gpu_parity.py::test_filter_kernel_equals_its_checker runs both against the oracle-side checker); the general
// buckets (windows with a degenerate target base) take the DEGEN instance below; phase B, oligos beyond 64 bases and the
// unit-level routes stay with filter_kernel.  Survivors leave as one dense list per bucket or -- for the second tier,
// tnt_sbound.hip.h -- group by group (UFilterArgs::bin_fill).
#ifndef TNT_UFILTER_HIP_H
#define TNT_UFILTER_HIP_H

#include <type_traits>
#include "tnt_filter.hip.h"

namespace tnt {

constexpr uint32_t RG_HOLE = 0xFFFFFFFFu;   // == QUEUE_HOLE (tnt_seed2.hip.h): a queue slot without an item
constexpr int RG_MAX_BINS = 4096;           // (oligo, class) pairs of one length bucket the regrouping handles (LDS histogram); buckets with more keep filter_kernel
constexpr int RG_THREADS = 256;
constexpr int RG_TILE = 8 * RG_THREADS;     // entries a workgroup of the scatter pass ranks at a time
constexpr uint32_t UF_BLOCK = 64;           // windows of one (oligo, class) a wavefront takes at a time
constexpr uint32_t UF_CHUNK = 8;            // blocks a wavefront grabs per atomic

// target dinucleotides of a plain window: k = x' * FILT_X + x with x', x in A C G T NONE
constexpr int UF_K = FILT_NONE * FILT_X + FILT_NONE + 1;
// row pitch of the per-oligo table in dwords: even (two rows per ds_read_b64), half of it odd (the bank pair of (k, i) is
// (k * S / 2 + i / 2) mod 32: different k, different pair, for every k below 32)
constexpr int uf_stride(int lmax)
{
	int h = (lmax + 1) / 2;
	if (!(h & 1)) h++;
	return 2 * h;
}

// One length bucket of a regrouping launch (the launch serves all buckets of a phase at once: three launches per batch
// instead of three per bucket, each of which had a start-up and a tail of its own)
struct RegroupBucket {
	uint32_t src_base, src_cap;  // the bucket's region of the phase-A queue as the seed kernel filled it (blocks with holes)
	uint32_t dst_base;           // ... of the regrouped queue: src_cap + UF_BLOCK * n_bins slots
	uint32_t bin_base, n_bins;   // its share of the histogram / cursor arrays
	uint32_t general;            // 1: a general bucket (windows with a degenerate target base): bins by Slot::bucket_general
	uint32_t counter;            // word offset of its counter line: [0] slots handed out, [4] out: regrouped entries, [5] work counter of its filter launch
	uint32_t pad;
};

struct RegroupArgs {
	const RegroupBucket *buckets;
	uint32_t n_buckets;
	const uint4 *src;            // phase-A queue buffer
	uint4 *dst;                  // regrouped queue buffer
	uint32_t *counters;          // bucket counter lines of the batch
	const uint32_t *slot_bin;    // [slot][2] rank of the slot's (oligo, threshold class) inside its plain / its general bucket
	uint32_t *hist;              // [bins of all buckets] entries per bin (zeroed per batch)
	uint32_t *cursor;            // next free slot of every bin's run
};

#ifdef TNT_REGROUP_KERNELS // (the three regrouping kernels are compiled once, in tnt_host.hip's translation unit)
// chunk c of a launch -> (bucket, first entry): the buckets' chunk counts are prefix-summed by every workgroup (a few dozen words)
struct RegroupMap {
	uint32_t first[MAX_BUCKETS + 1]; // first chunk of every bucket
	uint32_t n[MAX_BUCKETS];         // entries of every bucket
};
__device__ __forceinline__ void regroup_map(const RegroupArgs &a, uint32_t chunk, RegroupMap &m)
{
	if (threadIdx.x == 0) {
		uint32_t run = 0;
		for (uint32_t b = 0; b < a.n_buckets; b++) {
			uint32_t n = a.counters[a.buckets[b].counter];
			n = n < a.buckets[b].src_cap ? n : a.buckets[b].src_cap;
			m.first[b] = run, m.n[b] = n;
			run += (n + chunk - 1) / chunk;
		}
		m.first[a.n_buckets] = run;
	}
	__syncthreads();
}
__device__ __forceinline__ uint32_t regroup_bucket_of(const RegroupMap &m, uint32_t n_buckets, uint32_t c)
{
	uint32_t b = 0;
	while (b + 1 < n_buckets && m.first[b + 1] <= c) b++;
	return b;
}

// pass 1: entries per bin
constexpr uint32_t RG_COUNT_CHUNK = 64 * RG_THREADS;
__global__ __launch_bounds__(RG_THREADS) void regroup_count_kernel(RegroupArgs a)
{
	__shared__ uint32_t s_hist[RG_MAX_BINS];
	__shared__ RegroupMap s_map;
	regroup_map(a, RG_COUNT_CHUNK, s_map);
	const uint32_t n_chunks = s_map.first[a.n_buckets];
	for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
		const uint32_t b = regroup_bucket_of(s_map, a.n_buckets, c);
		const RegroupBucket B = a.buckets[b];
		const uint32_t lo = (c - s_map.first[b]) * RG_COUNT_CHUNK, hi = lo + RG_COUNT_CHUNK < s_map.n[b] ? lo + RG_COUNT_CHUNK : s_map.n[b];
		for (uint32_t x = threadIdx.x; x < B.n_bins; x += RG_THREADS) s_hist[x] = 0;
		__syncthreads();
		const uint4 *src = a.src + B.src_base;
		for (uint32_t i = lo + threadIdx.x; i < hi; i += RG_THREADS) {
			const uint4 e = src[i];
			if (e.x != RG_HOLE) atomicAdd(&s_hist[a.slot_bin[2u * (e.w >> ITEM_SLOT_SHIFT) + B.general]], 1u);
		}
		__syncthreads();
		for (uint32_t x = threadIdx.x; x < B.n_bins; x += RG_THREADS)
			if (s_hist[x]) atomicAdd(a.hist + B.bin_base + x, s_hist[x]);
		__syncthreads();
	}
}

// pass 2 (one workgroup per bucket): the runs -- every bin's count rounded up to whole blocks --, their padding, the total
__global__ __launch_bounds__(1024) void regroup_scan_kernel(RegroupArgs a)
{
	__shared__ uint32_t s_part[1024];
	__shared__ uint32_t s_carry;
	const RegroupBucket B = a.buckets[blockIdx.x];
	uint4 *dst = a.dst + B.dst_base;
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (uint32_t b0 = 0; b0 < B.n_bins; b0 += 1024) {
		const uint32_t b = b0 + threadIdx.x;
		const uint32_t cnt = b < B.n_bins ? a.hist[B.bin_base + b] : 0u;
		const uint32_t padded = (cnt + UF_BLOCK - 1) & ~(UF_BLOCK - 1);
		s_part[threadIdx.x] = padded;
		__syncthreads();
		for (uint32_t d = 1; d < 1024; d <<= 1) { // inclusive scan
			const uint32_t v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0u;
			__syncthreads();
			s_part[threadIdx.x] += v;
			__syncthreads();
		}
		const uint32_t start = s_carry + s_part[threadIdx.x] - padded;
		if (b < B.n_bins) {
			a.cursor[B.bin_base + b] = start;
			for (uint32_t x = cnt; x < padded; x++) dst[start + x] = make_uint4(RG_HOLE, 0, 0, 0);
		}
		__syncthreads();
		if (threadIdx.x == 1023) s_carry += s_part[1023];
		__syncthreads();
	}
	if (threadIdx.x == 0) a.counters[B.counter + 4] = s_carry;
}

// pass 3: every entry to its bin's run.  A workgroup ranks a tile of entries per bin in LDS, reserves the tile's share of
// every run with one atomic per (tile, bin) and stores the entries -- neighbours in a run come from one tile, so the 16-byte
// stores of a tile complete each other's lines in L2.
__global__ __launch_bounds__(RG_THREADS) void regroup_scatter_kernel(RegroupArgs a)
{
	__shared__ uint32_t s_cnt[RG_MAX_BINS], s_base[RG_MAX_BINS];
	__shared__ RegroupMap s_map;
	regroup_map(a, RG_TILE, s_map);
	const uint32_t n_chunks = s_map.first[a.n_buckets];
	for (uint32_t x = threadIdx.x; x < RG_MAX_BINS; x += RG_THREADS) s_cnt[x] = 0;
	__syncthreads();
	for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
		const uint32_t b = regroup_bucket_of(s_map, a.n_buckets, c);
		const RegroupBucket B = a.buckets[b];
		const uint32_t t0 = (c - s_map.first[b]) * RG_TILE, n = s_map.n[b];
		const uint4 *src = a.src + B.src_base;
		uint4 *dst = a.dst + B.dst_base;
		constexpr int PER = RG_TILE / RG_THREADS;
		uint4 e[PER];
		uint32_t br[PER]; // bin | rank << 12
#pragma unroll
		for (int r = 0; r < PER; r++) {
			const uint32_t i = t0 + r * RG_THREADS + threadIdx.x;
			e[r] = i < n ? src[i] : make_uint4(RG_HOLE, 0, 0, 0);
		}
#pragma unroll
		for (int r = 0; r < PER; r++) {
			br[r] = ~0u;
			if (e[r].x != RG_HOLE) {
				const uint32_t bin = a.slot_bin[2u * (e[r].w >> ITEM_SLOT_SHIFT) + B.general];
				br[r] = bin | (atomicAdd(&s_cnt[bin], 1u) << 12);
			}
		}
		__syncthreads();
		for (uint32_t x = threadIdx.x; x < B.n_bins; x += RG_THREADS) {
			const uint32_t k = s_cnt[x];
			if (k) s_base[x] = atomicAdd(a.cursor + B.bin_base + x, k), s_cnt[x] = 0;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < PER; r++)
			if (br[r] != ~0u) dst[s_base[br[r] & 4095u] + (br[r] >> 12)] = e[r];
		__syncthreads();
	}
}
static_assert(RG_MAX_BINS == 4096 && RG_TILE <= (1 << 20), "bin | rank << 12 in one word");
#endif

struct UFilterArgs {
	const FilterTables *tab;
	const uint32_t *packed;       // DbView::packed
	const uint32_t *olig_q4;      // [n][MAX_OLIGO / 4] reversed oligo, one byte per row = 4 * base code
	const int32_t *olig_len;
	const Slot *slots;
	const uint4 *fat;             // the regrouped queue: blocks of UF_BLOCK entries of one (oligo, class), holes behind a run's last entry
	const uint32_t *n_items_dev;  // its length
	uint32_t n_items;             // ... capped at the region
	uint32_t *next_block;         // work counter of the launch (zeroed per batch)
	uint32_t *survivors;          // out: item indices that go on to the full kernel
	uint32_t *n_survivors;
	uint32_t *n_rejected;         // statistics
	uint8_t *audit;               // optional: audit mode marks what the filter would reject and lets everything through
	int32_t *score_out;           // optional (unit-level entry): bound and threshold per item, [2 * idx]
	int32_t status_exact;         // reject only where the status is certain (bounds against the minimum Tm)
	// the instance for the general buckets (windows that hold a non-ACGT target code): the exception lists go by target and position
	DbView db;
	const Item *items;
	// Survivors BY GROUP (for the second tier, tnt_sbound.hip.h, whose wavefronts want one oligo each as this kernel's do): set,
	// `survivors` has the regrouped queue's geometry and the survivors of a group go to the front of the group's own run --
	// survivors[bin_end[g] - bin_count[g] + (position handed out by bin_fill[g])].  Null: one dense list behind n_survivors.
	const uint32_t *slot_bin;     // [slot][2] group of the slot inside its plain / its general bucket
	const uint32_t *bin_end;      // [groups of the bucket] end of the group's entries in the regrouped queue (the scatter pass's cursor)
	const uint32_t *bin_count;    // ... their number (the histogram)
	uint32_t *bin_fill;           // ... survivors so far (zeroed per batch)
	uint32_t general;
};

// DEGEN: the instance for the general buckets -- windows with a non-ACGT target code.  As in filter_kernel<L, true>: an IUPAC code
// or an inosine is the column value ANY (entry by entry the maximum over the bases it can stand for), a code the window loader
// skips (gap, unknown: the window shrinks, Q6) makes the window pass unfiltered.
template <int LMAX, bool DEGEN> __global__ __launch_bounds__(256) void ufilter_kernel(UFilterArgs a)
{
	static_assert(LMAX <= 64, "the 128-row instances update one register set in place: filter_kernel");
	constexpr int UF_KK = DEGEN ? FILT_X * FILT_X : UF_K; // target dinucleotides: ANY joins A C G T NONE
	constexpr int THREADS = 256, WAVES = THREADS / 64;
	constexpr int WMAX = LMAX + 2 * FLANK;
	constexpr int NW = (WMAX + 15) / 16; // packed words of a window in column order
	constexpr int S = uf_stride(LMAX);
	static_assert(UF_K <= 32 && (S % 2) == 0 && ((S / 2) % 2) == 1 && S >= LMAX, "table pitch"); // (ANY columns reach k = 35: k and k - 32 share a bank pair)
	__shared__ __attribute__((aligned(16))) uint32_t s_row[WAVES][UF_KK * S];
	__shared__ uint32_t s_keep[WAVES][128]; // survivors of a wavefront on their way to the list
	__shared__ uint32_t s_wcount[WAVES], s_wbase;
	typedef short s2 __attribute__((ext_vector_type(2))); // (X, M2) of a cell, or (stack, loop entry) of a table entry: one register
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	// (volatile: the optimiser would merge neighbouring reads into ds_read2_b64, which moves half the bytes per LDS cycle of
	// ds_read_b64 and banks modulo 32, MI355X_MICROARCH.md LDS table)
	typedef __attribute__((address_space(3))) const volatile u32x2 lds_cu2;
	typedef __attribute__((address_space(3))) const uint32_t lds_cu32;

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t n_items = *a.n_items_dev;
	n_items = n_items < a.n_items ? n_items : a.n_items;
	const uint32_t n_blocks = n_items / UF_BLOCK;
	uint32_t n_rej = 0, n_keep = 0; // windows this wavefront rejected / holds in s_keep (wave-uniform)
	uint32_t keep_bin = 0;          // a.bin_fill: the group the held survivors belong to
	int cur_oligo = -1, cur_cls = -1;
	// the held survivors (n of them, the first ones) to their group's run / to the dense list
	auto flush = [&](const uint32_t n) __attribute__((always_inline)) {
		filt_wave_sync();
		uint32_t at = 0;
		if (lane == 0) at = a.bin_fill ? a.bin_end[keep_bin] - a.bin_count[keep_bin] + atomicAdd(a.bin_fill + keep_bin, n) : atomicAdd(a.n_survivors, n);
		at = uint32_t(__builtin_amdgcn_readfirstlane(int(at)));
		if (uint32_t(lane) < n) a.survivors[at + uint32_t(lane)] = s_keep[wave][lane];
		const uint32_t rest = n_keep - n; // < 64
		const uint32_t moved = uint32_t(lane) < rest ? s_keep[wave][n + uint32_t(lane)] : 0u;
		filt_wave_sync();
		if (uint32_t(lane) < rest) s_keep[wave][lane] = moved;
		n_keep = rest;
		filt_wave_sync();
	};
	uint32_t *row = s_row[wave];
	const int row_base = int(uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)row)));

	for (;;) {
		uint32_t c0 = 0;
		if (lane == 0) c0 = atomicAdd(a.next_block, UF_CHUNK);
		c0 = uint32_t(__builtin_amdgcn_readfirstlane(int(c0)));
		if (c0 >= n_blocks) break;
		const uint32_t c1 = c0 + UF_CHUNK < n_blocks ? c0 + UF_CHUNK : n_blocks;
		for (uint32_t blk = c0; blk < c1; blk++) {
			const uint4 fe = a.fat[blk * UF_BLOCK + uint32_t(lane)];
			const uint32_t idx = fe.x;
			const bool active = idx != RG_HOLE;
			const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
			if (am == 0) continue;
			// the block's oligo and threshold class: those of its first entry (a run holds one (oligo, class))
			const int lead = __builtin_ctzll(am);
			const uint32_t lead_slot = uint32_t(__builtin_amdgcn_readlane(int(fe.w), lead)) >> ITEM_SLOT_SHIFT;
			const Slot ls = a.slots[lead_slot];
			const int oligo = __builtin_amdgcn_readfirstlane(ls.oligo), cls = __builtin_amdgcn_readfirstlane(filter_class(ls.thresh));
			const int L = __builtin_amdgcn_readfirstlane(a.olig_len[oligo]);
			if (a.bin_fill) { // the survivors held so far belong to another group: out with them
				const uint32_t bin = uint32_t(__builtin_amdgcn_readfirstlane(int(a.slot_bin[2u * lead_slot + a.general])));
				if (bin != keep_bin && n_keep) flush(n_keep);
				keep_bin = bin;
			}
			if (oligo != cur_oligo || cls != cur_cls) {
				// T[k][i] = plane[cls][k * 49 + 7 * q' + q]: q the base of row i (none behind the oligo: such rows add nothing and
				// open nothing), q' the base of the row above (none above the first)
				filt_wave_sync(); // (the lanes' reads of the last block's table are through)
				const uint32_t *plane = a.tab->plane[cls];
				const uint32_t *q4 = a.olig_q4 + size_t(oligo) * (MAX_OLIGO / 4);
				for (int e = lane; e < UF_KK * S; e += 64) {
					const int k = e / S, i = e - k * S;
					uint32_t q = bGAP, qp = bGAP;
					if (i < L) q = (q4[i >> 2] >> ((i & 3) * 8 + 2)) & 7u;
					if (i >= 1 && i - 1 < L) qp = (q4[(i - 1) >> 2] >> (((i - 1) & 3) * 8 + 2)) & 7u;
					row[e] = plane[k * NBP + int(7u * qp + q)];
				}
				cur_oligo = oligo, cur_cls = cls;
				filt_wave_sync();
			}
			const Slot slot = a.slots[active ? fe.w >> ITEM_SLOT_SHIFT : lead_slot];
			const bool minus = (fe.w & ITEM_MINUS) != 0;
			const int W = active ? int((fe.z >> 4) & 0xFFu) : 0;

			// The window in column order, 2 bits per column, column j at bits 2 (j - 1): plus-strand binding reads the bases as
			// they lie (bind_oligo.cpp:1227-1276), minus-strand binding the complement of base W - j (:524-573)
			uint32_t ov[NW];
			uint32_t om[DEGEN ? NW : 1]; // DEGEN: bit 2 (j - 1) set: column j holds an IUPAC code or an inosine
			bool unfilterable = false;
			{
				const uint32_t *pk = a.packed + fe.y;
				uint32_t raw[NW + 1];
#pragma unroll
				for (int k = 0; k <= NW; k++) raw[k] = active ? pk[k] : 0u; // (rows are padded and the array has sixteen words of slack behind its end)
				const uint32_t sh = (fe.z & 15u) * 2u;
#pragma unroll
				for (int k = 0; k < NW; k++) ov[k] = __builtin_amdgcn_alignbit(raw[k + 1], raw[k], sh);
				if (DEGEN) {
					// the window's non-ACGT codes from the target's exception list (window: bind_oligo.cpp:502-505)
#pragma unroll
					for (int k = 0; k < NW; k++) om[DEGEN ? k : 0] = 0u;
					if (active) {
						const Item it = a.items[idx];
						int start = int(it.tloc) - (int(fe.w & 127u) + FLANK);
						if (start < 0) start = 0;
						const int stop = start + W;
						const uint32_t e0 = a.db.exc_off[it.target], e1 = a.db.exc_off[it.target + 1];
						uint32_t lo = e0, hi = e1;
						while (lo < hi) {
							const uint32_t mid = (lo + hi) >> 1;
							if (a.db.exc_pos[mid] < uint32_t(start)) lo = mid + 1;
							else hi = mid;
						}
						for (uint32_t x = lo; x < e1 && a.db.exc_pos[x] < uint32_t(stop); x++) {
							if (a.db.exc_code[x] > 15) unfilterable = true; // skipped codes; inosine (4) and the IUPAC codes read as ANY
							const int r = int(a.db.exc_pos[x]) - start;
#pragma unroll
							for (int k = 0; k < NW; k++)
								if ((r >> 4) == k) om[DEGEN ? k : 0] |= 1u << ((r & 15) * 2);
						}
					}
				}
				if (__builtin_amdgcn_ballot_w64(minus) != 0) {
					// reversed (and, the bases, complemented): bit-reverse every word and swap the words, put the two bits of a base
					// back in order, shift the 16 * NW - W bases of slack out
					auto reversed = [&](uint32_t *x, const uint32_t flip) __attribute__((always_inline)) {
						uint32_t rv[NW + 1];
#pragma unroll
						for (int k = 0; k < NW; k++) {
							const uint32_t r = __builtin_bitreverse32(x[NW - 1 - k]);
							rv[k] = ((((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u))) ^ flip;
						}
						rv[NW] = 0;
						const uint32_t slack = uint32_t(16 * NW - W) * 2u; // bits; below 32 * NW for a window of at least one base
						// whole words first (NW <= 5: each pass drops one word where `slack >> 5` still asks for it), then the bits
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
							x[k] = minus ? v : x[k];
						}
					};
					reversed(ov, ~0u);
					if (DEGEN) reversed(om, 0u);
				}
			}

			int Wwave = W;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const int other = __shfl_xor(Wwave, o);
				Wwave = other > Wwave ? other : Wwave;
			}
			Wwave = __builtin_amdgcn_readfirstlane(Wwave);
			// lanes whose window is narrower than the widest of the block (clipped at an end of the target; holes) see "no base" behind it
			const bool ragged = __builtin_amdgcn_ballot_w64(W != Wwave) != 0;

			// The sweep: filter_kernel's, two columns per trip on alternate register sets; per cell three packed instructions
			//   a = entry + (dX, dX)   v_pk_add_i16, saturating        m = max(up, old)   v_pk_max_i16        n = max(a, (m.hi, m.hi))   v_pk_max_i16
			// and the entry read: T + k * S (one register per column) + 4 i (immediate).
			const s2 EDGE = {0, short(FILT_NEG)}; // a cell outside the matrix: X = 0, M2 = none
			s2 PA[LMAX], PB[LMAX];
#pragma unroll
			for (int i = 0; i < LMAX; i++) PA[i] = EDGE;
			s2 bestpk = EDGE;
			auto column = [&](const s2 *Pin, s2 *Pout, const int colbase) __attribute__((always_inline)) {
				s2 e[LMAX + 1];
#pragma unroll
				for (int i = 0; i + 1 < LMAX; i += 2) {
					const u32x2 v = *(lds_cu2 *)(uintptr_t(uint32_t(colbase + 4 * i)));
					const uint32_t v0 = v.x, v1 = v.y; // (copies: __builtin_bit_cast of a vector element reads element 0 whatever the element)
					e[i] = __builtin_bit_cast(s2, v0), e[i + 1] = __builtin_bit_cast(s2, v1);
				}
				if (LMAX & 1) e[LMAX - 1] = __builtin_bit_cast(s2, *(lds_cu32 *)(uintptr_t(uint32_t(colbase + 4 * (LMAX - 1)))));
				s2 D = EDGE, U = EDGE; // (1, j) sees the edge above it and diagonally
#pragma unroll
				for (int i = 1; i <= LMAX; i++) {
					const s2 O = Pin[i - 1]; // (i, j-1)
					const s2 sum = __builtin_elementwise_add_sat(e[i - 1], (s2){D.x, D.x});
					const s2 m = __builtin_elementwise_max(U, O);
					const s2 n = __builtin_elementwise_max(sum, (s2){m.y, m.y});
					Pout[i - 1] = n;
					D = O, U = n;
				}
				bestpk = __builtin_elementwise_max(bestpk, U); // the last row of this column
			};
			// column base of (x', x): row_base + (x' * FILT_X + x) * S * 4; `tpS` carries x' * FILT_X * S * 4 from column to column
			constexpr int KS = S * 4;
			int tpS = FILT_NONE * FILT_X * KS; // no column before the first
			auto sweep = [&](auto ragged_tag) __attribute__((always_inline)) {
				constexpr bool RAGGED = decltype(ragged_tag)::value;
#pragma unroll
				for (int w = 0; w < NW; w++) {
					const int j0 = 16 * w; // this word holds columns j0 + 1 .. j0 + 16
					if (j0 >= Wwave) break;
					const int ncol = Wwave - j0 < 16 ? Wwave - j0 : 16;
					const uint32_t word = ov[w];
#pragma unroll 1
					for (int c = 0; c < ncol; c += 2) {
						int t0 = int(__builtin_amdgcn_ubfe(word, uint32_t(2 * c), 2u));
						int t1 = int(__builtin_amdgcn_ubfe(word, uint32_t(2 * c + 2), 2u));
						if (DEGEN) {
							t0 = ((om[DEGEN ? w : 0] >> (2 * c)) & 1u) ? FILT_ANY : t0;
							t1 = ((om[DEGEN ? w : 0] >> (2 * c + 2)) & 1u) ? FILT_ANY : t1;
						}
						if (RAGGED) {
							t0 = j0 + c + 1 <= W ? t0 : FILT_NONE;
							t1 = j0 + c + 2 <= W ? t1 : FILT_NONE;
						} else if (c + 1 >= ncol) {
							t1 = FILT_NONE; // (the padding column of an odd width: wave-uniform)
						}
						const int a0 = t0 * KS, a1 = t1 * KS;
						column(PA, PB, row_base + tpS + a0);
						column(PB, PA, row_base + a0 * FILT_X + a1);
						tpS = a1 * FILT_X;
					}
				}
			};
			if (ragged) sweep(std::true_type());
			else sweep(std::false_type());
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
			if (a.survivors) {
				// survivors collect in a per-wavefront LDS buffer and leave for the list 64 at a time (filter_kernel)
				const unsigned long long m = __builtin_amdgcn_ballot_w64(active && pass);
				n_rej += uint32_t(__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && !pass)));
				if (active && pass) s_keep[wave][n_keep + uint32_t(__builtin_popcountll(m & ((1ull << lane) - 1ull)))] = idx;
				n_keep += uint32_t(__builtin_popcountll(m));
				if (n_keep >= 64u) flush(64u);
				filt_wave_sync();
			}
		}
	}
	if (a.survivors && a.bin_fill) {
		if (n_keep) flush(n_keep);
	} else if (a.survivors) { // what is left in the buffers: one atomic per workgroup
		if (lane == 0) s_wcount[wave] = n_keep;
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t run = 0;
			for (int w = 0; w < WAVES; w++) {
				const uint32_t c = s_wcount[w];
				s_wcount[w] = run;
				run += c;
			}
			s_wbase = run ? atomicAdd(a.n_survivors, run) : 0u;
		}
		__syncthreads();
		if (uint32_t(lane) < n_keep) a.survivors[s_wbase + s_wcount[wave] + lane] = s_keep[wave][lane];
	}
	if (a.n_rejected && lane == 0 && n_rej) atomicAdd(a.n_rejected, n_rej); // statistics: once per wavefront
}

// host-side launchers, defined in tnt_bind_inst.hip next to the filter instances of the same lengths (TNT_FILTER_LADDER)
#define X(N) void launch_ufilter_##N(const UFilterArgs &a, int blocks, hipStream_t stream);
TNT_FILTER_LADDER(X)
#undef X
#define X(N) void launch_ufilter_degen_##N(const UFilterArgs &a, int blocks, hipStream_t stream);
TNT_LMAX_LADDER(X)
#undef X

} // namespace tnt
#endif
