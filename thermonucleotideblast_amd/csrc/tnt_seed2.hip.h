// tnt_seed2.hip.h -- gfx950 seed kernel of the search path (round 3): scan, diagonal de-duplication, first cull and
// emission of bind work items for WHOLE-TARGET work units, written for a low instruction count per target.
//
// Same contract as seed_kernel<false, SEED_SEARCH> (tnt_seed.hip.h, which stays for the seed lists of windowed targets,
// the unit-level API and the shared-oligo marking): DNAHash::hash + find / find_complement (seq_hash.h:524-779),
// match_oligo_to_{minus,plus}_strand (bind_oligo.cpp:84-122), the early exits of amplicon() (amplicon_search.cpp:96-126)
// and the first cull_oligo_match (amplicon_search.cpp:128,679-765).  What changed, and why (round-2 counters: 5 200
// wave-instructions per 1-kb target, 38 % of them scalar EXEC bookkeeping of branchy per-seed code, three returning
// global atomics and a dozen returning LDS atomics per target on the critical path):
//  * 32-bit keys  la(7) | target_loc(15) | assembly group(3) | word index(7): a work unit of this kernel is a whole
//    target of at most 32 768 positions and a chunk holds at most 128 assays, so the numeric order of two keys of one
//    assay IS the reference's list order (target_loc, F- R- F+ R+ P- P+, word index).  Half the LDS traffic, no 64-bit
//    shifts; candidates are (position | table entry index) in one word, the entry itself is fetched by the dense pass;
//  * kept seeds, items, groups and queue slots are numbered with wave-wide ballots / prefix counts, not with returning
//    LDS atomics; items and groups go to a region PRIVATE to the wavefront (no global atomic per target);
//  * the cull needs, per seed, "a partner exists on the other strand within max_len + 50" and the strand of the list
//    SUCCESSOR (amplicon_search.cpp:748-755).  When the whole target is shorter than that reach (1-kb and
//    2-kb targets against the default -l 2000) the first is a comparison with the per-(assay, strand, role) extreme
//    keys, collected with LDS min / max atomics while the seeds are sorted; the loop over the assay's segment that is
//    left computes the successor (and the nearest probe, for TaqMan assays) with two instructions per key.  Longer
//    targets take the general loop (reach tested per pair), written with selects;
//  * phase-A items go STRAIGHT into the queue of the length bucket that binds them (blocks of 64 slots handed out per
//    bucket, unused slots hold QUEUE_HOLE): the bucket pass over the phase-A queue (two passes over 52 M entries on
//    config 2) is gone and the pre-filter reads the windows of one target back to back.
#ifndef TNT_SEED2_HIP_H
#define TNT_SEED2_HIP_H

#include <hip/hip_runtime.h>
#include "tnt_dev.h"
#include "tnt_seed.hip.h"

namespace tnt {

constexpr int S2_WAVES = 4;            // wavefronts (= targets in flight) per workgroup
constexpr int S2_THREADS = 64 * S2_WAVES;
constexpr int S2_CAP = 512;            // seeds of one target x one assay chunk held in LDS
constexpr int S2_ASSAYS = 128;         // assays per chunk (7-bit local index in the key)
constexpr int S2_SWEEP = 16;           // positions per lane and sweep
constexpr uint32_t QUEUE_BLOCK = 64;   // queue slots handed out at a time (one filter / bind wavefront reads one block)
constexpr uint32_t QUEUE_HOLE = 0xFFFFFFFFu; // a queue slot without an item

struct Seed2Args {
	DbView db;
	const uint32_t *unit_target; // work units = whole targets
	uint32_t unit0, unit1;
	int word_size;
	const uint16_t *wt_off;      // this chunk's bucket offsets [4^k + 1], relative to wt_ent
	const uint32_t *wt_ent;      // this chunk's entries (tnt_dev.h)
	const uint16_t *wlist;       // concatenated word lists
	const uint32_t *wl_off;      // [((assay * 3 + role) * 2 + minus)] -> first word
	const unsigned long long *wl_seq; // [list][2]: the strand-specific oligo, 2 bits per base
	uint32_t assay_begin, assay_end;
	const AssayDev *assays;
	int chunk_has_probe;         // some PCR assay of the chunk has a probe (TaqMan)
	int max_len;                 // -l
	int single_primer_pcr;
	const uint32_t *slot_pack;   // [slot] bucket | bucket_general << 8 | oligo length << 16
	int has_exceptions;          // the database holds non-ACGT codes: windows that hold one go to Slot::bucket_general
	Item *items;
	uint32_t item_region;        // items of one wavefront's private region
	Group *groups;
	uint32_t group_region;
	uint32_t *region_counts;     // [2 * regions]: items, groups in every region (kept across the chunk launches of a batch)
	uint32_t *queue;             // phase-A queues, one region per length bucket
	const uint32_t *q_base;      // [bucket] first slot of the bucket's region
	const uint32_t *q_cap;       // [bucket] slots of the region (a multiple of QUEUE_BLOCK)
	uint32_t *q_count;           // [bucket * BUCKET_STRIDE] slots handed out (device counter)
	uint32_t *q_state;           // [region][MAX_BUCKETS][2]: the block every wavefront is filling (next slot, end), kept across the chunk launches
	int last_chunk;              // the last launch of the batch marks the unused slots of the open blocks
	int n_buckets;
	uint32_t *stat_items, *stat_groups, *stat_phase_a; // statistics (one add per wavefront and launch)
	uint32_t *big_count;         // groups of more than 64 items
	uint32_t *err;
};

__device__ __forceinline__ uint32_t s2_key(uint32_t la, uint32_t ent, uint32_t p)
{
	const uint32_t wi = ent & 127u, role = (ent >> 8) & 3u;
	return (la << 25) | (p << 10) | (group_rank(role, (ent & WT_MINUS) != 0) << 7) | wi;
}
__device__ __forceinline__ uint32_t s2_rank(unsigned long long m) // set bits of m below this lane
{
	return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}

__global__ __launch_bounds__(S2_THREADS, 4) void seed2_kernel(Seed2Args a)
{
	__shared__ uint32_t s_raw[S2_WAVES][S2_CAP];      // kept seeds (keys) as the dense pass finds them
	__shared__ uint32_t s_srt[S2_WAVES][S2_CAP];      // candidates of the scan, then the keys grouped by assay
	__shared__ uint16_t s_hist[S2_WAVES][S2_ASSAYS];  // seeds per assay
	__shared__ uint16_t s_off[S2_WAVES][S2_ASSAYS];   // scatter cursors, then the END of every assay's segment
	__shared__ __attribute__((aligned(16))) uint32_t s_ext[S2_WAVES][S2_ASSAYS * 4]; // per assay: smallest key of F-, of R-, largest key of F+, of R+
	__shared__ uint32_t s_agg[S2_WAVES][S2_ASSAYS];   // per assay: successors on the minus strand | on the plus strand << 10 | valid seeds << 20
	__shared__ unsigned long long s_vmask[S2_WAVES][S2_CAP / 64]; // valid seeds of every round of 64
	__shared__ uint32_t s_qh[S2_WAVES][MAX_BUCKETS], s_qcur[S2_WAVES][MAX_BUCKETS], s_qend[S2_WAVES][MAX_BUCKETS];
	__shared__ uint32_t s_qpos[S2_WAVES][MAX_BUCKETS], s_qsplit[S2_WAVES][MAX_BUCKETS], s_qnew[S2_WAVES][MAX_BUCKETS];
	__shared__ uint8_t s_af[S2_ASSAYS];               // per assay of the chunk: 1 probe / padlock assay (hash-overload lists), 2 PCR with a probe

	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t *raw = s_raw[wv], *srt = s_srt[wv];
	uint16_t *hist = s_hist[wv], *off = s_off[wv];
	uint32_t *ext = s_ext[wv], *agg = s_agg[wv];
	const uint32_t wmask = (1u << (2 * a.word_size)) - 1u;
	const uint32_t n_chunk = a.assay_end - a.assay_begin;
	const uint32_t threshold = uint32_t(a.max_len) + 50u;
	const bool single = a.single_primer_pcr != 0;

	for (uint32_t i = threadIdx.x; i < n_chunk; i += S2_THREADS) {
		const AssayDev A = a.assays[a.assay_begin + i];
		s_af[i] = uint8_t((A.kind != 0 ? 1 : 0) | ((A.kind == 0 && A.has_probe) ? 2 : 0));
	}
	// the wavefront's private output region (items, groups) and the queue blocks it is filling; both persist across the
	// chunk launches of a batch
	const uint32_t region = blockIdx.x * S2_WAVES + wv;
	if (lane < MAX_BUCKETS) {
		s_qh[wv][lane] = 0;
		s_qcur[wv][lane] = a.q_state[(region * MAX_BUCKETS + lane) * 2];
		s_qend[wv][lane] = a.q_state[(region * MAX_BUCKETS + lane) * 2 + 1];
	}
	__syncthreads();

	const uint32_t item_lo = region * a.item_region, group_lo = region * a.group_region;
	uint32_t item_cur = item_lo + a.region_counts[2 * region], group_cur = group_lo + a.region_counts[2 * region + 1];
	const uint32_t item_first = item_cur, group_first = group_cur;
	uint32_t n_phase_a = 0;

	for (uint32_t unit = a.unit0 + region; unit < a.unit1; unit += gridDim.x * S2_WAVES) {
		const uint32_t t = a.unit_target[unit];
		const uint32_t len = a.db.len[t];
		if (len < uint32_t(a.word_size)) continue; // tntblast_local.cpp:513
		const uint32_t *pk = a.db.packed + a.db.word_off[t];
		const uint32_t npos = len - a.word_size + 1;
		// the first cull's reach covers the whole target: "a partner within max_len + 50" is "a partner" (see the head)
		const bool in_reach = npos - 1u <= threshold;
		for (uint32_t i = lane; i < (n_chunk + 1u) / 2u; i += 64) reinterpret_cast<uint32_t *>(hist)[i] = 0;
		uint32_t nkept = 0; // kept seeds so far (wave-uniform)

		// ---------------------------------------------------------------- scan + diagonal de-duplication
		uint32_t *cand = srt; // position | entry index << 15
		// dense pass over the candidate list: the first seed of every (oligo, strand, diagonal) survives
		auto drain = [&](uint32_t ncand) __attribute__((always_inline)) {
			wave_sync(); // the list was written by other lanes
			for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
				const uint32_t c = c0 + uint32_t(lane);
				bool first = false;
				uint32_t key = 0, la = 0;
				if (c < ncand) {
					const uint32_t cv = cand[c];
					const uint32_t cp = cv & 0x7FFFu, ent = a.wt_ent[cv >> 15];
					const uint32_t wi = ent & 127u;
					const uint32_t assay = ent >> WT_ASSAY_SHIFT;
					const uint32_t li = (assay * 3u + ((ent >> 8) & 3u)) * 2u + ((ent & WT_MINUS) ? 1u : 0u);
					const uint32_t kmax = wi < cp ? wi : cp;
					if ((ent & WT_PLAIN) && kmax + uint32_t(a.word_size) <= 32u) first = first_on_diagonal_packed(pk, a.wl_seq + 2 * li, wi, cp, a.word_size);
					else first = first_on_diagonal_words(pk, a.wlist + a.wl_off[li], wi, cp, wmask);
					la = assay - a.assay_begin;
					key = s2_key(la, ent, cp);
				}
				const unsigned long long m = __builtin_amdgcn_ballot_w64(first);
				if (first) {
					const uint32_t slot = nkept + s2_rank(m);
					if (slot < uint32_t(S2_CAP)) {
						raw[slot] = key;
						atomicAdd(reinterpret_cast<unsigned int *>(hist) + (la >> 1), (la & 1u) ? 0x10000u : 1u);
					}
				}
				nkept += uint32_t(__builtin_popcountll(m));
			}
			wave_sync();
		};
		// position-by-position scan of [p_begin, p_end): the overflow path of the sweep (bursts of shared words)
		auto scan_slow = [&](uint32_t p_begin, uint32_t p_end) __attribute__((always_inline)) {
			uint32_t p = p_begin + lane, e_done = 0;
			while (true) {
				uint32_t filled = 0; // wave-uniform
				while (true) {
					uint32_t e0 = 0, cnt = 0;
					if (p < p_end) {
						const uint32_t w = target_word(pk, p, wmask);
						e0 = uint32_t(a.wt_off[w]) + e_done;
						const uint32_t e1 = a.wt_off[w + 1];
						cnt = e1 > e0 ? e1 - e0 : 0u;
					}
					// exclusive prefix of cnt over the lanes
					uint32_t incl = cnt;
#pragma unroll
					for (int d = 1; d < 64; d <<= 1) {
						const uint32_t y = __shfl_up(incl, d);
						if (lane >= d) incl += y;
					}
					const uint32_t at = filled + incl - cnt;
					const uint32_t fit = at < uint32_t(S2_CAP) ? (cnt < uint32_t(S2_CAP) - at ? cnt : uint32_t(S2_CAP) - at) : 0u;
					for (uint32_t x = 0; x < fit; x++) cand[at + x] = p | ((e0 + x) << 15);
					bool over = false;
					if (p < p_end) {
						if (fit == cnt) p += 64, e_done = 0;
						else e_done += fit, over = true;
					}
					const uint32_t total = __shfl(incl, 63);
					filled = filled + total < uint32_t(S2_CAP) ? filled + total : uint32_t(S2_CAP);
					if (__builtin_amdgcn_ballot_w64(over) != 0 || __builtin_amdgcn_ballot_w64(p < p_end) == 0) break;
				}
				drain(filled);
				if (__builtin_amdgcn_ballot_w64(p < p_end) == 0) break;
			}
		};
		{
			// Sweeps of 64 x 16 positions: a lane loads the two packed words that hold its positions (and the k - 1 bases
			// behind them), extracts the k-mers with shifts and reads their bucket bounds -- both bounds with one 2-byte
			// aligned dword gather, all 16 gathers in flight at once; one wave-wide prefix sum gives every lane its slice of
			// the candidate list.
			uint32_t ncand = 0;
			for (uint32_t base = 0; base < npos; base += 64 * S2_SWEEP) {
				const uint32_t p0 = base + uint32_t(lane) * S2_SWEEP;
				unsigned long long v = 0;
				if (p0 < npos) v = (unsigned long long)pk[p0 >> 4] | ((unsigned long long)pk[(p0 >> 4) + 1] << 32);
				uint32_t bnd[S2_SWEEP]; // first entry | count << 16
				uint32_t mine = 0;
#pragma unroll
				for (int q = 0; q < S2_SWEEP; q++) {
					const uint32_t w = uint32_t(v >> (2 * q)) & wmask;
					const uint32_t two = *reinterpret_cast<const u32_a2 *>(a.wt_off + w);
					const uint32_t e0 = two & 0xFFFFu, e1 = two >> 16;
					const uint32_t n = (p0 + q < npos) ? e1 - e0 : 0u;
					bnd[q] = e0 | (n << 16);
					mine += n;
				}
				uint32_t incl = mine;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t y = __shfl_up(incl, d);
					if (lane >= d) incl += y;
				}
				const uint32_t total = __shfl(incl, 63);
				if (total == 0) continue;
				if (ncand + total > uint32_t(S2_CAP)) { // make room
					drain(ncand);
					ncand = 0;
				}
				if (total > uint32_t(S2_CAP)) { // a burst (words shared by many expanded assays): position by position
					const uint32_t e = base + 64 * S2_SWEEP;
					scan_slow(base, e < npos ? e : npos);
					continue;
				}
				uint32_t o = ncand + incl - mine;
#pragma unroll
				for (int q = 0; q < S2_SWEEP; q++) {
					uint32_t e = bnd[q] & 0xFFFFu;
					for (uint32_t n = bnd[q] >> 16; n > 0; n--, e++) cand[o++] = (p0 + q) | (e << 15);
				}
				ncand += total;
			}
			drain(ncand);
		}
		uint32_t n = nkept;
		if (n > uint32_t(S2_CAP)) {
			if (lane == 0) atomicOr(a.err, ERR_SEED_OVERFLOW);
			n = S2_CAP;
		}
		if (n == 0) continue;

		// ---------------------------------------------------------------- counting sort by assay + extreme keys
		{
			constexpr int PER = S2_ASSAYS / 64;
			uint32_t c[PER], sum = 0;
#pragma unroll
			for (int m = 0; m < PER; m++) {
				const uint32_t la = uint32_t(lane) * PER + m;
				c[m] = la < n_chunk ? hist[la] : 0u;
				sum += c[m];
			}
			uint32_t incl = sum;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t y = __shfl_up(incl, d);
				if (lane >= d) incl += y;
			}
			uint32_t run = incl - sum;
#pragma unroll
			for (int m = 0; m < PER; m++) {
				const uint32_t la = uint32_t(lane) * PER + m;
				if (la < n_chunk) {
					off[la] = uint16_t(run);
					agg[la] = 0;
					if (in_reach) *reinterpret_cast<uint4 *>(ext + 4 * la) = make_uint4(~0u, ~0u, 0u, 0u);
				}
				run += c[m];
			}
		}
		wave_sync();
		for (uint32_t i = lane; i < n; i += 64) {
			const uint32_t k = raw[i];
			const uint32_t la = k >> 25, g = (k >> 7) & 7u;
			const uint32_t old = atomicAdd(reinterpret_cast<unsigned int *>(off) + (la >> 1), (la & 1u) ? 0x10000u : 1u);
			srt[(la & 1u) ? (old >> 16) : (old & 0xFFFFu)] = k;
			if (in_reach && g < 4u) { // primers: the extreme key of every (strand, role)
				if (g < 2u) atomicMin(ext + 4 * la + g, k);
				else atomicMax(ext + 4 * la + g, k);
			}
		}
		wave_sync();

		// ---------------------------------------------------------------- first cull, one lane per seed
		// The seeds sit in `srt`, grouped by assay: segment of assay la = [off[la] - hist[la], off[la]), keys of one
		// segment compare like the reference's list after the cull's stable sort by target_loc (amplicon_search.cpp:688).
		// Per PCR seed (cull_oligo_match, amplicon_search.cpp:679-765):
		//  * a primer is valid iff a partner exists: other strand, allowed oligo, on its 3' side, within max_len + 50, and
		//    -- in an assay with a probe -- beyond the nearest probe on that side;
		//  * a probe is valid iff primers flank it: a minus-strand primer before it and a plus-strand primer behind it
		//    (the closest of each oligo), allowed as a pair, within reach of each other;
		//  * the strand of the list element FOLLOWING a valid seed is counted (:748-755; past-the-end counts nothing, Q4):
		//    the counts decide which strand is bound first (:131).
		// Probe / padlock lists are bound as a whole (hash overload, bind_oligo.cpp:124-454): every seed is valid.
		for (uint32_t i0 = 0; i0 < n; i0 += 64) {
			const uint32_t i = i0 + uint32_t(lane);
			bool valid = false;
			if (i < n) {
				const uint32_t k = srt[i];
				const uint32_t la = k >> 25, g = (k >> 7) & 7u;
				const uint32_t af = s_af[la];
				if (af & 1u) {
					valid = true;
				} else {
					const uint32_t cnt = hist[la], s0 = uint32_t(off[la]) - cnt;
					if (cnt > uint32_t(LIST_MAX)) atomicOr(a.err, ERR_LIST_OVERFLOW);
					const bool minus = (0x13u >> g) & 1u;      // F-, R-, P-
					const bool probe_me = g >= 4u;
					const uint32_t role = g & 1u;              // primers: 0 F, 1 R
					const uint32_t sm = minus ? 0u : ~0u;      // primers look towards higher keys on the minus strand, lower on the plus strand
					// groups a primer may pair with: the other strand, the other oligo (any oligo with single-primer PCR)
					const uint32_t pmask = minus ? (single ? 0xCu : (4u << (1u - role))) : (single ? 0x3u : (1u << (1u - role)));
					uint32_t next_d = ~0u;  // distance - 1 to the list successor (the minimum over all keys of (key - k - 1) as an unsigned number)
					uint32_t far_d = 0;     // distance to the farthest partner within reach (0: none)
					uint32_t pn_d = ~0u;    // distance to the nearest probe on the partner's side
					if (in_reach) {
						// successor (and nearest probe): two (five) instructions per key of the segment
						if (a.chunk_has_probe) {
							for (uint32_t x = 0; x < cnt; x++) {
								const uint32_t o = srt[s0 + x];
								const uint32_t d = o - k;
								next_d = min(next_d, d - 1u);
								const uint32_t dir = (d ^ sm) - sm; // > 0 as a signed number: on the partner's side
								const bool op = ((o >> 7) & 7u) >= 4u && int32_t(dir) > 0;
								pn_d = min(pn_d, op ? dir : ~0u);
							}
						} else {
							for (uint32_t x = 0; x < cnt; x++) next_d = min(next_d, srt[s0 + x] - k - 1u);
						}
						const uint4 e = *reinterpret_cast<const uint4 *>(ext + 4 * la);
						if (!probe_me) {
							if (minus) { // partners: plus-strand primers with a larger key
								const uint32_t f = max((pmask & 4u) ? e.z : 0u, (pmask & 8u) ? e.w : 0u);
								far_d = f > k ? f - k : 0u;
							} else {     // partners: minus-strand primers with a smaller key
								const uint32_t f = min((pmask & 1u) ? e.x : ~0u, (pmask & 2u) ? e.y : ~0u);
								far_d = f < k ? k - f : 0u;
							}
							valid = (af & 2u) ? (pn_d != ~0u && far_d > pn_d) : far_d != 0u;
						} else {
							// a minus-strand primer of oligo rf before the probe, a plus-strand primer of oligo rr behind it
							const bool f0 = e.x < k, f1 = e.y < k, r0 = e.z > k, r1 = e.w > k;
							valid = (f0 && r1) || (f1 && r0) || (single && ((f0 && r0) || (f1 && r1)));
						}
					} else {
						// general loop: reach tested per pair
						const uint32_t loc = (k >> 10) & 0x7FFFu;
						uint32_t fmax[2] = {0, 0}, rmin[2] = {~0u, ~0u}; // probes: closest primers, loc + 1 / loc (0 / ~0: none)
						for (uint32_t x = 0; x < cnt; x++) {
							const uint32_t o = srt[s0 + x];
							const uint32_t d = o - k;
							next_d = min(next_d, d - 1u);
							const uint32_t og = (o >> 7) & 7u, oloc = (o >> 10) & 0x7FFFu;
							const uint32_t dir = (d ^ sm) - sm;
							const uint32_t dl = ((oloc - loc) ^ sm) - sm; // distance of the positions towards the partner's side
							const bool side = int32_t(dir) > 0;
							const bool partner = side && ((pmask >> og) & 1u) && dl <= threshold;
							far_d = max(far_d, partner ? dir : 0u);
							pn_d = min(pn_d, (side && og >= 4u) ? dir : ~0u);
							if (a.chunk_has_probe) {
								const bool before = int32_t(d) < 0 && og < 2u, behind = int32_t(d) > 0 && (og & 6u) == 2u;
								const uint32_t orole = og & 1u;
								fmax[0] = max(fmax[0], (before && orole == 0u) ? oloc + 1u : 0u);
								fmax[1] = max(fmax[1], (before && orole == 1u) ? oloc + 1u : 0u);
								rmin[0] = min(rmin[0], (behind && orole == 0u) ? oloc : ~0u);
								rmin[1] = min(rmin[1], (behind && orole == 1u) ? oloc : ~0u);
							}
						}
						if (!probe_me) {
							valid = (af & 2u) ? (pn_d != ~0u && far_d > pn_d) : far_d != 0u;
						} else {
#pragma unroll
							for (int rf = 0; rf < 2; rf++)
#pragma unroll
								for (int rr = 0; rr < 2; rr++)
									if ((single || rf != rr) && fmax[rf] != 0u && rmin[rr] != ~0u && rmin[rr] - (fmax[rf] - 1u) <= threshold) valid = true;
						}
					}
					if (valid) {
						// one LDS add per valid seed: its assay's count, and the strand of its successor
						uint32_t add = 1u << 20;
						if (int32_t(next_d) >= 0) add += // (keys below this one leave huge values behind, not ~0: a successor is less than 2^25 away)
							 ((0x13u >> (((k + next_d + 1u) >> 7) & 7u)) & 1u) ? 1u : (1u << 10);
						atomicAdd(agg + la, add);
					}
				}
				if (valid && (af & 1u)) atomicAdd(agg + la, 1u << 20);
			}
			const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
			if (lane == 0) s_vmask[wv][i0 >> 6] = vm;
		}
		wave_sync();

		// ---------------------------------------------------------------- emission: items, groups, phase-A queues
		// Phase B = what the reference binds only after both primers were tried on the first strand
		// (amplicon_search.cpp:131-353): the other strand's primers and the probe.
		uint32_t exc0 = 0, exc1 = 0;
		if (a.has_exceptions) exc0 = a.db.exc_off[t], exc1 = a.db.exc_off[t + 1];
		uint32_t carry_la = ~0u; // assay of the last valid seed so far
		bool overflow = false;
		for (uint32_t i0 = 0; i0 < n && !overflow; i0 += 64) {
			const uint32_t i = i0 + uint32_t(lane);
			const unsigned long long vm = s_vmask[wv][i0 >> 6];
			if (vm == 0) continue;
			const bool valid = (vm >> lane) & 1ull;
			const uint32_t k = i < n ? srt[i] : 0u;
			const uint32_t la = k >> 25, g = (k >> 7) & 7u, wi = k & 127u, p = (k >> 10) & 0x7FFFu;
			const uint32_t ag = valid ? agg[la] : 0u;
			const bool minus_first = (ag & 1023u) < ((ag >> 10) & 1023u);
			const bool minus = (0x13u >> g) & 1u;
			const uint32_t role = g >= 4u ? 2u : (g & 1u);
			const bool nonpcr = valid && (s_af[la] & 1u);
			const bool pb = valid && !nonpcr && (role == 2u || minus != minus_first);
			const uint32_t n_valid = uint32_t(__builtin_popcountll(vm));
			// the first valid seed of every assay opens the group
			const unsigned long long before = vm & ((1ull << lane) - 1ull);
			uint32_t la_prev = carry_la;
			{
				const int pl = before ? 63 - __builtin_clzll(before) : lane;
				const uint32_t got = uint32_t(__shfl(int(la), pl));
				if (before) la_prev = got;
			}
			const bool first = valid && la_prev != la;
			const unsigned long long fm = __builtin_amdgcn_ballot_w64(first);
			const uint32_t n_first = uint32_t(__builtin_popcountll(fm));
			if (item_cur + n_valid > item_lo + a.item_region) {
				if (lane == 0) atomicOr(a.err, ERR_ITEM_OVERFLOW);
				overflow = true;
				continue;
			}
			if (group_cur + n_first > group_lo + a.group_region) {
				if (lane == 0) atomicOr(a.err, ERR_GROUP_OVERFLOW);
				overflow = true;
				continue;
			}
			const uint32_t o = item_cur + s2_rank(vm);
			const uint32_t gidx = group_cur + uint32_t(__builtin_popcountll(fm & ((2ull << lane) - 1ull))) - 1u;
			const uint32_t assay = a.assay_begin + la;
			if (first) {
				const uint32_t nv = ag >> 20;
				a.groups[gidx] = Group{unit, assay | (minus_first ? GROUP_MINUS_FIRST : 0u), o, nv};
				if (nv > 64u) atomicAdd(a.big_count, 1u);
			}
			uint32_t bucket = ~0u;
			if (valid) {
				const uint32_t slot = assay * 3u + role; // the Slot table is laid out [assay][role]
				Item it;
				it.target = t;
				it.tloc = p;
				it.info = wi | (minus ? ITEM_MINUS : 0u) | (pb ? ITEM_PHASE_B : 0u) | (slot << ITEM_SLOT_SHIFT);
				it.group = gidx;
				a.items[o] = it;
				if (!pb) {
					const uint32_t sp = a.slot_pack[slot];
					bucket = sp & 0xFFu;
					if (exc1 > exc0) { // same window as the bind kernel, bind_oligo.cpp:502-505
						int start = int(p) - (int(wi) + FLANK);
						if (start < 0) start = 0;
						const uint32_t stop = uint32_t(start) + (sp >> 16) + 2u * FLANK;
						uint32_t lo = exc0, hi = exc1;
						while (lo < hi) {
							const uint32_t mid = (lo + hi) >> 1;
							if (a.db.exc_pos[mid] < uint32_t(start)) lo = mid + 1;
							else hi = mid;
						}
						if (lo < exc1 && a.db.exc_pos[lo] < stop) bucket = (sp >> 8) & 0xFFu;
					}
				}
			}
			// phase-A items of this round into the queues of their buckets
			const unsigned long long am = __builtin_amdgcn_ballot_w64(bucket != ~0u);
			if (am) {
				uint32_t r = 0;
				if (bucket != ~0u) r = atomicAdd(&s_qh[wv][bucket], 1u);
				wave_sync();
				if (lane < a.n_buckets) {
					const uint32_t h = s_qh[wv][lane];
					if (h) {
						uint32_t cur = s_qcur[wv][lane];
						const uint32_t end = s_qend[wv][lane];
						s_qpos[wv][lane] = cur;
						s_qsplit[wv][lane] = end;
						if (cur + h > end) { // the block is full: the rest of the round goes to a new one
							const uint32_t at = atomicAdd(a.q_count + lane * BUCKET_STRIDE, QUEUE_BLOCK);
							if (at + QUEUE_BLOCK > a.q_cap[lane]) {
								atomicOr(a.err, ERR_ITEM_OVERFLOW);
								s_qnew[wv][lane] = ~0u;
							} else {
								const uint32_t nb = a.q_base[lane] + at;
								s_qnew[wv][lane] = nb;
								s_qend[wv][lane] = nb + QUEUE_BLOCK;
								cur = nb - (end - cur); // so that cur + h lands h - (end - cur) slots into the new block
							}
						}
						s_qcur[wv][lane] = cur + h;
						s_qh[wv][lane] = 0;
					}
				}
				wave_sync();
				if (bucket != ~0u) {
					uint32_t pos = s_qpos[wv][bucket] + r;
					const uint32_t split = s_qsplit[wv][bucket];
					if (pos >= split) {
						const uint32_t nb = s_qnew[wv][bucket];
						pos = nb == ~0u ? ~0u : nb + (pos - split);
					}
					if (pos != ~0u) a.queue[pos] = o;
				}
				n_phase_a += uint32_t(__builtin_popcountll(am));
			}
			item_cur += n_valid;
			group_cur += n_first;
			carry_la = uint32_t(__shfl(int(la), 63 - __builtin_clzll(vm)));
		}
		wave_sync();
	}

	// what is left of the wavefront's last blocks holds no item
	wave_sync();
	if (lane < a.n_buckets) {
		const uint32_t cur = s_qcur[wv][lane], end = s_qend[wv][lane];
		if (a.last_chunk)
			for (uint32_t x = cur; x < end; x++) a.queue[x] = QUEUE_HOLE;
		a.q_state[(region * MAX_BUCKETS + lane) * 2] = cur;
		a.q_state[(region * MAX_BUCKETS + lane) * 2 + 1] = end;
	}
	if (lane == 0) {
		a.region_counts[2 * region] = item_cur - item_lo;
		a.region_counts[2 * region + 1] = group_cur - group_lo;
		const uint32_t shard = region & (N_SHARD - 1);
		if (item_cur != item_first) atomicAdd(a.stat_items + shard * SHARD_STRIDE, item_cur - item_first);
		if (group_cur != group_first) atomicAdd(a.stat_groups + shard * SHARD_STRIDE, group_cur - group_first);
		if (n_phase_a) atomicAdd(a.stat_phase_a + shard * SHARD_STRIDE, n_phase_a);
	}
}

} // namespace tnt
#endif
