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
//  * the stage is a chain of dependent steps per target and was parked in s_waitcnt for 60 % of its wave-cycles (PMC,
//    profiles/r03b_pmc_*): what it waits for is memory, so the global round trips on a target's critical path went from
//    about two dozen to about eight -- the unit's record (target, length, first word) comes with ONE scalar load issued
//    two units ahead, its packed words one unit ahead (they are parked in LDS: scan and diagonal test read them there),
//    a table entry and the packed oligo it belongs to are fetched with the same index (one round trip per dense pass,
//    two passes in flight), the per-slot and per-assay constants sit in LDS;
//  * phase-A items go STRAIGHT into the queue of the length bucket that binds them (blocks of 64 slots handed out per
//    bucket, unused slots hold QUEUE_HOLE): the bucket pass over the phase-A queue (two passes over 52 M entries on
//    config 2) is gone and the pre-filter reads the windows of one target back to back.
// Round 6 (the stage is a chain of dependent steps per target that four wavefronts per SIMD do not hide: shorter chains):
//  * wave prefix sums and neighbour reads in the VALU's DPP forms (row_shr / row_bcast, wave_shr) instead of LDS crossbar
//    shuffles; the first cull's walk to the list successor only for the seeds that turned out valid (one round per target
//    instead of one per 64 seeds), four keys per trip; the counting sort four rounds at a time;
//  * the dense pass: one 16-byte record per candidate (table entry + the first 32 bases of its oligo), six candidates per
//    lane in flight, the common case as straight-line code (selects), the rare cases behind one wave-uniform branch;
//  * instances for databases of short targets (SHORT: every unit in LDS and within the cull's reach, the per-unit tests
//    and the general loops compiled out) and for assay sets of many chunks (BIG: lists of 896 seeds, three workgroups per CU).
// Config 2: 5.29 -> 4.6 ms per step; config 4 (54 chunk launches per batch -> 28): 472 -> 369 ms per shard.
#ifndef TNT_SEED2_HIP_H
#define TNT_SEED2_HIP_H

#include <hip/hip_runtime.h>
#include "tnt_dev.h"
#include "tnt_seed.hip.h"

namespace tnt {

// wavefronts (= targets in flight) per workgroup.  LTAB (round 4, opt-in: TNT_MI355X_SEED2_LDS_TABLE): the chunk's bucket offsets
// sit in LDS (32 KB at k = 7, staged once per launch) and ONE 1024-thread workgroup per CU shares them -- the scan's sixteen
// gathers per lane and sweep become LDS reads; the same 16 wavefronts per CU fit next to the table because two per-wavefront
// arrays share their space with others that are dead while they are in use (the target's words with the extreme keys, the
// emission's scratch words with the candidate list).  The scan on its own runs 3.6 x faster that way (0.28 ms per pass over
// config 2, tnt_seed_scan_only) -- the whole kernel runs SLOWER (config 2: 5.9 against 5.3 ms; config 4, sixteen chunk launches
// per batch: 50.7 against 37.2 ms per 100 k targets): the table reads were hitting L1 anyway, and a 1024-thread workgroup only
// starts when a whole CU has drained.  Kept as an instance, not used by default.
#ifndef TNT_S2_WAVES
#define TNT_S2_WAVES 4
#endif
constexpr int s2_waves(bool ltab) { return ltab ? 16 : TNT_S2_WAVES; }
constexpr int S2_TAB_WORDS = (1 << 14) + 2; // bucket offsets of a chunk for hash words of up to 7 bases
#ifndef TNT_S2_CAP
#define TNT_S2_CAP 512
#endif
constexpr int S2_CAP = TNT_S2_CAP;     // seeds of one target x one assay chunk held in LDS
constexpr int S2_CAP_LTAB = 512;       // ... by the instance that also keeps the chunk's bucket offsets there (sixteen wavefronts per workgroup)
// ... by the instances for assay sets that need many chunks (BIG): every chunk launch scans the database again, so a context
// whose assays fill six chunks or more at 512 seeds takes lists of 896 -- three workgroups per CU instead of four, but four
// chunks in ten fewer (config 4, 54 chunks -> 28: seed stage -16 % and more; config 5, ten chunks -> eight at +10 %, so it keeps 512)
constexpr int S2_CAP_BIG = 896;
constexpr int s2_cap(bool ltab, bool big = false) { return ltab ? S2_CAP_LTAB : (big ? S2_CAP_BIG : S2_CAP); }
constexpr int S2_OCC_BIG = 3;          // workgroups per CU of the BIG instances (LDS: 53 KB per workgroup)
constexpr int S2_ASSAYS = 128;         // assays per chunk (7-bit local index in the key)
#ifndef TNT_S2_SWEEP
#define TNT_S2_SWEEP 16
#endif
constexpr int S2_SWEEP = TNT_S2_SWEEP; // positions per lane and sweep (16 or 8)
constexpr int S2_TGT_WORDS = 128;       // packed words of a target parked in LDS (2 048 bases); longer targets are read from HBM
constexpr uint32_t QUEUE_BLOCK = 64;   // queue slots handed out at a time (one filter / bind wavefront reads one block)
constexpr uint32_t QUEUE_HOLE = 0xFFFFFFFFu; // a queue slot without an item
#ifndef TNT_S2_FLIGHT
#define TNT_S2_FLIGHT 6
#endif
constexpr int S2_FLIGHT = TNT_S2_FLIGHT; // candidates per lane whose table records are in flight together in the dense pass
constexpr int S2_BINS = 448;              // (oligo, threshold class) groups a workgroup counts in LDS (what is left of a quarter of a CU's 160 KB next to the other arrays)

struct Seed2Args {
	DbView db;
	const uint4 *unit_rec;       // work units = whole targets: {target, bases, first packed word lo, hi}; readable two records past unit1
	uint32_t unit0, unit1;
	int word_size;
	const uint16_t *wt_off;      // this chunk's bucket offsets [4^k + 1], relative to wt_ent
	const uint32_t *wt_ent;      // this chunk's entries (tnt_dev.h)
	const uint16_t *wlist;       // concatenated word lists
	const uint32_t *wl_off;      // [((assay * 3 + role) * 2 + minus)] -> first word
	const uint4 *wt_seq;         // [entry] the strand-specific oligo of the entry's list, 2 bits per base, base 0 in the low bits
	const uint4 *wt_rec;         // [entry] {the entry, the first 32 bases of that oligo, 0}: what the dense pass reads per candidate, as one load
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
	uint4 *queue;                // phase-A queues, one region per length bucket: self-contained entries (FilterArgs2::fat)
	int32_t *result;             // [item] set to "not bound" here: the pre-filter writes nothing for the windows it rejects
	const uint32_t *q_base;      // [bucket] first slot of the bucket's region
	const uint32_t *q_cap;       // [bucket] slots of the region (a multiple of QUEUE_BLOCK)
	uint32_t *q_count;           // [bucket * BUCKET_STRIDE] slots handed out (device counter)
	uint32_t *q_state;           // [region][MAX_BUCKETS][2]: the block every wavefront is filling (next slot, end), kept across the chunk launches
	int last_chunk;              // the last launch of the batch marks the unused slots of the open blocks
	int n_buckets;
	uint32_t *stat_items, *stat_groups, *stat_phase_a; // statistics (one add per wavefront and launch)
	unsigned long long *stat_cells; // ... DP cells (rows x window columns) of the phase-A windows
	// Round 5: the pre-filter reads its queues regrouped by (oligo, threshold class) (tnt_ufilter.hip.h); the entries per group
	// are counted here, as the entries are written -- an LDS histogram per workgroup, flushed once per launch -- which spares the
	// regrouping its counting pass over the queues.  Null: not counted (more groups than S2_BINS: the counting pass runs).
	const uint32_t *slot_bin;    // [slot][2] group of the slot inside its plain / its general bucket
	const uint32_t *bin_base;    // [bucket] first group of the bucket in bin_hist
	uint32_t *bin_hist;          // [groups of all buckets] entries per group (zeroed per batch)
	uint32_t n_bins;
	uint32_t *big_count;         // groups of more than 64 items
	uint32_t *err;
	uint32_t *debug;             // experimental builds only (TNT_EXP_STATS): cycles per phase
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

// Is (wi, p) the first seed of its diagonal?  Packed form for plain A/C/G/T oligos whose compared stretch fits 32 bases:
// XOR the target and the oligo along the diagonal and look for a run of `ws` matching bases that starts before the seed
// (bind_oligo.cpp:98-99: stable sort by diagonal + unique keeps the smallest word index).  W0..W2 = the three packed
// target words from word (p - kmax) >> 4 on; (o_lo, o_hi) = the oligo from base wi - kmax on (base 0 of the oligo
// whenever the seed lies at least wi bases into the target, i.e. nearly always).  All in 32-bit halves: v_alignbit
// funnels.  WS: the word size when the instance is compiled for one (7), 0 = the run-time value `ws`.
template <int WS> __device__ __forceinline__ bool s2_first_packed(uint32_t W0, uint32_t W1, uint32_t W2, uint32_t o_lo, uint32_t o_hi, uint32_t p, uint32_t kmax, int ws)
{
	const uint32_t sh = ((p - kmax) & 15u) * 2u;
	const uint32_t t_lo = __builtin_amdgcn_alignbit(W1, W0, sh), t_hi = __builtin_amdgcn_alignbit(W2, W1, sh);
	const uint32_t x_lo = t_lo ^ o_lo, x_hi = t_hi ^ o_hi;
	// one bit per matching base (even bit positions), 32 bases in two words
	const uint32_t z_lo = ~(x_lo | (x_lo >> 1)) & 0x55555555u, z_hi = ~(x_hi | (x_hi >> 1)) & 0x55555555u;
	// runs of `ws` matching bases: r = z & z>>2 & ... & z>>2(ws-1) over the 64-bit value (hi:lo)
	uint32_t r_lo, r_hi;
	if (WS == 7) { // runs of 2, of 4, then 4 + 2 + 1
		const uint32_t a_lo = z_lo & __builtin_amdgcn_alignbit(z_hi, z_lo, 2), a_hi = z_hi & (z_hi >> 2);
		const uint32_t b_lo = a_lo & __builtin_amdgcn_alignbit(a_hi, a_lo, 4), b_hi = a_hi & (a_hi >> 4);
		r_lo = b_lo & __builtin_amdgcn_alignbit(a_hi, a_lo, 8) & __builtin_amdgcn_alignbit(z_hi, z_lo, 12);
		r_hi = b_hi & (a_hi >> 8) & (z_hi >> 12);
	} else {
		r_lo = z_lo, r_hi = z_hi;
		for (int i = 1; i < ws; i++) {
			r_lo &= __builtin_amdgcn_alignbit(z_hi, z_lo, 2 * i);
			r_hi &= z_hi >> (2 * i);
		}
	}
	// ... that start before the seed: bases [0, kmax)
	const uint32_t b_lo = kmax >= 16u ? ~0u : ((1u << (2u * kmax)) - 1u);
	const uint32_t b_hi = kmax <= 16u ? 0u : (kmax >= 32u ? ~0u : ((1u << (2u * kmax - 32u)) - 1u));
	return ((r_lo & b_lo) | (r_hi & b_hi)) == 0u;
}

#ifndef TNT_EXP_S2_SKIP
#define TNT_EXP_S2_SKIP 0 // timing experiments (wrong results): 1 no item stores, 2 no queue stores, 4 no group stores, 8 no emission, 16 no cull loop, 32 nothing behind the dense pass, 64 no dense pass, 128 candidates counted, not listed
#endif
#ifndef TNT_S2_OCC
#define TNT_S2_OCC 4 // workgroups per CU the kernel is compiled for (5: registers capped at 96, the aliased LDS arrays of LTAB)
#endif
// SHORT: the host vouches that every unit of the launch fits the LDS copy of a target (S2_TGT_WORDS packed words) and lies within
// the first cull's reach as a whole (bases - k <= max_len + 50): the tests per unit, the reads of target words from HBM and the
// cull's general loop are compiled out (1-kb and 2-kb targets against the default -l 2000: every BASELINE configuration).
template <int WS, bool LTAB, bool SHORT, bool BIG = false> __global__ __launch_bounds__(64 * s2_waves(LTAB), LTAB ? 1 : (BIG ? S2_OCC_BIG : TNT_S2_OCC)) void seed2_kernel(Seed2Args a)
{
	constexpr int S2_WAVES = s2_waves(LTAB), S2_THREADS = 64 * S2_WAVES;
	constexpr int S2_CAP = s2_cap(LTAB, BIG); // (the instance's list capacity: shadows the namespace constant)
	static_assert(!(LTAB && BIG), "the large lists do not fit beside the bucket offsets");
	constexpr bool ALIAS = LTAB || TNT_S2_OCC > 4; // the target's words inside s_ext, the emission's scratch words inside s_srt
	static_assert(S2_ASSAYS * 4 >= S2_TGT_WORDS + 4 && S2_CAP >= 3 * MAX_BUCKETS, "the shared arrays hold their lodgers");
	__shared__ uint32_t s_raw[S2_WAVES][S2_CAP];      // kept seeds (keys) as the dense pass finds them
	__shared__ uint32_t s_srt[S2_WAVES][S2_CAP];      // candidates of the scan, then the keys grouped by assay (LTAB: the emission's scratch words live here too)
	__shared__ uint32_t s_tgt[S2_WAVES][ALIAS ? 1 : S2_TGT_WORDS + 4]; // the packed words of the target (targets of up to 2 048 bases); LTAB: inside s_ext
	__shared__ uint16_t s_hist[S2_WAVES][S2_ASSAYS];  // seeds per assay
	__shared__ uint16_t s_off[S2_WAVES][S2_ASSAYS];   // scatter cursors, then the END of every assay's segment
	__shared__ __attribute__((aligned(16))) uint32_t s_ext[S2_WAVES][S2_ASSAYS * 4]; // per assay: smallest key of F-, of R-, largest key of F+, of R+
	__shared__ uint32_t s_agg[S2_WAVES][S2_ASSAYS];   // per assay: successors on the minus strand | on the plus strand << 10 | valid seeds << 20
	__shared__ uint32_t s_qh[S2_WAVES][MAX_BUCKETS], s_qcur[S2_WAVES][MAX_BUCKETS], s_qend[S2_WAVES][MAX_BUCKETS];
	__shared__ uint32_t s_qtmp[S2_WAVES][ALIAS ? 1 : 3 * MAX_BUCKETS]; // per emission round: first slot, end of the open block, the new block (per bucket)
	__shared__ uint16_t s_tab[LTAB ? S2_TAB_WORDS : 2]; // LTAB: the chunk's bucket offsets [4^k + 1]
	__shared__ uint8_t s_af[S2_ASSAYS];               // per assay of the chunk: 1 probe / padlock assay (hash-overload lists), 2 PCR with a probe
	__shared__ uint32_t s_slot[S2_ASSAYS * 3];        // slot_pack of the chunk's slots
	__shared__ uint32_t s_qbase[MAX_BUCKETS], s_qcap[MAX_BUCKETS];
	__shared__ uint32_t s_bhist[S2_BINS];             // entries per (oligo, class) group written by this workgroup (Seed2Args::bin_hist)
	__shared__ uint32_t s_bin[S2_ASSAYS * 3];         // per slot of the chunk: group in bin_hist of its plain bucket | of its general bucket << 16
	// pointers the emission needs: read from LDS where they are used (kept in registers they spill to scratch memory,
	// and a scratch reload is a vector-memory load -- see the emission)
	enum { COLD_ITEMS = 0, COLD_GROUPS, COLD_QUEUE, COLD_Q_COUNT, COLD_BIG, COLD_ERR, COLD_RESULT, COLD_N };
	__shared__ unsigned long long s_cold[COLD_N];
	typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64; // (volatile: not hoisted out of the loops, not kept in registers)
	auto cold = [&](int k) __attribute__((always_inline)) { return *(lds_vu64 *)(&s_cold[k]); };

	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	// LTAB: the target's words are dead once the dense pass is through (the extreme keys are collected behind it), the candidate
	// list once the valid seeds sit in `raw` (the emission's scratch words are written and read inside one round)
	uint32_t *raw = s_raw[wv], *srt = s_srt[wv], *tgt = ALIAS ? s_ext[wv] : s_tgt[wv];
	uint32_t *qpos = ALIAS ? s_srt[wv] : s_qtmp[wv], *qsplit = qpos + MAX_BUCKETS, *qnew = qpos + 2 * MAX_BUCKETS;
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
	for (uint32_t i = threadIdx.x; i < n_chunk * 3u; i += S2_THREADS) s_slot[i] = a.slot_pack[a.assay_begin * 3u + i];
	if (int(threadIdx.x) < a.n_buckets) s_qbase[threadIdx.x] = a.q_base[threadIdx.x], s_qcap[threadIdx.x] = a.q_cap[threadIdx.x];
	if (a.bin_hist) {
		for (uint32_t i = threadIdx.x; i < uint32_t(S2_BINS); i += S2_THREADS) s_bhist[i] = 0;
		for (uint32_t i = threadIdx.x; i < n_chunk * 3u; i += S2_THREADS) {
			const uint32_t sp = a.slot_pack[a.assay_begin * 3u + i], sl = a.assay_begin * 3u + i;
			s_bin[i] = (a.bin_base[sp & 0xFFu] + a.slot_bin[2u * sl]) | ((a.bin_base[(sp >> 8) & 0xFFu] + a.slot_bin[2u * sl + 1u]) << 16);
		}
	}
	if (LTAB) // the chunk's bucket offsets, as dwords (the table starts on a dword, tnt_host.hip plan())
		for (uint32_t i = threadIdx.x; i < ((1u << (2 * a.word_size)) + 2u) / 2u; i += S2_THREADS)
			reinterpret_cast<uint32_t *>(s_tab)[i] = reinterpret_cast<const uint32_t *>(a.wt_off)[i];
	if (threadIdx.x == 0) {
		s_cold[COLD_ITEMS] = (unsigned long long)a.items, s_cold[COLD_GROUPS] = (unsigned long long)a.groups, s_cold[COLD_QUEUE] = (unsigned long long)a.queue;
		s_cold[COLD_Q_COUNT] = (unsigned long long)a.q_count, s_cold[COLD_BIG] = (unsigned long long)a.big_count, s_cold[COLD_ERR] = (unsigned long long)a.err;
		s_cold[COLD_RESULT] = (unsigned long long)a.result;
	}
	// the wavefront's private output region (items, groups) and the queue blocks it is filling; both persist across the
	// chunk launches of a batch
	const uint32_t region = uint32_t(__builtin_amdgcn_readfirstlane(int(blockIdx.x * S2_WAVES + wv)));
	if (lane < MAX_BUCKETS) {
		s_qh[wv][lane] = 0;
		s_qcur[wv][lane] = a.q_state[(region * MAX_BUCKETS + lane) * 2];
		s_qend[wv][lane] = a.q_state[(region * MAX_BUCKETS + lane) * 2 + 1];
	}
	if (lane < 4) tgt[S2_TGT_WORDS + lane] = 0; // (slack behind the last word: the scan and the diagonal test read ahead)
	__syncthreads();

	const uint32_t item_lo = region * a.item_region, group_lo = region * a.group_region;
	uint32_t item_cur = item_lo + a.region_counts[2 * region], group_cur = group_lo + a.region_counts[2 * region + 1];
	const uint32_t item_first = item_cur, group_first = group_cur;
	uint32_t n_phase_a = 0;
	uint32_t my_cells = 0; // DP cells of the windows this lane queued (statistics)

	// Software pipeline over the wavefront's units: the record of unit i + 2 (one scalar load) and the packed words of
	// unit i + 1 (two dwords per lane) are in flight while unit i is processed.
	const uint32_t stride = gridDim.x * S2_WAVES;
	auto load_rec = [&](uint32_t u) __attribute__((always_inline)) { return a.unit_rec[u < a.unit1 ? u : a.unit1]; }; // (a zero record behind the last unit)
	auto load_words = [&](const uint4 rec, uint32_t &w0, uint32_t &w1) __attribute__((always_inline)) {
		const uint32_t nw = (rec.y + 15u) >> 4; // words of the target
		const uint32_t *pk = a.db.packed + (uint64_t(rec.z) | (uint64_t(rec.w) << 32));
		w0 = uint32_t(lane) < nw && (SHORT || nw <= uint32_t(S2_TGT_WORDS)) ? pk[lane] : 0u;
		w1 = uint32_t(lane) + 64u < nw && (SHORT || nw <= uint32_t(S2_TGT_WORDS)) ? pk[lane + 64] : 0u;
	};
#ifdef TNT_EXP_STATS
	long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#define S2_STAMP(K)                                               \
	do {                                                          \
		const long long now_ = __builtin_readcyclecounter();      \
		st_acc[K] += now_ - st_last;                              \
		st_last = now_;                                           \
	} while (0)
#else
#define S2_STAMP(K) do { } while (0)
#endif
	uint32_t unit = a.unit0 + region;
	uint4 rec_cur = load_rec(unit), rec_next = load_rec(unit + stride);
	uint32_t wc0, wc1;
	load_words(rec_cur, wc0, wc1);

	for (; unit < a.unit1; unit += stride) {
		const uint32_t t = rec_cur.x, len = rec_cur.y;
		const uint32_t *pk = a.db.packed + (uint64_t(rec_cur.z) | (uint64_t(rec_cur.w) << 32));
		const bool in_lds = SHORT ? true : ((len + 15u) >> 4) <= uint32_t(S2_TGT_WORDS);
		const uint32_t rec_words = rec_cur.z; // first packed word of the target (the host checks that the packed array stays below 2^32 words)
		// this unit's words -> LDS; the next unit's words and the record behind it -> in flight
		tgt[lane] = wc0, tgt[lane + 64] = wc1;
		if (ALIAS && lane < 4) tgt[S2_TGT_WORDS + lane] = 0; // (the slack words were the last unit's extreme keys)
		rec_cur = rec_next;
		load_words(rec_next, wc0, wc1);
		rec_next = load_rec(unit + 2 * stride);
		if (len < uint32_t(a.word_size)) continue; // tntblast_local.cpp:513
		const uint32_t npos = len - a.word_size + 1;
		// the first cull's reach covers the whole target: "a partner within max_len + 50" is "a partner" (see the head)
		const bool in_reach = SHORT ? true : npos - 1u <= threshold;
		for (uint32_t i = lane; i < (n_chunk + 1u) / 2u; i += 64) reinterpret_cast<uint32_t *>(hist)[i] = 0;
		uint32_t nkept = 0; // kept seeds so far (wave-uniform)
		wave_sync();
		S2_STAMP(0);
		// (address spaces spelled out: merged into one generic pointer the two reads become FLAT loads, which count against
		// both the vector-memory and the LDS counter)
		typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
		typedef __attribute__((address_space(1))) const uint32_t glb_cu32;
		auto tw = [&](uint32_t w) __attribute__((always_inline)) {
			uint32_t v;
			if (in_lds) v = *((lds_cu32 *)tgt + w);
			else v = *((glb_cu32 *)pk + w);
			return v;
		};

		// ---------------------------------------------------------------- scan + diagonal de-duplication
		uint32_t *cand = srt; // position | entry index << 15
		// One candidate: its table entry and the first 32 bases of the oligo the entry belongs to come as one 16-byte record
		// (Seed2Args::wt_rec), S2_FLIGHT candidates per lane in flight; the target words come from LDS.
		struct Cand {
			uint32_t cv;  // position | entry index << 15
			uint32_t ent;
			uint2 seq;    // the first 32 bases of the entry's oligo
		};
		auto fetch = [&](uint32_t c, uint32_t ncand) __attribute__((always_inline)) {
			Cand x;
			x.cv = 0, x.ent = 0, x.seq = make_uint2(0, 0);
			if (c < ncand) {
				x.cv = cand[c];
				const uint4 r = a.wt_rec[x.cv >> 15];
				x.ent = r.x, x.seq = make_uint2(r.y, r.z);
			}
			return x;
		};
		// The common case -- a plain A/C/G/T oligo whose compared stretch lies in its first 32 bases -- is straight-line code for
		// every lane (selects, no branches: the candidates of a trip interleave in the instruction stream); the others (oligos
		// with inosine or IUPAC codes, more than 25 bases before the seed) take a branch the wavefront seldom meets.
		auto settle = [&](uint32_t c, uint32_t ncand, const Cand &x) __attribute__((always_inline)) {
			const bool live = c < ncand;
			const uint32_t wi = x.ent & 127u, cp = x.cv & 0x7FFFu;
			const uint32_t assay = x.ent >> WT_ASSAY_SHIFT;
			const uint32_t kmax = wi < cp ? wi : cp;
			// (the compared stretch -- kmax bases and the word -- fits 32 bases, and starts at base 0 of the oligo or lies in its first 32)
			const bool packed = (x.ent & WT_PLAIN) && kmax + uint32_t(a.word_size) <= 32u && (kmax == wi || wi + uint32_t(a.word_size) <= 32u);
			bool first;
			{
				// every compared base lies in the first 32 of the oligo (x.seq): the oligo from base wi - kmax on is a shift of
				// those two words -- by nothing unless the seed sits less than wi bases into the target.  kmax == 0 (the first
				// word of the oligo, or the first position of the target): nothing lies before the seed, the test says "first".
				const uint32_t w0 = (cp - kmax) >> 4;
				const uint32_t s2 = (wi - kmax) * 2u;
				const uint32_t a_lo = __builtin_amdgcn_alignbit(x.seq.y, x.seq.x, s2 & 31u), a_hi = x.seq.y >> (s2 & 31u);
				const uint32_t o_lo = s2 < 32u ? a_lo : a_hi, o_hi = s2 < 32u ? a_hi : 0u;
				first = s2_first_packed<WS>(tw(w0), tw(w0 + 1), tw(w0 + 2), o_lo, o_hi, cp, kmax, a.word_size);
			}
			const bool other = live && !packed && kmax != 0u;
			if (__builtin_amdgcn_ballot_w64(other) != 0ull) {
				if (other) {
					if ((x.ent & WT_PLAIN) && kmax + uint32_t(a.word_size) <= 32u) {
						// a long oligo's seed less than wi bases into the target: the oligo from base wi - kmax on
						const uint4 q = a.wt_seq[x.cv >> 15];
						const uint32_t w0 = (cp - kmax) >> 4;
						const uint32_t s2 = (wi - kmax) * 2u, word = s2 >> 5, sh2 = s2 & 31u;
						const uint32_t oa = word == 0 ? q.x : (word == 1 ? q.y : (word == 2 ? q.z : q.w));
						const uint32_t ob = word == 0 ? q.y : (word == 1 ? q.z : (word == 2 ? q.w : 0u));
						const uint32_t oc = word == 0 ? q.z : (word == 1 ? q.w : 0u);
						first = s2_first_packed<WS>(tw(w0), tw(w0 + 1), tw(w0 + 2), __builtin_amdgcn_alignbit(ob, oa, sh2), __builtin_amdgcn_alignbit(oc, ob, sh2), cp, kmax, a.word_size);
					} else {
						const uint32_t li = (assay * 3u + ((x.ent >> 8) & 3u)) * 2u + ((x.ent & WT_MINUS) ? 1u : 0u);
						first = first_on_diagonal_words(pk, a.wlist + a.wl_off[li], wi, cp, wmask);
					}
				}
			}
			first = first && live;
			const uint32_t la = assay - a.assay_begin;
			const uint32_t key = s2_key(la, x.ent, cp);
			const unsigned long long m = __builtin_amdgcn_ballot_w64(first);
			if (first) {
				const uint32_t slot = nkept + s2_rank(m);
				if (slot < uint32_t(S2_CAP)) {
					raw[slot] = key;
					atomicAdd(reinterpret_cast<unsigned int *>(hist) + (la >> 1), (la & 1u) ? 0x10000u : 1u);
				}
			}
			nkept += uint32_t(__builtin_popcountll(m));
		};
		// Sweeps of 64 x 16 positions: a lane takes the two packed words that hold its positions (and the k - 1 bases behind
		// them), extracts the k-mers with shifts and reads their bucket bounds -- both bounds with one 2-byte aligned dword
		// gather, all 16 gathers in flight at once; one wave-wide prefix sum gives every lane its slice of the candidate
		// list.  The list is drained (dense pass below) when the next sweep does not fit, and at the end; a sweep that does
		// not fit an empty list (a burst of words shared by many expanded assays) is scanned position by position, the list
		// drained as often as it fills.
		uint32_t base = 0;          // first position of the next sweep
		uint32_t slow_p = ~0u, slow_end = 0, slow_done = 0; // position-by-position scan of [.., slow_end): this lane's position, entries of it already taken
		bool slow = false;          // wave-uniform
		for (;;) {
			uint32_t ncand = 0;
			while (!slow && base < npos) {
				const uint32_t p0 = base + uint32_t(lane) * S2_SWEEP;
				unsigned long long v = 0;
				if (p0 < npos) v = (unsigned long long)tw(p0 >> 4) | ((unsigned long long)tw((p0 >> 4) + 1) << 32);
				if (S2_SWEEP == 8) v >>= 16 * (lane & 1); // (two lanes share a packed word)
				uint32_t bnd[S2_SWEEP]; // first entry | count << 16
				uint32_t mine = 0;
#pragma unroll
				for (int q = 0; q < S2_SWEEP; q++) {
					const uint32_t w = uint32_t(v >> (2 * q)) & wmask;
					typedef __attribute__((address_space(3))) const u32_a2 lds_u32_a2; // (gfx950 reads a dword at a 2-byte boundary of LDS in one instruction)
					const uint32_t two = LTAB ? *((lds_u32_a2 *)(s_tab + w)) : *reinterpret_cast<const u32_a2 *>(a.wt_off + w);
					const uint32_t e0 = two & 0xFFFFu, e1 = two >> 16;
					const uint32_t n = (p0 + q < npos) ? e1 - e0 : 0u;
					bnd[q] = e0 | (n << 16);
					mine += n;
				}
				const uint32_t incl = wave_prefix_sum(mine);
				const uint32_t total = wave_last(incl);
				if (ncand + total > uint32_t(S2_CAP)) {
					if (ncand == 0) { // a burst: position by position
						slow = true;
						slow_p = base + uint32_t(lane), slow_done = 0;
						slow_end = base + 64 * S2_SWEEP < npos ? base + 64 * S2_SWEEP : npos;
						base += 64 * S2_SWEEP;
					}
					break; // drain, then take this sweep again (or scan it slowly)
				}
				uint32_t o = ncand + incl - mine;
				if (!(TNT_EXP_S2_SKIP & 128)) {
#pragma unroll
				for (int q = 0; q < S2_SWEEP; q++) {
					uint32_t e = bnd[q] & 0xFFFFu;
					for (uint32_t n = bnd[q] >> 16; n > 0; n--, e++) cand[o++] = (p0 + q) | (e << 15);
				}
				}
				ncand += total;
				base += 64 * S2_SWEEP;
			}
			if (slow) {
				while (true) {
					uint32_t e0 = 0, cnt = 0;
					if (slow_p < slow_end) {
						const uint32_t wd = slow_p >> 4, shp = (slow_p & 15u) * 2u;
						const uint32_t w = uint32_t(((unsigned long long)tw(wd) | ((unsigned long long)tw(wd + 1) << 32)) >> shp) & wmask;
						e0 = uint32_t(LTAB ? s_tab[w] : a.wt_off[w]) + slow_done;
						const uint32_t e1 = LTAB ? s_tab[w + 1] : a.wt_off[w + 1];
						cnt = e1 > e0 ? e1 - e0 : 0u;
					}
					const uint32_t incl = wave_prefix_sum(cnt); // inclusive prefix of cnt over the lanes
					const uint32_t at = ncand + incl - cnt;
					const uint32_t fit = at < uint32_t(S2_CAP) ? (cnt < uint32_t(S2_CAP) - at ? cnt : uint32_t(S2_CAP) - at) : 0u;
					for (uint32_t x = 0; x < fit; x++) cand[at + x] = slow_p | ((e0 + x) << 15);
					bool over = false;
					if (slow_p < slow_end) {
						if (fit == cnt) slow_p += 64, slow_done = 0;
						else slow_done += fit, over = true;
					}
					const uint32_t total = wave_last(incl);
					ncand = ncand + total < uint32_t(S2_CAP) ? ncand + total : uint32_t(S2_CAP);
					if (__builtin_amdgcn_ballot_w64(over) != 0) break;
					if (__builtin_amdgcn_ballot_w64(slow_p < slow_end) == 0) {
						slow = false;
						break;
					}
				}
			}
			// dense pass over the candidate list, S2_FLIGHT rounds of 64 in flight: the first seed of every (oligo, strand,
			// diagonal) survives
			wave_sync(); // the list was written by other lanes
			S2_STAMP(1);
			for (uint32_t c0 = 0; c0 < ((TNT_EXP_S2_SKIP & 64) ? 0u : ncand); c0 += 64 * S2_FLIGHT) {
				Cand x[S2_FLIGHT];
#pragma unroll
				for (int f = 0; f < S2_FLIGHT; f++) x[f] = fetch(c0 + 64u * uint32_t(f) + uint32_t(lane), ncand);
#pragma unroll
				for (int f = 0; f < S2_FLIGHT; f++)
					if (f == 0 || c0 + 64u * uint32_t(f) < ncand) settle(c0 + 64u * uint32_t(f) + uint32_t(lane), ncand, x[f]);
			}
			wave_sync();
			S2_STAMP(2);
			if (!slow && base >= npos) break;
		}
		uint32_t n = nkept;
		if (n > uint32_t(S2_CAP)) {
			if (lane == 0) atomicOr(a.err, ERR_SEED_OVERFLOW);
			n = S2_CAP;
		}
		if (n == 0 || (TNT_EXP_S2_SKIP & 32)) continue;

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
			uint32_t run = wave_prefix_sum(sum) - sum;
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
		// (four rounds of 64 keys at a time: their reads, then their returning adds, are in flight together -- a round on its own
		// is a chain of two LDS latencies)
		for (uint32_t i0 = 0; i0 < n; i0 += 256) {
			uint32_t k4[4], old4[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t i = i0 + 64u * uint32_t(u) + uint32_t(lane);
				k4[u] = i < n ? raw[i] : ~0u; // (no key has all bits set: group 7 does not exist)
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t la = k4[u] >> 25;
				old4[u] = 0;
				if (k4[u] != ~0u) old4[u] = atomicAdd(reinterpret_cast<unsigned int *>(off) + (la >> 1), (la & 1u) ? 0x10000u : 1u);
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t k = k4[u], la = k >> 25, g = (k >> 7) & 7u;
				if (k != ~0u) {
					srt[(la & 1u) ? (old4[u] >> 16) : (old4[u] & 0xFFFFu)] = k;
					if (in_reach && g < 4u) { // primers: the extreme key of every (strand, role)
						if (g < 2u) atomicMin(ext + 4 * la + g, k);
						else atomicMax(ext + 4 * la + g, k);
					}
				}
			}
		}
		wave_sync();

		S2_STAMP(3);
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
		uint32_t n_valid_all = 0; // valid seeds so far (wave-uniform)
		for (uint32_t i0 = 0; i0 < ((TNT_EXP_S2_SKIP & 16) ? 0u : n); i0 += 64) {
			const uint32_t i = i0 + uint32_t(lane);
			bool valid = false;
			uint32_t kk = 0;
			if (i < n) {
				const uint32_t k = srt[i];
				kk = k;
				const uint32_t la = k >> 25, g = (k >> 7) & 7u;
				const uint32_t af = s_af[la];
				if (af & 1u) {
					valid = true;
				} else {
					const uint32_t cnt = hist[la], s0 = uint32_t(off[la]) - cnt;
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
						// Validity comes from the extreme keys; only a primer of an assay with a probe has to walk its segment (the nearest
						// probe on the partner's side).  The list successor, which only a VALID seed needs, is found behind the
						// compaction below -- one round over the valid seeds instead of one walk per seed of the target.
						if ((af & 2u) && !probe_me) {
							// (four keys per trip, their reads in flight together: a trip costs one LDS latency; the last key repeats
							// where the segment ends -- a minimum does not mind)
							for (uint32_t x = 0; x < cnt; x += 4) {
								uint32_t o4[4];
#pragma unroll
								for (int u = 0; u < 4; u++) o4[u] = srt[s0 + min(x + uint32_t(u), cnt - 1u)];
#pragma unroll
								for (int u = 0; u < 4; u++) {
									const uint32_t dir = ((o4[u] - k) ^ sm) - sm; // > 0 as a signed number: on the partner's side
									const bool op = ((o4[u] >> 7) & 7u) >= 4u && int32_t(dir) > 0;
									pn_d = min(pn_d, op ? dir : ~0u);
								}
							}
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
					if (valid && !in_reach) {
						// one LDS add per valid seed: its assay's count, and the strand of its successor
						uint32_t add = 1u << 20;
						if (int32_t(next_d) >= 0) add += // (keys below this one leave huge values behind, not ~0: a successor is less than 2^25 away)
							 ((0x13u >> (((k + next_d + 1u) >> 7) & 7u)) & 1u) ? 1u : (1u << 10);
						atomicAdd(agg + la, add);
					}
				}
				if (valid && (af & 1u) && !in_reach) atomicAdd(agg + la, 1u << 20);
			}
			// the valid seeds, still grouped by assay, move to `raw` (dead since the counting sort): the emission works on them only
			const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
			if (valid) raw[n_valid_all + s2_rank(vm)] = kk;
			n_valid_all += uint32_t(__builtin_popcountll(vm));
		}
		wave_sync();
		if (in_reach) {
			// The valid seeds' counts: one per assay, and the strand of every valid PCR seed's list successor -- the smallest key of
			// its assay's segment above its own (the minimum over all keys of (key - k - 1) as an unsigned number).
			for (uint32_t i0 = 0; i0 < n_valid_all; i0 += 64) {
				const uint32_t i = i0 + uint32_t(lane);
				if (i < n_valid_all) {
					const uint32_t k = raw[i];
					const uint32_t la = k >> 25;
					uint32_t add = 1u << 20;
					if (!(s_af[la] & 1u)) {
						const uint32_t cnt = hist[la], s0 = uint32_t(off[la]) - cnt;
						uint32_t next_d = ~0u;
						for (uint32_t x = 0; x < cnt; x += 4) {
							uint32_t o4[4];
#pragma unroll
							for (int u = 0; u < 4; u++) o4[u] = srt[s0 + min(x + uint32_t(u), cnt - 1u)];
							next_d = min(min(next_d, min(o4[0] - k - 1u, o4[1] - k - 1u)), min(o4[2] - k - 1u, o4[3] - k - 1u));
						}
						if (int32_t(next_d) >= 0) add += // (keys below this one leave huge values behind, not ~0: a successor is less than 2^25 away)
							 ((0x13u >> (((k + next_d + 1u) >> 7) & 7u)) & 1u) ? 1u : (1u << 10);
					}
					atomicAdd(agg + la, add);
				}
			}
			wave_sync();
		}

		S2_STAMP(4);
		// ---------------------------------------------------------------- emission: items, groups, phase-A queues
		// Phase B = what the reference binds only after both primers were tried on the first strand
		// (amplicon_search.cpp:131-353): the other strand's primers and the probe.
		// No vector-memory LOAD in here: a load's wait would also wait for the stores in front of it (loads, stores and
		// atomics retire in issue order), i.e. for the items and groups of the round before.  The pointers come from LDS.
		uint32_t exc0 = 0, exc1 = 0;
		if (a.has_exceptions) exc0 = a.db.exc_off[t], exc1 = a.db.exc_off[t + 1];
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		typedef __attribute__((address_space(1))) u32x4 glb_u32x4; // an Item / a Group as one 16-byte store
		typedef __attribute__((address_space(1))) uint32_t glb_u32;
		static_assert(sizeof(Item) == 16 && sizeof(Group) == 16, "items and groups are written as one dwordx4");
		glb_u32x4 *const g_items = (glb_u32x4 *)cold(COLD_ITEMS);
		glb_u32x4 *const g_groups = (glb_u32x4 *)cold(COLD_GROUPS);
		glb_u32x4 *const g_queue = (glb_u32x4 *)cold(COLD_QUEUE);
		glb_u32 *const g_result = (glb_u32 *)cold(COLD_RESULT);
		uint32_t carry_la = ~0u; // assay of the last seed of the round before
		for (uint32_t i0 = 0; i0 < ((TNT_EXP_S2_SKIP & 8) ? 0u : n_valid_all); i0 += 64) {
			const uint32_t i = i0 + uint32_t(lane);
			const bool valid = i < n_valid_all;
			const uint32_t n_valid = n_valid_all - i0 < 64u ? n_valid_all - i0 : 64u;
			const uint32_t k = valid ? raw[i] : 0u;
			const uint32_t la = k >> 25, g = (k >> 7) & 7u, wi = k & 127u, p = (k >> 10) & 0x7FFFu;
			const uint32_t ag = valid ? agg[la] : 0u;
			const bool minus_first = (ag & 1023u) < ((ag >> 10) & 1023u);
			const bool minus = (0x13u >> g) & 1u;
			const uint32_t role = g >= 4u ? 2u : (g & 1u);
			const bool nonpcr = valid && (s_af[la] & 1u);
			const bool pb = valid && !nonpcr && (role == 2u || minus != minus_first);
			// the first seed of every assay opens the group
			const uint32_t la_prev = uint32_t(__builtin_amdgcn_update_dpp(int(carry_la), int(la), 0x138, 0xF, 0xF, false)); // wave_shr:1; lane 0 keeps the carry
			const bool first = valid && la_prev != la;
			const unsigned long long fm = __builtin_amdgcn_ballot_w64(first);
			const uint32_t n_first = uint32_t(__builtin_popcountll(fm));
			if (item_cur + n_valid > item_lo + a.item_region || group_cur + n_first > group_lo + a.group_region) {
				if (lane == 0) __hip_atomic_fetch_or((glb_u32 *)cold(COLD_ERR), item_cur + n_valid > item_lo + a.item_region ? ERR_ITEM_OVERFLOW : ERR_GROUP_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				break;
			}
			const uint32_t o = item_cur + uint32_t(lane);
			const uint32_t gidx = group_cur + uint32_t(__builtin_popcountll(fm & ((2ull << lane) - 1ull))) - 1u;
			const uint32_t assay = a.assay_begin + la;
			if (first) {
				const uint32_t nv = ag >> 20;
#if !(TNT_EXP_S2_SKIP & 4)
				g_groups[gidx] = u32x4{unit, assay | (minus_first ? GROUP_MINUS_FIRST : 0u), o, nv}; // Group{target (unit), assay | flags, first, count}
#endif
				if (nv > 64u) __hip_atomic_fetch_add((glb_u32 *)cold(COLD_BIG), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			uint32_t bucket = ~0u;
			u32x4 fat = u32x4{0xFFFFFFFFu, 0u, 0u, 0u};
			if (valid) {
				const uint32_t slot = assay * 3u + role; // the Slot table is laid out [assay][role]
				// Item{target, tloc, info, group}
				const uint32_t info = wi | (minus ? ITEM_MINUS : 0u) | (pb ? ITEM_PHASE_B : 0u) | (slot << ITEM_SLOT_SHIFT);
#if !(TNT_EXP_S2_SKIP & 1)
				g_items[o] = u32x4{t, p, info, gidx};
				g_result[o] = 0xFFFFFFFFu; // -1: not bound (what the pre-filter's rejection means; a bind overwrites it)
#endif
				if (!pb) {
					const uint32_t sp = s_slot[la * 3u + role];
					bucket = sp & 0xFFu;
					// the window, bind_oligo.cpp:502-505
					int start = int(p) - (int(wi) + FLANK);
					if (start < 0) start = 0;
					uint32_t stop = uint32_t(start) + (sp >> 16) + 2u * FLANK;
					if (stop > len) stop = len;
					fat = u32x4{o, uint32_t(rec_words + uint32_t(start >> 4)), uint32_t(start & 15) | ((stop - uint32_t(start)) << 4) | ((sp >> 16) << 12), info};
					my_cells += (sp >> 16) * (stop - uint32_t(start)); // rows x columns of the window (statistics; below 2^32 per lane and launch)
					if (exc1 > exc0) { // windows with a degenerate target base go to the general instance of their length
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
						qpos[lane] = cur;
						qsplit[lane] = end;
						if (cur + h > end) { // the block is full: the rest of the round goes to a new one
							const uint32_t at = __hip_atomic_fetch_add((glb_u32 *)cold(COLD_Q_COUNT) + lane * BUCKET_STRIDE, QUEUE_BLOCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							if (at + QUEUE_BLOCK > s_qcap[lane]) {
								__hip_atomic_fetch_or((glb_u32 *)cold(COLD_ERR), ERR_ITEM_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
								qnew[lane] = ~0u;
							} else {
								const uint32_t nb = s_qbase[lane] + at;
								qnew[lane] = nb;
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
					uint32_t pos = qpos[bucket] + r;
					const uint32_t split = qsplit[bucket];
					if (pos >= split) {
						const uint32_t nb = qnew[bucket];
						pos = nb == ~0u ? ~0u : nb + (pos - split);
					}
#if !(TNT_EXP_S2_SKIP & 2)
					if (pos != ~0u) g_queue[pos] = fat;
#endif
					if (a.bin_hist && pos != ~0u) { // one LDS add per entry written: the regrouping's histogram, by the bucket the entry went to
						const uint32_t sb = s_bin[la * 3u + role];
						atomicAdd(&s_bhist[bucket != (s_slot[la * 3u + role] & 0xFFu) ? (sb >> 16) : (sb & 0xFFFFu)], 1u);
					}
				}
				n_phase_a += uint32_t(__builtin_popcountll(am));
			}
			item_cur += n_valid;
			group_cur += n_first;
			carry_la = uint32_t(__builtin_amdgcn_readlane(int(la), __builtin_amdgcn_readfirstlane(int(n_valid) - 1)));
		}
		wave_sync();
		S2_STAMP(5);
	}
#ifdef TNT_EXP_STATS
	if (lane == 0 && a.debug) { // setup, scan, dense pass, sort, cull, emission: kilo-cycles
		for (int q = 0; q < 6; q++) atomicAdd(a.debug + q, uint32_t(st_acc[q] >> 10));
		atomicAdd(a.debug + 7, 1u);
	}
#endif

	// what is left of the wavefront's last blocks holds no item
	wave_sync();
	if (lane < a.n_buckets) {
		const uint32_t cur = s_qcur[wv][lane], end = s_qend[wv][lane];
		if (a.last_chunk)
			for (uint32_t x = cur; x < end; x++) a.queue[x] = make_uint4(QUEUE_HOLE, 0, 0, 0);
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
	if (a.bin_hist) { // the workgroup's counts -> the batch's histogram
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < a.n_bins; i += S2_THREADS)
			if (s_bhist[i]) atomicAdd(a.bin_hist + i, s_bhist[i]);
	}
	wave_sync();
	if (a.stat_cells) { // DP cells of the windows this wavefront queued (statistics): one add per wavefront and launch
		unsigned long long cells = my_cells;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) cells += (unsigned long long)__shfl_xor((long long)cells, o);
		if (lane == 0 && cells) atomicAdd(a.stat_cells, cells);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// The scan on its own (a measurement launch, not part of a search): what SURVEY 8(d) calls the streaming half of the seed
// stage -- every k-mer of every target looked up in the inverted query-word table, the candidates counted -- with nothing
// behind it.  The table of one assay chunk sits in LDS as 16-bit bucket sizes (4^k entries: 32 KB at k = 7); one
// wavefront per target, a lane takes 16 consecutive positions (two packed dwords), the next target's words are in flight
// while this one's are looked up.  tnt_seed_scan_only() times it with events on the context's stream; bench.py reports
// it as roofline_seed.scan_only (source: measured), profiles/ holds the rocprofv3 summary of the same launch.
constexpr int SCAN_THREADS = 1024;
struct ScanOnlyArgs {
	DbView db;
	const uint4 *unit_rec; // whole-target work units (as for seed2_kernel)
	uint32_t unit0, unit1;
	int word_size;
	const uint16_t *wt_off; // bucket offsets [4^k + 1] of the chunk
	unsigned long long *n_candidates;
};

__global__ __launch_bounds__(SCAN_THREADS) void scan_only_kernel(ScanOnlyArgs a)
{
	extern __shared__ uint16_t s_cnt[]; // [4^k] candidates per word
	const uint32_t n_words = 1u << (2 * a.word_size), wmask = n_words - 1u;
	for (uint32_t w = threadIdx.x; w < n_words; w += SCAN_THREADS) s_cnt[w] = uint16_t(a.wt_off[w + 1] - a.wt_off[w]);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * SCAN_THREADS + threadIdx.x) >> 6, n_waves = (gridDim.x * SCAN_THREADS) >> 6;
	unsigned long long mine = 0;
	auto rec_of = [&](uint32_t u) __attribute__((always_inline)) { return a.unit_rec[u < a.unit1 ? u : a.unit1]; }; // (a zero record behind the last)
	uint32_t unit = a.unit0 + wave;
	uint4 rec = rec_of(unit);
	for (; unit < a.unit1; unit += n_waves) {
		const uint4 next = rec_of(unit + n_waves);
		const uint32_t len = rec.y;
		const uint32_t *pk = a.db.packed + (uint64_t(rec.z) | (uint64_t(rec.w) << 32));
		const uint32_t npos = len >= uint32_t(a.word_size) ? len - uint32_t(a.word_size) + 1u : 0u;
		for (uint32_t base = 0; base < npos; base += 64 * 16) {
			const uint32_t p0 = base + uint32_t(lane) * 16u;
			if (p0 < npos) {
				const unsigned long long v = (unsigned long long)pk[p0 >> 4] | ((unsigned long long)pk[(p0 >> 4) + 1] << 32); // (the rows end in slack words)
				uint32_t sum = 0;
#pragma unroll
				for (int q = 0; q < 16; q++) {
					const uint32_t c = s_cnt[uint32_t(v >> (2 * q)) & wmask];
					sum += (p0 + uint32_t(q) < npos) ? c : 0u;
				}
				mine += sum;
			}
		}
		rec = next;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) mine += (unsigned long long)__shfl_xor((long long)mine, o);
	if (lane == 0 && mine) atomicAdd(a.n_candidates, mine);
}

} // namespace tnt
#endif
