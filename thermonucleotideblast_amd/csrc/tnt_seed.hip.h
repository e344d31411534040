// tnt_seed.hip.h -- gfx950 seed-scan kernel: streams 2-bit packed targets, looks every k-mer up in
// the inverted query-word table, keeps the first seed of every (oligo, strand, diagonal), groups the
// seeds of one target per assay in LDS, runs the first cull and emits bind work items.
//
// Replaces DNAHash::hash + DNAHash::find/find_complement + iterator (seq_hash.h:524-779),
// match_oligo_to_{minus,plus}_strand (bind_oligo.cpp:84-122), the early exits of amplicon()
// (amplicon_search.cpp:96-126) and the first cull_oligo_match (amplicon_search.cpp:128,679-765).
//
// The reference indexes the TARGET and probes it with the query words; here the QUERY words are
// indexed once per context (word -> list of (assay, role, strand, word index)) and the target is
// streamed once, coalesced, 0.25 byte per base.  Semantics kept exactly:
//  * every target position carries a word: non-ACGT codes are hashed as their low two bits
//    (seq_hash.h:571-583); they are stored that way in the packed stream;
//  * `query_loc` is the index in the oligo's word list, not the base offset (Q2, seq_hash.h:214);
//  * per (oligo, strand) only the first seed of each diagonal (query_loc - target_loc) survives,
//    i.e. the one with the smallest word index (stable sort + unique, bind_oligo.cpp:98-99): a seed
//    (w, p) is dropped iff some k in 1..min(w,p) has word_list[w-k] == target_word(p-k).
//
// ONE WAVEFRONT PER TARGET, wave-synchronous (no workgroup barrier): the stage is a chain of short
// dependent steps per target, so throughput comes from the number of targets in flight (up to 18
// wavefronts per CU with 8.5 KB of LDS each), not from the width of one target's scan.
#ifndef TNT_SEED_HIP_H
#define TNT_SEED_HIP_H

#include <hip/hip_runtime.h>
#include "tnt_dev.h"

namespace tnt {

constexpr int SEED_WAVES = 4;         // wavefronts (= targets in flight) per workgroup of the instance that reads its tables from HBM / L2
constexpr int SEED_THREADS = 64 * SEED_WAVES;
// The streaming instance: the query-word table of the chunk (bucket offsets + entries + the packed oligos) sits in LDS once
// per workgroup and is shared by SEED_LDS_WAVES wavefronts; every lane walks SEED_SWEEP consecutive positions out of two registers.
constexpr int SEED_LDS_WAVES = 8;
constexpr int SEED_LDS_WORD_SIZE = 7;                 // hash words up to 7 bases: 4^7 + 1 16-bit offsets = 32 KB
constexpr int SEED_LDS_OFF = (1 << (2 * SEED_LDS_WORD_SIZE)) + 2;
constexpr int SEED_LDS_ENT = 10240;                   // entries of one assay chunk (40 KB)
constexpr int SEED_LDS_SEQ = 1536;                    // 64-bit words of packed oligos: 128 assays x 6 lists x 2 (12 KB); larger chunks read them from HBM
#ifndef TNT_SEED_SWEEP
#define TNT_SEED_SWEEP 16
#endif
#ifndef TNT_SEED_MINBLOCKS
#define TNT_SEED_MINBLOCKS 4
#endif
constexpr int SEED_SWEEP = TNT_SEED_SWEEP;               // positions per lane and sweep
constexpr int SEED_CAP = 512;         // seeds of one target x one assay chunk held in LDS
constexpr int SEED_MAX_ASSAYS = 256;  // assays per chunk (8-bit local index in the key)
constexpr uint32_t ITEM_NO_REP = 0xFFFFFFFFu;
constexpr int LIST_MAX = 512;         // seeds of one (work unit, assay): bounded by the seed list itself (PAIR_BIG of the pair kernel)

typedef uint32_t __attribute__((aligned(2))) u32_a2; // a dword at a 2-byte aligned address (two neighbouring 16-bit bucket offsets)

struct SeedOut { // debug / unit-level output: every kept seed before the cull
	uint32_t target, assay_role, strand_minus, query_loc, target_loc;
};

struct SeedArgs {
	DbView db;
	// Work units: a target, or one window of k-mer start positions [begin, end) of a long target.  Windows of one
	// target overlap by max_len + 50 (rounded up), the reach of the first cull, so every seed pair the reference's
	// cull keeps lies inside one window; seeds keep their true target_loc, bind windows are never clipped, and what
	// two overlapping windows both find is found twice, identically (the host drops the copies).
	const uint32_t *unit_target, *unit_begin, *unit_end;
	uint32_t unit0, unit1;
	int word_size;
	const uint16_t *wt_off;    // this chunk's bucket offsets [4^k + 1], relative to wt_ent (< 65536 entries per chunk)
	const uint32_t *wt_ent;    // this chunk's entries (see tnt_dev.h)
	uint32_t n_ent;            // number of entries of the chunk
	const uint16_t *wlist;     // concatenated word lists (little-endian 2-bit words)
	const uint32_t *wl_off;    // [((assay * 3 + role) * 2 + minus)] -> first word
	const unsigned long long *wl_seq; // [list][2]: the strand-specific oligo, 2 bits per base, base 0 in the low bits
	uint32_t assay_begin, assay_end; // chunk of expanded assays handled by this launch
	int max_len;               // -l
	int single_primer_pcr;
	Item *items;
	uint32_t item_cap;
	uint32_t *item_count;      // [N_SHARD * SHARD_STRIDE] sharded counters, region r = [r * cap / N_SHARD, ...)
	Group *groups;
	uint32_t group_cap;
	uint32_t *group_count;
	uint32_t *qa;              // indices of the items bound in phase A
	uint32_t *qa_count;
	uint32_t *big_count;       // groups of more than 64 items (the pair kernel's second instance only runs for them)
	// Shared oligos (degenerate expansion, --plex): the reference memoises every (oligo, window) it evaluates on a target
	// (BindCache, tntblast.h:247-324).  Here the seed kernel, which holds all seeds of a target in LDS, marks the
	// phase-A items that repeat an earlier one -- same canonical slot (oligo, strand concentration, thresholds), same
	// seed -- and the bind stage evaluates the first and hands its result to the others.
	const uint32_t *slot_canon; // [slot] first slot with the same (oligo, log_ct, thresh)
	const uint32_t *slot_next;  // [slot] next slot of that class (ascending), ITEM_NO_REP at the end
	uint32_t *item_rep;         // [item] the item whose bind result this one takes, ITEM_NO_REP otherwise
	SeedOut *seeds_out;        // optional
	uint32_t seeds_cap;
	uint32_t *seeds_count;
	uint32_t *err;
	uint32_t *debug; // experimental builds only (TNT_EXP_STATS)
};

__device__ __forceinline__ uint32_t target_word(const uint32_t *pk, uint32_t pos, uint32_t wmask)
{
	const uint32_t w = pos >> 4, sh = (pos & 15u) * 2u;
	const unsigned long long v = (unsigned long long)pk[w] | ((unsigned long long)pk[w + 1] << 32);
	return uint32_t(v >> sh) & wmask;
}

// role groups in the order amplicon() assembles match_list (amplicon_search.cpp:96-120):
// F-, R-, F+, R+, P-, P+
__device__ __forceinline__ uint32_t group_rank(uint32_t role, bool minus)
{
	if (role == 2) return minus ? 4u : 5u;
	return (minus ? 0u : 2u) + role;
}

// 64-bit key of one seed.  bits 63-56 local assay; low five bits: has_probe (16) | non-PCR (8) | role (6) | minus (1).
//  PCR:      bits 55-24 target_loc, 23-21 assembly group, 20-14 word index
//     = order of the list after the first cull's stable sort by target_loc (amplicon_search.cpp:688)
//  non-PCR:  bits 55-54 list id (minus: up, down; plus: up, down), 53-20 query_loc - target_loc + 2^33,
//            19-13 word index = per (strand, oligo) list in diagonal order (sort_by_delta, bind_oligo.cpp:157)
__device__ __forceinline__ unsigned long long seed_key(uint32_t local_assay, uint32_t ent, uint32_t p)
{
	const uint32_t wi = ent & 127u, role = (ent >> 8) & 3u;
	const bool minus = (ent & WT_MINUS) != 0, nonpcr = (ent & WT_NONPCR) != 0;
	unsigned long long k = (unsigned long long)local_assay << 56;
	if (!nonpcr) {
		k |= ((unsigned long long)p << 24) | ((unsigned long long)group_rank(role, minus) << 21) | ((unsigned long long)wi << 14);
	} else {
		const unsigned long long listid = (minus ? 0u : 2u) + (role == 1 ? 0u : 1u);
		const unsigned long long diag = (unsigned long long)(long(wi) - long(p) + (1l << 33));
		k |= (listid << 54) | (diag << 20) | ((unsigned long long)wi << 13);
	}
	return k | ((ent & WT_HASPROBE) ? 16u : 0u) | (nonpcr ? 8u : 0u) | (role << 1) | (minus ? 1u : 0u);
}

__device__ __forceinline__ void key_decode(unsigned long long k, uint32_t &wi, uint32_t &p)
{
	if (!(k & 8u)) {
		wi = uint32_t(k >> 14) & 127u;
		p = uint32_t(k >> 24);
	} else {
		wi = uint32_t(k >> 13) & 127u;
		const long diag = long((k >> 20) & ((1ull << 34) - 1ull)) - (1l << 33);
		p = uint32_t(long(wi) - diag);
	}
}

__device__ __forceinline__ uint32_t key_role(unsigned long long k) { return uint32_t(k >> 1) & 3u; }

// Is (wi, p) the first seed of its diagonal?  Word-list form (any oligo).
__device__ __forceinline__ bool first_on_diagonal_words(const uint32_t *pk, const uint16_t *wl, uint32_t wi, uint32_t p, uint32_t wmask)
{
	const uint32_t kmax = wi < p ? wi : p;
	for (uint32_t k = 1; k <= kmax; k++)
		if (uint32_t(wl[wi - k]) == target_word(pk, p - k, wmask)) return false;
	return true;
}

// Same test on packed bases for plain A/C/G/T oligos (word index == base offset) when the compared
// stretch fits 32 bases: XOR the two 64-bit windows and look for a run of `ws` matching bases that
// starts before the seed.
__device__ __forceinline__ bool first_on_diagonal_packed(const uint32_t *pk, const unsigned long long *seq, uint32_t wi, uint32_t p,
                                                         int ws)
{
	const uint32_t kmax = wi < p ? wi : p;
	if (kmax == 0) return true;
	const uint32_t t0 = p - kmax, w0 = t0 >> 4, sh = (t0 & 15u) * 2u;
	const unsigned long long lo = (unsigned long long)pk[w0] | ((unsigned long long)pk[w0 + 1] << 32);
	const unsigned long long T = sh ? ((lo >> sh) | ((unsigned long long)pk[w0 + 2] << (64u - sh))) : lo;
	const uint32_t s2 = (wi - kmax) * 2u;
	const unsigned long long olo = seq[0], ohi = seq[1];
	const unsigned long long O = s2 == 0 ? olo : (s2 < 64u ? ((olo >> s2) | (ohi << (64u - s2))) : (ohi >> (s2 - 64u)));
	const unsigned long long X = T ^ O;
	const unsigned long long Z = ~(X | (X >> 1)) & 0x5555555555555555ull; // one bit per matching base
	unsigned long long R = Z;
	for (int i = 1; i < ws; i++) R &= Z >> (2 * i);
	const unsigned long long before = (kmax >= 32u) ? ~0ull : ((1ull << (2u * kmax)) - 1ull);
	return (R & before) == 0;
}

// LDS_TABLES: the chunk's query-word table is staged in LDS once per workgroup (word size <= SEED_LDS_WORD_SIZE, at most
// SEED_LDS_ENT entries) and the scan walks SEED_SWEEP positions per lane out of registers; otherwise the tables are read
// from HBM / L2 position by position (word size 8, over-long chunks).
// MODE: what a launch is for -- SEED_SEARCH: culls and emits bind work items; SEED_SEARCH_SHARED: the same plus the marking of
// repeated (oligo, window) items of assays that share oligos; SEED_LIST: the seeds themselves (unit-level API, windows of long
// targets).  Separate instances: the common one does not carry the other two's code and pointers (this kernel spills scalar
// registers as it is).
enum { SEED_SEARCH = 0, SEED_SEARCH_SHARED = 1, SEED_LIST = 2 };
template <bool LDS_TABLES, int MODE> __global__ __launch_bounds__(64 * (LDS_TABLES ? SEED_LDS_WAVES : SEED_WAVES), (LDS_TABLES ? 2 : TNT_SEED_MINBLOCKS)) void seed_kernel(SeedArgs a)
{
	constexpr int WAVES = LDS_TABLES ? SEED_LDS_WAVES : SEED_WAVES;
	__shared__ unsigned long long s_raw[WAVES][SEED_CAP];
	__shared__ unsigned long long s_srt[WAVES][SEED_CAP];
	__shared__ uint16_t s_hist[WAVES][SEED_MAX_ASSAYS];
	__shared__ uint16_t s_off[WAVES][SEED_MAX_ASSAYS];
	__shared__ uint32_t s_cnt[WAVES];
	__shared__ uint32_t s_ncand[WAVES];
	__shared__ uint16_t s_wt_off[LDS_TABLES ? SEED_LDS_OFF : 2];
	__shared__ uint32_t s_wt_ent[LDS_TABLES ? SEED_LDS_ENT : 1];
	__shared__ unsigned long long s_seq[LDS_TABLES ? SEED_LDS_SEQ : 1];

	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned long long *raw = s_raw[wv], *srt = s_srt[wv];
	uint16_t *hist = s_hist[wv], *off = s_off[wv];
	const uint32_t wmask = (1u << (2 * a.word_size)) - 1u;
	const uint32_t n_chunk = a.assay_end - a.assay_begin;

	const uint32_t shard = (blockIdx.x * WAVES + wv) & (N_SHARD - 1);
	// packed oligos of the chunk's lists: from LDS when they fit
	const bool seq_lds = LDS_TABLES && n_chunk * 12u <= uint32_t(SEED_LDS_SEQ);
	if (LDS_TABLES) {
		const uint32_t n_off = (wmask + 2u + 1u) / 2u; // 4^k + 1 offsets, copied as dwords
		const uint32_t *src = reinterpret_cast<const uint32_t *>(a.wt_off);
		uint32_t *dst = reinterpret_cast<uint32_t *>(s_wt_off);
		for (uint32_t i = threadIdx.x; i < n_off; i += blockDim.x) dst[i] = src[i];
		for (uint32_t i = threadIdx.x; i < a.n_ent; i += blockDim.x) s_wt_ent[i] = a.wt_ent[i];
		if (seq_lds)
			for (uint32_t i = threadIdx.x; i < n_chunk * 12u; i += blockDim.x) s_seq[i] = a.wl_seq[size_t(a.assay_begin) * 12u + i];
		__syncthreads();
	}

#ifdef TNT_EXP_STATS
	long long st_t[7] = {0, 0, 0, 0, 0, 0, 0};
	long long st_acc[6] = {0, 0, 0, 0, 0, 0};
	long long st_drain = 0; // part of the scan slot spent in the de-duplication
#define ST(K) st_t[K] = __builtin_readcyclecounter()
#endif
	for (uint32_t unit = a.unit0 + blockIdx.x * WAVES + wv; unit < a.unit1; unit += gridDim.x * WAVES) {
		const uint32_t t = a.unit_target[unit];
#ifdef TNT_EXP_STATS
		for (int q = 1; q < 7; q++) st_t[q] = 0;
#endif
		const uint32_t len = a.db.len[t];
		for (uint32_t i = lane; i < uint32_t(SEED_MAX_ASSAYS); i += 64) hist[i] = 0; // all of it: the tail may hold representatives (see below)
		if (lane == 0) s_cnt[wv] = 0;
		wave_sync();
#ifdef TNT_EXP_STATS
		ST(0);
#endif

		// ---------------------------------------------------------------- scan + diagonal de-duplication
		if (len >= uint32_t(a.word_size)) { // tntblast_local.cpp:513
			const uint32_t *pk = a.db.packed + a.db.word_off[t];
			const uint32_t npos_all = len - a.word_size + 1;
			const uint32_t npos = a.unit_end[unit] < npos_all ? a.unit_end[unit] : npos_all;
			const uint32_t pos0 = a.unit_begin[unit];
			// Stage 1: every lane looks its positions up and appends the (position, entry) candidates to a wave-local
			// list (kept in `srt`, free until the counting sort).  Only ~0.2 candidates fall on a position, so testing
			// them in place would run the test with a handful of active lanes; stage 2 works through the list with
			// all lanes busy.
			// The list holds SEED_CAP candidates; when it fills up (buckets shared by many expanded assays come in
			// bursts) it is drained and the scan resumes where every lane stopped.
			uint2 *cand = reinterpret_cast<uint2 *>(srt);
			// Stage 2: diagonal de-duplication of the candidates, kept seeds -> `raw`
			auto drain = [&](uint32_t ncand) __attribute__((always_inline)) {
				wave_sync(); // the list was written by other lanes
#ifdef TNT_EXP_STATS
				const long long st_d0 = __builtin_readcyclecounter();
#endif
				for (uint32_t c = lane; c < ncand; c += 64) {
					const uint32_t cp = cand[c].x, ent = cand[c].y;
					const uint32_t wi = ent & 127u;
					const uint32_t assay = ent >> WT_ASSAY_SHIFT;
					const uint32_t li = (assay * 3u + ((ent >> 8) & 3u)) * 2u + ((ent & WT_MINUS) ? 1u : 0u);
					const uint32_t kmax = wi < cp ? wi : cp;
#ifndef TNT_EXP_NODEDUPE
					bool first;
					if ((ent & WT_PLAIN) && kmax + uint32_t(a.word_size) <= 32u) {
						if (seq_lds) first = first_on_diagonal_packed(pk, s_seq + 2 * (li - a.assay_begin * 6u), wi, cp, a.word_size);
						else first = first_on_diagonal_packed(pk, a.wl_seq + 2 * li, wi, cp, a.word_size);
					} else {
						first = first_on_diagonal_words(pk, a.wlist + a.wl_off[li], wi, cp, wmask);
					}
					if (!first) continue;
#endif
					const uint32_t la = assay - a.assay_begin;
					const uint32_t slot = atomicAdd(&s_cnt[wv], 1u);
					if (slot < uint32_t(SEED_CAP)) {
						raw[slot] = seed_key(la, ent, cp);
						atomicAdd(reinterpret_cast<unsigned int *>(hist) + (la >> 1), (la & 1u) ? 0x10000u : 1u);
					}
				}
				wave_sync();
#ifdef TNT_EXP_STATS
				st_drain += __builtin_readcyclecounter() - st_d0;
#endif
			};
			// The position-by-position scan of [p_begin, p_end): every lane looks one position up per step and appends its
			// (position, entry) candidates through an LDS counter; a full list is drained and the scan resumes where every
			// lane stopped.  The only scan of the instance without LDS tables; the overflow path of the sweep below.
			auto scan_slow = [&](uint32_t p_begin, uint32_t p_end) __attribute__((always_inline)) {
				uint32_t p = p_begin + lane, e_done = 0;
				while (true) {
					if (lane == 0) s_ncand[wv] = 0;
					wave_sync();
					while (true) {
						bool over = false;
						if (p < p_end) {
							const uint32_t w = target_word(pk, p, wmask);
							const uint32_t e0 = uint32_t(LDS_TABLES ? s_wt_off[w] : a.wt_off[w]) + e_done, e1 = LDS_TABLES ? s_wt_off[w + 1] : a.wt_off[w + 1];
							uint32_t fit = 0;
							if (e1 > e0) {
								const uint32_t at = atomicAdd(&s_ncand[wv], e1 - e0);
								fit = at < uint32_t(SEED_CAP) ? (e1 - e0 < uint32_t(SEED_CAP) - at ? e1 - e0 : uint32_t(SEED_CAP) - at) : 0u;
								for (uint32_t x = 0; x < fit; x++) cand[at + x] = make_uint2(p, LDS_TABLES ? s_wt_ent[e0 + x] : a.wt_ent[e0 + x]);
							}
							if (fit == e1 - e0) p += 64, e_done = 0;
							else e_done += fit, over = true;
						}
						if (__builtin_amdgcn_ballot_w64(over) != 0 || __builtin_amdgcn_ballot_w64(p < p_end) == 0) break;
					}
					wave_sync();
					uint32_t ncand = s_ncand[wv];
					if (ncand > uint32_t(SEED_CAP)) ncand = SEED_CAP;
					drain(ncand);
					if (__builtin_amdgcn_ballot_w64(p < p_end) == 0) break;
				}
			};
#ifdef TNT_EXP_OLDSCAN
			if (!LDS_TABLES) {
				scan_slow(pos0, npos);
			} else
#endif
			{
				// Sweeps of 64 x SEED_SWEEP positions: a lane loads the two packed words that hold its positions (and the
				// k - 1 bases behind them), extracts the 16 k-mers with shifts, reads their bucket bounds (all 32 reads in
				// flight at once: from LDS, or from L2 in the instance without LDS tables) and counts its candidates; one
				// wave-wide prefix sum gives every lane its slice of the candidate list.
				uint32_t ncand = 0; // candidates in the list (wave-uniform)
				for (uint32_t base = pos0 & ~uint32_t(SEED_SWEEP - 1); base < npos; base += 64 * SEED_SWEEP) {
					const uint32_t p0 = base + uint32_t(lane) * SEED_SWEEP;
					unsigned long long v = 0;
					if (p0 < npos) v = ((unsigned long long)pk[p0 >> 4] | ((unsigned long long)pk[(p0 >> 4) + 1] << 32)) >> ((p0 & 15u) * 2u);
					uint32_t bnd[SEED_SWEEP]; // first entry | count << 16
					uint32_t mine = 0;
#pragma unroll
					for (int q = 0; q < SEED_SWEEP; q++) {
						const uint32_t w = uint32_t(v >> (2 * q)) & wmask;
						// both bounds of the bucket with one (2-byte aligned) dword read: the gathers of this scan are bound by
						// the number of scattered requests, not by bytes
						uint32_t e0, e1;
						if (LDS_TABLES) {
							e0 = s_wt_off[w], e1 = s_wt_off[w + 1];
						} else {
							const uint32_t two = *reinterpret_cast<const u32_a2 *>(a.wt_off + w);
							e0 = two & 0xFFFFu, e1 = two >> 16;
						}
						const uint32_t n = (p0 + q >= pos0 && p0 + q < npos) ? e1 - e0 : 0u;
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
					if (ncand + total > uint32_t(SEED_CAP)) { // make room
						drain(ncand);
						ncand = 0;
					}
					if (total > uint32_t(SEED_CAP)) { // a burst (buckets shared by many expanded assays): position by position
						const uint32_t e = base + 64 * SEED_SWEEP;
						scan_slow(base > pos0 ? base : pos0, e < npos ? e : npos);
						continue;
					}
					uint32_t o = ncand + incl - mine;
#pragma unroll
					for (int q = 0; q < SEED_SWEEP; q++) {
						uint32_t e = bnd[q] & 0xFFFFu;
						for (uint32_t n = bnd[q] >> 16; n > 0; n--, e++) cand[o++] = make_uint2(p0 + q, LDS_TABLES ? s_wt_ent[e] : a.wt_ent[e]);
					}
					ncand += total;
				}
				wave_sync();
				drain(ncand);
			}
		}
		wave_sync();
#ifdef TNT_EXP_STATS
		ST(1);
#endif
		uint32_t n = s_cnt[wv];
		if (n > uint32_t(SEED_CAP)) {
			if (lane == 0) atomicOr(a.err, ERR_SEED_OVERFLOW);
			n = SEED_CAP;
		}

		if (MODE == SEED_LIST && a.seeds_out) {
			for (uint32_t i = lane; i < n; i += 64) {
				const uint32_t o = atomicAdd(a.seeds_count, 1u);
				if (o < a.seeds_cap) {
					const unsigned long long k = raw[i];
					uint32_t wi, p;
					key_decode(k, wi, p);
					a.seeds_out[o] = SeedOut{t, (a.assay_begin + uint32_t(k >> 56)) * 4u + key_role(k), uint32_t(k & 1u), wi, p};
				}
			}
		}
#ifdef TNT_EXP_SCANONLY
		if (false) {
#else
		if (MODE != SEED_LIST && a.items && n > 0) {
#endif
			// ------------------------------------------------------------ counting sort by assay (wave-wide)
			// exclusive prefix of hist -> off (4 assays per lane)
			{
				uint32_t c[SEED_MAX_ASSAYS / 64], sum = 0;
#pragma unroll
				for (int m = 0; m < SEED_MAX_ASSAYS / 64; m++) {
					const uint32_t la = uint32_t(lane) * (SEED_MAX_ASSAYS / 64) + m;
					c[m] = la < n_chunk ? hist[la] : 0u;
					sum += c[m];
				}
				uint32_t incl = sum;
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t y = __shfl_up(incl, d);
					if (lane >= d) incl += y;
				}
				uint32_t run = incl - sum;
#pragma unroll
				for (int m = 0; m < SEED_MAX_ASSAYS / 64; m++) {
					const uint32_t la = uint32_t(lane) * (SEED_MAX_ASSAYS / 64) + m;
					if (la < n_chunk) off[la] = uint16_t(run);
					run += c[m];
				}
			}
			wave_sync();
			for (uint32_t i = lane; i < n; i += 64) {
				const unsigned long long k = raw[i];
				const uint32_t la = uint32_t(k >> 56);
				const uint32_t old = atomicAdd(reinterpret_cast<unsigned int *>(off) + (la >> 1), (la & 1u) ? 0x10000u : 1u);
				srt[(la & 1u) ? (old >> 16) : (old & 0xFFFFu)] = k;
			}
			wave_sync();
#ifdef TNT_EXP_STATS
		ST(2);
#endif

			// ------------------------------------------------------------ first cull + emission, one lane per seed
			// The seeds sit in `srt`, grouped by assay (segment of assay la = [off[la] - hist[la], off[la])).  The
			// reference sorts each assay's list by target_loc and culls it with nested loops over the sorted list
			// (cull_oligo_match, amplicon_search.cpp:679-765).  The outcome per seed is a predicate over the other
			// seeds of its segment that only needs the ORDER RELATION of two keys (keys are unique and their numeric
			// order is the list order), so every seed evaluates it for itself with one or two passes over its
			// segment, no sort.  Items are emitted in arbitrary order inside their group; the pair kernel (which
			// only sees the ~1 % of groups that bound) restores the list order from the item fields.
			//
			// `raw` is dead after the counting sort; it holds the per-assay scratch of this phase:
			uint32_t *cmcp = reinterpret_cast<uint32_t *>(raw);            // [la] cm | cp << 16 (then: item | qa cursor)
			uint32_t *nvna = cmcp + SEED_MAX_ASSAYS;                        // [la] valid seeds | phase-A seeds << 16
			uint16_t *ibase = reinterpret_cast<uint16_t *>(nvna + SEED_MAX_ASSAYS); // [la] first item, relative to the wave's
			uint16_t *abase = ibase + SEED_MAX_ASSAYS;                      // [la] first phase-A queue entry
			uint16_t *gbase = abase + SEED_MAX_ASSAYS;                      // [la] group index
			static_assert(SEED_MAX_ASSAYS * (4 + 4 + 2 + 2 + 2) <= SEED_CAP * 8, "per-assay scratch must fit the raw key buffer");
			for (uint32_t i = lane; i < n_chunk; i += 64) cmcp[i] = 0, nvna[i] = 0;
			wave_sync();

			constexpr unsigned long long K_VALID = 1ull << 5, K_PHASE_B = 1ull << 6; // free key bits (see seed_key)
			const uint32_t threshold = uint32_t(a.max_len) + 50u;
			for (uint32_t i = lane; i < n; i += 64) {
				unsigned long long k = srt[i];
				const uint32_t la = uint32_t(k >> 56);
				const uint32_t cnt = hist[la], s0 = uint32_t(off[la]) - cnt;
				bool valid = false;
				if (cnt > uint32_t(LIST_MAX)) {
					atomicOr(a.err, ERR_LIST_OVERFLOW);
				} else if (k & 8u) {
					valid = true; // probe / padlock lists are bound as a whole (hash overload, bind_oligo.cpp:124-454)
				} else {
					const bool has_probe = (k & 16u) != 0, single = a.single_primer_pcr != 0;
					const uint32_t role = key_role(k);
					const bool minus = k & 1u;
					// The order relation of two keys of a segment is the order of bits 14..55 (target_loc, assembly group, word
					// index).  A work unit that is culled here is a whole target of at most 32 768 positions (build_units), so
					// the 32 bits from bit 14 up hold all of it: 32-bit comparisons (the 64-bit ones were most of this phase).
					auto order_of = [](unsigned long long key) __attribute__((always_inline)) {
						return __builtin_amdgcn_alignbit(uint32_t(key >> 32), uint32_t(key), 14); // wi | group << 7 | loc << 10
					};
					const uint32_t ko = order_of(k), loc = ko >> 10;
					// four keys of the segment per step (a key read twice changes nothing: every use is a minimum, a maximum or an "exists")
					const uint32_t last = s0 + cnt - 1u;
#define SEG4(X, O0, O1, O2, O3)                                                    \
	const unsigned long long O0 = srt[s0 + (X)],                                    \
	                         O1 = srt[(s0 + (X) + 1u < last ? s0 + (X) + 1u : last)], \
	                         O2 = srt[(s0 + (X) + 2u < last ? s0 + (X) + 2u : last)], \
	                         O3 = srt[(s0 + (X) + 3u < last ? s0 + (X) + 3u : last)]
					// One pass over the segment:
					//  * the element FOLLOWING this one in list order (amplicon_search.cpp:748-755 counts its strand when this
					//    seed is valid; past-the-end counts nothing);
					//  * a primer is valid iff a partner exists: other strand, allowed oligo, within reach (the reference takes
					//    the closest partner per oligo; the closest one is within reach iff any is) and, with a probe in the
					//    assay, beyond the nearest probe on that side (a pair only counts with a probe between).  The pass
					//    keeps the nearest probe (`pnear`) and the partner within reach that lies farthest out (`far`): some
					//    partner lies beyond the probe iff that one does;
					//  * a probe needs the closest primers on either side (per oligo).
					// (An order word is never 0 for a primer on the plus strand or a probe -- their groups are 2..5 -- and never
					// all ones, so both serve as "none".)
					uint32_t next = ~0u;
					uint32_t pnear = minus ? ~0u : 0u, far = minus ? 0u : ~0u;
					uint32_t f_loc[2] = {0, 0}, r_loc[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
					bool f_ok[2] = {false, false}, r_ok[2] = {false, false};
					// (as it stands, with branches: the same body written with selects only was measured 6 % slower)
					for (uint32_t x = 0; x < cnt; x += 4) {
						SEG4(x, o0, o1, o2, o3);
						const unsigned long long os[4] = {o0, o1, o2, o3};
#pragma unroll
						for (int j = 0; j < 4; j++) {
							const uint32_t oo = order_of(os[j]), olow = uint32_t(os[j]);
							if (oo > ko && oo < next) next = oo;
							const uint32_t orole = (olow >> 1) & 3u;
							if (orole == 2) {
								if (minus ? (oo > ko && oo < pnear) : (oo < ko && oo > pnear)) pnear = oo;
								continue;
							}
							const uint32_t oloc = oo >> 10;
							const bool ominus = olow & 1u;
							if (role == 2) { // probe: primers on either side of me
								if (ominus && oo < ko && (!f_ok[orole] || oloc > f_loc[orole])) f_ok[orole] = true, f_loc[orole] = oloc;
								if (!ominus && oo > ko && (!r_ok[orole] || oloc < r_loc[orole])) r_ok[orole] = true, r_loc[orole] = oloc;
							} else if (single || orole != role) {
								if (minus) { // I am an f: partners r to my right
									if (!ominus && oo > ko && oloc - loc <= threshold && oo > far) far = oo;
								} else {     // I am an r: partners f to my left
									if (ominus && oo < ko && loc - oloc <= threshold && oo < far) far = oo;
								}
							}
						}
					}
					if (role != 2) {
						if (!has_probe) valid = minus ? far != 0u : far != ~0u;
						else valid = minus ? (pnear != ~0u && far > pnear) : (pnear != 0u && far < pnear);
					}
#undef SEG4
					if (role == 2) {
						valid = false;
						for (int rf = 0; rf < 2; rf++)
							for (int rr = 0; rr < 2; rr++)
								if ((single || rf != rr) && f_ok[rf] && r_ok[rr] && r_loc[rr] - f_loc[rf] <= threshold) valid = true;
					}
					if (valid) {
						// strand of the list element FOLLOWING a valid one decides which strand is bound first (:131); its
						// assembly group tells the strand: F-, R-, F+, R+, P-, P+
						if (next != ~0u) atomicAdd(&cmcp[la], ((0x13u >> ((next >> 7) & 7u)) & 1u) ? 1u : 0x10000u);
					}
				}
				if (valid) {
					k |= K_VALID;
					atomicAdd(&nvna[la], 1u);
				}
				srt[i] = k;
			}
			wave_sync();
#ifdef TNT_EXP_STATS
		ST(3);
#endif
			// Phase B = what the reference binds only after both primers were tried on the first strand
			// (amplicon_search.cpp:131-353): the other strand's primers and the probe.
			for (uint32_t i = lane; i < n; i += 64) {
				unsigned long long k = srt[i];
				if (!(k & K_VALID)) continue;
				const uint32_t la = uint32_t(k >> 56);
				bool pb = false;
				if (!(k & 8u)) {
					const uint32_t c2 = cmcp[la];
					const bool minus_first = (c2 & 0xFFFFu) < (c2 >> 16);
					pb = key_role(k) == 2 || bool(k & 1u) != minus_first;
				}
				if (pb) srt[i] = k | K_PHASE_B;
				else atomicAdd(&nvna[la], 0x10000u);
			}
			wave_sync();
#ifdef TNT_EXP_STATS
		ST(4);
#endif

			// wave-wide exclusive prefix sums of (items, phase-A items, groups) over the assays: one atomic per counter
			// and wavefront.  The counters are sharded (a single hot word sustains only ~88 atomics/us): every
			// wavefront appends to the output region of its shard; consumers map a linear index back to
			// (shard, offset) with the shard prefix sums (shard_locate).
			uint32_t bi = 0, ba = 0, bg = 0;
			{
				constexpr int PER = SEED_MAX_ASSAYS / 64;
				uint32_t nv[PER], na[PER], gf[PER];
				uint32_t ii = 0, ia = 0, ig = 0;
#pragma unroll
				for (int m = 0; m < PER; m++) {
					const uint32_t la = uint32_t(lane) * PER + m;
					const uint32_t v = la < n_chunk ? nvna[la] : 0u;
					nv[m] = v & 0xFFFFu, na[m] = v >> 16;
					gf[m] = la < n_chunk ? (((cmcp[la] & 0xFFFFu) < (cmcp[la] >> 16)) ? GROUP_MINUS_FIRST : 0u) : 0u;
					ii += nv[m], ia += na[m], ig += nv[m] ? 1u : 0u;
				}
				const uint32_t si = ii, sa = ia, sg = ig;
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t yi = __shfl_up(ii, d), ya = __shfl_up(ia, d), yg = __shfl_up(ig, d);
					if (lane >= d) ii += yi, ia += ya, ig += yg;
				}
				if (lane == 63) {
					if (ii) bi = atomicAdd(a.item_count + shard * SHARD_STRIDE, ii);
					if (ia) ba = atomicAdd(a.qa_count + shard * SHARD_STRIDE, ia);
					if (ig) bg = atomicAdd(a.group_count + shard * SHARD_STRIDE, ig);
				}
				const uint32_t ti = __shfl(ii, 63), tg = __shfl(ig, 63);
				bi = __shfl(bi, 63);
				ba = __shfl(ba, 63);
				bg = __shfl(bg, 63);
				if (ti == 0) goto next_target; // nothing survives the cull in this chunk
				if (bi + ti > a.item_cap / N_SHARD) {
					if (lane == 0) atomicOr(a.err, ERR_ITEM_OVERFLOW);
					goto next_target;
				}
				if (bg + tg > a.group_cap / N_SHARD) {
					if (lane == 0) atomicOr(a.err, ERR_GROUP_OVERFLOW);
					goto next_target;
				}
				bi += shard * (a.item_cap / N_SHARD);
				ba += shard * (a.item_cap / N_SHARD);
				bg += shard * (a.group_cap / N_SHARD);
				uint32_t ri = ii - si, ra = ia - sa, rg = ig - sg; // exclusive prefixes of this lane's first assay
#pragma unroll
				for (int m = 0; m < PER; m++) {
					const uint32_t la = uint32_t(lane) * PER + m;
					if (la < n_chunk) {
						ibase[la] = uint16_t(ri), abase[la] = uint16_t(ra), gbase[la] = uint16_t(rg);
						if (nv[m]) a.groups[bg + rg] = Group{unit, (a.assay_begin + la) | gf[m], bi + ri, nv[m]};
						if (nv[m] > 64u) atomicAdd(a.big_count, 1u);
					}
					ri += nv[m], ra += na[m], rg += nv[m] ? 1u : 0u;
				}
			}
			wave_sync();
#ifdef TNT_EXP_STATS
		ST(5);
#endif
			for (uint32_t i = lane; i < n_chunk; i += 64) cmcp[i] = 0; // now the (item | phase-A << 16) cursors
			constexpr bool DEDUPE = MODE == SEED_SEARCH_SHARED;
			static_assert(SEED_CAP == 2 * SEED_MAX_ASSAYS, "hist + off hold one representative index per seed");
			if (DEDUPE) {
				// Expanded assays that share an oligo see the same seeds of it.  A phase-A seed of such an assay looks for
				// its twin -- same strand, word and position -- in the segments of the assays before it in its class (the
				// slots with this oligo, strand concentration and threshold set, ascending); the first twin that is a
				// phase-A item too is bound for everybody behind it.  The representative's index goes where the segment
				// bounds were (hist / off are not needed after this).
				uint32_t reps[SEED_CAP / 64];
#pragma unroll
				for (int m = 0; m < SEED_CAP / 64; m++) {
					const uint32_t i = uint32_t(lane) + 64u * uint32_t(m);
					reps[m] = i;
					if (i >= n) continue;
					const unsigned long long k = srt[i];
					if (!(k & K_VALID) || (k & K_PHASE_B)) continue;
					const uint32_t slot = (a.assay_begin + uint32_t(k >> 56)) * 3u + key_role(k);
					const uint32_t canon = a.slot_canon[slot];
					if (canon == slot) continue;
					uint32_t wi, p;
					key_decode(k, wi, p);
					// the first usable twin along the class (ascending slots) stands for everybody behind it
					for (uint32_t cur = canon; cur != slot && cur != ITEM_NO_REP && reps[m] == i; cur = a.slot_next[cur]) {
						if (cur / 3u < a.assay_begin || cur / 3u >= a.assay_end) continue; // bound in another launch
						const uint32_t la0 = cur / 3u - a.assay_begin, role0 = cur % 3u;
						const uint32_t cnt0 = hist[la0], s0 = uint32_t(off[la0]) - cnt0;
						for (uint32_t x = 0; x < cnt0; x++) {
							const unsigned long long o = srt[s0 + x];
							if (!(o & K_VALID) || (o & K_PHASE_B) || key_role(o) != role0 || ((o ^ k) & 9u)) continue; // strand, PCR / non-PCR
							uint32_t owi, op;
							key_decode(o, owi, op);
							if (owi == wi && op == p) {
								reps[m] = s0 + x;
								break;
							}
						}
					}
				}
				wave_sync();
#pragma unroll
				for (int m = 0; m < SEED_CAP / 64; m++) {
					const uint32_t i = uint32_t(lane) + 64u * uint32_t(m);
					if (i < n) (i < uint32_t(SEED_MAX_ASSAYS) ? hist[i] : off[i - SEED_MAX_ASSAYS]) = uint16_t(reps[m]);
				}
			}
			wave_sync();
			for (uint32_t i = lane; i < n; i += 64) {
				const unsigned long long k = srt[i];
				if (!(k & K_VALID)) {
					if (DEDUPE) srt[i] = ~0ull;
					continue;
				}
				const uint32_t la = uint32_t(k >> 56), assay = a.assay_begin + la;
				const bool pb = (k & K_PHASE_B) != 0;
				const uint32_t cur = atomicAdd(&cmcp[la], pb ? 1u : 0x10001u);
				const uint32_t o = bi + ibase[la] + (cur & 0xFFFFu);
				uint32_t wi, p;
				key_decode(k, wi, p);
				if (DEDUPE) {
					const uint32_t rep = i < uint32_t(SEED_MAX_ASSAYS) ? hist[i] : off[i - SEED_MAX_ASSAYS];
					srt[i] = ((unsigned long long)rep << 32) | o; // the key is not needed any more
				}
				Item it;
				it.target = t;
				it.tloc = p;
				// Slot table is laid out [assay][role]
				it.info = wi | ((k & 1u) ? ITEM_MINUS : 0u) | (pb ? ITEM_PHASE_B : 0u) | ((assay * 3u + key_role(k)) << ITEM_SLOT_SHIFT);
				it.group = bg + gbase[la];
				a.items[o] = it;
				if (!pb) a.qa[ba + abase[la] + (cur >> 16)] = o;
			}
			if (DEDUPE) {
				wave_sync();
				for (uint32_t i = lane; i < n; i += 64) {
					const unsigned long long v = srt[i];
					if (v == ~0ull) continue;
					const uint32_t rep = uint32_t(v >> 32);
					a.item_rep[uint32_t(v)] = rep != i ? uint32_t(srt[rep]) : ITEM_NO_REP;
				}
			}
		}
	next_target:
		wave_sync();
#ifdef TNT_EXP_STATS
		ST(6);
		{
			long long prev = st_t[0];
			for (int q = 1; q < 7; q++) {
				if (st_t[q] == 0) continue; // phase skipped (no seeds)
				st_acc[q - 1] += st_t[q] - prev;
				prev = st_t[q];
			}
		}
#endif
	}
#ifdef TNT_EXP_STATS
	if (lane == 0 && a.debug) {
		for (int q = 0; q < 6; q++) atomicAdd(a.debug + q, uint32_t(st_acc[q] >> 10));
		atomicAdd(a.debug + 6, uint32_t(st_drain >> 10));
		atomicAdd(a.debug + 7, 1u);
	}
#endif
}

} // namespace tnt
#endif
