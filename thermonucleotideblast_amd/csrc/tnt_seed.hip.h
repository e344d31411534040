// tnt_seed.hip.h -- gfx950 seed-scan kernel: streams 2-bit packed targets, looks every k-mer up in
// the inverted query-word table, keeps the first seed of every (oligo, strand, diagonal), sorts the
// seeds of one target in LDS, runs the first cull per assay and emits bind work items.
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
// One 256-thread workgroup per target: the four wavefronts scan the target's positions, kept seeds
// go to an LDS array as 64-bit sort keys, a bitonic sort groups them by assay in the order the
// reference's lists have, and one lane per non-empty assay segment runs the cull and emits items.
#ifndef TNT_SEED_HIP_H
#define TNT_SEED_HIP_H

#include <hip/hip_runtime.h>
#include "tnt_dev.h"

namespace tnt {

constexpr int SEED_THREADS = 256;
constexpr int SEED_CAP = 2048;        // seeds of one target x one assay chunk held in LDS (power of two)
constexpr int SEED_MAX_ASSAYS = 1024; // assays per chunk (10-bit local index in the sort key)
constexpr int LIST_MAX = 64;          // seeds of one (target, assay)
constexpr int WT_LDS_WORDS = (1 << 14) + 1; // bucket offsets staged in LDS for k <= 7

struct SeedOut { // debug / unit-level output: every kept seed before the cull
	uint32_t target, assay_role, strand_minus, query_loc, target_loc;
};

struct SeedArgs {
	DbView db;
	int word_size;
	const uint16_t *wt_off;    // this chunk's bucket offsets [4^k + 1], relative to wt_ent (< 65536 entries per chunk)
	const uint32_t *wt_ent;    // this chunk's entries (see tnt_dev.h)
	const uint16_t *wlist;     // concatenated word lists (little-endian 2-bit words)
	const uint32_t *wl_off;    // [((assay * 3 + role) * 2 + minus)] -> first word
	const AssayDev *assays;
	uint32_t assay_begin, assay_end; // chunk of expanded assays handled by this launch
	int max_len;               // -l
	int single_primer_pcr;
	Item *items;
	uint32_t item_cap;
	uint32_t *item_count;
	Group *groups;
	uint32_t group_cap;
	uint32_t *group_count;
	SeedOut *seeds_out;        // optional
	uint32_t seeds_cap;
	uint32_t *seeds_count;
	uint32_t *err;
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

// 64-bit sort key of one seed.  bits 63-54 local assay; low three bits role (2) | minus (1).
//  kind 0 (PCR):   bits 53-22 target_loc, 21-19 assembly group, 18-12 word index
//     = order of the list after the first cull's stable sort by target_loc (amplicon_search.cpp:688)
//  kind 1/2:       bits 53-52 list id (minus: up, down; plus: up, down), 51-18 query_loc - target_loc + 2^33,
//                  17-11 word index = per (strand, oligo) list in diagonal order (sort_by_delta, bind_oligo.cpp:157)
__device__ __forceinline__ unsigned long long seed_key(uint32_t local_assay, bool pcr, uint32_t role, bool minus, uint32_t wi,
                                                       uint32_t p)
{
	unsigned long long k = (unsigned long long)local_assay << 54;
	if (pcr) {
		k |= ((unsigned long long)p << 22) | ((unsigned long long)group_rank(role, minus) << 19) | ((unsigned long long)wi << 12);
	} else {
		const unsigned long long listid = (minus ? 0u : 2u) + (role == 1 ? 0u : 1u);
		const unsigned long long diag = (unsigned long long)(long(wi) - long(p) + (1l << 33));
		k |= (listid << 52) | (diag << 18) | ((unsigned long long)wi << 11);
	}
	return k | (role << 1) | (minus ? 1u : 0u);
}

__device__ __forceinline__ void key_decode(unsigned long long k, bool pcr, uint32_t &wi, uint32_t &p)
{
	if (pcr) {
		wi = uint32_t(k >> 12) & 127u;
		p = uint32_t(k >> 22);
	} else {
		wi = uint32_t(k >> 11) & 127u;
		const long diag = long((k >> 18) & ((1ull << 34) - 1ull)) - (1l << 33);
		p = uint32_t(long(wi) - diag);
	}
}

template <bool LDS_TABLE> __global__ __launch_bounds__(SEED_THREADS) void seed_kernel(SeedArgs a)
{
	__shared__ unsigned long long s_key[SEED_CAP];
	__shared__ uint16_t s_wt[LDS_TABLE ? WT_LDS_WORDS + 1 : 2];
	__shared__ uint32_t s_n;

	const int tid = threadIdx.x;
	const uint32_t wmask = (1u << (2 * a.word_size)) - 1u;
	if (LDS_TABLE) {
		for (uint32_t i = tid; i <= wmask + 1u; i += SEED_THREADS) s_wt[i] = a.wt_off[i];
	}
	const uint16_t *wt = LDS_TABLE ? s_wt : a.wt_off;
	__syncthreads();

	for (uint32_t t = blockIdx.x; t < a.db.n_targets; t += gridDim.x) {
		const uint32_t len = a.db.len[t];
		if (tid == 0) s_n = 0;
		__syncthreads();

		if (len >= uint32_t(a.word_size)) { // tntblast_local.cpp:513
			const uint32_t *pk = a.db.packed + a.db.word_off[t];
			const uint32_t npos = len - a.word_size + 1;
			for (uint32_t p = tid; p < npos; p += SEED_THREADS) {
				const uint32_t w = target_word(pk, p, wmask);
				const uint32_t e0 = wt[w], e1 = wt[w + 1];
				for (uint32_t e = e0; e < e1; e++) {
					const uint32_t ent = a.wt_ent[e];
					const uint32_t wi = ent & 127u;
					const uint32_t role = (ent >> 8) & 3u;
					const bool minus = (ent & WT_MINUS) != 0;
					const uint32_t assay = ent >> 11;
					// first seed of its diagonal?
					const uint16_t *wl = a.wlist + a.wl_off[(assay * 3u + role) * 2u + (minus ? 1u : 0u)];
					const uint32_t kmax = wi < p ? wi : p;
					bool first = true;
					for (uint32_t k = 1; k <= kmax; k++)
						if (uint32_t(wl[wi - k]) == target_word(pk, p - k, wmask)) {
							first = false;
							break;
						}
					if (!first) continue;
					const uint32_t slot = atomicAdd(&s_n, 1u);
					if (slot < SEED_CAP) s_key[slot] = seed_key(assay - a.assay_begin, ((ent >> 10) & 1u) == 0, role, minus, wi, p);
				}
			}
		}
		__syncthreads();
		uint32_t n = s_n;
		if (n > SEED_CAP) {
			if (tid == 0) atomicOr(a.err, ERR_SEED_OVERFLOW);
			n = SEED_CAP;
		}

		if (a.seeds_out) {
			for (uint32_t i = tid; i < n; i += SEED_THREADS) {
				const uint32_t o = atomicAdd(a.seeds_count, 1u);
				if (o < a.seeds_cap) {
					const unsigned long long k = s_key[i];
					const uint32_t assay = a.assay_begin + uint32_t(k >> 54);
					uint32_t wi, p;
					key_decode(k, a.assays[assay].kind == 0, wi, p);
					a.seeds_out[o] = SeedOut{t, assay * 4u + (uint32_t(k >> 1) & 3u), uint32_t(k & 1u), wi, p};
				}
			}
		}

		if (a.items && n > 0) {
			// ------------------------------------------------------------ bitonic sort of the keys in LDS
			uint32_t np2 = 1;
			while (np2 < n) np2 <<= 1;
			for (uint32_t i = n + tid; i < np2; i += SEED_THREADS) s_key[i] = ~0ull;
			__syncthreads();
			for (uint32_t k = 2; k <= np2; k <<= 1) {
				for (uint32_t j = k >> 1; j > 0; j >>= 1) {
					for (uint32_t i = tid; i < np2; i += SEED_THREADS) {
						const uint32_t l = i ^ j;
						if (l > i) {
							const unsigned long long x = s_key[i], y = s_key[l];
							const bool up = (i & k) == 0;
							if ((x > y) == up) {
								s_key[i] = y;
								s_key[l] = x;
							}
						}
					}
					__syncthreads();
				}
			}

			// ------------------------------------------------------------ one lane per assay segment
			for (uint32_t i0 = tid; i0 < n; i0 += SEED_THREADS) {
				const uint32_t la = uint32_t(s_key[i0] >> 54);
				if (i0 > 0 && uint32_t(s_key[i0 - 1] >> 54) == la) continue; // not the head of a segment
				const uint32_t assay = a.assay_begin + la;
				const AssayDev A = a.assays[assay];

				uint32_t cnt = 0;
				int n_minus = 0, n_plus = 0, n_probe = 0;
				while (i0 + cnt < n && uint32_t(s_key[i0 + cnt] >> 54) == la) {
					const unsigned long long k = s_key[i0 + cnt];
					if ((uint32_t(k >> 1) & 3u) == 2) n_probe++;
					else if (k & 1u) n_minus++;
					else n_plus++;
					cnt++;
				}
				if (cnt > uint32_t(LIST_MAX)) {
					atomicOr(a.err, ERR_LIST_OVERFLOW);
					continue;
				}
				const unsigned long long *key = s_key + i0;

				unsigned long long valid = (cnt >= 64) ? ~0ull : ((1ull << cnt) - 1ull);
				uint32_t gflags = 0;
				if (A.kind == 0) {
					// early exits of amplicon(), amplicon_search.cpp:101-126
					if (n_minus == 0 || n_plus == 0) continue;
					if (A.has_probe && n_probe == 0) continue;
					// cull_oligo_match, amplicon_search.cpp:679-765 (every entry is still unbound here)
					valid = 0;
					const uint32_t threshold = uint32_t(a.max_len) + 50u;
					for (uint32_t f = 0; f < cnt; f++) {
						const uint32_t fr = uint32_t(key[f] >> 1) & 3u;
						const bool fminus = key[f] & 1u;
						if (!fminus || fr == 2) continue;
						const uint32_t floc = uint32_t(key[f] >> 22);
						for (uint32_t r = f + 1; r < cnt; r++) {
							const uint32_t rloc = uint32_t(key[r] >> 22);
							if (rloc - floc > threshold) break;
							const uint32_t rr = uint32_t(key[r] >> 1) & 3u;
							const bool rminus = key[r] & 1u;
							if (rminus || rr == 2) continue;
							if (!a.single_primer_pcr && fr == rr) continue;
							if (A.has_probe) {
								for (uint32_t p = f + 1; p < r; p++)
									if ((uint32_t(key[p] >> 1) & 3u) == 2) valid |= (1ull << p) | (1ull << f) | (1ull << r);
							} else {
								valid |= (1ull << f) | (1ull << r);
							}
						}
					}
					// strand counts of the first cull (amplicon_search.cpp:748-755): the reference looks at the element
					// FOLLOWING each valid one (past-the-end counts nothing here)
					int cm = 0, cp = 0;
					for (uint32_t i = 0; i + 1 < cnt; i++)
						if ((valid >> i) & 1ull) {
							const bool nminus = key[i + 1] & 1u;
							cm += nminus ? 1 : 0;
							cp += nminus ? 0 : 1;
						}
					if (cm < cp) gflags |= GROUP_MINUS_FIRST; // amplicon_search.cpp:131
				}
				const int n_valid = __popcll(valid);
				if (n_valid == 0) continue;

				const uint32_t first = atomicAdd(a.item_count, uint32_t(n_valid));
				const uint32_t g = atomicAdd(a.group_count, 1u);
				if (first + n_valid > a.item_cap) {
					atomicOr(a.err, ERR_ITEM_OVERFLOW);
					continue;
				}
				if (g >= a.group_cap) {
					atomicOr(a.err, ERR_GROUP_OVERFLOW);
					continue;
				}
				a.groups[g] = Group{t, assay | gflags, first, uint32_t(n_valid)};
				uint32_t o = first;
				for (uint32_t i = 0; i < cnt; i++) {
					if (!((valid >> i) & 1ull)) continue;
					const unsigned long long k = key[i];
					const uint32_t role = uint32_t(k >> 1) & 3u;
					const bool minus = k & 1u;
					uint32_t wi, p;
					key_decode(k, A.kind == 0, wi, p);
					Item it;
					it.target = t;
					it.tloc = p;
					it.slot = uint32_t(role == 0 ? A.slot_f : (role == 1 ? A.slot_r : A.slot_p));
					it.info = wi | (minus ? ITEM_MINUS : 0u) | (role << 9);
					a.items[o++] = it;
				}
			}
		}
		__syncthreads();
	}
}

} // namespace tnt
#endif
