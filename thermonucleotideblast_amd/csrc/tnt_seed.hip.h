// tnt_seed.hip.h -- gfx950 seed-scan kernel: streams 2-bit packed targets, looks every k-mer up in
// the inverted query-word table, keeps the first seed of every (oligo, strand, diagonal), groups the
// seeds per assay in LDS, runs the first cull and emits bind work items.
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
#ifndef TNT_SEED_HIP_H
#define TNT_SEED_HIP_H

#include <hip/hip_runtime.h>
#include "tnt_dev.h"

namespace tnt {

constexpr int SEED_THREADS = 64;     // one wavefront per target
constexpr int SEED_CAP = 4096;       // seeds of one target x one assay chunk held in LDS
constexpr int SEED_MAX_ASSAYS = 512; // assays per chunk
constexpr int LIST_MAX = 64;         // seeds of one (target, assay)

struct SeedOut { // debug / unit-level output: every kept seed before the cull
	uint32_t target, assay_role, strand_minus, query_loc, target_loc;
};

struct SeedArgs {
	DbView db;
	int word_size;
	const uint32_t *wt_off;    // [4^k + 1]
	const uint32_t *wt_ent;    // entries (see tnt_dev.h)
	const uint16_t *wlist;     // concatenated word lists (little-endian 2-bit words)
	const uint32_t *wl_off;    // [((assay * 3 + role) * 2 + minus)] -> first word; [+1] -> end
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
__device__ __forceinline__ int group_rank(int role, bool minus)
{
	if (role == 2) return minus ? 4 : 5;
	return (minus ? 0 : 2) + role;
}

__global__ __launch_bounds__(SEED_THREADS) void seed_kernel(SeedArgs a)
{
	__shared__ uint32_t s_ent[SEED_CAP];
	__shared__ uint32_t s_pos[SEED_CAP];
	__shared__ uint16_t s_next[SEED_CAP];
	__shared__ uint32_t s_head[SEED_MAX_ASSAYS];
	__shared__ uint32_t s_n;

	const int lane = threadIdx.x;
	const uint32_t wmask = (1u << (2 * a.word_size)) - 1u;
	const uint32_t n_chunk = a.assay_end - a.assay_begin;

	for (uint32_t t = blockIdx.x; t < a.db.n_targets; t += gridDim.x) {
		const uint32_t len = a.db.len[t];
		for (uint32_t i = lane; i < n_chunk; i += SEED_THREADS) s_head[i] = 0xFFFFFFFFu;
		if (lane == 0) s_n = 0;
		__syncthreads();

		if (len >= uint32_t(a.word_size)) { // tntblast_local.cpp:513
			const uint32_t *pk = a.db.packed + a.db.word_off[t];
			const uint32_t npos = len - a.word_size + 1;
			for (uint32_t p = lane; p < npos; p += SEED_THREADS) {
				const uint32_t w = target_word(pk, p, wmask);
				const uint32_t e0 = a.wt_off[w], e1 = a.wt_off[w + 1];
				for (uint32_t e = e0; e < e1; e++) {
					const uint32_t ent = a.wt_ent[e];
					const uint32_t wi = ent & 127u;
					const uint32_t assay = ent >> 10;
					if (assay < a.assay_begin || assay >= a.assay_end) continue;
					// first seed of its diagonal?
					const uint32_t li = (assay * 3u + ((ent >> 8) & 3u)) * 2u + ((ent & WT_MINUS) ? 1u : 0u);
					const uint16_t *wl = a.wlist + a.wl_off[li];
					const uint32_t kmax = wi < p ? wi : p;
					bool first = true;
					for (uint32_t k = 1; k <= kmax; k++)
						if (uint32_t(wl[wi - k]) == target_word(pk, p - k, wmask)) {
							first = false;
							break;
						}
					if (!first) continue;
					const uint32_t slot = atomicAdd(&s_n, 1u);
					if (slot < SEED_CAP) {
						s_ent[slot] = ent;
						s_pos[slot] = p;
						// push on the assay's list (order is restored by the sort below)
						const uint32_t old = atomicExch(&s_head[assay - a.assay_begin], slot);
						s_next[slot] = uint16_t(old);
					}
				}
			}
		}
		__syncthreads();
		uint32_t n_seeds = s_n;
		if (n_seeds > SEED_CAP) {
			if (lane == 0) atomicOr(a.err, ERR_SEED_OVERFLOW);
			n_seeds = SEED_CAP;
		}

		if (a.seeds_out) {
			for (uint32_t i = lane; i < n_seeds; i += SEED_THREADS) {
				const uint32_t o = atomicAdd(a.seeds_count, 1u);
				if (o < a.seeds_cap) {
					const uint32_t ent = s_ent[i];
					a.seeds_out[o] = SeedOut{t, (ent >> 10) * 4u + ((ent >> 8) & 3u), (ent & WT_MINUS) ? 1u : 0u, ent & 127u, s_pos[i]};
				}
			}
		}

		// ---------------------------------------------------------------- per-assay processing
		if (a.items) {
			for (uint32_t la = lane; la < n_chunk; la += SEED_THREADS) {
				if (s_head[la] == 0xFFFFFFFFu) continue;
				uint16_t h = uint16_t(s_head[la]);
				const uint32_t assay = a.assay_begin + la;
				const AssayDev A = a.assays[assay];

				// gather: key = sort key, val = (role, minus, word index)
				unsigned long long key[LIST_MAX];
				int n = 0;
				int n_minus = 0, n_plus = 0, n_probe = 0;
				bool overflow = false;
				while (h != 0xFFFFu) {
					const uint32_t ent = s_ent[h];
					const uint32_t role = (ent >> 8) & 3u;
					const bool minus = (ent & WT_MINUS) != 0;
					const uint32_t wi = ent & 127u;
					const uint32_t p = s_pos[h];
					unsigned long long k;
					if (A.kind == 0) {
						// order after the first cull's stable sort by target_loc (amplicon_search.cpp:688):
						// (target_loc, assembly group, word index)
						k = ((unsigned long long)p << 24) | ((unsigned long long)group_rank(int(role), minus) << 16) | (wi << 8);
					} else {
						// hash overloads (probe / padlock): per (strand, oligo) list in diagonal order
						// (sort_by_delta, bind_oligo.cpp:157): list id, then query_loc - target_loc ascending
						const int listid = (minus ? 0 : 2) + (role == 1 ? 0 : 1); // minus: up(R), down(F|P); plus: up, down
						const long diag = long(wi) - long(p) + (1l << 33);
						k = ((unsigned long long)listid << 56) | ((unsigned long long)diag << 16) | (wi << 8);
					}
					k |= (role << 1) | (minus ? 1u : 0u); // low byte carries role/strand
					if (n < LIST_MAX) key[n++] = k;
					else overflow = true;
					if (role == 2) n_probe++;
					else if (minus) n_minus++;
					else n_plus++;
					h = s_next[h];
				}
				if (overflow) atomicOr(a.err, ERR_LIST_OVERFLOW);

				// early exits of amplicon(), amplicon_search.cpp:101-126
				if (A.kind == 0) {
					if (n_minus == 0 || n_plus == 0) continue;
					if (A.has_probe && n_probe == 0) continue;
				}
				// insertion sort (keys are unique)
				for (int i = 1; i < n; i++) {
					const unsigned long long x = key[i];
					int j = i - 1;
					while (j >= 0 && key[j] > x) {
						key[j + 1] = key[j];
						j--;
					}
					key[j + 1] = x;
				}

				unsigned long long valid = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
				if (A.kind == 0) { // cull_oligo_match, amplicon_search.cpp:679-765 (all entries are unbound here)
					valid = 0;
					const uint32_t threshold = uint32_t(a.max_len) + 50u;
					for (int f = 0; f < n; f++) {
						const uint32_t fr = uint32_t(key[f] >> 1) & 3u;
						const bool fminus = key[f] & 1u;
						if (!fminus || fr == 2) continue;
						const uint32_t floc = uint32_t(key[f] >> 24);
						for (int r = f + 1; r < n; r++) {
							const uint32_t rloc = uint32_t(key[r] >> 24);
							if (rloc - floc > threshold) break;
							const uint32_t rr = uint32_t(key[r] >> 1) & 3u;
							const bool rminus = key[r] & 1u;
							if (rminus || rr == 2) continue;
							if (!a.single_primer_pcr && fr == rr) continue;
							if (A.has_probe) {
								for (int p = f + 1; p < r; p++)
									if ((uint32_t(key[p] >> 1) & 3u) == 2) valid |= (1ull << p) | (1ull << f) | (1ull << r);
							} else {
								valid |= (1ull << f) | (1ull << r);
							}
						}
					}
				}
				const int n_valid = __popcll(valid);
				if (n_valid == 0) continue;
				// strand counts of the first cull (amplicon_search.cpp:748-755): the reference looks at the element
				// FOLLOWING each valid one (past-the-end counts nothing here)
				uint32_t gflags = 0;
				if (A.kind == 0) {
					int cm = 0, cp = 0;
					for (int i = 0; i + 1 < n; i++)
						if ((valid >> i) & 1ull) {
							const bool nminus = key[i + 1] & 1u;
							cm += nminus ? 1 : 0;
							cp += nminus ? 0 : 1;
						}
					if (cm < cp) gflags |= GROUP_MINUS_FIRST; // amplicon_search.cpp:131
				}

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
				for (int i = 0; i < n; i++) {
					if (!((valid >> i) & 1ull)) continue;
					const uint32_t role = uint32_t(key[i] >> 1) & 3u;
					const bool minus = key[i] & 1u;
					const uint32_t wi = uint32_t(key[i] >> 8) & 255u;
					uint32_t p;
					if (A.kind == 0) {
						p = uint32_t(key[i] >> 24);
					} else {
						const long diag = long((key[i] >> 16) & ((1ull << 40) - 1ull)) - (1l << 33);
						p = uint32_t(long(wi) - diag);
					}
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
