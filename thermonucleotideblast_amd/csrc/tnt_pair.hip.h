// tnt_pair.hip.h -- gfx950 pairing kernel: one lane per (target, assay) group of bind results.
//
// Replaces, given the bind outcome of every culled seed:
//  * PCR / TaqMan: the bind / cull sequence and the O(F x R [x P]) pairing loops of amplicon()
//    (amplicon_search.cpp:131-674) including sort_by_oligo_loc (:12-26), cull_oligo_match
//    (:679-765), the per-oligo de-duplication of bind_oligo_to_*_strand's mask overload
//    (bind_oligo.cpp:806-826, sort_by_bound_match :49-82);
//  * padlock / MIPS: padlock() (padlock_search.cpp:62-361) on top of the hash overload's
//    de-duplication (bind_oligo.cpp:431-453);
//  * probes: hybrid() (probe_search.cpp:67-230).
//
// The reference interleaves binding and culling (bind F, cull, bind R, cull, ...).  Binding a seed
// does not depend on the list it sits in, so every seed that survives the FIRST cull is bound up
// front by the bind kernel and this kernel replays the reference's exact sequence of list
// operations on the results.  Culls only ever remove entries, so the replay sees the same lists.
#ifndef TNT_PAIR_HIP_H
#define TNT_PAIR_HIP_H

#include <hip/hip_runtime.h>
#include "tnt_dev.h"
#include "tnt_hitrec.h"
#include <map>
#include <utility>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <algorithm>
#include <vector>
#endif

namespace tnt {

// The list surgery below (sorts, culls, pairing loops) also runs on the host, for targets too long for one seed work
// unit: their (target, assay) lists are assembled from all windows and replayed whole (tnt_host.hip, search_long_targets).
#define TNT_HD __host__ __device__
#if defined(__HIP_DEVICE_COMPILE__)
#define TNT_SET_ERR(PTR, CODE) atomicOr((PTR), (CODE))
#else
#define TNT_SET_ERR(PTR, CODE) ((void)__atomic_fetch_or((PTR), (CODE), __ATOMIC_RELAXED)) // (the host replay runs on several threads)
#endif

constexpr int PAIR_THREADS = 64;
constexpr int PAIR_HOST = 1 << 20; // entries of one (target, assay) list in the host replay (arrays on 256-MB thread stacks: only the pages a list touches exist)
#ifndef TNT_PAIR_SMALL
#define TNT_PAIR_SMALL 16
#endif
constexpr int PAIR_SMALL = TNT_PAIR_SMALL; // entries of one (target, assay) group handled by the common instance of the pair kernel (its per-lane lists live in
                               // scratch memory: a quarter of the 64-entry instance's, which serves the groups of 17 to 64 entries)
constexpr int PAIR_MAX = 64;
constexpr int PAIR_BIG = 512; // ... and by the instance for crowded groups (short hash words, repeats); == LIST_MAX of the seed kernel

struct PairArgs {
	const Group *groups;
	uint32_t n_groups;
	const uint32_t *active;     // indices of the groups with at least one bound item (filter_kernel)
	const uint32_t *n_active;
	const Item *items;
	const int32_t *result;
	const BoundRec *pool;
	const AssayDev *assays;
	DbView db;
	int max_len;
	int single_primer_pcr;
	int min_max_primer_clamp;
	int assay_format; // TNT_ASSAY_*
	int target_strand;
	Hit *hits;
	uint32_t hit_cap;
	uint32_t *hit_count;
	uint32_t *err;
	const uint32_t *n_big;      // groups of more than PAIR_MAX items seen by the seed kernel
	// seed work units: a hit is reported by the unit that OWNS its first coordinate (windows of a long target overlap;
	// the owner's window holds every seed within reach of the hit, so its lists are complete there)
	const uint32_t *unit_target;
	const int32_t *unit_own_lo, *unit_own_hi;
	int big_enabled;            // the PAIR_BIG instance is launched as well (else a crowded group is an error the host repairs)
	// Replay of the reference's per-target BindCache (row a6; cache_replay_kernel): the groups of the assays that share an
	// oligo with another strand concentration / threshold set are paired by pair_group<N, true>, whose bind steps ask
	// cache_resolve() for every entry in the order the reference evaluates them.
	const Slot *slots;
	const Thresh *thresh;
	const Item *citems;         // the items of those groups, copied side by side ...
	const BindFull *cfull;      // ... and their full bind records (unit-level kernel instance: every field, whatever the status)
	uint32_t *corigin;          // per citem: 0xFFFFFFFF = not evaluated yet, else the citem whose evaluation the memo of its key holds
	BoundRec *cpool;            // bound records as the memo hands them out
	uint32_t cpool_cap;
	uint32_t *cpool_count;
	void *cmemo;                // host replay of windowed targets (tnt_long.hip.h): the target's memo as a map, CacheMemo * (the lists hold 10^5 entries there; null: the evaluated citems of the run are searched)
};

// one group of an assay that goes through the cache replay: where its items were copied to
struct ConflictRec {
	uint32_t group, cfirst, count, pad;
};

enum : uint8_t { MF = 1, MR = 2, MP = 4, MPLUS = 8, MMINUS = 16, MVALID = 32 }; // oligo_info, tntblast.h:147-154

struct Entry {
	int32_t loc5, loc3;
	uint32_t tloc;
	int32_t rec;  // pool index or < 0
	uint8_t mask;
};

// sort_by_oligo_loc, amplicon_search.cpp:12-26 (mixed keys)
TNT_HD inline bool oligo_loc_less(const Entry &x, const Entry &y)
{
	if (!(x.loc5 + x.loc3) || !(y.loc5 + y.loc3)) return x.tloc < y.tloc;
	if (x.loc5 == y.loc5) return x.loc3 < y.loc3;
	return x.loc5 < y.loc5;
}

// sort_by_bound_match, bind_oligo.cpp:49-82 (alignment.size() is 3 * columns + constant)
TNT_HD inline bool bound_less(const Entry &x, const Entry &y, const BoundRec *pool)
{
	if (x.loc5 != y.loc5) return x.loc5 < y.loc5;
	if (x.loc3 != y.loc3) return x.loc3 < y.loc3;
	const BoundRec &a = pool[x.rec], &b = pool[y.rec];
	if (a.tm == b.tm) {
		if (a.mm == b.mm) return a.text_len > b.text_len;
		return a.mm > b.mm;
	}
	return a.tm > b.tm;
}

// oligo_info::operator<, tntblast.h:230-242
TNT_HD inline bool info_less(const Entry &x, const Entry &y, const BoundRec *pool)
{
	if (x.loc5 != y.loc5) return x.loc5 < y.loc5;
	if (x.loc3 != y.loc3) return x.loc3 < y.loc3;
	return pool[x.rec].tm > pool[y.rec].tm;
}

// A (target, assay) list and the working space of its sorts and bind steps: N entries each, in storage the caller owns -- arrays
// of the function's own (scratch memory on the device, the worker's stack on the host) or, for the common instance of the pair
// kernel, the workgroup's LDS (a lane's chain of dependent accesses runs at LDS latency instead of scratch-memory latency).
template <int N> struct ListT {
	Entry *e;
	Entry *scratch;
	int n;
};

// stable insertion sort == std::list::sort for a strict weak order
template <class Less> TNT_HD void stable_sort(Entry *e, int n, Less less)
{
#if !defined(__HIP_DEVICE_COMPILE__)
	if (n > 64) { // host replay of long lists
		std::stable_sort(e, e + n, less);
		return;
	}
#endif
	for (int i = 1; i < n; i++) {
		const Entry x = e[i];
		int j = i - 1;
		while (j >= 0 && less(x, e[j])) {
			e[j + 1] = e[j];
			j--;
		}
		e[j + 1] = x;
	}
}

// std::list::sort as libstdc++ performs it.  sort_by_oligo_loc (amplicon_search.cpp:12-26) compares bound entries by
// loc_5 / loc_3 and anything else by target_loc: not a strict weak order, so WHICH order comes out depends
// on the algorithm, and the cull that follows breaks out of its inner loop on that order (:707).  libstdc++'s
// list::sort is a bottom-up merge sort over a binary counter of runs: the list is cut, in its current order, into
// blocks following the binary digits of n from the most significant one; every block is merged up from single
// elements in aligned halves; the blocks are then merged from the last one leftwards.  A merge takes from the right
// run iff comp(right, left), else from the left one.
template <class Less> TNT_HD void merge_runs(Entry *e, Entry *tmp, int lo, int mid, int hi, Less less)
{
	int a = lo, b = mid, o = lo;
	while (a < mid && b < hi) {
		if (less(e[b], e[a])) tmp[o++] = e[b++];
		else tmp[o++] = e[a++];
	}
	while (a < mid) tmp[o++] = e[a++];
	while (b < hi) tmp[o++] = e[b++];
	for (int i = lo; i < hi; i++) e[i] = tmp[i];
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host replay of long lists (windowed targets: up to a million entries): the same merges in an order that keeps the two
// halves of a block in the cache -- a block is finished before its neighbour is begun, instead of one width after the
// other across the whole list --, and a merge that would hand back its input (no element of the left run is preceded by
// the head of the right run: the loop of merge_runs takes the whole left run first, then the right one) is skipped.
// 96 lists of 165 000 entries (a 250-Mb record against config 2) otherwise stream 1 GB each through every core's memory
// channel: 0.8 s.
// `plain` (optional, oligo_loc_less only): plain[i + 1] - plain[lo + 1] counts, over (lo, i], the entries that are bound or lie
// before their left neighbour in target_loc.  A block without any, whose first entry is unbound too, holds unbound entries in
// ascending target_loc order: there sort_by_oligo_loc is a strict weak order (it compares target_loc) and every merge inside the
// block hands back its input -- the whole subtree is skipped.  The lists of a windowed target arrive in target_loc order and the bind
// steps append what they bound at the end, so nearly all of a 700 000-entry list is such blocks.
template <class Less> void sort_block(Entry *e, Entry *tmp, int lo, int size, Less less, const int *plain = nullptr)
{
	if (size < 2) return;
	if (plain && plain[lo + size] - plain[lo + 1] == 0 && !(e[lo].loc5 + e[lo].loc3)) return;
	const int half = size / 2;
	sort_block(e, tmp, lo, half, less, plain);
	sort_block(e, tmp, lo + half, half, less, plain);
	const Entry &head = e[lo + half];
	int a = lo;
	while (a < lo + half && !less(head, e[a])) a++;
	if (a < lo + half) merge_runs(e, tmp, lo, lo + half, lo + size, less);
}
#endif

template <class Less> TNT_HD void list_sort(Entry *e, Entry *tmp, int n, Less less, bool by_oligo_loc = false)
{
	if (n < 2) return;
#if !defined(__HIP_DEVICE_COMPILE__)
	std::vector<int> plain_prefix;
#ifndef TNT_EXP_NO_PLAIN_SKIP
	if (by_oligo_loc && n >= 4096) {
		plain_prefix.resize(size_t(n) + 1);
		plain_prefix[0] = 0;
		for (int i = 0; i < n; i++)
			plain_prefix[i + 1] = plain_prefix[i] + (((e[i].loc5 + e[i].loc3) != 0 || (i > 0 && e[i].tloc < e[i - 1].tloc)) ? 1 : 0);
	}
#endif
	const int *plain = plain_prefix.empty() ? nullptr : plain_prefix.data();
#endif
	// blocks by the binary digits of n, most significant first; each sorted bottom-up in aligned halves
	int start = 0;
	for (int bit = 30; bit >= 0; bit--) {
		const int size = 1 << bit;
		if (!(n & size)) continue;
#if !defined(__HIP_DEVICE_COMPILE__)
		sort_block(e, tmp, start, size, less, plain);
#else
		for (int width = 1; width < size; width <<= 1)
			for (int lo = start; lo < start + size; lo += 2 * width) merge_runs(e, tmp, lo, lo + width, lo + 2 * width, less);
#endif
		start += size;
	}
	// final merges: the last (smallest) block into its left neighbour, and so on
	int lo_bit = 0;
	while (!(n & (1 << lo_bit))) lo_bit++;
	int right = n - (1 << lo_bit); // start of the merged tail so far
	for (int bit = lo_bit + 1; bit <= 30; bit++) {
		const int size = 1 << bit;
		if (!(n & size)) continue;
		const int left = right - size;
		merge_runs(e, tmp, left, right, n, less);
		right = left;
	}
}

// cull_oligo_match, amplicon_search.cpp:679-765; returns the (minus, plus) "strand counts" the way the
// reference computes them (mask of the element FOLLOWING each valid one, :748-755; end() counts nothing)
template <int N> TNT_HD void cull(ListT<N> &l, int max_len, bool has_probe, bool single_primer, int &cnt_minus, int &cnt_plus)
{
	const uint32_t threshold = uint32_t(max_len) + 50u;
	Entry *const tmp = l.scratch;
	list_sort(l.e, tmp, l.n, [](const Entry &x, const Entry &y) { return oligo_loc_less(x, y); }, true);
	// (no entry carries MVALID here: the flag is taken off again where the survivors are compacted below -- one pass over a
	// 700 000-entry list of a windowed target less)
	for (int f = 0; f < l.n; f++) {
		if (l.e[f].mask & (MPLUS | MP)) continue;
		for (int r = f + 1; r < l.n; r++) {
			if ((l.e[r].tloc - l.e[f].tloc) > threshold) break;
			if (l.e[r].mask & (MMINUS | MP)) continue;
			if (!single_primer && ((l.e[f].mask & (MR | MF)) == (l.e[r].mask & (MR | MF)))) continue;
			if (has_probe) {
				for (int p = f + 1; p < r; p++)
					if (l.e[p].mask & MP) {
						l.e[p].mask |= MVALID;
						l.e[f].mask |= MVALID;
						l.e[r].mask |= MVALID;
					}
			} else {
				l.e[f].mask |= MVALID;
				l.e[r].mask |= MVALID;
			}
		}
	}
	cnt_minus = cnt_plus = 0;
	// The reference advances the iterator first and then reads ->mask, i.e. it looks at the element that
	// FOLLOWS each valid one in the list as it stands (invalid successors are still present at that time).
	int o = 0;
	for (int i = 0; i < l.n; i++) {
		if (!(l.e[i].mask & MVALID)) continue;
		if (i + 1 < l.n) { // (the successor is read before the compaction can have written it: o <= i)
			cnt_minus += (l.e[i + 1].mask & MMINUS) ? 1 : 0;
			cnt_plus += (l.e[i + 1].mask & MPLUS) ? 1 : 0;
		}
		l.e[o] = l.e[i];
		l.e[o++].mask &= uint8_t(~MVALID);
	}
	l.n = o;
}

// bind_oligo_to_{minus,plus}_strand, mask overload: list surgery of bind_oligo.cpp:476-500,806-826
template <int N> TNT_HD void bind_step(ListT<N> &l, uint8_t want, const BoundRec *pool)
{
	Entry *const cur = l.scratch;
	int nc = 0;
	int o = 0;
	for (int i = 0; i < l.n; i++) {
		if ((l.e[i].mask & want) != want) {
			l.e[o++] = l.e[i];
			continue;
		}
		if (l.e[i].rec >= 0) { // bound (unbound seeds are dropped: curr_oligo.pop_front); a bound entry takes its coordinates from the bind result
			cur[nc] = l.e[i];
			cur[nc].loc5 = pool[l.e[i].rec].loc5;
			cur[nc++].loc3 = pool[l.e[i].rec].loc3;
		}
	}
	l.n = o;
	if (nc == 0) return;
	// curr_oligo was filled with push_front: reverse, then stable sort
	for (int i = 0; i < nc / 2; i++) {
		const Entry t = cur[i];
		cur[i] = cur[nc - 1 - i];
		cur[nc - 1 - i] = t;
	}
	stable_sort(cur, nc, [pool](const Entry &x, const Entry &y) { return bound_less(x, y, pool); });
	l.e[l.n++] = cur[0];
	for (int i = 1; i < nc; i++)
		if (l.e[l.n - 1].loc5 != cur[i].loc5 || l.e[l.n - 1].loc3 != cur[i].loc3) l.e[l.n++] = cur[i];
}

struct PairArgs;
struct ConflictRec;
// The groups of one target that go through the cache replay, in assay order (a window of the sorted (target, assay) list).
struct ConflictRun {
	const uint32_t *vals;
	const ConflictRec *recs;
	TNT_HD const ConflictRec &operator[](int g) const { return recs[vals[g]]; }
};
template <int N, bool CACHED = false, bool WORK = false> TNT_HD void pair_group(const PairArgs &a, const Group G, ConflictRun run = ConflictRun{nullptr, nullptr}, int n_run = 0, uint32_t cfirst = 0, Entry *work = nullptr);

TNT_HD inline Hit *alloc_hit(const PairArgs &a)
{
	if (TNT_EXP_PAIR_SKIP & 2) return nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
	// One atomic for all the lanes of the wavefront that stand here together: 32 000 hits per step on one counter is 0.4 ms of
	// serialised atomics otherwise (a hot word takes some 90 of them per microsecond), and behind pair_select_kernel most lanes of
	// a wavefront do report a hit.
	const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
	const int leader = __builtin_ctzll(here);
	const int lane = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
	uint32_t first = 0;
	if (lane == leader) first = atomicAdd(a.hit_count, uint32_t(__builtin_popcountll(here)));
	first = uint32_t(__builtin_amdgcn_readlane(int(first), leader));
	const uint32_t o = first + uint32_t(__builtin_popcountll(here & ((1ull << lane) - 1ull)));
#else
	const uint32_t o = __atomic_fetch_add(a.hit_count, 1u, __ATOMIC_RELAXED); // (the host replay runs on several threads)
#endif
	if (o >= a.hit_cap) {
		TNT_SET_ERR(a.err, ERR_HIT_OVERFLOW);
		return nullptr;
	}
	return a.hits + o;
}

// After phase A (first strand of the PCR groups, everything of the probe / padlock groups) the bind
// kernel has put every group with a bound item on the active list.  A PCR group that is not on it is dead:
// the reference's list is empty after the second cull (amplicon_search.cpp:151-179 / :238-266) and its
// phase-B seeds are never bound.  This kernel queues the phase-B items of the groups that are alive.
struct FilterArgs {
	const Group *groups;
	const Item *items;
	const uint32_t *active;
	const uint32_t *n_active;
	uint32_t *qb;
	uint32_t *qb_count;
};

__global__ __launch_bounds__(256) void filter_kernel(FilterArgs a)
{
	const uint32_t n = *a.n_active;
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
		const Group G = a.groups[a.active[x]];
		const uint32_t count = G.count & ~GROUP_ALIVE;
		uint32_t nb = 0;
		for (uint32_t i = 0; i < count; i++) nb += (a.items[G.first + i].info & ITEM_PHASE_B) ? 1u : 0u;
		if (nb) {
			uint32_t o = atomicAdd(a.qb_count, nb);
			for (uint32_t i = 0; i < count; i++)
				if (a.items[G.first + i].info & ITEM_PHASE_B) a.qb[o++] = G.first + i;
		}
	}
}

// After phase A: the items that repeat another item's bind (shared oligos, see SeedArgs::slot_canon) take its result
// -- the pool record is shared, nothing downstream writes to it -- and, like a bind of their own would, put their group
// on the active list.
struct PropagateArgs {
	const uint32_t *src;          // the phase-A queue (sharded, as the seed kernel wrote it)
	const uint32_t *shard_counts;
	uint32_t shard_region;
	uint32_t cap;
	const Item *items;
	const uint32_t *item_rep;
	int32_t *result;
	Group *groups;
	uint32_t *active;
	uint32_t *n_active;
	uint32_t *n_dups; // statistics: binds saved
};

__global__ __launch_bounds__(256) void propagate_kernel(PropagateArgs a)
{
	__shared__ uint32_t s_pre[N_SHARD + 1];
	__shared__ uint32_t s_dups;
	shard_prefix(a.shard_counts, s_pre, threadIdx.x);
	if (threadIdx.x == 0) s_dups = 0;
	__syncthreads();
	const uint32_t n_all = s_pre[N_SHARD];
	const uint32_t n = n_all < a.cap ? n_all : a.cap;
	uint32_t mine = 0;
	for (uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x; lin < n; lin += gridDim.x * blockDim.x) {
		const uint32_t idx = a.src[shard_locate(s_pre, lin, a.shard_region)];
		const uint32_t rep = a.item_rep[idx];
		if (rep == 0xFFFFFFFFu) continue;
		mine++;
		const int32_t r = a.result[rep];
		a.result[idx] = r;
		if (r >= 0) {
			const uint32_t g = a.items[idx].group;
			if (!(atomicOr(&a.groups[g].count, GROUP_ALIVE) & GROUP_ALIVE)) a.active[atomicAdd(a.n_active, 1u)] = g;
		}
	}
	if (mine) atomicAdd(&s_dups, mine);
	__syncthreads();
	if (threadIdx.x == 0 && s_dups) atomicAdd(a.n_dups, s_dups);
}

// Distribute a queue of item indices over the length buckets of their oligos: every bucket is bound by the
// kernel instance with exactly that many rows (no padding rows, uniform trip counts inside a wavefront).
// Two passes: count (bucket sizes), scatter (exclusive offsets -> one compact array, bucket after bucket).
struct BucketArgs {
	const uint32_t *src;          // item indices
	const uint32_t *src_count;    // flat source: number of entries
	const uint32_t *shard_counts; // or: sharded source (seed kernel output) of `shard_region` entries per shard
	uint32_t shard_region;
	const Item *items;
	const Slot *slots;
	const int32_t *olig_len;
	DbView db;                    // exception lists: windows with a degenerate target base go to Slot::bucket_general
	int has_exceptions;
	uint32_t *dst;                // compact output, `cap` entries
	uint32_t cap;
	uint32_t *counters;           // per bucket BUCKET_STRIDE words: [0] size, [1] scatter cursor, [2] first entry in dst
	int n_buckets;
	uint32_t merge_below;         // a queue shorter than this goes to the last (longest) bucket as a whole: one launch
	                              // with padding rows beats one under-filled launch per bucket
	uint32_t *err;
	const uint32_t *item_rep;     // optional: items that take the result of another item are not queued (see SeedArgs)
	unsigned long long *cells;    // optional statistics: DP cells (rows x window columns) of the queued items, added by the scatter pass
};

constexpr int BUCKET_PER_THREAD = 16; // (one global atomic per bucket and 4 096 items: a hot counter sustains only ~88 atomics per microsecond)

template <bool SCATTER> __global__ __launch_bounds__(256) void bucket_kernel(BucketArgs a)
{
	__shared__ uint32_t s_pre[N_SHARD + 1];
	__shared__ uint32_t s_cnt[MAX_BUCKETS], s_base[MAX_BUCKETS], s_first[MAX_BUCKETS];
	const int tid = threadIdx.x;
	if (a.shard_counts) shard_prefix(a.shard_counts, s_pre, tid);
	if (SCATTER && tid == 0) {
		uint32_t run = 0;
		for (int b = 0; b < a.n_buckets; b++) {
			s_first[b] = run;
			if (blockIdx.x == 0) a.counters[b * BUCKET_STRIDE + 2] = run;
			run += a.counters[b * BUCKET_STRIDE];
		}
	}
	__syncthreads();
	const uint32_t n_all = a.shard_counts ? s_pre[N_SHARD] : *a.src_count;
	const uint32_t n = n_all < a.cap ? n_all : a.cap;
	if (n_all > a.cap && blockIdx.x == 0 && tid == 0) atomicOr(a.err, ERR_ITEM_OVERFLOW);
	constexpr uint32_t STEP = 256 * BUCKET_PER_THREAD;
	unsigned long long my_cells = 0;
	for (uint32_t base = blockIdx.x * STEP; base < n; base += gridDim.x * STEP) {
		if (tid < MAX_BUCKETS) s_cnt[tid] = 0;
		__syncthreads();
		uint32_t idx[BUCKET_PER_THREAD], bkt[BUCKET_PER_THREAD], pos[BUCKET_PER_THREAD];
#pragma unroll
		for (int m = 0; m < BUCKET_PER_THREAD; m++) {
			const uint32_t lin = base + uint32_t(m) * 256u + uint32_t(tid);
			bkt[m] = 0xFFFFFFFFu;
			if (lin < n) {
				idx[m] = a.src[a.shard_counts ? shard_locate(s_pre, lin, a.shard_region) : lin];
				if (a.item_rep && a.item_rep[idx[m]] != 0xFFFFFFFFu) continue; // bound through its representative
				const Item it = a.items[idx[m]];
				const Slot sl = a.slots[it.info >> ITEM_SLOT_SHIFT];
				bool degenerate = false;
				if (a.has_exceptions) { // same window as the bind kernel, bind_oligo.cpp:502-505
					const uint32_t e0 = a.db.exc_off[it.target], e1 = a.db.exc_off[it.target + 1];
					if (e1 > e0) {
						int start = int(it.tloc) - (int(it.info & 127u) + FLANK);
						if (start < 0) start = 0;
						const uint32_t stop = uint32_t(start + a.olig_len[sl.oligo] + 2 * FLANK);
						uint32_t lo = e0, hi = e1;
						while (lo < hi) {
							const uint32_t mid = (lo + hi) >> 1;
							if (a.db.exc_pos[mid] < uint32_t(start)) lo = mid + 1;
							else hi = mid;
						}
						degenerate = lo < e1 && a.db.exc_pos[lo] < stop;
					}
				}
				bkt[m] = n < a.merge_below ? uint32_t(a.n_buckets - 1) : uint32_t(degenerate ? sl.bucket_general : sl.bucket);
				pos[m] = atomicAdd(&s_cnt[bkt[m]], 1u);
				if (SCATTER && a.cells) { // rows x columns of the window, bind_oligo.cpp:502-505
					const int L = a.olig_len[sl.oligo];
					int start = int(it.tloc) - (int(it.info & 127u) + sl.qloc_bias + FLANK);
					if (start < 0) start = 0;
					int stop = start + L + 2 * FLANK;
					const int tlen = int(a.db.len[it.target]);
					if (stop > tlen) stop = tlen;
					my_cells += (unsigned long long)(L) * (unsigned long long)(stop > start ? stop - start : 0);
				}
			}
		}
		__syncthreads();
		if (tid < a.n_buckets && s_cnt[tid]) s_base[tid] = atomicAdd(a.counters + tid * BUCKET_STRIDE + (SCATTER ? 1 : 0), s_cnt[tid]);
		__syncthreads();
		if (SCATTER) {
#pragma unroll
			for (int m = 0; m < BUCKET_PER_THREAD; m++)
				if (bkt[m] != 0xFFFFFFFFu) a.dst[s_first[bkt[m]] + s_base[bkt[m]] + pos[m]] = idx[m];
			__syncthreads();
		}
	}
	if (SCATTER && a.cells) { // one atomic per wavefront
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) my_cells += (unsigned long long)__shfl_xor((long long)my_cells, o);
		if ((tid & 63) == 0 && my_cells) atomicAdd(a.cells, my_cells);
	}
}

typedef std::map<std::pair<unsigned long long, unsigned long long>, uint32_t> CacheMemo; // host replay: (oligo, strand), window -> the citem evaluated first
// One request to the reference's BindCache (tntblast.h:247-324; bind_oligo.cpp:508-512 miss, :750-803 / :396-428 hit).  The
// entry stands for citem ci = -2 - rec.  Its key is (oligo string, window, strand); the memo of a key is the FIRST
// evaluation of it on this target -- whatever strand concentration and threshold set asked --, holding what that caller's
// threshold cascade had looked at when it stopped (the rest zero / empty).  A later request applies ITS thresholds to the
// memo.  `run` = the target's groups of such assays up to and including the current one, in assay order: the citems
// evaluated so far are found there (corigin != 0xFFFFFFFF).
TNT_HD inline void cache_resolve(const PairArgs &a, Entry &e, const ConflictRun run, int n_run)
{
	if (e.rec >= -1) return;
	const uint32_t ci = uint32_t(-2 - e.rec);
	const Item it = a.citems[ci];
	const Slot sl = a.slots[it.info >> ITEM_SLOT_SHIFT];
	const BindFull &own = a.cfull[ci];
	uint32_t origin = ci;
#if !defined(__HIP_DEVICE_COMPILE__)
	if (a.cmemo) { // every evaluated citem of a key has the key's first evaluation as its origin: the map holds exactly that
		CacheMemo &memo = *static_cast<CacheMemo *>(a.cmemo);
		const std::pair<unsigned long long, unsigned long long> key((unsigned long long)(uint32_t(sl.oligo)) << 1 | ((it.info & ITEM_MINUS) ? 1u : 0u),
		                                                             ((unsigned long long)uint32_t(own.target_start) << 32) | uint32_t(own.target_stop));
		origin = memo.emplace(key, ci).first->second;
	} else
#endif
	for (int g = 0; g < n_run && origin == ci; g++)
		for (uint32_t x = run[g].cfirst; x < run[g].cfirst + run[g].count; x++) {
			if (x == ci || a.corigin[x] == 0xFFFFFFFFu) continue;
			const Item ix = a.citems[x];
			if ((ix.info ^ it.info) & ITEM_MINUS) continue; // one memo per strand (tntblast_local.cpp:371-372)
			if (a.slots[ix.info >> ITEM_SLOT_SHIFT].oligo != sl.oligo) continue;
			const BindFull &fx = a.cfull[x];
			if (fx.target_start != own.target_start || fx.target_stop != own.target_stop) continue;
			origin = a.corigin[x];
			break;
		}
	a.corigin[ci] = origin;
	const BindFull &U = a.cfull[origin];
	const Thresh th = a.thresh[sl.thresh];
	int status = own.status;
	const int s1 = U.status; // where the first caller's cascade stopped: 0 bound, 1 Tm, 2 dG, 3 / 4 anchors, 5 mismatches, 6 gaps, 7 poly-degenerate, 8 no alignment
	const bool full = s1 == 0;
	const float tm = U.tm, dg = s1 == 1 ? 0.0f : U.dg;
	const int a5 = (full || (s1 >= 3 && s1 <= 7)) ? U.a5 : 0, a3 = (full || (s1 >= 4 && s1 <= 7)) ? U.a3 : 0;
	const int mm = (full || (s1 >= 5 && s1 <= 7)) ? U.mm : 0, gap = (full || (s1 >= 6 && s1 <= 7)) ? U.gap : 0, poly = (full || s1 == 7) ? U.poly : 0;
	if (origin != ci) { // a hit: the caller's thresholds against the memo
		status = 0;
		if (tm < th.min_tm || tm > th.max_tm) status = 1;
		else if (dg < th.min_dg || dg > th.max_dg) status = 2;
		else if (s1 == 8) status = 8;
		else if (a5 < th.clamp5 || a3 < th.clamp3) status = 3;
		else if (mm > th.max_mm) status = 5;
		else if (gap > th.max_gap) status = 6;
		else if (poly > th.max_poly) status = 7;
	}
	if (status != 0) {
		e.rec = -1;
		return;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t at = atomicAdd(a.cpool_count, 1u);
#else
	const uint32_t at = __atomic_fetch_add(a.cpool_count, 1u, __ATOMIC_RELAXED);
#endif
	if (at >= a.cpool_cap) {
		TNT_SET_ERR(a.err, ERR_POOL_OVERFLOW);
		e.rec = -1;
		return;
	}
	BoundRec &r = a.cpool[at];
	r.loc5 = full ? U.loc5 : 0, r.loc3 = full ? U.loc3 : 0;
	r.tm = tm, r.dH = full ? U.dH : 0.0f, r.dS = full ? U.dS : 0.0f;
	r.a5 = a5, r.a3 = a3, r.mm = mm, r.gap = gap;
	int n = 0;
	if (full)
		for (; n < ALIGN_TEXT - 1 && U.text[n]; n++) r.text[n] = U.text[n];
	for (int z = n; z < (n / 4 + 1) * 4; z++) r.text[z] = 0; // (zeros to the end of the dword, as the bind kernel leaves its texts: copy_text moves dwords)
	r.text_len = n;
	// a text that lives in the long-text arena: the field holds 0x7F 'L' 'T', eight hex digits of offset, six of length (bind_tail).
	// sort_by_bound_match compares the text's OWN length where Tm and the mismatch count tie (bound_less), so take it from there
	if (n == LTEXT_REF_CHARS && uint8_t(r.text[0]) == 0x7Fu && r.text[1] == 'L' && r.text[2] == 'T') {
		int len = 0;
		for (int d = 0; d < 6; d++) {
			const int ch = r.text[11 + d];
			len = len * 16 + (ch >= 'a' ? ch - 'a' + 10 : ch - '0');
		}
		r.text_len = len;
	}
	e.rec = int32_t(at);
	e.loc5 = r.loc5, e.loc3 = r.loc3;
}

// N = PAIR_MAX: the groups of up to PAIR_MAX items (all of them in practice); N = PAIR_BIG: the crowded rest.  The
// per-lane lists live in scratch memory, so the common instance keeps them small.
constexpr int PAIR_WORK_STRIDE = 2 * PAIR_SMALL * int(sizeof(Entry) / 4) + 1; // dwords per lane: an odd stride, so that the lanes' entries of one index fall on different banks
template <int N> __global__ __launch_bounds__(PAIR_THREADS) void pair_kernel(PairArgs a)
{
	static_assert(sizeof(Entry) % 4 == 0 && alignof(Entry) <= 4, "entries are whole dwords");
	__shared__ uint32_t s_work[N == PAIR_SMALL ? PAIR_THREADS * PAIR_WORK_STRIDE : 1];
	if (N == PAIR_BIG && *a.n_big == 0) return;
	const uint32_t n_active = *a.n_active;
	for (uint32_t gi = blockIdx.x * PAIR_THREADS + threadIdx.x; gi < n_active; gi += gridDim.x * PAIR_THREADS) {
		const Group G = a.groups[a.active[gi]];
		if (a.citems && (a.assays[G.assay & GROUP_ASSAY_MASK].cache_replay & 1)) continue; // paired by cache_replay_kernel
		const uint32_t count = G.count & ~GROUP_ALIVE;
		// every group is paired by the smallest instance that holds it
		const int fits = count <= uint32_t(PAIR_SMALL) ? PAIR_SMALL : (count <= uint32_t(PAIR_MAX) ? PAIR_MAX : PAIR_BIG);
		if (fits != N) {
			if (N == PAIR_MAX && fits == PAIR_BIG && !a.big_enabled) atomicOr(a.err, ERR_LIST_OVERFLOW);
			continue;
		}
#ifndef TNT_PAIR_NO_LDS
		if constexpr (N == PAIR_SMALL) pair_group<N, false, true>(a, G, ConflictRun{nullptr, nullptr}, 0, 0, reinterpret_cast<Entry *>(s_work + threadIdx.x * PAIR_WORK_STRIDE));
		else
#endif
			pair_group<N>(a, G);
	}
}

// CACHED: the group belongs to an assay that goes through the replay of the reference's BindCache: its items are
// a.citems[cfirst ...], nothing is known about them until a bind step asks (cache_resolve), records come from a.cpool.
// WORK: the list and its working space live in `work` (2 * N entries handed in by the caller: the LDS of pair_kernel<PAIR_SMALL>).
template <int N, bool CACHED, bool WORK> TNT_HD void pair_group(const PairArgs &a, const Group G, const ConflictRun run, int n_run, uint32_t cfirst, Entry *work)
{
	const AssayDev A = a.assays[G.assay & GROUP_ASSAY_MASK];
	const int seq_index = int(a.db.first_seq_index + a.unit_target[G.target]);
	const int own_lo = a.unit_own_lo[G.target], own_hi = a.unit_own_hi[G.target];
	const BoundRec *pool = CACHED ? a.cpool : a.pool;

	const uint32_t g_count = G.count & ~GROUP_ALIVE;
	if (g_count > uint32_t(N)) {
		TNT_SET_ERR(a.err, ERR_LIST_OVERFLOW);
		return;
	}

	// The seed kernel emits the items of a group in arbitrary order; the reference's list order (the order of the
	// seed keys: target_loc, assembly group, word index for amplicon(); list, diagonal, word index for the hash
	// overload) is restored here, where only the groups that bound something arrive.
	Entry own[WORK ? 1 : 2 * N];
	ListT<N> l;
	l.e = WORK ? work : own;
	l.scratch = l.e + N;
	l.n = int(g_count);
	bool any_bound = false;
	uint32_t bound_classes = 0; // 1: a minus-strand primer is bound, 2: a plus-strand primer, 4: a probe
	uint32_t bound_arms = 0;
	unsigned long long order[N];
	for (int i = 0; i < l.n; i++) {
		const Item it = CACHED ? a.citems[cfirst + uint32_t(i)] : a.items[G.first + i];
		const int32_t res = CACHED ? -2 - int32_t(cfirst + uint32_t(i)) : a.result[G.first + i];
		{
			// (word index over the whole oligo: the seeds of an oligo beyond 128 bases come from segments of 128 words, Slot::qloc_bias)
			const uint32_t role = (it.info >> ITEM_SLOT_SHIFT) % 3u, wi = uint32_t(int32_t(it.info & 127u) + a.slots[it.info >> ITEM_SLOT_SHIFT].qloc_bias);
			const bool minus = (it.info & ITEM_MINUS) != 0;
			unsigned long long key;
			if (A.kind == 0) {
				const unsigned long long rank = role == 2 ? (minus ? 4u : 5u) : ((minus ? 0u : 2u) + role); // F-, R-, F+, R+, P-, P+
				key = ((unsigned long long)it.tloc << 28) | (rank << 25) | ((unsigned long long)wi << 14);
			} else {
				const unsigned long long listid = (minus ? 0u : 2u) + (role == 1 ? 0u : 1u);
				key = (listid << 58) | ((unsigned long long)(long(wi) - long(it.tloc) + (1l << 33)) << 24) | ((unsigned long long)wi << 13);
			}
			order[i] = key;
		}
		Entry &e = l.e[i];
		e.tloc = it.tloc;
		// a probe seeded in segments (oligos beyond 128 bases): the entry carries its diagonal (query word - target position, over
		// the whole oligo) to the hit, where the merge of the segments' hits needs it (the hash-overload lists never look at tloc)
		if (A.kind == 2 && (A.cache_replay & 2)) e.tloc = uint32_t(int32_t(it.info & 127u) + a.slots[it.info >> ITEM_SLOT_SHIFT].qloc_bias - int32_t(it.tloc));
		e.rec = CACHED ? res : (res >= 0 ? res : -1);
		e.loc5 = e.loc3 = 0;
		const uint32_t role = (it.info >> ITEM_SLOT_SHIFT) % 3u;
		e.mask = uint8_t((role == 0 ? MF : (role == 1 ? MR : MP)) | ((it.info & ITEM_MINUS) ? MMINUS : MPLUS));
		if (res >= 0) {
			any_bound = true;
			bound_classes |= uint32_t(role == 2 ? 4u : ((it.info & ITEM_MINUS) ? 1u : 2u));
			bound_arms |= 1u << (((it.info & ITEM_MINUS) ? 0u : 2u) + (role == 1 ? 0u : 1u)); // the hash overload's lists: minus-up, minus-down, plus-up, plus-down
		}
	}
	if (!any_bound && !CACHED) return;
	// amplicon() reports a pair of a bound minus-strand primer and a bound plus-strand primer (with a bound probe between them in
	// a TaqMan assay), amplicon_search.cpp:356-420: a group without one of them ends in one of the early returns below or in an
	// empty pairing loop -- nine in ten of the groups that arrive here (one strand bound, nothing on the other): they leave
	// before the list is put in order.
	if (!CACHED && A.kind == 0 && ((bound_classes & 3u) != 3u || (A.has_probe && !(bound_classes & 4u)))) return;
	// padlock() pairs an upstream with a downstream arm of the same strand (padlock_search.cpp:62-361)
	if (!CACHED && A.kind == 1 && (bound_arms & 3u) != 3u && (bound_arms & 12u) != 12u) return;
	if (TNT_EXP_PAIR_SKIP & 4) return;
#if !defined(__HIP_DEVICE_COMPILE__)
	bool in_order = true; // (the lists of windowed targets arrive in list order: tnt_sort.hip)
	for (int i = 1; i < l.n && in_order; i++) in_order = order[i - 1] < order[i];
	if (in_order) {
	} else if (l.n > 64) { // host replay of long lists: the same order (keys are unique), not quadratic
		unsigned long long *idx = new unsigned long long[size_t(l.n)];
		Entry *cp = new Entry[size_t(l.n)];
		for (int i = 0; i < l.n; i++) idx[i] = uint32_t(i), cp[i] = l.e[i];
		std::sort(idx, idx + l.n, [&](unsigned long long x, unsigned long long y) { return order[x] < order[y]; });
		for (int i = 0; i < l.n; i++) l.e[i] = cp[idx[i]];
		delete[] idx;
		delete[] cp;
	} else
#endif
	for (int i = 1; i < l.n; i++) { // insertion sort by key (keys are unique)
		const Entry x = l.e[i];
		const unsigned long long kx = order[i];
		int j = i - 1;
		while (j >= 0 && order[j] > kx) {
			l.e[j + 1] = l.e[j];
			order[j + 1] = order[j];
			j--;
		}
		l.e[j + 1] = x;
		order[j + 1] = kx;
	}

	if (A.kind == 0) {
		// ---------------------------------------------------------------- amplicon(), amplicon_search.cpp:128-674
		const bool has_probe = A.has_probe != 0;
		const bool single = a.single_primer_pcr != 0;
		// strand counts of the first cull were taken by the seed kernel on the full list (:128-131)
		const bool minus_first = (G.assay & GROUP_MINUS_FIRST) != 0;
		int d0, d1;
		for (int s = 0; s < 4; s++) {
			const bool fwd = (s % 2) == 0;
			const bool minus = minus_first ? (s < 2) : (s >= 2);
			const uint8_t want = uint8_t((fwd ? MF : MR) | (minus ? MMINUS : MPLUS));
			if (CACHED) // the entries of this oligo and strand meet the memo in list order (bind_oligo.cpp:476-512)
				for (int i = 0; i < l.n; i++)
					if ((l.e[i].mask & want) == want) cache_resolve(a, l.e[i], run, n_run);
			bind_step(l, want, pool);
			if (s < 3) {
				cull(l, a.max_len, has_probe, single, d0, d1);
				if (l.n == 0 && !(minus_first && s == 2)) return;
			}
		}
		if (has_probe) {
			cull(l, a.max_len, has_probe, single, d0, d1);
			if (l.n == 0) return;
			for (int pass = 0; pass < 2; pass++) {
				const uint8_t want = uint8_t(MP | (pass == 0 ? MMINUS : MPLUS));
				if (CACHED)
					for (int i = 0; i < l.n; i++)
						if ((l.e[i].mask & want) == want) cache_resolve(a, l.e[i], run, n_run);
				bind_step(l, want, pool);
			}
		}
		list_sort(l.e, l.scratch, l.n, [](const Entry &x, const Entry &y) { return oligo_loc_less(x, y); }, true); // :356

		const bool apply_mm = a.min_max_primer_clamp >= 0;
		const uint32_t min_max_clamp = apply_mm ? uint32_t(a.min_max_primer_clamp) : 0u;
		// The reference walks every pair of the list (amplicon_search.cpp:356-420: `continue` on an amplicon beyond max_len, no
		// break): quadratic in the bound oligos, 0.3 s for one list of a 250-Mb record.  Where the list is in ascending loc5
		// order -- it is, behind the sort above, unless a footprint's coordinates sum to zero and sort_by_oligo_loc falls back to
		// target_loc -- every R behind the first one whose loc5 already puts the amplicon beyond max_len does so too:
		// R.loc3 >= R.loc5 - slack with slack = the largest (loc5 - loc3) of the list, so the walk stops there.  Same pairs, same order.
		bool ascending = true;
		long slack = -(1l << 40);
		for (int i = 0; i < l.n; i++) {
			if (i > 0 && l.e[i].loc5 < l.e[i - 1].loc5) ascending = false;
			const long d = long(l.e[i].loc5) - long(l.e[i].loc3);
			slack = d > slack ? d : slack;
		}
		for (int f = 0; f < l.n; f++) {
			const Entry &F = l.e[f];
			if (F.mask & (MPLUS | MP)) continue;
			for (int r = f + 1; r < l.n; r++) {
				const Entry &R = l.e[r];
#ifndef TNT_EXP_FULL_WALK // (tests/native/host_pair_walk_check.hip holds the bounded walks against the reference's walk over every pair)
				if (ascending && long(R.loc5) - slack - long(F.loc5) + 1 > long(a.max_len)) break;
#endif
				if (R.mask & (MMINUS | MP)) continue;
				if (!single && ((F.mask & (MR | MF)) == (R.mask & (MR | MF)))) continue;
				if (F.loc3 >= R.loc5) continue;
				if ((R.loc3 - F.loc5 + 1) > a.max_len) continue;
				const BoundRec &bf = pool[F.rec], &br = pool[R.rec];
				if (apply_mm && uint32_t(bf.a3 > br.a3 ? bf.a3 : br.a3) <= min_max_clamp) continue;
				if (F.loc5 < own_lo || F.loc5 >= own_hi) continue; // reported by the window that owns the amplicon start

				Hit h;
				init_hit(h, A, seq_index);
				if ((F.mask & MR) && (R.mask & MR)) h.forward_oligo = 1; // :453-457
				if ((F.mask & MF) && (R.mask & MF)) h.reverse_oligo = 0; // :459-463
				h.primer_strand = (F.mask & MF) ? 0 : 1;
				h.amplicon_start = F.loc5;
				h.amplicon_stop = R.loc3;
				const bool swap = (F.mask & MR) && (R.mask & MF); // :478-483
				const BoundRec &fo = swap ? br : bf, &ro = swap ? bf : br;
				h.forward_tm = fo.tm, h.reverse_tm = ro.tm;
				h.forward_dH = fo.dH, h.reverse_dH = ro.dH;
				h.forward_dS = fo.dS, h.reverse_dS = ro.dS;
				h.forward_mm = wrap8(fo.mm), h.reverse_mm = wrap8(ro.mm);
				h.forward_gap = wrap8(fo.gap), h.reverse_gap = wrap8(ro.gap);
				h.forward_primer_clamp = wrap8(fo.a3), h.reverse_primer_clamp = wrap8(ro.a3);

				if (has_probe) {
					for (int p = f + 1; p < r; p++) {
						const Entry &P = l.e[p];
						if (!(P.mask & MP)) continue;
						if (!(P.loc5 >= F.loc5 && P.loc3 <= R.loc3)) continue; // :418
						if ((P.mask & (MPLUS | MMINUS)) == (F.mask & (MPLUS | MMINUS))) {
							if (P.loc5 <= F.loc3) continue;
						} else {
							if (P.loc3 >= R.loc5) continue;
						}
						Hit *out = alloc_hit(a);
						if (!out) return;
						const BoundRec &bp = pool[P.rec];
						*out = h;
						out->probe_start = P.loc5, out->probe_stop = P.loc3;
						out->probe_tm = bp.tm, out->probe_dH = bp.dH, out->probe_dS = bp.dS;
						out->probe_mm = wrap8(bp.mm), out->probe_gap = wrap8(bp.gap);
						out->probe_strand = (P.mask & MPLUS) ? 0 : 1;
						copy_text(out->forward_align, fo);
						copy_text(out->reverse_align, ro);
						copy_text(out->probe_align, bp);
					}
				} else {
					Hit *out = alloc_hit(a);
					if (!out) return;
					*out = h;
					copy_text(out->forward_align, fo);
					copy_text(out->reverse_align, ro);
				}
			}
		}
		return;
	}

	// -------------------------------------------------------------------- hash-overload assays
	// items arrive per list (minus-up, minus-down, plus-up, plus-down) in diagonal order; each list keeps
	// its bound entries, sorted by oligo_info::operator< (stable) and de-duplicated by (loc5, loc3)
	// (bind_oligo.cpp:431-453)
	Entry lists[4][N];
	int ln[4] = {0, 0, 0, 0};
	if (CACHED) // the hash overload evaluates every seed of a list, list after list, in diagonal order (bind_oligo.cpp:124-175)
		for (int i = 0; i < l.n; i++) cache_resolve(a, l.e[i], run, n_run);
	for (int i = 0; i < l.n; i++) {
		Entry e = l.e[i];
		if (e.rec < 0) continue;
		e.loc5 = pool[e.rec].loc5;
		e.loc3 = pool[e.rec].loc3;
		const int id = ((e.mask & MMINUS) ? 0 : 2) + ((e.mask & MR) ? 0 : 1);
		lists[id][ln[id]++] = e;
	}
	for (int id = 0; id < 4; id++) {
		stable_sort(lists[id], ln[id], [pool](const Entry &x, const Entry &y) { return info_less(x, y, pool); });
		int o = 0;
		for (int i = 0; i < ln[id]; i++)
			if (o == 0 || lists[id][o - 1].loc5 != lists[id][i].loc5 || lists[id][o - 1].loc3 != lists[id][i].loc3)
				lists[id][o++] = lists[id][i];
		ln[id] = o;
	}

	if (A.kind == 2) { // hybrid(), probe_search.cpp:67-230
		for (int pass = 0; pass < 2; pass++) {
			const int id = pass == 0 ? 1 : 3;
			for (int i = 0; i < ln[id]; i++) {
				const Entry &e = lists[id][i];
				if (e.loc5 < own_lo || e.loc5 >= own_hi) continue;
				const BoundRec &b = pool[e.rec];
				Hit *out = alloc_hit(a);
				if (!out) return;
				init_hit(*out, A, seq_index);
				out->probe_tm = b.tm, out->probe_dH = b.dH, out->probe_dS = b.dS;
				out->probe_mm = wrap8(b.mm), out->probe_gap = wrap8(b.gap);
				out->probe_start = e.loc5, out->probe_stop = e.loc3;
				out->probe_strand = pass == 0 ? 1 : 0;
				if (A.cache_replay & 2) out->amplicon_start = int32_t(e.tloc); // (the diagonal: read and cleared by the merge of the segments' hits)
				copy_text(out->probe_align, b);
			}
		}
		return;
	}

	// padlock(), padlock_search.cpp:62-361: upstream = reverse oligo, downstream = forward oligo
	const int max_len = (a.assay_format == 3 /* MIPS */) ? a.max_len : 0; // tntblast_local.cpp:592,606
	for (int pass = 0; pass < 2; pass++) {
		const bool minus = pass == 0;
		const int up = minus ? 0 : 2, down = up + 1;
		// The reference tries every (upstream, downstream) pair (padlock_search.cpp:230-361): quadratic in the bound arms of a
		// long record.  The downstream list is in ascending loc5 order (oligo_info::operator< is a strict weak order with loc5
		// first), and a pair needs  0 <= len <= max_len  with len = U.loc5 - D.loc3 - 1 (minus) or D.loc5 - U.loc3 - 1 (plus):
		// that bounds D.loc5 from both sides -- through the smallest and largest footprint (loc3 - loc5) of the list on the minus
		// strand -- so only the stretch of the list between those bounds is walked: the same pairs in the same order.
		long span_lo = 0, span_hi = 0;
		for (int d = 0; d < ln[down]; d++) {
			const long sp = long(lists[down][d].loc3) - long(lists[down][d].loc5);
			span_lo = (d == 0 || sp < span_lo) ? sp : span_lo;
			span_hi = (d == 0 || sp > span_hi) ? sp : span_hi;
		}
		for (int u = 0; u < ln[up]; u++) {
			const Entry &U = lists[up][u];
			const long d5_lo = minus ? long(U.loc5) - 1 - long(max_len) - span_hi : long(U.loc3) + 1;
			const long d5_hi = minus ? long(U.loc5) - 1 - span_lo : long(U.loc3) + 1 + long(max_len);
			int d = 0;
#ifndef TNT_EXP_FULL_WALK
			if (ln[down] > 16) { // first entry with loc5 >= d5_lo
				int lo = 0, hi = ln[down];
				while (lo < hi) {
					const int mid = (lo + hi) >> 1;
					if (long(lists[down][mid].loc5) < d5_lo) lo = mid + 1;
					else hi = mid;
				}
				d = lo;
			}
#endif
			for (; d < ln[down]; d++) {
				const Entry &D = lists[down][d];
#ifndef TNT_EXP_FULL_WALK
				if (long(D.loc5) > d5_hi) break;
#endif
				const int len = minus ? (U.loc5 - D.loc3 - 1) : (D.loc5 - U.loc3 - 1);
				if (len < 0 || len > max_len) continue;
				{
					const int first = minus ? D.loc5 : U.loc5;
					if (first < own_lo || first >= own_hi) continue;
				}
				const BoundRec &bu = pool[U.rec], &bd = pool[D.rec];
				Hit *out = alloc_hit(a);
				if (!out) return;
				init_hit(*out, A, seq_index);
				out->primer_strand = minus ? 1 : 0;
				out->amplicon_start = minus ? D.loc5 : U.loc5;
				out->amplicon_stop = minus ? U.loc3 : D.loc3;
				out->forward_tm = bd.tm, out->reverse_tm = bu.tm;
				out->forward_dH = bd.dH, out->reverse_dH = bu.dH;
				out->forward_dS = bd.dS, out->reverse_dS = bu.dS;
				out->forward_mm = wrap8(bd.mm), out->reverse_mm = wrap8(bu.mm);
				out->forward_gap = wrap8(bd.gap), out->reverse_gap = wrap8(bu.gap);
				out->forward_primer_clamp = wrap8(bd.a3);
				out->reverse_primer_clamp = wrap8(bu.a5);
				copy_text(out->forward_align, bd);
				copy_text(out->reverse_align, bu);
			}
		}
	}
}

// Which of the groups that bound something can report anything at all?  amplicon() needs a bound primer on either strand (and a
// bound probe in a TaqMan assay), padlock() an upstream and a downstream arm on one strand (the early returns of pair_group):
// nine in ten of the groups the bind stage leaves alive hold one bound strand and nothing else.  One cheap lane per group looks
// at the results of its items and hands the pair kernels -- 97 registers and 1.8 KB of scratch memory per lane -- only the
// groups that need them.  Groups of the cache replay are left out (their results are not known yet), crowded groups are passed
// on unseen (the pair kernels decide which instance takes them).
struct PairSelectArgs {
	PairArgs p;
	uint32_t *selected;   // out: group indices, in place of PairArgs::active
	uint32_t *n_selected; // device counter, zeroed before the launch
};

__global__ __launch_bounds__(256) void pair_select_kernel(PairSelectArgs a)
{
	const uint32_t n_active = *a.p.n_active;
	for (uint32_t base = blockIdx.x * 256u; base < n_active; base += gridDim.x * 256u) {
		const uint32_t gi = base + threadIdx.x;
		bool keep = false;
		uint32_t g = 0;
		if (gi < n_active) {
			g = a.p.active[gi];
			const Group G = a.p.groups[g];
			const AssayDev A = a.p.assays[G.assay & GROUP_ASSAY_MASK];
			const uint32_t count = G.count & ~GROUP_ALIVE;
			if (a.p.citems && (A.cache_replay & 1)) {
				keep = false;
			} else if (count > uint32_t(PAIR_MAX) || A.kind == 2) {
				keep = true;
			} else {
				uint32_t classes = 0, arms = 0;
				for (uint32_t i = 0; i < count; i++) {
					if (a.p.result[G.first + i] < 0) continue;
					const uint32_t info = a.p.items[G.first + i].info;
					const uint32_t role = (info >> ITEM_SLOT_SHIFT) % 3u;
					classes |= role == 2 ? 4u : ((info & ITEM_MINUS) ? 1u : 2u);
					arms |= 1u << (((info & ITEM_MINUS) ? 0u : 2u) + (role == 1 ? 0u : 1u));
				}
				if (A.kind == 0) keep = (classes & 3u) == 3u && (!A.has_probe || (classes & 4u));
				else keep = (arms & 3u) == 3u || (arms & 12u) == 12u;
			}
		}
		const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
		if (m) {
			const int lane = threadIdx.x & 63;
			uint32_t at = 0;
			if (lane == __builtin_ctzll(m)) at = atomicAdd(a.n_selected, uint32_t(__builtin_popcountll(m)));
			at = uint32_t(__shfl(int(at), __builtin_ctzll(m)));
			if (keep) a.selected[at + uint32_t(__builtin_popcountll(m & ((1ull << lane) - 1ull)))] = g;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Row a6, the reference's per-target BindCache where it changes results: an oligo string used under two strand
// concentrations / threshold sets among the PCR / padlock assays of a context.  The assays that hold such an oligo are
// flagged (AssayDev::cache_replay); behind the ordinary stages of a batch
//   cache_collect_kernel  picks their groups out of the seed kernel's output, copies their items side by side (citems)
//                         and writes one sort key (target, assay) per group;
//   the unit-level bind kernel evaluates EVERY citem in full (BindFull: all fields, whatever the status);
//   a radix sort puts the groups in (target, assay) order;
//   cache_replay_kernel   walks the groups of a target in assay order, one lane per target, through pair_group<N, true>.
struct CacheCollectArgs {
	const Group *groups;
	const uint32_t *seg_counts; // groups in segment s: seg_counts[s * seg_count_stride]
	uint32_t seg_count_stride, n_segs, seg_size;
	const AssayDev *assays;
	const Item *items;
	unsigned long long *keys;   // [cap_groups], preset to ~0
	uint32_t *vals;
	ConflictRec *recs;
	uint32_t cap_groups;
	uint32_t *n_groups;
	Item *citems;
	uint32_t *corigin;
	uint32_t cap_items;
	uint32_t *n_items;
	uint32_t *n_big;
	uint32_t *err;
};

__global__ __launch_bounds__(256) void cache_collect_kernel(CacheCollectArgs a)
{
	for (uint32_t s = blockIdx.x; s < a.n_segs; s += gridDim.x) {
		const uint32_t n = a.seg_counts[size_t(s) * a.seg_count_stride];
		for (uint32_t i = threadIdx.x; i < n && i < a.seg_size; i += blockDim.x) {
			const uint32_t gi = s * a.seg_size + i;
			const Group G = a.groups[gi];
			const uint32_t assay = G.assay & GROUP_ASSAY_MASK;
			if (!(a.assays[assay].cache_replay & 1)) continue;
			const uint32_t count = G.count & ~GROUP_ALIVE;
			const uint32_t slot = atomicAdd(a.n_groups, 1u);
			const uint32_t cf = atomicAdd(a.n_items, count);
			if (slot >= a.cap_groups || cf + count > a.cap_items) {
				atomicOr(a.err, ERR_ITEM_OVERFLOW);
				continue;
			}
			if (count > uint32_t(PAIR_MAX)) atomicAdd(a.n_big, 1u);
			a.recs[slot] = ConflictRec{gi, cf, count, 0u};
			a.keys[slot] = ((unsigned long long)G.target << 32) | assay;
			a.vals[slot] = slot;
			for (uint32_t k = 0; k < count; k++) {
				Item it = a.items[G.first + k];
				it.info &= ~ITEM_PHASE_B;
				a.citems[cf + k] = it;
				a.corigin[cf + k] = 0xFFFFFFFFu;
			}
		}
	}
}

struct CacheReplayArgs {
	PairArgs p;
	const unsigned long long *keys; // sorted; unused slots hold ~0 and come last
	const uint32_t *vals;
	const ConflictRec *recs;
	uint32_t cap_groups;
	const uint32_t *n_big; // collected groups of more than PAIR_MAX items
};

// N = PAIR_MAX: the targets all of whose collected groups hold up to PAIR_MAX items (all of them in practice); N = PAIR_BIG:
// the targets with a crowded group (the replay of one target is one sequence, so a target goes to one instance whole).
template <int N> __global__ __launch_bounds__(PAIR_THREADS) void cache_replay_kernel(CacheReplayArgs a)
{
	if (N == PAIR_BIG && *a.n_big == 0) return;
	for (uint32_t i = blockIdx.x * PAIR_THREADS + threadIdx.x; i < a.cap_groups; i += gridDim.x * PAIR_THREADS) {
		const unsigned long long key = a.keys[i];
		if (key == ~0ull) continue;
		if (i > 0 && (a.keys[i - 1] >> 32) == (key >> 32)) continue; // not the first group of its target
		if (*a.n_big != 0 || N == PAIR_BIG) {
			uint32_t most = 0;
			for (uint32_t j = i; j < a.cap_groups && a.keys[j] != ~0ull && (a.keys[j] >> 32) == (key >> 32); j++) most = max(most, a.recs[a.vals[j]].count);
			if ((most > uint32_t(PAIR_MAX)) != (N == PAIR_BIG)) continue;
			if (most > uint32_t(PAIR_BIG)) {
				atomicOr(a.p.err, ERR_LIST_OVERFLOW);
				continue;
			}
		}
		const ConflictRun run{a.vals + i, a.recs};
		int n_run = 0;
		for (uint32_t j = i; j < a.cap_groups && a.keys[j] != ~0ull && (a.keys[j] >> 32) == (key >> 32); j++) {
			const ConflictRec r = run[n_run++];
			pair_group<N, true>(a.p, a.p.groups[r.group], run, n_run, r.cfirst);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Compact form of the hit records for the gather of a multi-GPU run (the reference's workers send packed hit lists to
// the master, tntblast_master.cpp:811-828): per hit the 120 bytes of numbers, three 16-bit text lengths and the text
// bytes themselves instead of three fixed 448-byte fields -- about 350 bytes instead of 1 464 for 20-mers.
//   blob = { uint32 n_hits, uint32 n_bytes } { record }*      record = 120 B numbers | u16 len[3] | u16 0 | texts | pad to 4
constexpr int HIT_POD_BYTES = 120;
static_assert(HIT_POD_BYTES == int(sizeof(Hit)) - 3 * ALIGN_TEXT, "the numbers of a hit come first, the three texts last");

struct HitPackArgs {
	const Hit *hits;
	uint32_t n;            // hits to pack ...
	const uint32_t *n_dev; // ... or, when set, the device counter that holds their number (capped by n)
	uint32_t *sizes;       // [n] record sizes, then (in place) exclusive offsets
	uint8_t *blob;         // output
	uint32_t *total;       // [0] bytes of the blob (header included), [1] hits in it
};
__device__ __forceinline__ uint32_t hit_pack_count(const HitPackArgs &a)
{
	if (!a.n_dev) return a.n;
	const uint32_t n = *a.n_dev;
	return n < a.n ? n : a.n;
}

// strlen of one 448-byte text field by a whole wavefront: every lane looks at two dwords of the field (read coalesced),
// the first zero byte is the minimum over the lanes
__device__ __forceinline__ uint32_t wave_text_len(const char *t, int lane)
{
	const uint32_t *w = reinterpret_cast<const uint32_t *>(t);
	uint32_t first = uint32_t(ALIGN_TEXT - 1);
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int d = lane + 64 * k;
		if (d < ALIGN_TEXT / 4) {
			const uint32_t v = w[d];
			const uint32_t z = (v - 0x01010101u) & ~v & 0x80808080u; // a set high bit marks a zero byte (the lowest one is exact)
			if (z) {
				const uint32_t at = uint32_t(d) * 4u + (uint32_t(__builtin_ctz(z)) >> 3);
				first = at < first ? at : first;
			}
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t other = uint32_t(__shfl_xor(int(first), o));
		first = other < first ? other : first;
	}
	return first;
}

// one wavefront per hit
__global__ __launch_bounds__(256) void hit_size_kernel(HitPackArgs a)
{
	const uint32_t n = hit_pack_count(a);
	const int lane = threadIdx.x & 63;
	for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += (gridDim.x * blockDim.x) >> 6) {
		const Hit &h = a.hits[i];
		const uint32_t bytes = uint32_t(HIT_POD_BYTES) + 8u + wave_text_len(h.forward_align, lane) + wave_text_len(h.reverse_align, lane) + wave_text_len(h.probe_align, lane);
		if (lane == 0) a.sizes[i] = (bytes + 3u) & ~3u;
	}
}

// exclusive prefix sum of sizes[] by one workgroup (hit lists are short: tens of thousands of records per search)
__global__ __launch_bounds__(1024) void hit_scan_kernel(HitPackArgs a)
{
	__shared__ uint32_t s_part[1024];
	const uint32_t n = hit_pack_count(a);
	const uint32_t per = (n + 1023u) / 1024u, b = threadIdx.x * per, e = b + per < n ? b + per : n;
	uint32_t sum = 0;
	for (uint32_t i = b; i < e; i++) sum += a.sizes[i];
	s_part[threadIdx.x] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = 8; // header
		for (int t = 0; t < 1024; t++) {
			const uint32_t v = s_part[t];
			s_part[t] = run;
			run += v;
		}
		a.total[0] = run;
		a.total[1] = n;
		reinterpret_cast<uint32_t *>(a.blob)[0] = n;
		reinterpret_cast<uint32_t *>(a.blob)[1] = run;
	}
	__syncthreads();
	uint32_t run = s_part[threadIdx.x];
	for (uint32_t i = b; i < e; i++) {
		const uint32_t v = a.sizes[i];
		a.sizes[i] = run;
		run += v;
	}
}

// one wavefront per hit: the numbers as dwords, the texts byte by byte with the lanes side by side
__global__ __launch_bounds__(256) void hit_pack_kernel(HitPackArgs a)
{
	const uint32_t n = hit_pack_count(a);
	const int lane = threadIdx.x & 63;
	for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += (gridDim.x * blockDim.x) >> 6) {
		const Hit &h = a.hits[i];
		uint8_t *o = a.blob + a.sizes[i];
		const uint32_t *src = reinterpret_cast<const uint32_t *>(&h);
		uint32_t *dst = reinterpret_cast<uint32_t *>(o);
		if (lane < HIT_POD_BYTES / 4) dst[lane] = src[lane];
		const char *txt[3] = {h.forward_align, h.reverse_align, h.probe_align};
		uint32_t len[3];
		for (int k = 0; k < 3; k++) len[k] = wave_text_len(txt[k], lane);
		if (lane == 0) {
			dst[HIT_POD_BYTES / 4] = len[0] | (len[1] << 16);
			dst[HIT_POD_BYTES / 4 + 1] = len[2];
		}
		uint8_t *w = o + HIT_POD_BYTES + 8;
		for (int k = 0; k < 3; k++) {
			for (uint32_t c = uint32_t(lane); c < len[k]; c += 64) w[c] = uint8_t(txt[k][c]);
			w += len[k];
		}
		const uint32_t pad = uint32_t(-int32_t(len[0] + len[1] + len[2])) & 3u;
		if (uint32_t(lane) < pad) w[lane] = 0;
	}
}

} // namespace tnt
#endif
