// tnt_sort.hip -- device side of the long-target path: the seeds of all windows of the targets that are seeded in windows,
// ordered into per-(target, assay) lists in the reference's list order (long_sort_seeds), their first cull (long_first_cull)
// and, behind the bind kernels, the bind / cull sequence and the pairing loops of the PCR lists (long_pair_device, round 6).
//
// The reference keeps one std::list per (target, assay) and sorts it by target_loc before the first cull
// (match_oligo_to_*_strand + cull_oligo_match, bind_oligo.cpp:84-122, amplicon_search.cpp:688; the hash-overload lists by
// diagonal, bind_oligo.cpp:157).  Round 2 assembled these lists on the host (counting sort + std::sort per list: 18 of
// 43 ms for a 5-Mb target against 96 primer pairs).  Here: one key kernel, two stable LSD radix sorts (rocPRIM: by the
// in-list key, then by the list), one gather.  Product code: nothing here includes or links anything under oracle/.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "tnt_dev.h"
#include "tnt_arena.h"
#include "tnt_seed.hip.h"
#include "tnt_hitrec.h"
#include <vector>

namespace tnt {

namespace {

// order of two seeds of one list = order of seed_key (tnt_seed.hip.h): PCR lists by (target_loc, assembly group F- R- F+ R+
// P- P+, word index); hash-overload lists by (list, diagonal, word index)
__global__ __launch_bounds__(256) void long_keys_kernel(const SeedOut *seeds, uint32_t n, const AssayDev *assays, unsigned long long *key, uint32_t *idx)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const SeedOut sd = seeds[i];
		const uint32_t assay = sd.assay_role >> 2, role = sd.assay_role & 3u;
		unsigned long long k;
		if (assays[assay].kind == 0) {
			const unsigned long long rank = role == 2 ? (sd.strand_minus ? 4u : 5u) : ((sd.strand_minus ? 0u : 2u) + role);
			k = ((unsigned long long)sd.target_loc << 10) | (rank << 7) | (unsigned long long)sd.query_loc;
		} else {
			const unsigned long long listid = (sd.strand_minus ? 0u : 2u) + (role == 1 ? 0u : 1u);
			k = (listid << 41) | ((unsigned long long)(long(sd.query_loc) - long(sd.target_loc) + (1l << 33)) << 7) | (unsigned long long)sd.query_loc;
		}
		key[i] = k;
		idx[i] = i;
	}
}

__global__ __launch_bounds__(256) void long_bucket_kernel(const SeedOut *seeds, const uint32_t *idx, uint32_t n, const uint32_t *target_rank, uint32_t n_assays, uint32_t *bucket)
{
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const SeedOut sd = seeds[idx[j]];
		bucket[j] = target_rank[sd.target] * n_assays + (sd.assay_role >> 2);
	}
}

__global__ __launch_bounds__(256) void long_gather_kernel(const SeedOut *seeds, const uint32_t *idx, uint32_t n, Item *items)
{
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const SeedOut sd = seeds[idx[j]];
		const uint32_t assay = sd.assay_role >> 2, role = sd.assay_role & 3u;
		items[j] = Item{sd.target, sd.target_loc, sd.query_loc | (sd.strand_minus ? ITEM_MINUS : 0u) | ((assay * 3u + role) << ITEM_SLOT_SHIFT), 0u};
	}
}

// ---- first cull of the assembled lists (cull_oligo_match, amplicon_search.cpp:679-765, on lists nothing is bound on yet)
// The list is in target_loc order, so the reference's loops -- F over the minus-strand primers, R over the plus-strand
// primers behind it until target_loc is more than max_len + 50 away, P over the probes between the two -- reduce to
// questions about the NEAREST element of a class before / behind a list position:
//   no probe:  a minus-strand primer stays iff the next plus-strand primer of an admissible role is within the threshold,
//              a plus-strand primer iff the last minus-strand primer of an admissible role is;
//   probe:     a minus-strand primer looks for that plus-strand primer behind the next probe, a plus-strand primer for the
//              minus-strand primer before the last probe, a probe for the nearest primers on either side of itself
// (admissible: the other role, any role with --single-primer-pcr, :716).  "Nearest of a class" is one max-scan over six
// running positions (three classes forwards, three backwards: the backward ones scan the mirrored array).
struct Near6 {
	uint32_t v[6]; // position + 1 (0: none): last minus F, last minus R, last probe | mirrored: next plus F, next plus R, next probe
};
struct Near6Max {
	__host__ __device__ Near6 operator()(const Near6 &a, const Near6 &b) const
	{
		Near6 r;
		for (int i = 0; i < 6; i++) r.v[i] = a.v[i] > b.v[i] ? a.v[i] : b.v[i];
		return r;
	}
};
struct Tally4 {
	uint32_t valid, head, minus, plus; // kept seeds, list heads, successors of kept seeds on the minus / plus strand (:748-755)
};
struct Tally4Sum {
	__host__ __device__ Tally4 operator()(const Tally4 &a, const Tally4 &b) const { return Tally4{a.valid + b.valid, a.head + b.head, a.minus + b.minus, a.plus + b.plus}; }
};

__device__ inline uint32_t item_role(const Item &it) { return (it.info >> ITEM_SLOT_SHIFT) % 3u; }

__global__ __launch_bounds__(256) void long_near_kernel(const Item *items, uint32_t n, Near6 *near)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		Near6 r;
		const Item f = items[k], b = items[n - 1u - k];
		const uint32_t rf = item_role(f), rb = item_role(b);
		const bool f_minus = (f.info & ITEM_MINUS) != 0, b_minus = (b.info & ITEM_MINUS) != 0;
		r.v[0] = (f_minus && rf == 0) ? k + 1u : 0u;
		r.v[1] = (f_minus && rf == 1) ? k + 1u : 0u;
		r.v[2] = rf == 2 ? k + 1u : 0u;
		r.v[3] = (!b_minus && rb == 0) ? k + 1u : 0u;
		r.v[4] = (!b_minus && rb == 1) ? k + 1u : 0u;
		r.v[5] = rb == 2 ? k + 1u : 0u;
		near[k] = r;
	}
}

struct NearView {
	const Near6 *near; // inclusive max-scan of long_near_kernel's output
	const uint32_t *bucket;
	uint32_t n;
	// the last element of class c (0 minus F, 1 minus R, 2 probe) before position i in i's list, or n
	__device__ uint32_t last(uint32_t i, int c) const
	{
		if (i == 0 || i >= n) return n;
		const uint32_t v = near[i - 1u].v[c];
		return (v && bucket[v - 1u] == bucket[i]) ? v - 1u : n;
	}
	// the next element of class c (0 plus F, 1 plus R, 2 probe) behind position i in i's list, or n
	__device__ uint32_t next(uint32_t i, int c) const
	{
		if (i + 1u >= n) return n;
		const uint32_t v = near[n - 2u - i].v[3 + c];
		return (v && bucket[n - v] == bucket[i]) ? n - v : n;
	}
};

__global__ __launch_bounds__(256) void long_cull_kernel(const Item *items, const uint32_t *bucket, const Near6 *near, uint32_t n, const AssayDev *assays, uint32_t n_assays,
                                                       uint32_t threshold, int single_primer, Tally4 *tally)
{
	const NearView nv{near, bucket, n};
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const Item it = items[i];
		const uint32_t list = bucket[i];
		const AssayDev A = assays[list % n_assays];
		const uint32_t role = item_role(it);
		const bool minus = (it.info & ITEM_MINUS) != 0;
		bool valid = false;
		if (A.kind != 0) {
			valid = true; // probe / padlock lists are bound as a whole (hash overload, bind_oligo.cpp:124-454)
		} else if (role < 2u) {
			// partner roles: the other one; both with single_primer
			for (int b = 0; b < 2 && !valid; b++) {
				if (!single_primer && uint32_t(b) == role) continue;
				if (minus) {
					uint32_t from = i;
					if (A.has_probe) from = nv.next(i, 2);
					const uint32_t r = from < n ? nv.next(from, b) : n;
					valid = r < n && items[r].tloc - it.tloc <= threshold;
				} else {
					uint32_t from = i;
					if (A.has_probe) from = nv.last(i, 2);
					const uint32_t f = from < n ? nv.last(from, b) : n;
					valid = f < n && it.tloc - items[f].tloc <= threshold;
				}
			}
		} else if (A.has_probe) {
			for (int a = 0; a < 2 && !valid; a++)
				for (int b = 0; b < 2 && !valid; b++) {
					if (!single_primer && a == b) continue;
					const uint32_t f = nv.last(i, a), r = nv.next(i, b);
					valid = f < n && r < n && items[r].tloc - items[f].tloc <= threshold;
				}
		}
		Tally4 t{valid ? 1u : 0u, (i == 0 || bucket[i - 1u] != list) ? 1u : 0u, 0u, 0u};
		if (valid && A.kind == 0 && i + 1u < n && bucket[i + 1u] == list) { // the reference reads the mask of the element BEHIND each kept one
			if (items[i + 1u].info & ITEM_MINUS) t.minus = 1u;
			else t.plus = 1u;
		}
		tally[i] = t;
	}
}

// kept seeds to their places; the head and the tail of every list leave what the list's group needs
__global__ __launch_bounds__(256) void long_keep_kernel(const Item *items, const uint32_t *bucket, const Tally4 *tally, uint32_t n, Item *kept, uint4 *seg_head, uint4 *seg_tail)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const Tally4 t = tally[i];
		const Tally4 before = i ? tally[i - 1u] : Tally4{0u, 0u, 0u, 0u};
		if (t.valid != before.valid) kept[before.valid] = items[i];
		const uint32_t list = bucket[i];
		if (t.head != before.head) seg_head[t.head - 1u] = make_uint4(list, before.valid, before.minus, before.plus);
		if (i + 1u == n || bucket[i + 1u] != list) seg_tail[t.head - 1u] = make_uint4(t.valid, t.minus, t.plus, 0u);
	}
}

__global__ __launch_bounds__(256) void long_group_kernel(const uint4 *seg_head, const uint4 *seg_tail, uint32_t n_seg, uint32_t n_assays, Group *groups)
{
	for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seg; s += gridDim.x * blockDim.x) {
		const uint4 h = seg_head[s], t = seg_tail[s];
		const uint32_t minus = t.y - h.z, plus = t.z - h.w;
		groups[s] = Group{h.x / n_assays, (h.x % n_assays) | (minus < plus ? GROUP_MINUS_FIRST : 0u), h.y, t.x - h.y}; // amplicon_search.cpp:131
	}
}

} // namespace

// seeds[n] (device) -> items[n], bucket[n] (device, caller's buffers): the seeds as bind work items, list after list
// (bucket = rank of the target among the windowed targets * n_assays + assay), every list in the reference's order.
// Scratch from `arena` (given back on return; long_sort_bytes(n) says how much).  Returns hipSuccess or the first error.
size_t long_sort_bytes(size_t n) { return n * 28 + (n * 12 + (size_t(16) << 20)) + 8 * 256; } // keys, indices, buckets + rocPRIM's double buffers and histograms

hipError_t long_sort_seeds(const SeedOut *seeds, uint32_t n, const uint32_t *target_rank, const AssayDev *assays, uint32_t n_assays,
                           uint32_t n_buckets, Item *items, uint32_t *bucket, DeviceArena &arena, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	const size_t mark = arena.mark();
	unsigned long long *k0 = arena.take<unsigned long long>(n), *k1 = arena.take<unsigned long long>(n);
	uint32_t *i0 = arena.take<uint32_t>(n), *i1 = arena.take<uint32_t>(n), *b0 = arena.take<uint32_t>(n);
	hipError_t e;
	const int blocks = int(std::min<size_t>(4096, (size_t(n) + 255) / 256));
	size_t need = 0, need2 = 0;
	unsigned bits = 1;
	while (bits < 32 && (1ull << bits) < (unsigned long long)n_buckets) bits++;
	if ((e = rocprim::radix_sort_pairs(nullptr, need, k0, k1, i0, i1, size_t(n), 0u, 45u, stream)) != hipSuccess) return e;
	if ((e = rocprim::radix_sort_pairs(nullptr, need2, b0, bucket, i1, i0, size_t(n), 0u, bits, stream)) != hipSuccess) return e;
	uint8_t *tmp = arena.take<uint8_t>(std::max(need, need2));
	if (!k0 || !k1 || !i0 || !i1 || !b0 || !tmp) return hipErrorOutOfMemory;
	hipLaunchKernelGGL(long_keys_kernel, dim3(blocks), dim3(256), 0, stream, seeds, n, assays, k0, i0);
	// stable LSD sorts: by the in-list key, then by the list
	if ((e = rocprim::radix_sort_pairs(tmp, need, k0, k1, i0, i1, size_t(n), 0u, 45u, stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(long_bucket_kernel, dim3(blocks), dim3(256), 0, stream, seeds, i1, n, target_rank, n_assays, b0);
	if ((e = rocprim::radix_sort_pairs(tmp, need2, b0, bucket, i1, i0, size_t(n), 0u, bits, stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(long_gather_kernel, dim3(blocks), dim3(256), 0, stream, seeds, i0, n, items);
	arena.rewind(mark); // (the stream is in order: what takes this room next runs behind the kernels above)
	return hipGetLastError();
}

// First cull of the assembled lists on the device: items[n] / bucket[n] as long_sort_seeds left them -> the kept seeds,
// list after list (kept[n], caller's buffer; *n_kept of them), and one group per list (groups[*n_groups], taken from the
// arena and left there: the caller copies them): Group::target = rank of the target among the windowed targets, ::assay
// with GROUP_MINUS_FIRST from the reference's successor counts, ::first / ::count = its kept seeds.  n_lists_max: upper
// bound of the number of lists (for the reservation); the scratch stays taken until the caller rewinds the arena.
size_t long_cull_bytes(size_t n, size_t n_lists_max) { return n * (sizeof(Near6) * 2 + sizeof(Tally4)) + (size_t(16) << 20) + std::min(n, n_lists_max) * 48 + 8 * 256; }

hipError_t long_first_cull(const Item *items, const uint32_t *bucket, uint32_t n, const AssayDev *assays, uint32_t n_assays, int max_len, int single_primer,
                           Item *kept, uint32_t *n_kept, Group **groups, uint32_t *n_groups, DeviceArena &arena, hipStream_t stream)
{
	*n_kept = *n_groups = 0;
	*groups = nullptr;
	if (n == 0) return hipSuccess;
	hipError_t e;
	Near6 *const nr_in = arena.take<Near6>(n), *const nr = arena.take<Near6>(n);
	Tally4 *const tl_in = reinterpret_cast<Tally4 *>(nr_in), *const tl = arena.take<Tally4>(n); // (the tallies reuse the scanned array's input)
	size_t need = 0, need2 = 0;
	if ((e = rocprim::inclusive_scan(nullptr, need, nr_in, nr, size_t(n), Near6Max(), stream)) != hipSuccess) return e;
	if ((e = rocprim::inclusive_scan(nullptr, need2, tl_in, tl, size_t(n), Tally4Sum(), stream)) != hipSuccess) return e;
	uint8_t *const tmp = arena.take<uint8_t>(std::max(need, need2));
	if (!nr_in || !nr || !tl || !tmp) return hipErrorOutOfMemory;
	const int blocks = int(std::min<size_t>(4096, (size_t(n) + 255) / 256));
	hipLaunchKernelGGL(long_near_kernel, dim3(blocks), dim3(256), 0, stream, items, n, nr_in);
	if ((e = rocprim::inclusive_scan(tmp, need, nr_in, nr, size_t(n), Near6Max(), stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(long_cull_kernel, dim3(blocks), dim3(256), 0, stream, items, bucket, nr, n, assays, n_assays, uint32_t(max_len) + 50u, single_primer, tl_in);
	if ((e = rocprim::inclusive_scan(tmp, need2, tl_in, tl, size_t(n), Tally4Sum(), stream)) != hipSuccess) return e;
	Tally4 total;
	if ((e = hipMemcpyAsync(&total, tl + (n - 1u), sizeof(total), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
	if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
	uint4 *const head = arena.take<uint4>(total.head), *const tail = arena.take<uint4>(total.head);
	*groups = arena.take<Group>(total.head);
	if (!head || !tail || !*groups) return hipErrorOutOfMemory;
	hipLaunchKernelGGL(long_keep_kernel, dim3(blocks), dim3(256), 0, stream, items, bucket, tl, n, kept, head, tail);
	hipLaunchKernelGGL(long_group_kernel, dim3(int(std::min<size_t>(1024, (size_t(total.head) + 255) / 256))), dim3(256), 0, stream, head, tail, total.head, n_assays, *groups);
	*n_kept = total.valid;
	*n_groups = total.head;
	return hipGetLastError();
}

// =====================================================================================================================
// The bind / cull sequence and the pairing loops of amplicon() (amplicon_search.cpp:131-674) for the PCR lists of targets
// that are seeded in windows, on the device, for all lists of a pass at once.
//
// What has to be reproduced.  Given the lists the first cull left (in target_loc order) and the bind outcome of every one of
// their seeds, the reference goes through four bind steps -- F and R on the strand the first cull's counts name, then on the
// other one --: the seeds of the step's (oligo, strand) class leave the list, the bound ones come back at its END with their
// footprints, ordered by sort_by_bound_match and reduced to one per footprint (bind_oligo.cpp:476-500, 806-826); behind each of
// the first three steps cull_oligo_match runs again (:679-765), and it begins with list.sort(sort_by_oligo_loc) (:688).  An
// assay with a probe culls once more and binds P-, P+.  A last sort_by_oligo_loc, then the pairing loops (:356-674).
//
// sort_by_oligo_loc (:12-26) compares two bound entries by their footprints (loc_5, loc_3) and any other pair by target_loc.
// That is not a strict weak order in general, and where it is not, what std::list::sort leaves depends on its
// merge tree: the host replay (tnt_pair.hip.h, list_sort) restates that tree, serially.  But it IS one -- "target_loc <" --
// whenever the bound entries of the list are CONSISTENT: for any two of them target_loc and (loc_5, loc_3) order the same way,
// ties included.  Then list.sort() is the stable sort by target_loc, i.e. the bound entries of the step are MERGED by
// target_loc into a list that is in that order already; the marking loops of the cull, which look at target_loc and at the
// list order only, are the first cull's "nearest element of a class" scans (above) run on the merged list; and the last sort
// sees bound entries only, where the comparison is a strict weak order whatever the footprints.  Consistency is checked
// exactly, per list and step: the bound entries of a list in list order must have target_loc and footprint either both
// strictly increasing or both equal from one to the next (transitive: it holds for every pair).  A list that fails -- two
// footprints crossing their seeds' order, a footprint whose coordinates sum to zero in the last sort -- is flagged and handed
// to the host replay whole; nothing of it is reported from here.
//
// Every step is a handful of streaming passes over all lists together (flag, scan, scatter, binary-search merge, the cull's
// two scans); the bound entries, a hundredth of the seeds, are sorted with rocPRIM's radix sort.  The pairing loops run one
// thread per minus-strand primer, twice: counting its hits, then writing them where the count's prefix sum says -- the hits
// come out in the order the reference's loops produce them.
// =====================================================================================================================
namespace {

struct LpList { // one list of the device pairing
	uint32_t assay;     // expanded assay
	uint32_t flags;     // LP_MINUS_FIRST | LP_HAS_PROBE
	int32_t seq_index;  // of its target
	uint32_t group;     // index of its group in the caller's array
};
constexpr uint32_t LP_MINUS_FIRST = 1u, LP_HAS_PROBE = 2u;

// step of the bind sequence that takes the entries of (role, strand): 0 .. 3 the primers (first strand F, R; other strand F, R), 4 P-, 5 P+
__device__ inline uint32_t lp_step_of(uint32_t role, bool minus, uint32_t lflags)
{
	if (role == 2u) return minus ? 4u : 5u;
	const bool first_strand = minus == ((lflags & LP_MINUS_FIRST) != 0u);
	return (first_strand ? 0u : 2u) + role;
}
// Is the list sorted again behind bind step `step` (a cull follows: behind the first three steps, and behind the fourth in an
// assay with a probe)?  Then the step's entries are merged in by target_loc and the list must be consistent; else they stay at
// the END of the list, as bind_oligo_to_*_strand leaves them, until the last sort.
__device__ inline bool lp_sorted_behind(uint32_t step, uint32_t lflags) { return step < 3u || (step == 3u && (lflags & LP_HAS_PROBE)); }
__device__ inline uint32_t lp_bias(int32_t v) { return uint32_t(v) ^ 0x80000000u; } // ascending order of a signed number as an unsigned one
__device__ inline uint32_t lp_float_desc(float f) // descending order of a float as an ascending unsigned key
{
	if (f == 0.0f) f = 0.0f; // (-0.0 and +0.0 compare equal)
	const uint32_t u = __float_as_uint(f);
	const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
	return ~asc;
}

// the kept seeds of the lists that take part, side by side: Item::group = the seed's index in `kept` (its bind result sits there)
__global__ __launch_bounds__(256) void lp_init_kernel(const Item *kept, uint32_t n_kept, const uint32_t *g_first, const uint32_t *g_dst, const uint32_t *g_list, uint32_t n_groups,
                                                     Item *items, uint32_t *bucket)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_kept; i += gridDim.x * blockDim.x) {
		uint32_t lo = 0, hi = n_groups; // last group whose first seed is at or before i
		while (hi - lo > 1u) {
			const uint32_t mid = (lo + hi) >> 1;
			if (g_first[mid] <= i) lo = mid;
			else hi = mid;
		}
		if (g_dst[lo] == 0xFFFFFFFFu || i >= g_first[lo + 1u]) continue;
		Item it = kept[i];
		it.group = i;
		const uint32_t at = g_dst[lo] + (i - g_first[lo]);
		items[at] = it;
		bucket[at] = g_list[lo];
	}
}

// bind step `step`: the entries of the step's class leave the list; the bound ones are collected (keep | collect << 32)
__global__ __launch_bounds__(256) void lp_flag_kernel(const Item *items, const uint32_t *bucket, uint32_t n, const LpList *lists, const int32_t *result, uint32_t step,
                                                     unsigned long long *flag)
{
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const Item it = items[j];
		const LpList L = lists[bucket[j]];
		const bool mine = lp_step_of(item_role(it), (it.info & ITEM_MINUS) != 0u, L.flags) == step && (step < 4u || (L.flags & LP_HAS_PROBE));
		unsigned long long f = 1ull;
		if (mine) f = result[it.group] >= 0 ? (1ull << 32) : 0ull;
		flag[j] = f;
	}
}

struct LpBound { // a bound entry of the current step
	Item it;
	uint32_t bucket;
	int32_t loc5, loc3;
	uint32_t tm_desc, mm, text_len;
};

// pos = exclusive prefix sums of the flags: the entries that stay close ranks; the collected ones are listed in REVERSE list
// order (bind_oligo.cpp:476-500 fills curr_oligo with push_front), which the stable sorts behind this keep among equals
__global__ __launch_bounds__(256) void lp_scatter_kernel(const Item *items, const uint32_t *bucket, uint32_t n, const unsigned long long *flag, const unsigned long long *pos,
                                                        const int32_t *result, const BoundRec *pool, uint32_t n_collect, Item *items_out, uint32_t *bucket_out, LpBound *bound)
{
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const unsigned long long f = flag[j], p = pos[j];
		if (f & 1ull) {
			items_out[uint32_t(p)] = items[j];
			bucket_out[uint32_t(p)] = bucket[j];
		} else if (f >> 32) {
			const Item it = items[j];
			const BoundRec &r = pool[result[it.group]];
			bound[n_collect - 1u - uint32_t(p >> 32)] = LpBound{it, bucket[j], r.loc5, r.loc3, lp_float_desc(r.tm), uint32_t(r.mm), uint32_t(r.text_len)};
		}
	}
}

// keys of the three stable passes that put the collected entries into sort_by_bound_match's order per list (bind_oligo.cpp:49-82:
// loc_5, loc_3, then Tm, mismatches, alignment size descending), least significant first
__global__ __launch_bounds__(256) void lp_bound_key_kernel(const LpBound *bound, const uint32_t *idx, uint32_t n, int pass, unsigned long long *key)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const LpBound &b = bound[idx ? idx[k] : k];
		unsigned long long v;
		if (pass == 0) v = ((unsigned long long)(~b.mm) << 32) | (unsigned long long)(~b.text_len);
		else if (pass == 1) v = ((unsigned long long)lp_bias(b.loc3) << 32) | (unsigned long long)b.tm_desc;
		else v = ((unsigned long long)b.bucket << 32) | (unsigned long long)lp_bias(b.loc5);
		key[k] = v;
	}
}
__global__ __launch_bounds__(256) void lp_iota_kernel(uint32_t *idx, uint32_t n)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) idx[k] = k;
}

// one entry per footprint: the first of every run of equal (list, loc_5, loc_3) (bind_oligo.cpp:806-826)
__global__ __launch_bounds__(256) void lp_unique_flag_kernel(const LpBound *bound, const uint32_t *idx, uint32_t n, uint32_t *first)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t f = 1u;
		if (k) {
			const LpBound &a = bound[idx[k - 1u]], &b = bound[idx[k]];
			if (a.bucket == b.bucket && a.loc5 == b.loc5 && a.loc3 == b.loc3) f = 0u;
		}
		first[k] = f;
	}
}
__global__ __launch_bounds__(256) void lp_unique_scatter_kernel(const LpBound *bound, const uint32_t *idx, uint32_t n, const uint32_t *first, const uint32_t *pos, const LpList *lists,
                                                               uint32_t step, Item *c_items, uint32_t *c_bucket, unsigned long long *c_key)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		if (!first[k]) continue;
		const LpBound &b = bound[idx[k]];
		c_items[pos[k]] = b.it;
		c_bucket[pos[k]] = b.bucket;
		// (the merge key: the entry's target_loc where the list is sorted behind this step, else "behind everything of its list")
		c_key[pos[k]] = ((unsigned long long)b.bucket << 32) | (unsigned long long)(lp_sorted_behind(step, lists[b.bucket].flags) ? b.it.tloc : 0xFFFFFFFFu);
	}
}
// the step's entries are in footprint order; as a run of the merge below they have to be in target_loc order as well
__global__ __launch_bounds__(256) void lp_run_check_kernel(const unsigned long long *c_key, uint32_t n, uint32_t *list_flag)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k + 1u < n; k += gridDim.x * blockDim.x)
		if ((c_key[k] >> 32) == (c_key[k + 1u] >> 32) && uint32_t(c_key[k]) != 0xFFFFFFFFu && uint32_t(c_key[k]) >= uint32_t(c_key[k + 1u])) atomicOr(list_flag + uint32_t(c_key[k] >> 32), 1u);
}

__global__ __launch_bounds__(256) void lp_run_gather_kernel(const Item *c_items, const uint32_t *c_bucket, const uint32_t *idx, uint32_t n, Item *out_items, uint32_t *out_bucket)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		out_items[k] = c_items[idx[k]];
		out_bucket[k] = c_bucket[idx[k]];
	}
}

// list.sort(sort_by_oligo_loc) of [the entries that stayed, in their order] + [the step's entries] for consistent lists: the
// stable merge by (list, target_loc), an entry that stayed before a step's entry of the same target_loc
__global__ __launch_bounds__(256) void lp_merge_kernel(const Item *b_items, const uint32_t *b_bucket, uint32_t n_b, const Item *c_items, const uint32_t *c_bucket,
                                                      const unsigned long long *c_key, uint32_t n_c, Item *out_items, uint32_t *out_bucket)
{
	const uint32_t n = n_b + n_c;
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
		if (x < n_b) {
			const Item it = b_items[x];
			const unsigned long long key = ((unsigned long long)b_bucket[x] << 32) | (unsigned long long)it.tloc;
			uint32_t lo = 0, hi = n_c; // the step's entries with a smaller key come first
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (c_key[mid] < key) lo = mid + 1u;
				else hi = mid;
			}
			out_items[x + lo] = it;
			out_bucket[x + lo] = b_bucket[x];
		} else {
			const uint32_t r = x - n_b;
			const unsigned long long key = c_key[r];
			uint32_t lo = 0, hi = n_b; // the entries that stayed with a key up to this one come first
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				const unsigned long long km = ((unsigned long long)b_bucket[mid] << 32) | (unsigned long long)b_items[mid].tloc;
				if (km <= key) lo = mid + 1u;
				else hi = mid;
			}
			out_items[r + lo] = c_items[r];
			out_bucket[r + lo] = c_bucket[r];
		}
	}
}

// consistency of a list's bound entries (see the head): those whose class the sequence has bound so far, in list order
struct LpFoot {
	uint32_t bucket, tloc;
	int32_t loc5, loc3;
};
__global__ __launch_bounds__(256) void lp_foot_flag_kernel(const Item *items, const uint32_t *bucket, uint32_t n, const LpList *lists, const int32_t *result, const BoundRec *pool,
                                                          uint32_t step, uint32_t *flag)
{
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const Item it = items[j];
		const LpList L = lists[bucket[j]];
		uint32_t f = 0u;
		if (lp_sorted_behind(step, L.flags) && lp_step_of(item_role(it), (it.info & ITEM_MINUS) != 0u, L.flags) <= step) { // (entries of a bound class that are still here are bound)
			const BoundRec &r = pool[result[it.group]];
			f = (r.loc5 + r.loc3) != 0 ? 1u : 0u; // (coordinates that sum to zero: sort_by_oligo_loc takes the entry for an unbound one)
		}
		flag[j] = f;
	}
}
__global__ __launch_bounds__(256) void lp_foot_scatter_kernel(const Item *items, const uint32_t *bucket, uint32_t n, const uint32_t *flag, const uint32_t *pos, const int32_t *result,
                                                             const BoundRec *pool, LpFoot *foot)
{
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		if (!flag[j]) continue;
		const Item it = items[j];
		const BoundRec &r = pool[result[it.group]];
		foot[pos[j]] = LpFoot{bucket[j], it.tloc, r.loc5, r.loc3};
	}
}
__global__ __launch_bounds__(256) void lp_foot_check_kernel(const LpFoot *foot, uint32_t n, uint32_t *list_flag, uint32_t *debug_count = nullptr)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k + 1u < n; k += gridDim.x * blockDim.x) {
		const LpFoot a = foot[k], b = foot[k + 1u];
		if (a.bucket != b.bucket) continue;
		const bool loc_less = a.loc5 != b.loc5 ? a.loc5 < b.loc5 : a.loc3 < b.loc3;
		const bool loc_same = a.loc5 == b.loc5 && a.loc3 == b.loc3;
		const bool ok = (a.tloc < b.tloc && loc_less) || (a.tloc == b.tloc && loc_same);
		if (!ok) {
			atomicOr(list_flag + a.bucket, 2u);
			if (debug_count && atomicAdd(debug_count, 1u) < 12u)
				printf("long_pair_device: list %u: target_loc %u footprint [%d, %d] before target_loc %u footprint [%d, %d]\n", a.bucket, a.tloc, a.loc5, a.loc3, b.tloc, b.loc5, b.loc3);
		}
	}
}

// cull_oligo_match on the lists as they stand (the first cull's kernel with the list's assay looked up in the list table;
// `probe_lists_only`: the cull in front of the probe's bind steps, which only assays with a probe run)
__global__ __launch_bounds__(256) void lp_cull_kernel(const Item *items, const uint32_t *bucket, const Near6 *near, uint32_t n, const LpList *lists, uint32_t threshold, int single_primer,
                                                     int probe_lists_only, uint32_t *valid_out)
{
	const NearView nv{near, bucket, n};
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const Item it = items[i];
		const LpList L = lists[bucket[i]];
		const bool has_probe = (L.flags & LP_HAS_PROBE) != 0u;
		const uint32_t role = item_role(it);
		const bool minus = (it.info & ITEM_MINUS) != 0;
		bool valid = false;
		if (probe_lists_only && !has_probe) {
			valid = true;
		} else if (role < 2u) {
			for (int b = 0; b < 2 && !valid; b++) {
				if (!single_primer && uint32_t(b) == role) continue;
				if (minus) {
					uint32_t from = i;
					if (has_probe) from = nv.next(i, 2);
					const uint32_t r = from < n ? nv.next(from, b) : n;
					valid = r < n && items[r].tloc - it.tloc <= threshold;
				} else {
					uint32_t from = i;
					if (has_probe) from = nv.last(i, 2);
					const uint32_t f = from < n ? nv.last(from, b) : n;
					valid = f < n && it.tloc - items[f].tloc <= threshold;
				}
			}
		} else if (has_probe) {
			for (int a = 0; a < 2 && !valid; a++)
				for (int b = 0; b < 2 && !valid; b++) {
					if (!single_primer && a == b) continue;
					const uint32_t f = nv.last(i, a), r = nv.next(i, b);
					valid = f < n && r < n && items[r].tloc - items[f].tloc <= threshold;
				}
		}
		valid_out[i] = valid ? 1u : 0u;
	}
}
__global__ __launch_bounds__(256) void lp_keep_kernel(const Item *items, const uint32_t *bucket, uint32_t n, const uint32_t *valid, const uint32_t *pos, Item *items_out, uint32_t *bucket_out)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		if (valid[i]) {
			items_out[pos[i]] = items[i];
			bucket_out[pos[i]] = bucket[i];
		}
}

// ---- the last sort and the pairing loops (amplicon_search.cpp:356-674) on what the sequence left: bound entries only
struct LpFinal {
	int32_t loc5, loc3;
	int32_t rec;       // pool record
	uint32_t bucket;
	uint32_t mask;     // bit 0 R (else F), bit 1 probe, bit 2 minus strand
};
__global__ __launch_bounds__(256) void lp_final_kernel(const Item *items, const uint32_t *bucket, uint32_t n, const int32_t *result, const BoundRec *pool, LpFinal *fin, uint32_t *list_flag)
{
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const Item it = items[j];
		const int32_t rec = result[it.group];
		const BoundRec &r = pool[rec];
		const uint32_t role = item_role(it);
		fin[j] = LpFinal{r.loc5, r.loc3, rec, bucket[j], (role == 1u ? 1u : 0u) | (role == 2u ? 2u : 0u) | ((it.info & ITEM_MINUS) ? 4u : 0u)};
		if (r.loc5 + r.loc3 == 0) atomicOr(list_flag + bucket[j], 4u); // sort_by_oligo_loc would compare this entry by target_loc: not a strict weak order
	}
}
__global__ __launch_bounds__(256) void lp_final_key_kernel(const LpFinal *fin, const uint32_t *idx, uint32_t n, int pass, unsigned long long *key)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const LpFinal &e = fin[idx ? idx[k] : k];
		key[k] = pass == 0 ? (unsigned long long)lp_bias(e.loc3) : (((unsigned long long)e.bucket << 32) | (unsigned long long)lp_bias(e.loc5));
	}
}
__global__ __launch_bounds__(256) void lp_final_gather_kernel(const LpFinal *fin, const uint32_t *idx, uint32_t n, LpFinal *out, long long *list_slack)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const LpFinal e = fin[idx[k]];
		out[k] = e;
		atomicMax(list_slack + e.bucket, (long long)e.loc5 - (long long)e.loc3); // the largest (loc5 - loc3) of the list: bounds the walk below
	}
}

struct LpPairArgs {
	const LpFinal *e;
	uint32_t n;
	const LpList *lists;
	const uint32_t *list_flag;
	const long long *list_slack;
	const AssayDev *assays;
	const BoundRec *pool;
	int max_len, single_primer, min_max_primer_clamp;
	uint32_t *count;            // [n] hits of every entry as the forward primer ...
	const uint32_t *offset;     // ... and, in the writing pass, where they go
	Hit *hits;
};
// One thread per entry F: its share of the reference's loops (F over the list, R behind it, P between them).  The list is in
// ascending loc5 order, so every R behind the first one whose loc5 puts the amplicon beyond max_len does so too (R.loc3 >=
// R.loc5 - slack): the walk stops there, as in pair_group (tnt_pair.hip.h).  WRITE = false counts, WRITE = true writes.
template <bool WRITE> __global__ __launch_bounds__(256) void lp_pair_kernel(LpPairArgs a)
{
	for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < a.n; f += gridDim.x * blockDim.x) {
		const LpFinal F = a.e[f];
		uint32_t hits = 0;
		const bool takes_part = !(F.mask & 2u) && (F.mask & 4u) && !a.list_flag[F.bucket];
		if (takes_part) {
			const LpList L = a.lists[F.bucket];
			const AssayDev A = a.assays[L.assay];
			const bool has_probe = (L.flags & LP_HAS_PROBE) != 0u;
			const long long slack = a.list_slack[F.bucket];
			const bool apply_mm = a.min_max_primer_clamp >= 0;
			Hit *out = WRITE ? a.hits + a.offset[f] : nullptr;
			for (uint32_t r = f + 1u; r < a.n; r++) {
				const LpFinal R = a.e[r];
				if (R.bucket != F.bucket) break;
				if ((long long)R.loc5 - slack - (long long)F.loc5 + 1 > (long long)a.max_len) break;
				if ((R.mask & 2u) || (R.mask & 4u)) continue;               // a probe, or on the minus strand as well
				if (!a.single_primer && ((F.mask & 1u) == (R.mask & 1u))) continue;
				if (F.loc3 >= R.loc5) continue;
				if ((R.loc3 - F.loc5 + 1) > a.max_len) continue;
				const BoundRec &bf = a.pool[F.rec], &br = a.pool[R.rec];
				if (apply_mm && uint32_t(bf.a3 > br.a3 ? bf.a3 : br.a3) <= uint32_t(a.min_max_primer_clamp)) continue;
				const bool f_is_r = (F.mask & 1u) != 0u, r_is_r = (R.mask & 1u) != 0u;
				const bool swap = f_is_r && !r_is_r; // :478-483
				const BoundRec &fo = swap ? br : bf, &ro = swap ? bf : br;
				auto fill = [&](Hit &h) __attribute__((always_inline)) {
					init_hit(h, A, L.seq_index);
					if (f_is_r && r_is_r) h.forward_oligo = 1; // :453-457
					if (!f_is_r && !r_is_r) h.reverse_oligo = 0; // :459-463
					h.primer_strand = f_is_r ? 1 : 0;
					h.amplicon_start = F.loc5;
					h.amplicon_stop = R.loc3;
					h.forward_tm = fo.tm, h.reverse_tm = ro.tm;
					h.forward_dH = fo.dH, h.reverse_dH = ro.dH;
					h.forward_dS = fo.dS, h.reverse_dS = ro.dS;
					h.forward_mm = wrap8(fo.mm), h.reverse_mm = wrap8(ro.mm);
					h.forward_gap = wrap8(fo.gap), h.reverse_gap = wrap8(ro.gap);
					h.forward_primer_clamp = wrap8(fo.a3), h.reverse_primer_clamp = wrap8(ro.a3);
					copy_text(h.forward_align, fo);
					copy_text(h.reverse_align, ro);
				};
				if (has_probe) {
					for (uint32_t p = f + 1u; p < r; p++) {
						const LpFinal P = a.e[p];
						if (!(P.mask & 2u)) continue;
						if (!(P.loc5 >= F.loc5 && P.loc3 <= R.loc3)) continue; // :418
						if ((P.mask & 4u) == (F.mask & 4u)) {
							if (P.loc5 <= F.loc3) continue;
						} else {
							if (P.loc3 >= R.loc5) continue;
						}
						if (WRITE) {
							Hit &h = out[hits];
							fill(h);
							const BoundRec &bp = a.pool[P.rec];
							h.probe_start = P.loc5, h.probe_stop = P.loc3;
							h.probe_tm = bp.tm, h.probe_dH = bp.dH, h.probe_dS = bp.dS;
							h.probe_mm = wrap8(bp.mm), h.probe_gap = wrap8(bp.gap);
							h.probe_strand = (P.mask & 4u) ? 1 : 0;
							copy_text(h.probe_align, bp);
						}
						hits++;
					}
				} else {
					if (WRITE) fill(out[hits]);
					hits++;
				}
			}
		}
		if (!WRITE) a.count[f] = hits;
	}
}

template <class T> hipError_t lp_scan(void *tmp, size_t &bytes, const T *in, T *out, size_t n, hipStream_t stream)
{
	return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), stream);
}

} // namespace

// Bytes of arena the device pairing takes for n seeds in its lists (the two list buffers, the flag / scan arrays, the cull's
// scans; the bound entries' arrays are a small part of it).
size_t long_pair_bytes(size_t n) { return n * (2 * (sizeof(Item) + 4) + 16 + 16 + 2 * sizeof(Near6) + 8) + n / 4 * (sizeof(LpBound) + 64) + (size_t(64) << 20); }

// kept[n_kept] (device): the seeds the first cull kept, list after list; h_groups[n_groups]: their groups (host copy: first, count,
// target rank, assay | GROUP_MINUS_FIRST); h_take[g] != 0: the list takes part (a PCR assay outside the cache replay);
// result[n_kept] / pool (device): the bind outcome of every seed.  Returns the hits of the lists that took part and were
// consistent, in list order, and h_flagged[g] = 1 for the lists whose sort_by_oligo_loc is not a strict weak order (host replay).
hipError_t long_pair_device(const Item *kept, uint32_t n_kept, const Group *h_groups, uint32_t n_groups, const uint8_t *h_take, const int32_t *result, const BoundRec *pool,
                            const AssayDev *assays, const AssayDev *h_assays, const uint32_t *h_unit_target, uint32_t first_seq_index, int max_len, int single_primer,
                            int min_max_primer_clamp, std::vector<Hit> &hits_out, std::vector<uint8_t> &h_flagged, DeviceArena &arena, hipStream_t stream)
{
	hits_out.clear();
	h_flagged.assign(n_groups, 0);
	// ---- the list table
	std::vector<uint32_t> g_first(n_groups + 1), g_dst(n_groups), g_list(n_groups);
	std::vector<LpList> lists;
	uint32_t n = 0;
	bool any_probe = false;
	for (uint32_t g = 0; g < n_groups; g++) {
		const uint32_t count = h_groups[g].count & ~GROUP_ALIVE;
		g_first[g] = h_groups[g].first;
		g_dst[g] = 0xFFFFFFFFu, g_list[g] = 0;
		if (!h_take[g] || !count) continue;
		const uint32_t assay = h_groups[g].assay & GROUP_ASSAY_MASK;
		const bool probe = h_assays[assay].has_probe != 0;
		any_probe = any_probe || probe;
		g_dst[g] = n, g_list[g] = uint32_t(lists.size());
		lists.push_back(LpList{assay, ((h_groups[g].assay & GROUP_MINUS_FIRST) ? LP_MINUS_FIRST : 0u) | (probe ? LP_HAS_PROBE : 0u),
		                       int32_t(first_seq_index + h_unit_target[h_groups[g].target]), g});
		n += count;
	}
	g_first[n_groups] = n_kept;
	if (n == 0) return hipSuccess;
	const uint32_t n_lists = uint32_t(lists.size());
	hipError_t e;
#define LP_TRY(X)                               \
	do {                                        \
		if ((e = (X)) != hipSuccess) return e;  \
	} while (0)
	const bool debug = getenv("TNT_MI355X_LONG_PAIR_DEBUG") != nullptr; // every stage synchronised and named on stderr
	auto stage = [&](const char *what, unsigned long long a0 = 0, unsigned long long a1 = 0) {
		if (!debug) return;
		const hipError_t se = hipStreamSynchronize(stream);
		fprintf(stderr, "long_pair_device: %-28s %llu %llu %s\n", what, a0, a1, se == hipSuccess ? "ok" : hipGetErrorString(se));
	};
	const size_t mark0 = arena.mark();
	uint32_t *const d_gfirst = arena.take<uint32_t>(n_groups + 1), *const d_gdst = arena.take<uint32_t>(n_groups), *const d_glist = arena.take<uint32_t>(n_groups);
	LpList *const d_lists = arena.take<LpList>(n_lists);
	uint32_t *const d_list_flag = arena.take<uint32_t>(n_lists);
	long long *const d_slack = arena.take<long long>(n_lists);
	Item *itm[2] = {arena.take<Item>(n), arena.take<Item>(n)};
	uint32_t *bkt[2] = {arena.take<uint32_t>(n), arena.take<uint32_t>(n)};
	unsigned long long *const d_flag = arena.take<unsigned long long>(n), *const d_pos = arena.take<unsigned long long>(n);
	Near6 *const nr_in = arena.take<Near6>(n), *const nr = arena.take<Near6>(n);
	uint32_t *const d_cnt = arena.take<uint32_t>(4);
	if (!d_gfirst || !d_gdst || !d_glist || !d_lists || !d_list_flag || !d_slack || !itm[0] || !itm[1] || !bkt[0] || !bkt[1] || !d_flag || !d_pos || !nr_in || !nr || !d_cnt) return hipErrorOutOfMemory;
	uint32_t *const d_flag32 = reinterpret_cast<uint32_t *>(d_flag), *const d_pos32 = reinterpret_cast<uint32_t *>(d_pos); // (the 32-bit flags and sums of the culls live in the same arrays)
	LP_TRY(hipMemcpyAsync(d_gfirst, g_first.data(), sizeof(uint32_t) * (n_groups + 1), hipMemcpyHostToDevice, stream));
	LP_TRY(hipMemcpyAsync(d_gdst, g_dst.data(), sizeof(uint32_t) * n_groups, hipMemcpyHostToDevice, stream));
	LP_TRY(hipMemcpyAsync(d_glist, g_list.data(), sizeof(uint32_t) * n_groups, hipMemcpyHostToDevice, stream));
	LP_TRY(hipMemcpyAsync(d_lists, lists.data(), sizeof(LpList) * n_lists, hipMemcpyHostToDevice, stream));
	LP_TRY(hipMemsetAsync(d_list_flag, 0, sizeof(uint32_t) * n_lists, stream));
	LP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * 4, stream));
	auto grid = [](size_t m) { return dim3(unsigned(std::min<size_t>(8192, std::max<size_t>(1, (m + 255) / 256)))); };
	hipLaunchKernelGGL(lp_init_kernel, grid(n_kept), dim3(256), 0, stream, kept, n_kept, d_gfirst, d_gdst, d_glist, n_groups, itm[0], bkt[0]);
	int cur = 0;
	uint32_t n_cur = n;
	stage("lists side by side", n, n_lists);
	// scratch of the scans and sorts: sized once for the largest calls
	size_t need = 0, need_a = 0;
	LP_TRY(lp_scan<unsigned long long>(nullptr, need_a, d_flag, d_pos, size_t(n), stream));
	need = std::max(need, need_a);
	LP_TRY(rocprim::inclusive_scan(nullptr, need_a, nr_in, nr, size_t(n), Near6Max(), stream));
	need = std::max(need, need_a);
	LP_TRY(lp_scan<uint32_t>(nullptr, need_a, d_flag32, d_pos32, size_t(n), stream));
	need = std::max(need, need_a) + 256;
	uint8_t *const tmp = arena.take<uint8_t>(need);
	if (!tmp) return hipErrorOutOfMemory;
	const size_t mark_step = arena.mark();

	// one cull of the lists as they stand
	auto cull = [&](int probe_lists_only) -> hipError_t {
		if (n_cur == 0) return hipSuccess;
		size_t b = need;
		hipLaunchKernelGGL(long_near_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], n_cur, nr_in);
		hipError_t ee;
		if ((ee = rocprim::inclusive_scan(tmp, b, nr_in, nr, size_t(n_cur), Near6Max(), stream)) != hipSuccess) return ee;
		hipLaunchKernelGGL(lp_cull_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], bkt[cur], nr, n_cur, d_lists, uint32_t(max_len) + 50u, single_primer, probe_lists_only, d_flag32);
		b = need;
		if ((ee = lp_scan<uint32_t>(tmp, b, d_flag32, d_pos32, size_t(n_cur), stream)) != hipSuccess) return ee;
		hipLaunchKernelGGL(lp_keep_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], bkt[cur], n_cur, d_flag32, d_pos32, itm[cur ^ 1], bkt[cur ^ 1]);
		uint32_t last[2];
		if ((ee = hipMemcpyAsync(&last[0], d_flag32 + (n_cur - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
		if ((ee = hipMemcpyAsync(&last[1], d_pos32 + (n_cur - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
		if ((ee = hipStreamSynchronize(stream)) != hipSuccess) return ee;
		n_cur = last[0] + last[1];
		cur ^= 1;
		stage("cull", n_cur, unsigned(probe_lists_only));
		return hipGetLastError();
	};
	// one bind step: the step's entries leave the list, the bound ones come back merged in
	auto bind_step = [&](uint32_t step) -> hipError_t {
		if (n_cur == 0) return hipSuccess;
		hipError_t ee;
		size_t b = need;
		hipLaunchKernelGGL(lp_flag_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], bkt[cur], n_cur, d_lists, result, step, d_flag);
		if ((ee = lp_scan<unsigned long long>(tmp, b, d_flag, d_pos, size_t(n_cur), stream)) != hipSuccess) return ee;
		unsigned long long last[2];
		if ((ee = hipMemcpyAsync(&last[0], d_flag + (n_cur - 1u), sizeof(last[0]), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
		if ((ee = hipMemcpyAsync(&last[1], d_pos + (n_cur - 1u), sizeof(last[1]), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
		if ((ee = hipStreamSynchronize(stream)) != hipSuccess) return ee;
		const unsigned long long tot = last[0] + last[1];
		const uint32_t n_stay = uint32_t(tot), n_bound = uint32_t(tot >> 32);
		stage("bind step: flags", n_stay, n_bound);
		arena.rewind(mark_step);
		LpBound *const bound = arena.take<LpBound>(n_bound);
		unsigned long long *const k0 = arena.take<unsigned long long>(n_bound), *const k1 = arena.take<unsigned long long>(n_bound);
		uint32_t *const i0 = arena.take<uint32_t>(n_bound), *const i1 = arena.take<uint32_t>(n_bound), *const first = arena.take<uint32_t>(n_bound), *const upos = arena.take<uint32_t>(n_bound);
		Item *const c_items = arena.take<Item>(n_bound);
		uint32_t *const c_bucket = arena.take<uint32_t>(n_bound);
		unsigned long long *const c_key = arena.take<unsigned long long>(n_bound);
		size_t sort_bytes = 0, scan_bytes = 0;
		if (n_bound) {
			if ((ee = rocprim::radix_sort_pairs(nullptr, sort_bytes, k0, k1, i0, i1, size_t(n_bound), 0u, 64u, stream)) != hipSuccess) return ee;
			if ((ee = lp_scan<uint32_t>(nullptr, scan_bytes, first, upos, size_t(n_bound), stream)) != hipSuccess) return ee;
		}
		uint8_t *const stmp = arena.take<uint8_t>(std::max(sort_bytes, scan_bytes) + 256);
		if (!bound || !k0 || !k1 || !i0 || !i1 || !first || !upos || !c_items || !c_bucket || !c_key || !stmp) return hipErrorOutOfMemory;
		hipLaunchKernelGGL(lp_scatter_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], bkt[cur], n_cur, d_flag, d_pos, result, pool, n_bound, itm[cur ^ 1], bkt[cur ^ 1], bound);
		uint32_t n_c = 0;
		if (n_bound) {
			// sort_by_bound_match's order per list: three stable passes, least significant key first (the entries start out in reverse list order)
			uint32_t *ia = i0, *ib = i1;
			hipLaunchKernelGGL(lp_iota_kernel, grid(n_bound), dim3(256), 0, stream, ia, n_bound);
			for (int pass = 0; pass < 3; pass++) {
				hipLaunchKernelGGL(lp_bound_key_kernel, grid(n_bound), dim3(256), 0, stream, bound, ia, n_bound, pass, k0);
				size_t sb = sort_bytes;
				if ((ee = rocprim::radix_sort_pairs(stmp, sb, k0, k1, ia, ib, size_t(n_bound), 0u, 64u, stream)) != hipSuccess) return ee;
				std::swap(ia, ib);
			}
			hipLaunchKernelGGL(lp_unique_flag_kernel, grid(n_bound), dim3(256), 0, stream, bound, ia, n_bound, first);
			size_t cb = scan_bytes;
			if ((ee = lp_scan<uint32_t>(stmp, cb, first, upos, size_t(n_bound), stream)) != hipSuccess) return ee;
			hipLaunchKernelGGL(lp_unique_scatter_kernel, grid(n_bound), dim3(256), 0, stream, bound, ia, n_bound, first, upos, d_lists, step, c_items, c_bucket, c_key);
			uint32_t lastu[2];
			if ((ee = hipMemcpyAsync(&lastu[0], first + (n_bound - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
			if ((ee = hipMemcpyAsync(&lastu[1], upos + (n_bound - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
			if ((ee = hipStreamSynchronize(stream)) != hipSuccess) return ee;
			n_c = lastu[0] + lastu[1];
			stage("bind step: sorted, unique", n_bound, n_c);
			hipLaunchKernelGGL(lp_run_check_kernel, grid(n_c), dim3(256), 0, stream, c_key, n_c, d_list_flag);
			// The merge below looks its places up by bisection: the run has to be in key order -- which it is, unless the check just
			// flagged its list (whose outcome nobody reads, but whose entries must still land on places of their own).  One more
			// stable pass by the merge key: nothing moves in a consistent list.
			if (n_c) {
				hipLaunchKernelGGL(lp_iota_kernel, grid(n_c), dim3(256), 0, stream, i0, n_c);
				size_t sb = sort_bytes;
				if ((ee = rocprim::radix_sort_pairs(stmp, sb, c_key, k0, i0, i1, size_t(n_c), 0u, 64u, stream)) != hipSuccess) return ee;
				// (bound[] is dead: its room takes the run in key order)
				Item *const s_items = reinterpret_cast<Item *>(bound);
				uint32_t *const s_bucket = first;
				hipLaunchKernelGGL(lp_run_gather_kernel, grid(n_c), dim3(256), 0, stream, c_items, c_bucket, i1, n_c, s_items, s_bucket);
				if ((ee = hipMemcpyAsync(c_items, s_items, sizeof(Item) * n_c, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return ee;
				if ((ee = hipMemcpyAsync(c_bucket, s_bucket, sizeof(uint32_t) * n_c, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return ee;
				if ((ee = hipMemcpyAsync(c_key, k0, sizeof(unsigned long long) * n_c, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return ee;
			}
		}
		// merge: itm[cur ^ 1] holds the entries that stayed; the merged list goes back to itm[cur]
		hipLaunchKernelGGL(lp_merge_kernel, grid(size_t(n_stay) + n_c), dim3(256), 0, stream, itm[cur ^ 1], bkt[cur ^ 1], n_stay, c_items, c_bucket, c_key, n_c, itm[cur], bkt[cur]);
		n_cur = n_stay + n_c;
		stage("bind step: merged", n_cur, step);
		// consistency of the bound entries now in the lists that are sorted behind this step
		if (n_cur && step < 4u) {
			hipLaunchKernelGGL(lp_foot_flag_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], bkt[cur], n_cur, d_lists, result, pool, step, d_flag32);
			size_t fb = need;
			if ((ee = lp_scan<uint32_t>(tmp, fb, d_flag32, d_pos32, size_t(n_cur), stream)) != hipSuccess) return ee;
			uint32_t lastf[2];
			if ((ee = hipMemcpyAsync(&lastf[0], d_flag32 + (n_cur - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
			if ((ee = hipMemcpyAsync(&lastf[1], d_pos32 + (n_cur - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return ee;
			if ((ee = hipStreamSynchronize(stream)) != hipSuccess) return ee;
			const uint32_t n_foot = lastf[0] + lastf[1];
			arena.rewind(mark_step);
			LpFoot *const foot = arena.take<LpFoot>(n_foot);
			if (!foot) return hipErrorOutOfMemory;
			hipLaunchKernelGGL(lp_foot_scatter_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], bkt[cur], n_cur, d_flag32, d_pos32, result, pool, foot);
			hipLaunchKernelGGL(lp_foot_check_kernel, grid(n_foot), dim3(256), 0, stream, foot, n_foot, d_list_flag, debug ? d_cnt : nullptr);
			stage("bind step: consistency", n_foot, step);
		}
		return hipGetLastError();
	};

	for (uint32_t s = 0; s < 4; s++) { // amplicon_search.cpp:131-353
		LP_TRY(bind_step(s));
		if (s < 3) LP_TRY(cull(0));
	}
	if (any_probe) {
		LP_TRY(cull(1));
		LP_TRY(bind_step(4));
		LP_TRY(bind_step(5));
	}
	// ---- the last sort (by footprint, stable) and the pairing loops
	std::vector<uint32_t> flags(n_lists, 0);
	if (n_cur) {
		arena.rewind(mark_step);
		LpFinal *const fin = arena.take<LpFinal>(n_cur), *const fin_sorted = arena.take<LpFinal>(n_cur);
		unsigned long long *const k0 = arena.take<unsigned long long>(n_cur), *const k1 = arena.take<unsigned long long>(n_cur);
		uint32_t *const i0 = arena.take<uint32_t>(n_cur), *const i1 = arena.take<uint32_t>(n_cur), *const cnt = arena.take<uint32_t>(n_cur), *const off = arena.take<uint32_t>(n_cur);
		size_t sort_bytes = 0, scan_bytes = 0;
		LP_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, k0, k1, i0, i1, size_t(n_cur), 0u, 64u, stream));
		LP_TRY(lp_scan<uint32_t>(nullptr, scan_bytes, cnt, off, size_t(n_cur), stream));
		uint8_t *const stmp = arena.take<uint8_t>(std::max(sort_bytes, scan_bytes) + 256);
		if (!fin || !fin_sorted || !k0 || !k1 || !i0 || !i1 || !cnt || !off || !stmp) return hipErrorOutOfMemory;
		hipLaunchKernelGGL(lp_final_kernel, grid(n_cur), dim3(256), 0, stream, itm[cur], bkt[cur], n_cur, result, pool, fin, d_list_flag);
		uint32_t *ia = i0, *ib = i1;
		hipLaunchKernelGGL(lp_iota_kernel, grid(n_cur), dim3(256), 0, stream, ia, n_cur);
		for (int pass = 0; pass < 2; pass++) {
			hipLaunchKernelGGL(lp_final_key_kernel, grid(n_cur), dim3(256), 0, stream, fin, ia, n_cur, pass, k0);
			size_t sb = sort_bytes;
			LP_TRY(rocprim::radix_sort_pairs(stmp, sb, k0, k1, ia, ib, size_t(n_cur), 0u, 64u, stream));
			std::swap(ia, ib);
		}
		{
			std::vector<long long> low(n_lists, -(1ll << 40));
			LP_TRY(hipMemcpyAsync(d_slack, low.data(), sizeof(long long) * n_lists, hipMemcpyHostToDevice, stream));
			LP_TRY(hipStreamSynchronize(stream)); // (`low` leaves scope)
		}
		hipLaunchKernelGGL(lp_final_gather_kernel, grid(n_cur), dim3(256), 0, stream, fin, ia, n_cur, fin_sorted, d_slack);
		stage("last sort", n_cur);
		LpPairArgs pa;
		pa.e = fin_sorted, pa.n = n_cur, pa.lists = d_lists, pa.list_flag = d_list_flag, pa.list_slack = d_slack, pa.assays = assays, pa.pool = pool;
		pa.max_len = max_len, pa.single_primer = single_primer, pa.min_max_primer_clamp = min_max_primer_clamp;
		pa.count = cnt, pa.offset = off, pa.hits = nullptr;
		hipLaunchKernelGGL(lp_pair_kernel<false>, grid(n_cur), dim3(256), 0, stream, pa);
		size_t cb = scan_bytes;
		LP_TRY(lp_scan<uint32_t>(stmp, cb, cnt, off, size_t(n_cur), stream));
		uint32_t last[2];
		LP_TRY(hipMemcpyAsync(&last[0], cnt + (n_cur - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		LP_TRY(hipMemcpyAsync(&last[1], off + (n_cur - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		LP_TRY(hipStreamSynchronize(stream));
		const size_t n_hits = size_t(last[0]) + last[1];
		stage("hits counted", n_hits);
		if (n_hits) {
			Hit *d_hits = nullptr; // (hit records are 1.4 KB: their array is an allocation of its own, not part of the arena's reservation)
			LP_TRY(hipMalloc(reinterpret_cast<void **>(&d_hits), n_hits * sizeof(Hit)));
			pa.hits = d_hits;
			hipLaunchKernelGGL(lp_pair_kernel<true>, grid(n_cur), dim3(256), 0, stream, pa);
			hits_out.resize(n_hits);
			e = hipMemcpyAsync(hits_out.data(), d_hits, n_hits * sizeof(Hit), hipMemcpyDeviceToHost, stream);
			if (e == hipSuccess) e = hipStreamSynchronize(stream);
			(void)hipFree(d_hits);
			if (e != hipSuccess) return e;
		}
	}
	LP_TRY(hipMemcpyAsync(flags.data(), d_list_flag, sizeof(uint32_t) * n_lists, hipMemcpyDeviceToHost, stream));
	LP_TRY(hipStreamSynchronize(stream));
	uint32_t why[3] = {0, 0, 0}; // lists handed back because of: a step's run out of target_loc order, crossing footprints, a footprint summing to zero
	for (uint32_t l = 0; l < n_lists; l++)
		if (flags[l]) {
			h_flagged[lists[l].group] = 1;
			for (int b = 0; b < 3; b++) why[b] += (flags[l] >> b) & 1u;
		}
	if (debug) fprintf(stderr, "long_pair_device: lists handed back: %u run order, %u crossing footprints, %u zero-sum footprint\n", why[0], why[1], why[2]);
	arena.rewind(mark0);
#undef LP_TRY
	return hipGetLastError();
}

// =====================================================================================================================
// The lists of probe / padlock assays of windowed targets (hybrid(), probe_search.cpp:67-230; padlock(),
// padlock_search.cpp:62-361, on the hash overload's lists, bind_oligo.cpp:124-454): nothing is culled there -- every seed
// is bound, the bound ones of a list are split by (strand, arm), stable-sorted by oligo_info::operator< (loc_5, loc_3, Tm
// descending: a strict weak order, tntblast.h:230-242) and reduced to one per footprint (bind_oligo.cpp:431-453); a probe
// assay reports every entry of its two probe lists, a padlock assay every (upstream, downstream) pair of one strand whose
// gap is 0 .. max_len (MIPS) bases.  All lists of a pass together: the bound seeds are picked out, sorted with three stable
// radix passes, reduced, and one thread per entry counts its hits, then writes them where the counts' prefix sum says -- the
// order of the reference's loops.
// =====================================================================================================================
namespace {

struct LhEntry { // a bound seed of a probe / padlock list
	int32_t loc5, loc3;
	int32_t rec;
	uint32_t sub;      // list << 2 | sublist (0 minus upstream, 1 minus downstream / probe, 2 plus upstream, 3 plus downstream / probe)
	uint32_t tm_desc;
};

__global__ __launch_bounds__(256) void lh_flag_kernel(const Item *kept, uint32_t n_kept, const uint32_t *g_first, const uint32_t *g_list, uint32_t n_groups, const int32_t *result,
                                                     uint32_t *flag, uint32_t *list_of)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_kept; i += gridDim.x * blockDim.x) {
		uint32_t lo = 0, hi = n_groups; // last group whose first seed is at or before i
		while (hi - lo > 1u) {
			const uint32_t mid = (lo + hi) >> 1;
			if (g_first[mid] <= i) lo = mid;
			else hi = mid;
		}
		const uint32_t l = g_list[lo];
		flag[i] = (l != 0xFFFFFFFFu && i < g_first[lo + 1u] && result[i] >= 0) ? 1u : 0u;
		list_of[i] = l;
	}
}
__global__ __launch_bounds__(256) void lh_gather_kernel(const Item *kept, uint32_t n_kept, const uint32_t *flag, const uint32_t *pos, const uint32_t *list_of, const int32_t *result,
                                                       const BoundRec *pool, LhEntry *out)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_kept; i += gridDim.x * blockDim.x) {
		if (!flag[i]) continue;
		const Item it = kept[i];
		const BoundRec &r = pool[result[i]];
		const uint32_t sub = ((it.info & ITEM_MINUS) ? 0u : 2u) + (item_role(it) == 1u ? 0u : 1u); // (the reverse oligo is the upstream arm)
		out[pos[i]] = LhEntry{r.loc5, r.loc3, result[i], (list_of[i] << 2) | sub, lp_float_desc(r.tm)};
	}
}
__global__ __launch_bounds__(256) void lh_key_kernel(const LhEntry *e, const uint32_t *idx, uint32_t n, int pass, unsigned long long *key)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const LhEntry &x = e[idx ? idx[k] : k];
		key[k] = pass == 0 ? (unsigned long long)x.tm_desc : (pass == 1 ? (unsigned long long)lp_bias(x.loc3) : (((unsigned long long)x.sub << 32) | (unsigned long long)lp_bias(x.loc5)));
	}
}
__global__ __launch_bounds__(256) void lh_unique_flag_kernel(const LhEntry *e, const uint32_t *idx, uint32_t n, uint32_t *first)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t f = 1u;
		if (k) {
			const LhEntry &a = e[idx[k - 1u]], &b = e[idx[k]];
			if (a.sub == b.sub && a.loc5 == b.loc5 && a.loc3 == b.loc3) f = 0u;
		}
		first[k] = f;
	}
}
__global__ __launch_bounds__(256) void lh_unique_scatter_kernel(const LhEntry *e, const uint32_t *idx, uint32_t n, const uint32_t *first, const uint32_t *pos, LhEntry *out, uint32_t *out_sub)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		if (first[k]) {
			out[pos[k]] = e[idx[k]];
			out_sub[pos[k]] = e[idx[k]].sub;
		}
}

// smallest and largest footprint (loc3 - loc5) of every sublist: they bound the stretch of a downstream list a ligation can come from
__global__ __launch_bounds__(256) void lh_span_kernel(const LhEntry *e, uint32_t n, int *span_lo, int *span_hi)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const LhEntry x = e[k];
		atomicMin(span_lo + x.sub, x.loc3 - x.loc5);
		atomicMax(span_hi + x.sub, x.loc3 - x.loc5);
	}
}

struct LhPairArgs {
	const LhEntry *e;
	const uint32_t *sub; // e[k].sub, ascending: the sublists are runs
	const int *span_lo, *span_hi; // [sublist]
	uint32_t n;
	const LpList *lists;
	const AssayDev *assays;
	const BoundRec *pool;
	int max_len;         // of a padlock pair's gap (0 unless MIPS)
	uint32_t *count;
	const uint32_t *offset;
	Hit *hits;
};
__device__ inline uint32_t lh_lower(const uint32_t *sub, uint32_t n, uint32_t key) // first k with sub[k] >= key
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (sub[mid] < key) lo = mid + 1u;
		else hi = mid;
	}
	return lo;
}
// One thread per entry: a probe-list entry of a probe assay is a hit; an upstream arm of a padlock assay pairs with the
// downstream arms of its strand whose gap fits (the walk pair_group bounds by the footprints, here by bisection on loc5 and
// the gap test itself: the same pairs in the same order).
template <bool WRITE> __global__ __launch_bounds__(256) void lh_pair_kernel(LhPairArgs a)
{
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < a.n; k += gridDim.x * blockDim.x) {
		const LhEntry E = a.e[k];
		const LpList L = a.lists[E.sub >> 2];
		const AssayDev A = a.assays[L.assay];
		const uint32_t sl = E.sub & 3u;
		uint32_t hits = 0;
		Hit *out = WRITE ? a.hits + a.offset[k] : nullptr;
		if (A.kind == 2) { // hybrid(): the probe lists (1: minus strand, 3: plus strand)
			if (sl & 1u) {
				if (WRITE) {
					Hit &h = out[0];
					const BoundRec &b = a.pool[E.rec];
					init_hit(h, A, L.seq_index);
					h.probe_tm = b.tm, h.probe_dH = b.dH, h.probe_dS = b.dS;
					h.probe_mm = wrap8(b.mm), h.probe_gap = wrap8(b.gap);
					h.probe_start = E.loc5, h.probe_stop = E.loc3;
					h.probe_strand = sl == 1u ? 1 : 0;
					copy_text(h.probe_align, b);
				}
				hits = 1;
			}
		} else if (!(sl & 1u)) { // padlock(): this entry is an upstream arm U
			const bool minus = sl == 0u;
			const uint32_t d0 = lh_lower(a.sub, a.n, E.sub + 1u), d1 = lh_lower(a.sub, a.n, E.sub + 2u); // the downstream arms of its strand
			// a pair needs 0 <= gap <= max_len: that bounds the downstream arm's loc5 from both sides, on the minus strand through
			// the smallest and largest footprint of the list (the walk of pair_group, tnt_pair.hip.h); the list is in ascending loc5 order
			const long span_lo = a.span_lo[E.sub + 1u], span_hi = a.span_hi[E.sub + 1u];
			const long d5_lo = minus ? long(E.loc5) - 1 - long(a.max_len) - span_hi : long(E.loc3) + 1;
			const long d5_hi = minus ? long(E.loc5) - 1 - span_lo : long(E.loc3) + 1 + long(a.max_len);
			uint32_t lo = d0, hi = d1; // first downstream arm with loc5 >= d5_lo
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (long(a.e[mid].loc5) < d5_lo) lo = mid + 1u;
				else hi = mid;
			}
			for (uint32_t d = lo; d < d1; d++) {
				const LhEntry D = a.e[d];
				if (long(D.loc5) > d5_hi) break;
				const int len = minus ? (E.loc5 - D.loc3 - 1) : (D.loc5 - E.loc3 - 1);
				if (len < 0 || len > a.max_len) continue;
				if (WRITE) {
					Hit &h = out[hits];
					const BoundRec &bu = a.pool[E.rec], &bd = a.pool[D.rec];
					init_hit(h, A, L.seq_index);
					h.primer_strand = minus ? 1 : 0;
					h.amplicon_start = minus ? D.loc5 : E.loc5;
					h.amplicon_stop = minus ? E.loc3 : D.loc3;
					h.forward_tm = bd.tm, h.reverse_tm = bu.tm;
					h.forward_dH = bd.dH, h.reverse_dH = bu.dH;
					h.forward_dS = bd.dS, h.reverse_dS = bu.dS;
					h.forward_mm = wrap8(bd.mm), h.reverse_mm = wrap8(bu.mm);
					h.forward_gap = wrap8(bd.gap), h.reverse_gap = wrap8(bu.gap);
					h.forward_primer_clamp = wrap8(bd.a3);
					h.reverse_primer_clamp = wrap8(bu.a5);
					copy_text(h.forward_align, bd);
					copy_text(h.reverse_align, bu);
				}
				hits++;
			}
		}
		if (!WRITE) a.count[k] = hits;
	}
}

} // namespace

size_t long_hash_bytes(size_t n_kept, size_t n_bound, size_t n_lists) { return n_kept * 16 + n_bound * (2 * sizeof(LhEntry) + 48) + n_lists * 64 + (size_t(48) << 20); }

// The probe / padlock lists among h_groups (h_take[g] != 0): their hits, list after list, in the order of the reference's loops.
hipError_t long_hash_device(const Item *kept, uint32_t n_kept, const Group *h_groups, uint32_t n_groups, const uint8_t *h_take, const int32_t *result, const BoundRec *pool,
                            const AssayDev *assays, const AssayDev *h_assays, const uint32_t *h_unit_target, uint32_t first_seq_index, int pair_max_len, std::vector<Hit> &hits_out,
                            DeviceArena &arena, hipStream_t stream)
{
	hits_out.clear();
	std::vector<uint32_t> g_first(n_groups + 1), g_list(n_groups);
	std::vector<LpList> lists;
	for (uint32_t g = 0; g < n_groups; g++) {
		g_first[g] = h_groups[g].first;
		g_list[g] = 0xFFFFFFFFu;
		if (!h_take[g] || !(h_groups[g].count & ~GROUP_ALIVE)) continue;
		g_list[g] = uint32_t(lists.size());
		lists.push_back(LpList{h_groups[g].assay & GROUP_ASSAY_MASK, 0u, int32_t(first_seq_index + h_unit_target[h_groups[g].target]), g});
	}
	g_first[n_groups] = n_kept;
	if (lists.empty() || n_kept == 0) return hipSuccess;
	if (lists.size() >= (size_t(1) << 30)) return hipErrorInvalidValue;
	hipError_t e;
#define LH_TRY(X)                               \
	do {                                        \
		if ((e = (X)) != hipSuccess) return e;  \
	} while (0)
	const size_t mark0 = arena.mark();
	auto grid = [](size_t m) { return dim3(unsigned(std::min<size_t>(8192, std::max<size_t>(1, (m + 255) / 256)))); };
	uint32_t *const d_gfirst = arena.take<uint32_t>(n_groups + 1), *const d_glist = arena.take<uint32_t>(n_groups);
	LpList *const d_lists = arena.take<LpList>(lists.size());
	uint32_t *const flag = arena.take<uint32_t>(n_kept), *const pos = arena.take<uint32_t>(n_kept), *const list_of = arena.take<uint32_t>(n_kept);
	size_t scan_bytes = 0;
	LH_TRY(lp_scan<uint32_t>(nullptr, scan_bytes, flag, pos, size_t(n_kept), stream));
	uint8_t *const tmp = arena.take<uint8_t>(scan_bytes + 256);
	if (!d_gfirst || !d_glist || !d_lists || !flag || !pos || !list_of || !tmp) return hipErrorOutOfMemory;
	LH_TRY(hipMemcpyAsync(d_gfirst, g_first.data(), sizeof(uint32_t) * (n_groups + 1), hipMemcpyHostToDevice, stream));
	LH_TRY(hipMemcpyAsync(d_glist, g_list.data(), sizeof(uint32_t) * n_groups, hipMemcpyHostToDevice, stream));
	LH_TRY(hipMemcpyAsync(d_lists, lists.data(), sizeof(LpList) * lists.size(), hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(lh_flag_kernel, grid(n_kept), dim3(256), 0, stream, kept, n_kept, d_gfirst, d_glist, n_groups, result, flag, list_of);
	size_t sb = scan_bytes;
	LH_TRY(lp_scan<uint32_t>(tmp, sb, flag, pos, size_t(n_kept), stream));
	uint32_t last[2];
	LH_TRY(hipMemcpyAsync(&last[0], flag + (n_kept - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	LH_TRY(hipMemcpyAsync(&last[1], pos + (n_kept - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	LH_TRY(hipStreamSynchronize(stream));
	const uint32_t n_b = last[0] + last[1];
	if (n_b) {
		LhEntry *const ent = arena.take<LhEntry>(n_b), *const uniq = arena.take<LhEntry>(n_b);
		unsigned long long *const k0 = arena.take<unsigned long long>(n_b), *const k1 = arena.take<unsigned long long>(n_b);
		uint32_t *const i0 = arena.take<uint32_t>(n_b), *const i1 = arena.take<uint32_t>(n_b), *const first = arena.take<uint32_t>(n_b), *const upos = arena.take<uint32_t>(n_b), *const usub = arena.take<uint32_t>(n_b);
		size_t sort_bytes = 0, scan2 = 0;
		LH_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, k0, k1, i0, i1, size_t(n_b), 0u, 64u, stream));
		LH_TRY(lp_scan<uint32_t>(nullptr, scan2, first, upos, size_t(n_b), stream));
		uint8_t *const stmp = arena.take<uint8_t>(std::max(sort_bytes, scan2) + 256);
		if (!ent || !uniq || !k0 || !k1 || !i0 || !i1 || !first || !upos || !usub || !stmp) return hipErrorOutOfMemory;
		hipLaunchKernelGGL(lh_gather_kernel, grid(n_kept), dim3(256), 0, stream, kept, n_kept, flag, pos, list_of, result, pool, ent);
		// oligo_info::operator< per sublist, stable in list order: Tm descending, loc_3, (list, sublist, loc_5), least significant first
		uint32_t *ia = i0, *ib = i1;
		hipLaunchKernelGGL(lp_iota_kernel, grid(n_b), dim3(256), 0, stream, ia, n_b);
		for (int pass = 0; pass < 3; pass++) {
			hipLaunchKernelGGL(lh_key_kernel, grid(n_b), dim3(256), 0, stream, ent, ia, n_b, pass, k0);
			size_t b2 = sort_bytes;
			LH_TRY(rocprim::radix_sort_pairs(stmp, b2, k0, k1, ia, ib, size_t(n_b), 0u, 64u, stream));
			std::swap(ia, ib);
		}
		hipLaunchKernelGGL(lh_unique_flag_kernel, grid(n_b), dim3(256), 0, stream, ent, ia, n_b, first);
		size_t b3 = scan2;
		LH_TRY(lp_scan<uint32_t>(stmp, b3, first, upos, size_t(n_b), stream));
		hipLaunchKernelGGL(lh_unique_scatter_kernel, grid(n_b), dim3(256), 0, stream, ent, ia, n_b, first, upos, uniq, usub);
		LH_TRY(hipMemcpyAsync(&last[0], first + (n_b - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		LH_TRY(hipMemcpyAsync(&last[1], upos + (n_b - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		LH_TRY(hipStreamSynchronize(stream));
		const uint32_t n_u = last[0] + last[1];
		// hits: counted, then written (first / upos are free again)
		const size_t n_sub = lists.size() * 4 + 4;
		int *const span_lo = arena.take<int>(n_sub), *const span_hi = arena.take<int>(n_sub);
		if (!span_lo || !span_hi) return hipErrorOutOfMemory;
		{
			std::vector<int> init(n_sub, INT32_MAX);
			LH_TRY(hipMemcpyAsync(span_lo, init.data(), sizeof(int) * n_sub, hipMemcpyHostToDevice, stream));
			LH_TRY(hipStreamSynchronize(stream));
			std::fill(init.begin(), init.end(), INT32_MIN);
			LH_TRY(hipMemcpyAsync(span_hi, init.data(), sizeof(int) * n_sub, hipMemcpyHostToDevice, stream));
			LH_TRY(hipStreamSynchronize(stream));
		}
		hipLaunchKernelGGL(lh_span_kernel, grid(n_u), dim3(256), 0, stream, uniq, n_u, span_lo, span_hi);
		LhPairArgs pa;
		pa.span_lo = span_lo, pa.span_hi = span_hi;
		pa.e = uniq, pa.sub = usub, pa.n = n_u, pa.lists = d_lists, pa.assays = assays, pa.pool = pool, pa.max_len = pair_max_len;
		pa.count = first, pa.offset = upos, pa.hits = nullptr;
		hipLaunchKernelGGL(lh_pair_kernel<false>, grid(n_u), dim3(256), 0, stream, pa);
		size_t b4 = scan2;
		LH_TRY(lp_scan<uint32_t>(stmp, b4, first, upos, size_t(n_u), stream));
		LH_TRY(hipMemcpyAsync(&last[0], first + (n_u - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		LH_TRY(hipMemcpyAsync(&last[1], upos + (n_u - 1u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		LH_TRY(hipStreamSynchronize(stream));
		const size_t n_hits = size_t(last[0]) + last[1];
		if (n_hits) {
			Hit *d_hits = nullptr;
			LH_TRY(hipMalloc(reinterpret_cast<void **>(&d_hits), n_hits * sizeof(Hit)));
			pa.hits = d_hits;
			hipLaunchKernelGGL(lh_pair_kernel<true>, grid(n_u), dim3(256), 0, stream, pa);
			hits_out.resize(n_hits);
			e = hipMemcpyAsync(hits_out.data(), d_hits, n_hits * sizeof(Hit), hipMemcpyDeviceToHost, stream);
			if (e == hipSuccess) e = hipStreamSynchronize(stream);
			(void)hipFree(d_hits);
			if (e != hipSuccess) return e;
		}
	}
	arena.rewind(mark0);
#undef LH_TRY
	return hipGetLastError();
}

// 64-bit keys with 32-bit values, ascending (the groups of the cache replay in (target, assay) order, tnt_pair.hip.h).  With
// tmp == nullptr only the scratch size is reported.
hipError_t sort_pairs_u64_u32(void *tmp, size_t &tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                              size_t n, hipStream_t stream)
{
	return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 64u, stream);
}

} // namespace tnt
