// tnt_sort.hip -- device side of the long-target path's list assembly: the seeds of all windows of the targets that
// are seeded in windows, ordered into per-(target, assay) lists in the reference's list order.
//
// The reference keeps one std::list per (target, assay) and sorts it by target_loc before the first cull
// (match_oligo_to_*_strand + cull_oligo_match, bind_oligo.cpp:84-122, amplicon_search.cpp:688; the hash-overload lists by
// diagonal, bind_oligo.cpp:157).  Round 2 assembled these lists on the host (counting sort + std::sort per list: 18 of
// 43 ms for a 5-Mb target against 96 primer pairs).  Here: one key kernel, two stable LSD radix sorts (rocPRIM: by the
// in-list key, then by the list), one gather.  Product code: nothing here includes or links anything under oracle/.
#include <hip/hip_runtime.h>
#include <cstring>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "tnt_dev.h"
#include "tnt_arena.h"
#include "tnt_seed.hip.h"

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

// 64-bit keys with 32-bit values, ascending (the groups of the cache replay in (target, assay) order, tnt_pair.hip.h).  With
// tmp == nullptr only the scratch size is reported.
hipError_t sort_pairs_u64_u32(void *tmp, size_t &tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                              size_t n, hipStream_t stream)
{
	return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 64u, stream);
}

} // namespace tnt
