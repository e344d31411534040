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

#include "tnt_dev.h"
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

struct Scratch {
	void *p = nullptr;
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
	~Scratch()
	{
		if (p) (void)hipFree(p);
	}
};

} // namespace

// seeds[n] (device) -> items[n], bucket[n] (device, caller's buffers): the seeds as bind work items, list after list
// (bucket = rank of the target among the windowed targets * n_assays + assay), every list in the reference's order.
// Returns hipSuccess or the first error.
hipError_t long_sort_seeds(const SeedOut *seeds, uint32_t n, const uint32_t *target_rank, const AssayDev *assays, uint32_t n_assays,
                           uint32_t n_buckets, Item *items, uint32_t *bucket, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	Scratch key0, key1, idx0, idx1, bkt0, tmp;
	hipError_t e;
	if ((e = key0.alloc(size_t(n) * 8)) != hipSuccess || (e = key1.alloc(size_t(n) * 8)) != hipSuccess || (e = idx0.alloc(size_t(n) * 4)) != hipSuccess ||
	    (e = idx1.alloc(size_t(n) * 4)) != hipSuccess || (e = bkt0.alloc(size_t(n) * 4)) != hipSuccess)
		return e;
	unsigned long long *k0 = static_cast<unsigned long long *>(key0.p), *k1 = static_cast<unsigned long long *>(key1.p);
	uint32_t *i0 = static_cast<uint32_t *>(idx0.p), *i1 = static_cast<uint32_t *>(idx1.p), *b0 = static_cast<uint32_t *>(bkt0.p);
	const int blocks = int(std::min<size_t>(4096, (size_t(n) + 255) / 256));
	hipLaunchKernelGGL(long_keys_kernel, dim3(blocks), dim3(256), 0, stream, seeds, n, assays, k0, i0);
	size_t need = 0, need2 = 0;
	if ((e = rocprim::radix_sort_pairs(nullptr, need, k0, k1, i0, i1, size_t(n), 0u, 45u, stream)) != hipSuccess) return e;
	unsigned bits = 1;
	while (bits < 32 && (1ull << bits) < (unsigned long long)n_buckets) bits++;
	if ((e = rocprim::radix_sort_pairs(nullptr, need2, b0, bucket, i1, i0, size_t(n), 0u, bits, stream)) != hipSuccess) return e;
	if ((e = tmp.alloc(std::max(need, need2))) != hipSuccess) return e;
	// stable LSD sorts: by the in-list key, then by the list
	if ((e = rocprim::radix_sort_pairs(tmp.p, need, k0, k1, i0, i1, size_t(n), 0u, 45u, stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(long_bucket_kernel, dim3(blocks), dim3(256), 0, stream, seeds, i1, n, target_rank, n_assays, b0);
	if ((e = rocprim::radix_sort_pairs(tmp.p, need2, b0, bucket, i1, i0, size_t(n), 0u, bits, stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(long_gather_kernel, dim3(blocks), dim3(256), 0, stream, seeds, i0, n, items);
	if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e; // (the scratch arrays are freed on return)
	return hipGetLastError();
}

} // namespace tnt
