// tnt_arena.h -- one grow-only device allocation that the long-target path carves its arrays from (tnt_host.hip,
// tnt_sort.hip): some twenty hipMalloc / hipFree pairs per search otherwise, each a device synchronisation.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace tnt {

struct DeviceArena {
	uint8_t *p = nullptr;
	size_t cap = 0, used = 0;

	// make room for `bytes` in total; only valid while nothing is taken (the arrays would move)
	hipError_t reserve(size_t bytes)
	{
		used = 0;
		if (bytes <= cap) return hipSuccess;
		release();
		const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), bytes);
		if (e == hipSuccess) cap = bytes;
		return e;
	}
	// the next `count` elements of T (256-byte aligned), or nullptr when the reservation was too small
	template <class T> T *take(size_t count)
	{
		const size_t at = (used + 255) & ~size_t(255), bytes = count * sizeof(T);
		if (at + bytes > cap) return nullptr;
		used = at + bytes;
		return reinterpret_cast<T *>(p + at);
	}
	size_t mark() const { return used; }
	void rewind(size_t m) { used = m; }
	void release()
	{
		if (p) (void)hipFree(p);
		p = nullptr, cap = used = 0;
	}
	~DeviceArena() { release(); }
	DeviceArena() = default;
	DeviceArena(const DeviceArena &) = delete;
	DeviceArena &operator=(const DeviceArena &) = delete;
};

} // namespace tnt
