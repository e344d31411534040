// tnt_wbind_inst.hip -- instances of the wavefront-per-window bind kernel (tnt_wbind.hip.h): rows per lane 1 (oligos up to
// 64 bases), 2 (up to 128), 4 / 8 / 16 (up to 256 / 512 / 1 024: the lengths only this kernel binds); the throughput form
// and the unit-level form (FULL: any request kind, dp_max / n_max reported).
#include "tnt_wbind.hip.h"

namespace tnt {

#define X(N)                                                                                                             \
	void launch_wbind_##N(const BindArgs &a, int blocks, hipStream_t stream)                                             \
	{                                                                                                                    \
		hipLaunchKernelGGL((wbind_kernel<N, false>), dim3(blocks), dim3(64 * wb_waves(N)), 0, stream, a);                 \
	}                                                                                                                    \
	void launch_wbind_full_##N(const BindArgs &a, int blocks, hipStream_t stream)                                        \
	{                                                                                                                    \
		hipLaunchKernelGGL((wbind_kernel<N, true>), dim3(blocks), dim3(64 * wb_waves(N)), 0, stream, a);                  \
	}                                                                                                                    \
	void launch_wbind_dink_##N(const BindArgs &a, int blocks, bool full, hipStream_t stream) /* --dinkelbach: one wavefront per workgroup */ \
	{                                                                                                                    \
		if (full) hipLaunchKernelGGL((wbind_kernel<N, true, true>), dim3(blocks), dim3(64), 0, stream, a);                \
		else hipLaunchKernelGGL((wbind_kernel<N, false, true>), dim3(blocks), dim3(64), 0, stream, a);                    \
	}
TNT_WBIND_LADDER(X)
#undef X

} // namespace tnt
