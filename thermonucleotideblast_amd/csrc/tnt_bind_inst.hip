// tnt_bind_inst.hip -- the throughput instances of the bind kernel, one per oligo-length bucket.
//
// The build compiles this file several times with -DTNT_INST_LIST="G(36) P(36) E(22) ..." (a share of the
// instances each: G = general ladder, P = plain-window ladder, E = exact length + plain windows, F / D = pre-filter for plain / any windows, U / V = pre-filter on queues regrouped by oligo, plain / any windows, S = the pre-filter's second tier, T = the same for the general buckets) so they build
// in parallel; compiled once without the macro it provides all of them.
#include "tnt_bind.hip.h"
#include "tnt_filter.hip.h"
#include "tnt_ufilter.hip.h"
#include "tnt_sbound.hip.h"

#ifndef TNT_INST_LIST
#define TNT_INST_LIST TNT_LMAX_LADDER(G) TNT_LMAX_LADDER(P) TNT_LONG_LADDER(G) TNT_LONG_LADDER(P) TNT_EXACT_LADDER(E) TNT_FILTER_LADDER(F) TNT_LMAX_LADDER(D) TNT_LONG_LADDER(F) TNT_LONG_LADDER(D) TNT_FILTER_LADDER(U) TNT_LMAX_LADDER(V) TNT_SBOUND_LADDER(S) TNT_SBOUND_DEGEN_LADDER(T)
#endif

namespace tnt {

#define TNT_LAUNCHER(NAME, N, EXACT, PLAIN)                                                                          \
	void NAME(const BindArgs &a, int blocks, hipStream_t stream)                                                     \
	{                                                                                                                \
		hipLaunchKernelGGL((bind_kernel<N, false, EXACT, PLAIN>), dim3(blocks), dim3(bind_threads_for(N, PLAIN)), 0, stream, a); \
	}
#define F(N)                                                                                                         \
	void launch_filter_##N(const FilterArgs2 &a, int blocks, hipStream_t stream)                                     \
	{                                                                                                                \
		hipLaunchKernelGGL((filter_kernel<N, false>), dim3(blocks), dim3(256), 0, stream, a);                         \
	}
#define D(N)                                                                                                         \
	void launch_filter_degen_##N(const FilterArgs2 &a, int blocks, hipStream_t stream)                               \
	{                                                                                                                \
		hipLaunchKernelGGL((filter_kernel<N, true>), dim3(blocks), dim3(256), 0, stream, a);                          \
	}
#define U(N)                                                                                                         \
	void launch_ufilter_##N(const UFilterArgs &a, int blocks, hipStream_t stream)                                    \
	{                                                                                                                \
		hipLaunchKernelGGL((ufilter_kernel<N, false>), dim3(blocks), dim3(256), 0, stream, a);                        \
	}
#define V(N)                                                                                                         \
	void launch_ufilter_degen_##N(const UFilterArgs &a, int blocks, hipStream_t stream)                              \
	{                                                                                                                \
		hipLaunchKernelGGL((ufilter_kernel<N, true>), dim3(blocks), dim3(256), 0, stream, a);                         \
	}
#define S(N)                                                                                                         \
	void launch_sbound_##N(const SBoundArgs &a, int blocks, hipStream_t stream)                                      \
	{                                                                                                                \
		hipLaunchKernelGGL((sbound_kernel<N, (N >= 17)>), dim3(blocks), dim3(SB_THREADS), 0, stream, a);                                \
	}
#define T(N)                                                                                                         \
	void launch_sbound_degen_##N(const SBoundDegenArgs &a, int blocks, hipStream_t stream)                           \
	{                                                                                                                \
		hipLaunchKernelGGL((sbound_degen_kernel<N>), dim3(blocks), dim3(SBD_THREADS), 0, stream, a);                  \
	}
#define G(N) TNT_LAUNCHER(launch_bind_##N, N, false, false)
#define P(N) TNT_LAUNCHER(launch_bind_plain_##N, N, false, true)
#define E(N) TNT_LAUNCHER(launch_bind_exact_##N, N, true, true)
TNT_INST_LIST
#undef G
#undef P
#undef E
#undef F
#undef D
#undef U
#undef V
#undef S
#undef T

} // namespace tnt
