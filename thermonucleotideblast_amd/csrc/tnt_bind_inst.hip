// tnt_bind_inst.hip -- the throughput instances of the bind kernel, one per oligo-length bucket.
//
// The build compiles this file several times with -DTNT_INST_LIST="G(36) E(22) ..." (a share of TNT_LMAX_LADDER
// as G(N) and of TNT_EXACT_LADDER as E(N) each) so the instances build in parallel; compiled once without the
// macro it provides all of them.
#include "tnt_bind.hip.h"

#ifndef TNT_INST_LIST
#define TNT_INST_LIST TNT_LMAX_LADDER(G) TNT_EXACT_LADDER(E)
#endif

namespace tnt {

#define G(N)                                                                                                        \
	void launch_bind_##N(const BindArgs &a, int blocks, hipStream_t stream)                                         \
	{                                                                                                               \
		hipLaunchKernelGGL((bind_kernel<N, false, false>), dim3(blocks), dim3(bind_threads_for(N)), 0, stream, a);  \
	}
#define E(N)                                                                                                        \
	void launch_bind_exact_##N(const BindArgs &a, int blocks, hipStream_t stream)                                   \
	{                                                                                                               \
		hipLaunchKernelGGL((bind_kernel<N, false, true>), dim3(blocks), dim3(bind_threads_for(N)), 0, stream, a);   \
	}
TNT_INST_LIST
#undef G
#undef E

} // namespace tnt
