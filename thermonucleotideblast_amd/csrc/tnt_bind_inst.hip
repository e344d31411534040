// tnt_bind_inst.hip -- the throughput instances of the bind kernel, one per oligo-length bucket.
//
// The build compiles this file several times with -DTNT_INST_LIST="X(16) X(22) ..." (a share of
// TNT_LMAX_LADDER each) so the instances build in parallel; compiled once without the macro it provides all of them.
#include "tnt_bind.hip.h"

#ifndef TNT_INST_LIST
#define TNT_INST_LIST TNT_LMAX_LADDER(X)
#endif

namespace tnt {

#define X(N)                                                                                                        \
	void launch_bind_##N(const BindArgs &a, int blocks, hipStream_t stream)                                         \
	{                                                                                                               \
		hipLaunchKernelGGL((bind_kernel<N, false>), dim3(blocks), dim3(BIND_THREADS), 0, stream, a);                \
	}
TNT_INST_LIST
#undef X

} // namespace tnt
