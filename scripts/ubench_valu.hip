// Micro-benchmark: issue cost of the integer VALU / LDS instructions the pre-filter and the seed kernel are made of, with
// several wavefronts per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o exp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define BODY64(X) REP8(REP8(X))

template <int WHICH> __global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
	__shared__ uint32_t tab[2048];
	for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = i * 2654435761u + seed;
	__syncthreads();
	uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
	uint32_t b = seed | 1, c = seed * 7 + 3;
	for (int it = 0; it < iters; it++) {
		if (WHICH == 0) { BODY64(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 1) { BODY64(asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 2) { BODY64(asm volatile("v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %1, %1, %8, %9\n v_max3_i32 %2, %2, %8, %9\n v_max3_i32 %3, %3, %8, %9\n v_max3_i32 %4, %4, %8, %9\n v_max3_i32 %5, %5, %8, %9\n v_max3_i32 %6, %6, %8, %9\n v_max3_i32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
		if (WHICH == 3) { BODY64(asm volatile("v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_add_i16 %3, %3, %8 clamp\n v_pk_add_i16 %4, %4, %8 clamp\n v_pk_add_i16 %5, %5, %8 clamp\n v_pk_add_i16 %6, %6, %8 clamp\n v_pk_add_i16 %7, %7, %8 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 4) { BODY64(asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 5) { BODY64(asm volatile("v_pk_max_i16 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_max_i16 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_max_i16 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_max_i16 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_max_i16 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_max_i16 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_max_i16 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_max_i16 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 6) { BODY64(asm volatile("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 7) { BODY64(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 8) { BODY64(asm volatile("v_alignbit_b32 %0, %0, %8, 3\n v_alignbit_b32 %1, %1, %8, 3\n v_alignbit_b32 %2, %2, %8, 3\n v_alignbit_b32 %3, %3, %8, 3\n v_alignbit_b32 %4, %4, %8, 3\n v_alignbit_b32 %5, %5, %8, 3\n v_alignbit_b32 %6, %6, %8, 3\n v_alignbit_b32 %7, %7, %8, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 9 || WHICH == 10 || WHICH == 11 || WHICH == 15 || WHICH == 16) {
			// LDS reads: 9 = random dword per lane in a 8 KB table, 10 = one address for the whole wavefront (broadcast), 11 = lane-consecutive
			uint32_t ad0, ad1, ad2, ad3;
			if (WHICH == 9) ad0 = (a0 & 2047u) * 4, ad1 = (a1 & 2047u) * 4, ad2 = (a2 & 2047u) * 4, ad3 = (a3 & 2047u) * 4;
			else if (WHICH == 15) { const uint32_t c = threadIdx.x & 3u; ad0 = ((a0 & 511u) * 4 + c) * 4, ad1 = ((a1 & 511u) * 4 + c) * 4, ad2 = ((a2 & 511u) * 4 + c) * 4, ad3 = ((a3 & 511u) * 4 + c) * 4; } // four interleaved copies: a lane only meets the lanes of its own residue
			else if (WHICH == 16) { const uint32_t c = threadIdx.x & 1u; ad0 = ((a0 & 1023u) * 2 + c) * 4, ad1 = ((a1 & 1023u) * 2 + c) * 4, ad2 = ((a2 & 1023u) * 2 + c) * 4, ad3 = ((a3 & 1023u) * 2 + c) * 4; }
			else if (WHICH == 10) ad0 = (uint32_t(it) & 2047u) * 4, ad1 = ad0 ^ 64, ad2 = ad0 ^ 128, ad3 = ad0 ^ 256;
			else ad0 = (threadIdx.x & 63) * 4, ad1 = ad0 + 256, ad2 = ad0 + 512, ad3 = ad0 + 1024;
			REP8(REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n s_waitcnt lgkmcnt(0)" : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3) : "memory");))
			a0 += a4 & 1, a1 += a5 & 1, a2 += a6 & 1, a3 += a7 & 1;
		}
		if (WHICH == 12) { BODY64(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
		if (WHICH == 13) { BODY64(asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
		if (WHICH == 14) { BODY64(asm volatile("v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n v_max_i16 %4, %4, %8\n v_max_i16 %5, %5, %8\n v_max_i16 %6, %6, %8\n v_max_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ tab[threadIdx.x];
}

template <int W> double run(uint32_t *out, int blocks, int iters, int per_iter)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, 4, 1u);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double wave_instr = double(blocks) * 4 * iters * per_iter;
	return wave_instr / (ms * 1e-3);
}

int main()
{
	uint32_t *out;
	const char *names[] = {"v_add_u32", "v_max_i32", "v_max3_i32", "v_pk_add_i16 clamp", "v_pk_max_i16", "v_pk_max_i16 op_sel", "v_add_u32_sdwa WORD_1", "v_lshl_add_u32", "v_alignbit_b32",
	                       "ds_read_b32 random", "ds_read_b32 broadcast", "ds_read_b32 consecutive", "v_pk_add_u16", "v_add3_u32", "v_max_i16", "ds_read_b32 random, 4 copies by lane", "ds_read_b32 random, 2 copies by lane"};
	for (int waves_per_simd : {1, 2, 4, 8}) {
		const int blocks = 256 * waves_per_simd; // 4 wavefronts per workgroup = one per SIMD of a CU
		hipMalloc(&out, size_t(blocks) * 256 * 4);
		double r[17];
		const int it = 2000;
		r[0] = run<0>(out, blocks, it, 512), r[1] = run<1>(out, blocks, it, 512), r[2] = run<2>(out, blocks, it, 512), r[3] = run<3>(out, blocks, it, 512);
		r[4] = run<4>(out, blocks, it, 512), r[5] = run<5>(out, blocks, it, 512), r[6] = run<6>(out, blocks, it, 512), r[7] = run<7>(out, blocks, it, 512);
		r[8] = run<8>(out, blocks, it, 512), r[9] = run<9>(out, blocks, it, 256), r[10] = run<10>(out, blocks, it, 256), r[11] = run<11>(out, blocks, it, 256);
		r[12] = run<12>(out, blocks, it, 512), r[13] = run<13>(out, blocks, it, 512), r[14] = run<14>(out, blocks, it, 512);
		r[15] = run<15>(out, blocks, it, 256), r[16] = run<16>(out, blocks, it, 256);
		printf("== %d wavefront(s) per SIMD (%d workgroups of 256)\n", waves_per_simd, blocks);
		for (int i = 0; i < 17; i++)
			printf("%-38s %8.1f G wave-instr/s  = %.2f SIMD-cycles per wave-instruction at 2.4 GHz, %.2f relative to v_add_u32\n", names[i], r[i] * 1e-9, 1024 * 2.4e9 / r[i], r[0] / r[i]);
		hipFree(out);
	}
	return 0;
}
