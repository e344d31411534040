// Issue rate of VALU instructions on gfx950 (MI355X), per SIMD, at 1 / 2 / 4 wavefronts per SIMD: whole-kernel
// time (HIP events) x shader clock / wave-instructions issued per SIMD.  The shader clock is measured in the same
// launch (s_memtime ticks of one wave over the event time), because it drops under sustained VALU load.  Every candidate for the bind kernel's fill loop is measured with
// independent register chains (8 chains, 64 instructions per loop trip).
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(S) S S S S S S S S
// I(r) expands to one instruction working on chain register r; operands %8 / %9 are loop invariant
#define CHAINS(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")

#define DEFINE(NAME, I)                                                                                            \
	__global__ void NAME(long long *out, int iters, int x, int y)                                                  \
	{                                                                                                              \
		int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
		asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_mov_b64 s[10:11], vcc" ::"v"(r0), "v"(y) : "vcc", "s10", "s11");                                        \
		const long long t0 = __builtin_readcyclecounter();                                                         \
		for (int it = 0; it < iters; it++)                                                                         \
			asm volatile(REP8(CHAINS(I)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "v"(y) : "vcc", "s10", "s11"); \
		const long long t1 = __builtin_readcyclecounter();                                                         \
		if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                 \
		if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0x7fffffff) out[1] = 1;                                      \
	}

#define I_FMA(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define I_SUB(r) "v_sub_u32 " r ", " r ", %8\n"
#define I_ADD(r) "v_add_u32 " r ", " r ", %8\n"
#define I_AND(r) "v_and_b32 " r ", " r ", %8\n"
#define I_OR(r) "v_or_b32 " r ", " r ", %8\n"
#define I_XOR(r) "v_xor_b32 " r ", " r ", %8\n"
#define I_LSHL(r) "v_lshlrev_b32 " r ", 1, " r "\n"
#define I_LSHR(r) "v_lshrrev_b32 " r ", 1, " r "\n"
#define I_ASHR(r) "v_ashrrev_i32 " r ", 31, " r "\n"
#define I_MAXI(r) "v_max_i32 " r ", " r ", %8\n"
#define I_MINI(r) "v_min_i32 " r ", " r ", %8\n"
#define I_MAXU(r) "v_max_u32 " r ", " r ", %8\n"
#define I_MAXF(r) "v_max_f32 " r ", " r ", %8\n"
#define I_MAX3(r) "v_max3_i32 " r ", " r ", %8, %9\n"
#define I_MED3(r) "v_med3_i32 " r ", " r ", %8, %9\n"
#define I_ALIGN(r) "v_alignbit_b32 " r ", " r ", %8, 31\n"
#define I_BFE(r) "v_bfe_u32 " r ", " r ", %8, 4\n"
#define I_BFI(r) "v_bfi_b32 " r ", " r ", %8, %9\n"
#define I_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define I_MUL24(r) "v_mul_u32_u24 " r ", " r ", %8\n"
#define I_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n"
#define I_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 2, %8\n"
#define I_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 1, %8\n"
#define I_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9\n"
#define I_MOV(r) "v_mov_b32 " r ", %8\n"
#define I_CMPEQ(r) "v_cmp_eq_u32 vcc, " r ", %8\n"
#define I_CMPGT(r) "v_cmp_gt_i32 vcc, " r ", %8\n"
#define I_CMPEQ_S(r) "v_cmp_eq_u32 s[10:11], " r ", %8\n"
#define I_ADDC(r) "v_addc_co_u32 " r ", vcc, " r ", " r ", vcc\n"
#define I_CMPADDC(r) "v_cmp_eq_u32 vcc, " r ", %8\n v_addc_co_u32 " r ", vcc, " r ", " r ", vcc\n"
#define I_SUBCO(r) "v_sub_co_u32 " r ", vcc, " r ", %8\n"
#define I_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define I_PKSUB(r) "v_pk_sub_i16 " r ", " r ", %8\n"
#define I_PKMAX(r) "v_pk_max_i16 " r ", " r ", %8\n"
#define I_PKADDU(r) "v_pk_add_u16 " r ", " r ", %8\n"
#define I_PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define I_SAD(r) "v_sad_u32 " r ", " r ", %8, %9\n"
#define I_SUBF(r) "v_sub_f32 " r ", " r ", %8\n"
#define I_ADDF(r) "v_add_f32 " r ", " r ", %8\n"
#define I_MULF(r) "v_mul_f32 " r ", " r ", %8\n"
#define I_MINF(r) "v_min_f32 " r ", " r ", %8\n"
#define I_MAX3F(r) "v_max3_f32 " r ", " r ", %8, %9\n"
#define I_CMPGTF(r) "v_cmp_gt_f32 vcc, " r ", %8\n"
#define I_CVTIF(r) "v_cvt_f32_i32 " r ", " r "\n"
#define I_SUBDEP(r) "v_sub_u32 %0, %0, %8\n"
#define I_MAXI16(r) "v_max_i16 " r ", " r ", %8\n"
#define I_SUBREV(r) "v_subrev_u32 " r ", " r ", %8\n"
#define I_DOT4(r) "v_dot4_i32_i8 " r ", " r ", %8, %9\n"
#define I_MULLO(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define I_ADDLSHL(r) "v_add_lshl_u32 " r ", " r ", %8, 1\n"
#define I_XAD(r) "v_xad_u32 " r ", " r ", %8, %9\n"
#define I_CNDMASK_S(r) "v_cndmask_b32_e64 " r ", " r ", %8, s[10:11]\n"
#define I_CMPCND(r) "v_cmp_gt_i32 vcc, " r ", %8\n v_cndmask_b32 " r ", " r ", %9, vcc\n"
#define I_CNDMASK_K(r) "v_cndmask_b32 " r ", %8, %9, vcc\n"
#define I_NOT(r) "v_not_b32 " r ", " r "\n"
#define I_BFREV(r) "v_bfrev_b32 " r ", " r "\n"
#define I_ADDSDWA(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_ADDSDWA_W(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_MAXI_VOP3(r) "v_max_i32_e64 " r ", " r ", %8\n"
#define I_SUB_VOP3(r) "v_sub_u32_e64 " r ", " r ", %8\n"

#define LIST(X)                                                                                                    \
	X(fma, I_FMA) X(sub, I_SUB) X(add, I_ADD) X(and_, I_AND) X(or_, I_OR) X(xor_, I_XOR) X(lshl, I_LSHL) X(lshr, I_LSHR)   \
	X(ashr, I_ASHR) X(maxi, I_MAXI) X(mini, I_MINI) X(maxu, I_MAXU) X(maxf, I_MAXF) X(max3, I_MAX3) X(med3, I_MED3)        \
	X(align, I_ALIGN) X(bfe, I_BFE) X(bfi, I_BFI) X(mad24, I_MAD24) X(mul24, I_MUL24) X(add3, I_ADD3) X(lshladd, I_LSHLADD) \
	X(lshlor, I_LSHLOR) X(andor, I_ANDOR) X(mov, I_MOV) X(cmpeq, I_CMPEQ) X(cmpgt, I_CMPGT) X(cmpeq_s, I_CMPEQ_S)          \
	X(addc, I_ADDC) X(cmpaddc, I_CMPADDC) X(subco, I_SUBCO) X(cndmask, I_CNDMASK) X(pksub, I_PKSUB) X(pkmax, I_PKMAX)      \
	X(pkaddu, I_PKADDU) X(perm, I_PERM) X(sad, I_SAD) X(subf, I_SUBF) X(addf, I_ADDF) X(mulf, I_MULF) X(minf, I_MINF)       \
	X(max3f, I_MAX3F) X(cmpgtf, I_CMPGTF) X(cvtif, I_CVTIF) X(subdep, I_SUBDEP) X(maxi16, I_MAXI16) X(subrev, I_SUBREV)    \
	X(dot4, I_DOT4) X(mullo, I_MULLO) X(addlshl, I_ADDLSHL) X(xad, I_XAD) X(cndmask_s, I_CNDMASK_S) X(cmpcnd, I_CMPCND) X(cndmask_k, I_CNDMASK_K) X(not_, I_NOT) X(bfrev, I_BFREV) X(add_sdwa_b, I_ADDSDWA) X(add_sdwa_w, I_ADDSDWA_W) X(maxi_vop3, I_MAXI_VOP3) X(sub_vop3, I_SUB_VOP3)

#define X(NAME, I) DEFINE(k_##NAME, I)
LIST(X)
#undef X

typedef void (*kern)(long long *, int, int, int);
struct Entry {
	const char *name;
	kern k;
	int per_chain;
};
#define X(NAME, I) {#NAME, k_##NAME, 1},
static Entry entries[] = {LIST(X)};
#undef X

int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	long long *out;
	hipMalloc(&out, 64);
	const int iters = 8000;
	printf("%d CUs; shader cycles per wave-instruction per SIMD (clock GHz during the 4 w/SIMD run)\n%-12s %8s %8s %8s %8s %6s\n", cus, "instruction", "1 w/SIMD", "2 w/SIMD", "3 w/SIMD", "4 w/SIMD", "GHz");
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (const Entry &e : entries) {
		printf("%-12s", e.name);
		const int n_inst = (std::string(e.name) == "cmpaddc" || std::string(e.name) == "cmpcnd") ? 128 : 64;
		double ghz = 0;
		for (int wps : {1, 2, 3, 4}) {
			hipLaunchKernelGGL(e.k, dim3(cus * wps), dim3(256), 0, 0, out, 200, 3, 5);
			hipEventRecord(e0);
			hipLaunchKernelGGL(e.k, dim3(cus * wps), dim3(256), 0, 0, out, iters, 3, 5);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			long long cyc = 0;
			hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
			// one wave's tick count / its own share of the kernel time ~ clock; use the 1 w/SIMD launch (the wave runs the whole time)
			if (wps == 1) ghz = double(cyc) / (ms * 1e6);
			printf(" %8.2f", ms * 1e6 * ghz / (double(iters) * n_inst * wps));
		}
		printf(" %6.2f\n", ghz);
	}
	return 0;
}
