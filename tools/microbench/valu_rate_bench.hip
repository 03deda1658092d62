// Microbenchmark (not part of the product): issue rate of the integer VALU instructions the Canny tile kernel is made of,
// on gfx950.  Each kernel runs a long chain-free stream of one opcode (8 independent accumulators) with 8 waves per SIMD;
// reported: SIMD cycles per wave64 instruction (4.0 = full rate) assuming 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int kIters = 4096;

#define DEFINE_KERNEL(NAME, ASMSTR)                                                            \
__global__ __launch_bounds__(256) void k_##NAME(int* out, int seed)                            \
{                                                                                              \
	int a[8], b = seed + threadIdx.x, c = seed * 3 + 1;                                        \
	for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                 \
	for (int it = 0; it < kIters; ++it) {                                                      \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c)); \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c)); \
	}                                                                                          \
	int s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                          \
	if (s == 0x12345678) out[0] = s;                                                           \
}

DEFINE_KERNEL(add, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(max3, "v_max3_i32 %0, %0, %1, %2")
DEFINE_KERNEL(sad, "v_sad_u32 %0, %0, %1, %2")
DEFINE_KERNEL(alignbit, "v_alignbit_b32 %0, %0, %1, 31")
DEFINE_KERNEL(mul24, "v_mul_u32_u24 %0, %0, %1")
DEFINE_KERNEL(mullo, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL(lshladd, "v_lshl_add_u32 %0, %0, 2, %1")
DEFINE_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
DEFINE_KERNEL(cmp_cndmask, "v_cmp_lt_i32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp_cndmask_sgpr, "v_cmp_lt_i32_e64 s[10:11], %0, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]")
DEFINE_KERNEL(max2, "v_max_i32 %0, %0, %1")
DEFINE_KERNEL(and2, "v_and_b32 %0, %0, %1")
DEFINE_KERNEL(sub2, "v_sub_u32 %0, %0, %1")
DEFINE_KERNEL(lshl2, "v_lshlrev_b32 %0, 1, %0")
DEFINE_KERNEL(bfi, "v_bfi_b32 %0, %0, %1, %2")
DEFINE_KERNEL(cmp, "v_cmp_lt_i32 vcc, %0, %1")
DEFINE_KERNEL(cmp_sgpr, "v_cmp_lt_i32_e64 s[10:11], %0, %1")
DEFINE_KERNEL(bfe, "v_bfe_u32 %0, %0, 8, 8")
DEFINE_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(writelane, "v_writelane_b32 %0, s4, 3")
DEFINE_KERNEL(mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_KERNEL(sdwa_add, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
DEFINE_KERNEL(pk_add, "v_pk_add_u16 %0, %0, %1")
DEFINE_KERNEL(pk_max, "v_pk_max_u16 %0, %0, %1")
DEFINE_KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
DEFINE_KERNEL(dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")

template <typename K> void run(const char* name, K kern, int* out)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	const double instr_per_simd = (double)blocks * 4 / 1024.0 * kIters * 16; // wave-instructions each SIMD executes
	printf("%-12s %7.3f ms  %5.2f SIMD cycles per wave64 instruction\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
	int* out; (void)hipMalloc(&out, 4);
#define RUN(N) run(#N, k_##N, out);
	RUN(cndmask_sgpr) RUN(cmp_cndmask) RUN(cmp_cndmask_sgpr) RUN(max2) RUN(and2) RUN(sub2) RUN(lshl2) RUN(bfi)
	RUN(add) RUN(max3) RUN(sad) RUN(alignbit) RUN(mul24) RUN(mullo) RUN(lshladd) RUN(cndmask) RUN(cmp) RUN(cmp_sgpr) RUN(bfe) RUN(and_or)
	RUN(writelane) RUN(mad24) RUN(sdwa_add) RUN(pk_add) RUN(pk_max) RUN(perm) RUN(dpp_mov)
	return 0;
}
