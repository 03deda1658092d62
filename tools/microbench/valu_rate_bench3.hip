// Microbenchmark (not part of the product): issue cost of the gfx950 integer / packed-16 / byte-SAD / DPP / SDWA opcodes that
// candidate designs of the Canny tile kernel are made of.  Each kernel runs a dependency-light stream of one opcode (8
// independent accumulators, 16 instances per loop trip) at W waves per SIMD; reported: SIMD cycles per wave64 instruction at the
// measured shader clock (s_memtime-free: wall time x 2.4 GHz, so treat absolute numbers as +-10 %, ratios as exact).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

constexpr int kIters = 2048;

#define K32(NAME, ASMSTR)                                                                       \
__global__ __launch_bounds__(256) void k_##NAME(int* out, int seed)                            \
{                                                                                              \
	int a[8], b = seed + threadIdx.x, c = seed * 3 + 1;                                        \
	for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                 \
	for (int it = 0; it < kIters; ++it) {                                                      \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); \
	}                                                                                          \
	int s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                          \
	if (s == 0x12345678) out[0] = s;                                                           \
}
// 64-bit destination / source forms: %0 = 64-bit accumulator, %1 = 64-bit b, %2 = 32-bit c
#define K64(NAME, ASMSTR)                                                                       \
__global__ __launch_bounds__(256) void k_##NAME(int* out, int seed)                            \
{                                                                                              \
	uint64_t a[8], b = seed + threadIdx.x; int c = seed * 3 + 1;                               \
	for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                 \
	for (int it = 0; it < kIters; ++it) {                                                      \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c)); \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c)); \
	}                                                                                          \
	uint64_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                     \
	if (s == 0x12345678) out[0] = (int)s;                                                      \
}

#define K64b(NAME, ASMSTR)                                                                       \
__global__ __launch_bounds__(256) void k_##NAME(int* out, int seed)                            \
{                                                                                              \
	uint64_t a[8], b = seed + threadIdx.x;                                                     \
	for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                 \
	for (int it = 0; it < kIters; ++it) {                                                      \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b)); \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(a[i]) : "v"(b)); \
	}                                                                                          \
	uint64_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                     \
	if (s == 0x12345678) out[0] = (int)s;                                                      \
}
K32(add, "v_add_u32 %0, %0, %1")
K32(max3, "v_max3_i32 %0, %0, %1, %2")
K32(min_u16, "v_min_u16 %0, %0, %1")
K32(max_i16, "v_max_i16 %0, %0, %1")
K32(mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
K32(lshr_b16, "v_lshrrev_b16 %0, 1, %0")
K32(ashr_i16, "v_ashrrev_i16 %0, 1, %0")
K32(subrev_u32, "v_subrev_u32 %0, %0, %1")
K32(and_lit, "v_and_b32 %0, 0x00ff00ff, %0")
K32(and_sgpr, "v_and_b32 %0, s4, %0")
K32(add_sgpr, "v_add_u32 %0, s4, %0")
K32(add_co, "v_add_co_u32 %0, vcc, %0, %1")
K32(addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
K32(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
K32(lshl_v, "v_lshlrev_b32 %0, %1, %0")
K32(lshr_v, "v_lshrrev_b32 %0, %1, %0")
K32(lshl_1, "v_lshlrev_b32 %0, 1, %0")
K32(lshl_16, "v_lshlrev_b32 %0, 16, %0")
K32(mul_i24, "v_mul_i32_i24 %0, %0, %1")
K32(dot2_i16, "v_dot2_i32_i16 %0, %0, %1, %2")
K32(dot2_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
K32(sdwa_lshl_w1, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
K32(sdwa_and_w1, "v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
K32(add_f32, "v_add_f32 %0, %0, %1")
K32(max_f32, "v_max_f32 %0, %0, %1")
K32(add_f32_abs, "v_add_f32 %0, |%0|, |%1|")
K32(fma_f32, "v_fma_f32 %0, %0, %1, %2")
K32(fmac_f32, "v_fmac_f32 %0, %1, %2")
K32(add_f16, "v_add_f16 %0, %0, %1")
K32(max_f16, "v_max_f16 %0, %0, %1")
K32(pk_add_f16, "v_pk_add_f16 %0, %0, %1")
K32(pk_max_f16, "v_pk_max_f16 %0, %0, %1")
K32(pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
K32(pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
K32(cvt_f32_ub0, "v_cvt_f32_ubyte0 %0, %0")
K32(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
K32(cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
K32(xnor, "v_xnor_b32 %0, %0, %1")
K32(mov_dpp_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
K64b(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
K64b(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %0")
struct Entry { const char* name; void (*k)(int*, int); int perTrip; };
static Entry entries[] = {
	{ "add", k_add, 16 },
	{ "max3", k_max3, 16 },
	{ "min_u16", k_min_u16, 16 },
	{ "max_i16", k_max_i16, 16 },
	{ "mul_lo_u16", k_mul_lo_u16, 16 },
	{ "lshr_b16", k_lshr_b16, 16 },
	{ "ashr_i16", k_ashr_i16, 16 },
	{ "subrev_u32", k_subrev_u32, 16 },
	{ "and_lit", k_and_lit, 16 },
	{ "and_sgpr", k_and_sgpr, 16 },
	{ "add_sgpr", k_add_sgpr, 16 },
	{ "add_co", k_add_co, 16 },
	{ "addc_co", k_addc_co, 16 },
	{ "cndmask_vcc", k_cndmask_vcc, 16 },
	{ "lshl_v", k_lshl_v, 16 },
	{ "lshr_v", k_lshr_v, 16 },
	{ "lshl_1", k_lshl_1, 16 },
	{ "lshl_16", k_lshl_16, 16 },
	{ "mul_i24", k_mul_i24, 16 },
	{ "dot2_i16", k_dot2_i16, 16 },
	{ "dot2_u16", k_dot2_u16, 16 },
	{ "sdwa_lshl_w1", k_sdwa_lshl_w1, 16 },
	{ "sdwa_and_w1", k_sdwa_and_w1, 16 },
	{ "add_f32", k_add_f32, 16 },
	{ "max_f32", k_max_f32, 16 },
	{ "add_f32_abs", k_add_f32_abs, 16 },
	{ "fma_f32", k_fma_f32, 16 },
	{ "fmac_f32", k_fmac_f32, 16 },
	{ "add_f16", k_add_f16, 16 },
	{ "max_f16", k_max_f16, 16 },
	{ "pk_add_f16", k_pk_add_f16, 16 },
	{ "pk_max_f16", k_pk_max_f16, 16 },
	{ "pk_fma_f16", k_pk_fma_f16, 16 },
	{ "pk_mul_f16", k_pk_mul_f16, 16 },
	{ "cvt_f32_ub0", k_cvt_f32_ub0, 16 },
	{ "cvt_u32_f32", k_cvt_u32_f32, 16 },
	{ "cvt_pkrtz", k_cvt_pkrtz, 16 },
	{ "xnor", k_xnor, 16 },
	{ "mov_dpp_shr", k_mov_dpp_shr, 16 },
	{ "pk_add_f32", k_pk_add_f32, 16 },
	{ "pk_fma_f32", k_pk_fma_f32, 16 },
};

static double run(const Entry& e, int* out, int wavesPerSimd)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * wavesPerSimd;
	hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	const double instr_per_simd = (double)blocks * 4 / 1024.0 * kIters * e.perTrip;
	return ms * 1e-3 * 2.4e9 / instr_per_simd;
}

int main()
{
	int* out; (void)hipMalloc(&out, 4);
	printf("%-16s %7s %7s %7s %7s %7s %7s\n", "opcode", "w=2", "w=3", "w=4", "w=5", "w=6", "w=8");
	for (const Entry& e : entries) {
		printf("%-16s", e.name);
		for (int w : {2, 3, 4, 5, 6, 8}) printf(" %7.2f", run(e, out, w));
		printf("\n");
	}
	return 0;
}
