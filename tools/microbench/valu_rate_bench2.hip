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

K32(add, "v_add_u32 %0, %0, %1")
K32(sub, "v_sub_u32 %0, %0, %1")
K32(and2, "v_and_b32 %0, %0, %1")
K32(or2, "v_or_b32 %0, %0, %1")
K32(xor2, "v_xor_b32 %0, %0, %1")
K32(lshl, "v_lshlrev_b32 %0, 1, %0")
K32(lshr, "v_lshrrev_b32 %0, 1, %0")
K32(ashr, "v_ashrrev_i32 %0, 1, %0")
K32(max_i32, "v_max_i32 %0, %0, %1")
K32(min_u32, "v_min_u32 %0, %0, %1")
K32(mov, "v_mov_b32 %0, %1")
K32(not1, "v_not_b32 %0, %0")
K32(bfrev, "v_bfrev_b32 %0, %0")
K32(ffbl, "v_ffbl_b32 %0, %0")
K32(bcnt, "v_bcnt_u32_b32 %0, %1, %0")
K32(mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
K32(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
K32(cmp, "v_cmp_lt_i32 vcc, %0, %1")
K32(cmp_u16, "v_cmp_lt_u16 vcc, %0, %1")
K32(cmp_e64, "v_cmp_lt_i32_e64 s[10:11], %0, %1")
K32(mul24, "v_mul_u32_u24 %0, %0, %1")
K32(mulhi24, "v_mul_hi_u32_u24 %0, %0, %1")
K32(mullo, "v_mul_lo_u32 %0, %0, %1")
K32(mad24, "v_mad_u32_u24 %0, %0, %1, %2")
K32(add_u16, "v_add_u16 %0, %0, %1")
K32(sub_u16, "v_sub_u16 %0, %0, %1")
K32(max_u16, "v_max_u16 %0, %0, %1")
K32(lshl_u16, "v_lshlrev_b16 %0, 1, %0")
K32(mad_u16, "v_mad_u16 %0, %0, %1, %2")
K32(max3, "v_max3_i32 %0, %0, %1, %2")
K32(med3, "v_med3_i32 %0, %0, %1, %2")
K32(sad_u32, "v_sad_u32 %0, %0, %1, %2")
K32(sad_u16, "v_sad_u16 %0, %0, %1, %2")
K32(sad_u8, "v_sad_u8 %0, %0, %1, %2")
K32(msad_u8, "v_msad_u8 %0, %0, %1, %2")
K32(dot4_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
K32(dot2_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
K32(alignbit, "v_alignbit_b32 %0, %0, %1, 16")
K32(alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
K32(perm, "v_perm_b32 %0, %0, %1, %2")
K32(bfe, "v_bfe_u32 %0, %0, 8, 8")
K32(bfi, "v_bfi_b32 %0, %0, %1, %2")
K32(and_or, "v_and_or_b32 %0, %0, %1, %2")
K32(or3, "v_or3_b32 %0, %0, %1, %2")
K32(add3, "v_add3_u32 %0, %0, %1, %2")
K32(lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
K32(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
K32(add_lshl, "v_add_lshl_u32 %0, %0, %1, 1")
K32(xad, "v_xad_u32 %0, %0, %1, %2")
K32(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
K32(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
K32(pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
K32(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
K32(pk_min_u16, "v_pk_min_u16 %0, %0, %1")
K32(pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
K32(pk_mad_u16_k, "v_pk_mad_u16 %0, %0, 2, %1 op_sel_hi:[1,0,1]")
K32(pk_mul_lo, "v_pk_mul_lo_u16 %0, %0, %1")
K32(pk_lshr, "v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]")
K32(pk_ashr, "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]")
K32(pack_b32, "v_pack_b32_f16 %0, %0, %1")
K32(cvt_pk_u16, "v_cvt_pk_u16_u32 %0, %0, %1")
K32(dpp_mov_rowshr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
K32(dpp_mov_waveshr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
K32(dpp_add_rowshr, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
K32(dpp_add_bcast15, "v_add_u32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf")
K32(sdwa_add_u32, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
K32(sdwa_add_u16_hi, "v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")
K32(sdwa_add_bytes, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2")
K32(readlane, "v_readlane_b32 s10, %0, 63")
K32(writelane, "v_writelane_b32 %0, s4, 3")
K32(mix_add_pk, "v_add_u32 %0, %0, %1\n\tv_pk_add_u16 %0, %0, %2")
K32(mix_add_max3, "v_add_u32 %0, %0, %1\n\tv_max3_i32 %0, %0, %1, %2")
K32(mix_add_and, "v_add_u32 %0, %0, %1\n\tv_and_b32 %0, %0, %2")
K32(snop, "s_nop 0")
K32(salu_add, "s_add_u32 s10, s10, s11")
K32(mix_valu_salu, "v_add_u32 %0, %0, %1\n\ts_add_u32 s10, s10, s11")
K32(mix_pk_salu, "v_pk_add_u16 %0, %0, %1\n\ts_add_u32 s10, s10, s11")
K64(qsad, "v_qsad_pk_u16_u8 %0, %1, %2, %0")
K64(mqsad, "v_mqsad_pk_u16_u8 %0, %1, %2, %0")
K64(lshl64, "v_lshlrev_b64 %0, 1, %0")
K64(add64, "v_lshl_add_u64 %0, %0, 0, %1")

struct Entry { const char* name; void (*k)(int*, int); int perTrip; };
#define E(N) { #N, k_##N, 16 },
#define E2(N) { #N, k_##N, 32 },
static Entry entries[] = {
	E(add) E(sub) E(and2) E(or2) E(xor2) E(lshl) E(lshr) E(ashr) E(max_i32) E(min_u32) E(mov) E(not1) E(bfrev) E(ffbl) E(bcnt) E(mbcnt)
	E(cndmask) E(cmp) E(cmp_u16) E(cmp_e64) E(mul24) E(mulhi24) E(mullo) E(mad24) E(add_u16) E(sub_u16) E(max_u16) E(lshl_u16) E(mad_u16)
	E(max3) E(med3) E(sad_u32) E(sad_u16) E(sad_u8) E(msad_u8) E(dot4_u8) E(dot2_u16) E(alignbit) E(alignbyte) E(perm) E(bfe) E(bfi)
	E(and_or) E(or3) E(add3) E(lshl_or) E(lshl_add) E(add_lshl) E(xad)
	E(pk_add_u16) E(pk_sub_i16) E(pk_sub_u16_clamp) E(pk_max_i16) E(pk_min_u16) E(pk_mad_u16) E(pk_mad_u16_k) E(pk_mul_lo) E(pk_lshr) E(pk_ashr)
	E(pack_b32) E(cvt_pk_u16)
	E(dpp_mov_rowshr) E(dpp_mov_waveshr) E(dpp_add_rowshr) E(dpp_add_bcast15) E(sdwa_add_u32) E(sdwa_add_u16_hi) E(sdwa_add_bytes)
	E(readlane) E(writelane) E2(mix_add_pk) E2(mix_add_max3) E2(mix_add_and) E(snop) E(salu_add) E2(mix_valu_salu) E2(mix_pk_salu)
	E(qsad) E(mqsad) E(lshl64) E(add64)
};

static double run(const Entry& e, int* out, int wavesPerSimd)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * wavesPerSimd; // blocks of 4 waves: wavesPerSimd blocks per CU
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
	printf("%-20s %8s %8s %8s   (SIMD cycles per wave64 instruction @2.4 GHz nominal; 1, 3, 8 waves per SIMD)\n", "opcode", "w=1", "w=3", "w=8");
	for (const Entry& e : entries) {
		const double c1 = run(e, out, 1), c3 = run(e, out, 3), c8 = run(e, out, 8);
		printf("%-20s %8.2f %8.2f %8.2f\n", e.name, c1, c3, c8);
	}
	return 0;
}
