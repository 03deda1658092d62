// Microbenchmark (not part of the product), round 4: what v_cndmask_b32 really costs on gfx950 (round 2's table showed 23 cycles for a stream of
// v_cndmask ... vcc), and the cost of the sequences the Canny candidate list is built from.  Same method as valu_rate_bench3: a
// dependency-light stream (8 accumulators, 16 instances per trip) at W waves per SIMD; SIMD cycles per wave64 instruction at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 2048;

#define KGEN(NAME, PRE, BODY, PERTRIP)                                                          \
__global__ __launch_bounds__(256) void k_##NAME(int* out, int seed)                            \
{                                                                                              \
	int a[8], b = seed + threadIdx.x, c = seed * 3 + 1;                                        \
	for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                 \
	PRE;                                                                                       \
	for (int it = 0; it < kIters; ++it) {                                                      \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) { BODY; }                                \
		_Pragma("unroll") for (int i = 0; i < 8; ++i) { BODY; }                                \
	}                                                                                          \
	int s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                          \
	if (s == 0x12345678) out[0] = s;                                                           \
}
#define A1(STR) asm volatile(STR : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s4", "s5", "s6", "s7")

KGEN(add, , A1("v_add_u32 %0, %0, %1"), 16)
// vcc written once before the loop (a real lane mask), then only read
KGEN(cnd_vcc_set, asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(b), "v"(c) : "vcc"), A1("v_cndmask_b32 %0, %0, %1, vcc"), 16)
// the lane mask in an SGPR pair
KGEN(cnd_sgpr, asm volatile("v_cmp_gt_i32 s[4:5], %0, %1" :: "v"(b), "v"(c) : "s4", "s5"), A1("v_cndmask_b32_e64 %0, %0, %1, s[4:5]"), 16)
// inline-constant operands (the form the compiler emits for flags)
KGEN(cnd_const, asm volatile("v_cmp_gt_i32 s[4:5], %0, %1" :: "v"(b), "v"(c) : "s4", "s5"), A1("v_cndmask_b32_e64 %0, 0, 2, s[4:5]"), 16)
// compare + select pairs, as in the NMS loop
KGEN(cmp_cnd, , A1("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc"), 32)
KGEN(cmp_e32, , A1("v_cmp_lt_u32 vcc, %0, %1\n\tv_add_u32 %0, %0, %2"), 32)
KGEN(cmp_e64, , A1("v_cmp_lt_u32 s[4:5], %0, %1\n\tv_add_u32 %0, %0, %2"), 32)
KGEN(cmp_sdwa, , A1("v_cmp_gt_u32_sdwa s[4:5], %0, %1 src0_sel:WORD_1 src1_sel:DWORD\n\tv_add_u32 %0, %0, %2"), 32)
KGEN(cmp_u16, , A1("v_cmp_gt_u16 vcc, %0, %1\n\tv_add_u32 %0, %0, %2"), 32)
// arithmetic select: mask = ashr(x, 31); r = (mask & y) -- three fast ops
KGEN(arith_sel, , A1("v_ashrrev_i32 %0, 31, %0\n\tv_and_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %2"), 48)
// rank of a lane in a ballot (mask in s[4:5]) and the pair with its compare
KGEN(mbcnt_pair, asm volatile("v_cmp_gt_i32 s[4:5], %0, %1" :: "v"(b), "v"(c) : "s4", "s5"), A1("v_mbcnt_lo_u32_b32 %0, s4, %0\n\tv_mbcnt_hi_u32_b32 %0, s5, %0"), 32)
KGEN(slot_seq, , A1("v_cmp_gt_u32_sdwa s[4:5], %0, %1 src0_sel:WORD_1 src1_sel:DWORD\n\tv_mov_b32 %0, %2\n\tv_mbcnt_lo_u32_b32 %0, s4, %0\n\tv_mbcnt_hi_u32_b32 %0, s5, %0\n\tv_add_u32 %0, %0, %0"), 80)
KGEN(max_u16, , A1("v_max_u16 %0, %0, %1"), 16)
KGEN(mul_lo_u16, , A1("v_mul_lo_u16 %0, %0, %1"), 16)
KGEN(mul_u24, , A1("v_mul_u32_u24 %0, %0, %1"), 16)
KGEN(perm, , A1("v_perm_b32 %0, %0, %1, %2"), 16)
KGEN(pk_max_u16, , A1("v_pk_max_u16 %0, %0, %1"), 16)
KGEN(bfi, , A1("v_bfi_b32 %0, %1, %0, %2"), 16)
KGEN(add3, , A1("v_add3_u32 %0, %0, %1, %2"), 16)
KGEN(readfirstlane, , A1("v_readfirstlane_b32 s4, %0\n\tv_add_u32 %0, %0, %1"), 32)
KGEN(mov_sgpr, asm volatile("s_mov_b32 s4, 17" ::: "s4"), A1("v_mov_b32 %0, s4"), 16)
KGEN(lshl_or, , A1("v_lshl_or_b32 %0, %0, 16, %1"), 16)
KGEN(and_or, , A1("v_and_or_b32 %0, %0, %1, %2"), 16)
KGEN(xor3, , A1("v_xor_b32 %0, %0, %1"), 16)
KGEN(sub_lit, , A1("v_sub_u32 %0, 0x8000800, %0"), 16)
KGEN(or_lit, , A1("v_or_b32 %0, 0x1000100, %0"), 16)
KGEN(add_inl, , A1("v_add_u32 %0, 2, %0"), 16)

struct Entry { const char* name; void (*k)(int*, int); int perTrip; };
#define E(NAME, N) { #NAME, k_##NAME, N }
static Entry entries[] = {
	E(add, 16), E(cnd_vcc_set, 16), E(cnd_sgpr, 16), E(cnd_const, 16), E(cmp_cnd, 32), E(cmp_e32, 32), E(cmp_e64, 32), E(cmp_sdwa, 32), E(cmp_u16, 32),
	E(arith_sel, 48), E(mbcnt_pair, 32), E(slot_seq, 80), E(max_u16, 16), E(mul_lo_u16, 16), E(mul_u24, 16), E(perm, 16), E(pk_max_u16, 16), E(bfi, 16), E(add3, 16),
	E(readfirstlane, 32), E(mov_sgpr, 16), E(lshl_or, 16), E(and_or, 16), E(xor3, 16), E(sub_lit, 16), E(or_lit, 16), E(add_inl, 16),
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
	printf("%-16s %7s %7s %7s   (SIMD cycles per wave64 instruction @2.4 GHz nominal; sequences: per instruction of the sequence)\n", "opcode", "w=2", "w=4", "w=8");
	for (const Entry& e : entries) {
		printf("%-16s", e.name);
		for (int w : {2, 4, 8}) printf(" %7.2f", run(e, out, w));
		printf("\n");
	}
	return 0;
}
