// Microbenchmark (not part of the product), round 4: how gfx950 issues MIXED streams of "fast" (v_add_u32 / v_xor_b32 ...: ~2.3 cycles per
// wave64 instruction in a homogeneous stream) and "slow" (v_perm_b32, v_pk_max_u16, v_add3_u32 ...: ~4.3) VALU instructions, with and without
// SALU / dependent operands in between.  Each kernel runs one 16-instruction pattern per loop trip over 16 independent accumulators at 8 waves
// per SIMD; reported: nominal-2.4-GHz SIMD cycles per PATTERN and what the homogeneous costs would predict.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 2048;
#define F(i) "v_add_u32 %" #i ", %" #i ", %16\n\t"
#define X(i) "v_xor_b32 %" #i ", %" #i ", %17\n\t"
#define S(i) "v_perm_b32 %" #i ", %" #i ", %16, %17\n\t"
#define P(i) "v_pk_max_u16 %" #i ", %" #i ", %16\n\t"
#define T(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n\t"
#define M(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n\t"
#define SA "s_add_u32 s4, s4, 1\n\t"
#define SN "s_nop 0\n\t"
// dependent fast op: every instruction reads the previous one's result (accumulator 0 only)
#define D "v_add_u32 %0, %0, %16\n\t"
#define DS "v_perm_b32 %0, %0, %16, %17\n\t"

#define KPAT(NAME, STR)                                                                         \
__global__ __launch_bounds__(256) void k_##NAME(int* out, int seed)                            \
{                                                                                              \
	int a[16], b = seed + threadIdx.x, c = seed * 3 + 1;                                       \
	for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;                                \
	for (int it = 0; it < kIters; ++it) {                                                      \
		asm volatile(STR : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
			"+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c) : "s4", "vcc", "scc");     \
	}                                                                                          \
	int s = 0; for (int i = 0; i < 16; ++i) s += a[i];                                         \
	if (s == 0x12345678) out[0] = s;                                                           \
}
KPAT(f16, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15))
KPAT(s16, S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15))
KPAT(fs_alt, F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7) F(8) S(9) F(10) S(11) F(12) S(13) F(14) S(15))
KPAT(fs_grp8, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15))
KPAT(fs_grp2, F(0) F(1) S(2) S(3) F(4) F(5) S(6) S(7) F(8) F(9) S(10) S(11) F(12) F(13) S(14) S(15))
KPAT(ffs, F(0) F(1) S(2) F(3) F(4) S(5) F(6) F(7) S(8) F(9) F(10) S(11) F(12) F(13) S(14) F(15))
KPAT(fffs, F(0) F(1) F(2) S(3) F(4) F(5) F(6) S(7) F(8) F(9) F(10) S(11) F(12) F(13) F(14) S(15))
KPAT(fp_alt, F(0) P(1) F(2) P(3) F(4) P(5) F(6) P(7) F(8) P(9) F(10) P(11) F(12) P(13) F(14) P(15))
KPAT(fp_grp8, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15))
KPAT(ft_alt, F(0) T(1) F(2) T(3) F(4) T(5) F(6) T(7) F(8) T(9) F(10) T(11) F(12) T(13) F(14) T(15))
KPAT(fm_alt, F(0) M(1) F(2) M(3) F(4) M(5) F(6) M(7) F(8) M(9) F(10) M(11) F(12) M(13) F(14) M(15))
KPAT(fx_alt, F(0) X(1) F(2) X(3) F(4) X(5) F(6) X(7) F(8) X(9) F(10) X(11) F(12) X(13) F(14) X(15))
KPAT(f16_salu4, F(0) F(1) F(2) F(3) SA F(4) F(5) F(6) F(7) SA F(8) F(9) F(10) F(11) SA F(12) F(13) F(14) F(15) SA)
KPAT(f16_salu8, F(0) F(1) SA F(2) F(3) SA F(4) F(5) SA F(6) F(7) SA F(8) F(9) SA F(10) F(11) SA F(12) F(13) SA F(14) F(15) SA)
KPAT(f16_salu16, F(0) SA F(1) SA F(2) SA F(3) SA F(4) SA F(5) SA F(6) SA F(7) SA F(8) SA F(9) SA F(10) SA F(11) SA F(12) SA F(13) SA F(14) SA F(15) SA)
KPAT(s16_salu8, S(0) S(1) SA S(2) S(3) SA S(4) S(5) SA S(6) S(7) SA S(8) S(9) SA S(10) S(11) SA S(12) S(13) SA S(14) S(15) SA)
KPAT(f16_nop8, F(0) F(1) SN F(2) F(3) SN F(4) F(5) SN F(6) F(7) SN F(8) F(9) SN F(10) F(11) SN F(12) F(13) SN F(14) F(15) SN)
KPAT(dep16, D D D D D D D D D D D D D D D D)
KPAT(dep_s16, DS DS DS DS DS DS DS DS DS DS DS DS DS DS DS DS)
KPAT(dep_pairs, F(0) F(0) F(1) F(1) F(2) F(2) F(3) F(3) F(4) F(4) F(5) F(5) F(6) F(6) F(7) F(7))
KPAT(dep_fs, F(0) S(0) F(1) S(1) F(2) S(2) F(3) S(3) F(4) S(4) F(5) S(5) F(6) S(6) F(7) S(7))
KPAT(salu16, SA SA SA SA SA SA SA SA SA SA SA SA SA SA SA SA)

struct Entry { const char* name; void (*k)(int*, int); double predicted; };
#define E(NAME, NF, NS) { #NAME, k_##NAME, (NF) * 2.3 + (NS) * 4.3 }
static Entry entries[] = {
	E(f16, 16, 0), E(s16, 0, 16), E(fs_alt, 8, 8), E(fs_grp8, 8, 8), E(fs_grp2, 8, 8), E(ffs, 11, 5), E(fffs, 12, 4), E(fp_alt, 8, 8), E(fp_grp8, 8, 8),
	E(ft_alt, 8, 8), E(fm_alt, 8, 8), E(fx_alt, 16, 0), E(f16_salu4, 16, 0), E(f16_salu8, 16, 0), E(f16_salu16, 16, 0), E(s16_salu8, 0, 16), E(f16_nop8, 16, 0),
	E(dep16, 16, 0), E(dep_s16, 0, 16), E(dep_pairs, 16, 0), E(dep_fs, 8, 8), E(salu16, 0, 0),
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
	const double patterns_per_simd = (double)blocks * 4 / 1024.0 * kIters;
	return ms * 1e-3 * 2.4e9 / patterns_per_simd;
}

int main()
{
	int* out; (void)hipMalloc(&out, 4);
	printf("%-14s %9s %9s %9s %9s   (nominal-2.4-GHz SIMD cycles per 16-instruction pattern; predicted = 2.3 per fast + 4.3 per slow)\n", "pattern", "w=4", "w=6", "w=8", "predicted");
	for (const Entry& e : entries) {
		printf("%-14s", e.name);
		for (int w : {4, 6, 8}) printf(" %9.1f", run(e, out, w));
		printf(" %9.1f\n", e.predicted); fflush(stdout);
	}
	return 0;
}
