// Microbenchmark (not part of the product): the voting loop's instruction mix around ds_add_u32 -- what does each ingredient cost on the LDS-atomic floor?
//   A  ds_add only (addresses in registers)                               B  + the vote's 4 VALU (2 v_mad_i32_i24 with an SGPR operand, v_lshrrev, v_lshl_add) + 2 SALU per vote
//   C  B + s_waitcnt lgkmcnt(0) every 32 votes (what a scalar load's wait does: lgkmcnt counts LDS operations too, so the wave drains its own atomics)
//   D  B + a real s_load_dwordx16 pair per 32 votes, issued a block ahead, waited with lgkmcnt(0)      E  D with the wait placed after 16 of the block's 32 votes
// One 1024-thread workgroup per CU, 1216-row window (the 4K plan), bank = lane & 31.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRows = 1216;
constexpr int kBlocks = 192;   // blocks of 32 votes per wave

template <int MODE>
__global__ __launch_bounds__(1024) void bench(unsigned* out, const unsigned* list, int n)
{
	extern __shared__ unsigned hist[];
	for (int i = threadIdx.x; i < kRows * 32; i += 1024) hist[i] = 0;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	int nc = -(int)(20000 + lane * 300), ns = -(int)(50000 - lane * 500), K = (kRows - 1) * 65536 + 65535;   // rows stay inside the window: no clamp needed
	asm volatile("" : "+v"(nc), "+v"(ns), "+v"(K));
	unsigned inc = (lane & 32) ? 0x10000u : 1u;
	const unsigned lane4 = (lane & 31) * 4u;
	__syncthreads();
	typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
	const unsigned* base = list + (size_t)(blockIdx.x * 16 + wave) * kBlocks * 32;
	auto vote = [&](unsigned e) {
		if (MODE == 0) {
			unsigned ad = ((e * 2654435761u) >> 8) % kRows * 128u + lane4;   // (hoisted by the caller for mode A: see below)
			asm volatile("ds_add_u32 %0, %1" : : "v"(ad), "v"(inc) : "memory");
			return;
		}
		const int lx = (int)(e & 0xffffu), ly = (int)(e >> 16);
		int val, ad;
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(val) : "s"(lx), "v"(nc), "v"(K));
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(val) : "s"(ly), "v"(ns), "v"(val));
		const unsigned row = (unsigned)val >> 16;
		asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(ad) : "v"(row), "v"(lane4));
		asm volatile("ds_add_u32 %0, %1" : : "v"(ad), "v"(inc) : "memory");
	};
	auto vote2 = [&](unsigned e1, unsigned e2) {   // two votes, their instructions interleaved: two dependency chains per wave
		const int lx1 = (int)(e1 & 0xffffu), ly1 = (int)(e1 >> 16), lx2 = (int)(e2 & 0xffffu), ly2 = (int)(e2 >> 16);
		int v1, v2, a1, a2;
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(v1) : "s"(lx1), "v"(nc), "v"(K));
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(v2) : "s"(lx2), "v"(nc), "v"(K));
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(v1) : "s"(ly1), "v"(ns), "v"(v1));
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(v2) : "s"(ly2), "v"(ns), "v"(v2));
		unsigned r1, r2;
		asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r1) : "v"(v1));
		asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r2) : "v"(v2));
		asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(a1) : "v"(r1), "v"(lane4));
		asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(a2) : "v"(r2), "v"(lane4));
		asm volatile("ds_add_u32 %0, %1" : : "v"(a1), "v"(inc) : "memory");
		asm volatile("ds_add_u32 %0, %1" : : "v"(a2), "v"(inc) : "memory");
	};
	if (MODE == 10) {
		// K: no scalar unpacking -- two v_mul_i32_i24_sdwa pick lx / ly out of the packed SGPR entry (WORD_0 / WORD_1), v_add3 adds K: 5 VALU, 0 SALU
		u32x16 a0, a1, b0, b1;
		auto vote10 = [&](unsigned e) {
			int p1, p2, val, ad;
			asm volatile("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(p1) : "s"(e), "v"(nc));
			asm volatile("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(p2) : "s"(e), "v"(ns));
			asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(val) : "v"(p1), "v"(p2), "v"(K));
			unsigned row;
			asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(row) : "v"(val));
			asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(ad) : "v"(row), "v"(lane4));
			asm volatile("ds_add_u32 %0, %1" : : "v"(ad), "v"(inc) : "memory");
		};
#define LD10(r0, r1, blk) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r0), "=&s"(r1) : "s"(base + (size_t)(blk) * 32) : "memory")
#define WT10(r0, r1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1) : : "memory")
#define V10(r) _Pragma("unroll") for (int u = 0; u < 16; ++u) vote10(r[u])
		LD10(a0, a1, 0); WT10(a0, a1);
		for (int b = 0; b < kBlocks; b += 2) {
			LD10(b0, b1, b + 1);
			V10(a0); V10(a1); WT10(b0, b1);
			LD10(a0, a1, (b + 2) % kBlocks);
			V10(b0); V10(b1); WT10(a0, a1);
		}
	}
	else if (MODE == 9) {
		// J: no scalar unpacking at all -- the packed entry (ly << 16 | lx) goes straight into two v_dot2_i32_i16 against per-lane constants split into a high and a
		// low byte part (cos, sin have 17 bits): val = (dot2(e, hi) << 8) + (dot2(e, lo) + K); 5 VALU, 0 SALU
		u32x16 a0, a1, b0, b1;
		const int nch = nc >> 8, ncl = nc & 255, nsh = ns >> 8, nsl = ns & 255;
		unsigned Bhi = ((unsigned)(nsh & 0xffff) << 16) | (unsigned)(nch & 0xffff), Blo = ((unsigned)nsl << 16) | (unsigned)ncl;
		asm volatile("" : "+v"(Bhi), "+v"(Blo));
		auto vote9 = [&](unsigned e) {
			int d1, d2, val, ad;
			asm volatile("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d1) : "s"(e), "v"(Bhi));
			asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d2) : "s"(e), "v"(Blo), "v"(K));
			asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(val) : "v"(d1), "v"(d2));
			unsigned row;
			asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(row) : "v"(val));
			asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(ad) : "v"(row), "v"(lane4));
			asm volatile("ds_add_u32 %0, %1" : : "v"(ad), "v"(inc) : "memory");
		};
#define LD9(r0, r1, blk) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r0), "=&s"(r1) : "s"(base + (size_t)(blk) * 32) : "memory")
#define WT9(r0, r1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1) : : "memory")
#define V9(r) _Pragma("unroll") for (int u = 0; u < 16; ++u) vote9(r[u])
		LD9(a0, a1, 0); WT9(a0, a1);
		for (int b = 0; b < kBlocks; b += 2) {
			LD9(b0, b1, b + 1);
			V9(a0); V9(a1); WT9(b0, b1);
			LD9(a0, a1, (b + 2) % kBlocks);
			V9(b0); V9(b1); WT9(a0, a1);
		}
	}
	else if (MODE == 6 || MODE == 7 || MODE == 8) {
		// G: 4 VALU, no SALU (coordinates already unpacked in SGPRs)   H: 2 VALU (shift + lshl_add on a value in a register)   I: 1 VALU (lshl_add only)
		u32x16 r0, r1;
		asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r0), "=&s"(r1) : "s"(base) : "memory");
		int valr = K - 1234567; unsigned rowr = 77;
		asm volatile("" : "+v"(valr), "+v"(rowr));
		for (int b = 0; b < kBlocks; ++b) {
#pragma unroll
			for (int u = 0; u < 16; ++u) {
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const int lx = (int)((h ? r1[u] : r0[u]) & 0x3ffu);   // (one SALU for the mask: the loaded words are packed; G keeps it out of the VALU count only)
					int val, ad;
					if (MODE == 6) {
						asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(val) : "s"(lx), "v"(nc), "v"(K));
						asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(val) : "s"(lx), "v"(ns), "v"(val));
						const unsigned row = (unsigned)val >> 16;
						asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(ad) : "v"(row), "v"(lane4));
					}
					else if (MODE == 7) {
						unsigned row;
						asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(row) : "v"(valr));
						asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(ad) : "v"(row), "v"(lane4));
					}
					else asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(ad) : "v"(rowr), "v"(lane4));
					asm volatile("ds_add_u32 %0, %1" : : "v"(ad), "v"(inc) : "memory");
				}
			}
		}
	}
	else if (MODE == 0) {
		unsigned a[16];
		for (int k = 0; k < 16; ++k) a[k] = ((base[k * 64 + lane] * 2654435761u) >> 8) % kRows * 128u + lane4;
		for (int b = 0; b < kBlocks * 2; ++b)
#pragma unroll
			for (int k = 0; k < 16; ++k) asm volatile("ds_add_u32 %0, %1" : : "v"(a[k]), "v"(inc) : "memory");
	}
	else if (MODE == 1 || MODE == 2) {
		u32x16 r0, r1;
		asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r0), "=&s"(r1) : "s"(base) : "memory");
		for (int b = 0; b < kBlocks; ++b) {
#pragma unroll
			for (int u = 0; u < 16; ++u) vote(r0[u]);
#pragma unroll
			for (int u = 0; u < 16; ++u) vote(r1[u]);
			if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
	}
	else {
		u32x16 a0, a1, b0, b1;
#define LD(r0, r1, blk) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r0), "=&s"(r1) : "s"(base + (size_t)(blk) * 32) : "memory")
#define WT(r0, r1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1) : : "memory")
#define V16(r) _Pragma("unroll") for (int u = 0; u < 16; ++u) vote(r[u])
		LD(a0, a1, 0); WT(a0, a1);
		for (int b = 0; b < kBlocks; b += 2) {
			LD(b0, b1, b + 1);
#define V32P(r0, r1) _Pragma("unroll") for (int u = 0; u < 16; ++u) vote2(r0[u], r1[u])
			if (MODE == 3) { V16(a0); V16(a1); WT(b0, b1); }
			else if (MODE == 5) { V32P(a0, a1); WT(b0, b1); }
			else { V16(a0); WT(b0, b1); V16(a1); }
			LD(a0, a1, (b + 2) % kBlocks);
			if (MODE == 3) { V16(b0); V16(b1); WT(a0, a1); }
			else if (MODE == 5) { V32P(b0, b1); WT(a0, a1); }
			else { V16(b0); WT(a0, a1); V16(b1); }
		}
	}
	__builtin_amdgcn_s_waitcnt(0xc07f);
	__syncthreads();
	unsigned v = 0;
	for (int i = threadIdx.x; i < kRows * 32; i += 1024) v += hist[i];
	if (v == 0xdeadbeefu) out[0] = v;
}

template <int MODE> void run(const char* name, unsigned* out, unsigned* list, size_t lds)
{
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(bench<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * 2;
	hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(1024), lds, 0, out, list, 0);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(1024), lds, 0, out, list, 0);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	const double instr = (double)blocks * 16 * kBlocks * 32;
	printf("%-100s %8.3f ms  %6.2f clk per vote instruction per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (instr / 256));
}

int main()
{
	const size_t n = (size_t)512 * 16 * kBlocks * 32;
	std::vector<unsigned> h(n);
	srand(3);
	for (size_t i = 0; i < n; ++i) h[i] = (((unsigned)rand() % 720) << 16) | ((unsigned)rand() % 960);
	unsigned *out, *list;
	(void)hipMalloc(&out, 4); (void)hipMalloc(&list, n * 4);
	(void)hipMemcpy(list, h.data(), n * 4, hipMemcpyHostToDevice);
	const size_t lds = (size_t)kRows * 128;
	run<0>("A  ds_add_u32 only", out, list, lds);
	run<1>("B  + 4 VALU + 2 SALU per vote (edges in SGPRs, loaded once)", out, list, lds);
	run<2>("C  B + s_waitcnt lgkmcnt(0) every 32 votes", out, list, lds);
	run<3>("D  B + s_load_dwordx16 x 2 per 32 votes, a block ahead, lgkmcnt(0) after the block (the product's loop)", out, list, lds);
	run<4>("E  D with the wait after 16 of the 32 votes", out, list, lds);
	run<5>("F  D with two votes interleaved instruction by instruction (two dependency chains per wave)", out, list, lds);
	run<6>("G  4 VALU + 1 SALU per vote (one coordinate, no shift on the scalar unit)", out, list, lds);
	run<7>("H  2 VALU per vote (v_lshrrev + v_lshl_add on a register), no SALU", out, list, lds);
	run<8>("I  1 VALU per vote (v_lshl_add), no SALU", out, list, lds);
	run<10>("K  no scalar unpacking: 2 v_mul_i32_i24_sdwa (WORD_0 / WORD_1 of the packed SGPR entry) + v_add3 + shift + lshl_add (5 VALU, 0 SALU)", out, list, lds);
	run<9>("J  the product's loop without scalar unpacking: 2 v_dot2_i32_i16 on the packed entry + combine + shift + lshl_add (5 VALU, 0 SALU)", out, list, lds);
	return 0;
}
// Round 6 result (one MI355X): A 4.36, B 5.31, C 5.35, D 5.25, E 5.29, F 5.37 clk per vote instruction per CU; the cost curve -- I (1 VALU) 4.36, H (2 VALU) 4.46,
// G (4 VALU + 1 SALU) 4.99, B (4 VALU + 2 SALU) 5.31: a scalar instruction costs the loop 0.3 clk, the two multiply-adds 0.5 together; getting rid of the scalar
// unpacking costs more than it saves -- K (v_mul_i32_i24_sdwa picking WORD_0 / WORD_1 of the packed SGPR entry + v_add3: 5 VALU, 0 SALU) 5.5 - 5.8, J (two
// v_dot2_i32_i16 on the packed entry against byte-split constants) 6.4 - 6.8.  The product's voting kernel: 0.345 ms x 2.4 GHz / 130 k
// instructions per CU = 6.37 -- i.e. its loop runs at the rate of D, the rest is the five rounds 1152 workgroups take on 256 CUs where 4.5 would do (x 1.11) and
// 0.026 ms of zeroing / flushing / workgroup turnaround.  lgkmcnt(0) drains (C) and the placement of the wait (E) cost nothing; what the loop pays over the bare
// atomic (22 %) is in-order issue of 7 instructions per vote at 4 waves per SIMD -- a 1024-thread workgroup that owns the CU's LDS cannot have more; a second
// dependency chain per wave (F) does not buy it back.
