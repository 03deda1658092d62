// canny_lab -- experiment harness (not part of the product): times variants of the kernel-size-3 Canny tile kernel on the benchmark's
// 32 x 4K frames and compares their outputs (E / U masks, edge bytes) with variant 0 bit for bit.
//   build: tools/canny_lab/build.sh      run: tools/canny_lab/canny_lab [frames] [reps] [W H]
#include "../../compv_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

using namespace compvhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef hipError_t (*launch_fn)(const CannyArgs&, int, bool, hipStream_t);
#define DECL(n) namespace lab_##n { hipError_t launch_canny_tiles_swar(const CannyArgs&, int, bool, hipStream_t); }
LAB_DECLS
struct Variant { const char* name; launch_fn fn; };
static Variant variants[] = { LAB_TABLE };

static void synth(uint8_t* out, int W, int H, uint32_t seed)
{
	uint32_t s = seed;
	for (int j = 0; j < H; ++j) for (int i = 0; i < W; ++i) {
		s = s * 1664525u + 1013904223u;
		int v = 40 + (((i / 64 + j / 64) & 1) * 150) + (int)(s >> 28);
		if (((i + 2 * j) % 257) < 3) v = 255;
		out[(size_t)j * W + i] = (uint8_t)v;
	}
}

int main(int argc, char** argv)
{
	const int frames = argc > 1 ? atoi(argv[1]) : 32;
	const int reps = argc > 2 ? atoi(argv[2]) : 20;
	const int W = argc > 4 ? atoi(argv[3]) : 3840, H = argc > 4 ? atoi(argv[4]) : 2160, S = W;   // canny_lab frames reps [W H]  (W % 8 == 0)
	const int tilesX512 = (W + 511) / 512;
	const int wb = tilesX512 * 16;
	const size_t bitsStride = (size_t)wb * H;
	std::vector<uint8_t> h((size_t)frames * W * H);
	for (int f = 0; f < frames; ++f) synth(h.data() + (size_t)f * W * H, W, H, 12345u + f);
	uint8_t *din, *dout; uint32_t *de, *du;
	CK(hipMalloc(&din, h.size())); CK(hipMalloc(&dout, h.size()));
	CK(hipMalloc(&de, bitsStride * frames * 4 + (8u << 20))); CK(hipMalloc(&du, bitsStride * frames * 4 + (8u << 20)));
	CK(hipMemcpy(din, h.data(), h.size(), hipMemcpyHostToDevice));
	CannyArgs a; memset(&a, 0, sizeof(a));
	a.in = din; a.out = dout; a.ebits = de; a.ubits = du; a.thrDev = nullptr;
	a.inFrameStride = (size_t)S * H; a.outFrameStride = (size_t)S * H; a.bitsFrameStride = bitsStride;
	a.W = W; a.H = H; a.S = S; a.So = S; a.wb = wb; a.tLow = 59; a.tHigh = 119; a.simdEnd = W; a.cStart = 1; a.ksize = 3;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	std::vector<uint8_t> refOut, out(h.size());
	std::vector<uint32_t> refE, refU, E(bitsStride * frames), U(bitsStride * frames);
	const int nv = (int)(sizeof(variants) / sizeof(variants[0]));
	{
		for (int i = 0; i < 3; ++i) CK(hipMemcpyAsync(dout, din, h.size(), hipMemcpyDeviceToDevice, 0));
		CK(hipEventRecord(e0, 0));
		for (int i = 0; i < 10; ++i) CK(hipMemcpyAsync(dout, din, h.size(), hipMemcpyDeviceToDevice, 0));
		CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("device copy of %zu MB: %.4f ms (%.2f TB/s read+write)\n", h.size() >> 20, ms / 10, 2.0 * h.size() / (ms / 10 * 1e-3) / 1e12);
	}
	// pass 1: outputs of every variant against variant 0
	for (int v = 0; v < nv; ++v) {
		CK(hipMemset(dout, 0xAA, h.size())); CK(hipMemset(de, 0, bitsStride * frames * 4)); CK(hipMemset(du, 0, bitsStride * frames * 4));
		CK(variants[v].fn(a, frames, false, 0));
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(out.data(), dout, out.size(), hipMemcpyDeviceToHost));
		CK(hipMemcpy(E.data(), de, E.size() * 4, hipMemcpyDeviceToHost));
		CK(hipMemcpy(U.data(), du, U.size() * 4, hipMemcpyDeviceToHost));
		const char* verdict;
		if (v == 0) { refOut = out; refE = E; refU = U; verdict = "reference"; }
		else verdict = (out == refOut && E == refE && U == refU) ? "MATCH" : "differs";
		printf("%-28s %s\n", variants[v].name, verdict);
	}
	// pass 2: timing, variants interleaved launch by launch (clock / thermal drift hits all of them alike)
	std::vector<std::vector<float>> ts(nv);
	for (int i = 0; i < 5; ++i) for (int v = 0; v < nv; ++v) CK(variants[v].fn(a, frames, false, 0));
	CK(hipDeviceSynchronize());
	for (int i = 0; i < reps; ++i)
		for (int v = 0; v < nv; ++v) {
			CK(hipEventRecord(e0, 0));
			CK(variants[v].fn(a, frames, false, 0));
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts[v].push_back(ms);
		}
	for (int v = 0; v < nv; ++v) {
		std::sort(ts[v].begin(), ts[v].end());
		printf("%-28s median %.4f ms  p25 %.4f  min %.4f  max %.4f   (%.3f of %s)\n", variants[v].name, ts[v][ts[v].size() / 2], ts[v][ts[v].size() / 4], ts[v].front(), ts[v].back(),
		       ts[v][ts[v].size() / 2] / ts[0][ts[0].size() / 2], variants[0].name);
	}
	return 0;
}
