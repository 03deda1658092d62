// Times the KHT host linker (pack + khtLink) on a raw W x H edge map, points written to ordinary or to pinned host memory.
//   tools/kht_lab/build.sh && python tools/kht_lab/make_edges.py /tmp/edges4k.raw && tools/kht_lab/link_bench /tmp/edges4k.raw 3840 2160 40
#include "../../compv_amd/csrc/kht.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
using namespace compvhip;
int main(int argc, char** argv)
{
	if (argc < 4) return 1;
	const size_t W = atoi(argv[2]), H = atoi(argv[3]);
	const int reps = argc > 4 ? atoi(argv[4]) : 40;
	FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
	std::vector<uint8_t> e(W * H); if (fread(e.data(), 1, e.size(), f) != e.size()) return 2; fclose(f);
	using clk = std::chrono::steady_clock;
	for (int pinned = 0; pinned < 2; ++pinned) {
		KhtBitPlane plane0, plane; std::vector<KhtRange> strings;
		khtPackBytes(e.data(), W, H, W, plane0);
		const size_t most = khtPlaneCount(plane0) + 1;
		KhtPoint* pts = nullptr;
		if (pinned) { if (hipHostMalloc(reinterpret_cast<void**>(&pts), most * sizeof(KhtPoint)) != hipSuccess) { printf("no pinned memory (no device?)\n"); return 0; } }
		else pts = new KhtPoint[most];
		double bestPack = 1e9, bestLink = 1e9; size_t n = 0;
		for (int r = 0; r < reps; ++r) {
			auto t0 = clk::now();
			khtPackBytes(e.data(), W, H, W, plane0);
			auto t1 = clk::now();
			plane = plane0;
			auto t2 = clk::now();
			n = khtLink(plane, 10, pts, strings);
			auto t3 = clk::now();
			bestPack = std::min(bestPack, std::chrono::duration<double, std::milli>(t1 - t0).count());
			bestLink = std::min(bestLink, std::chrono::duration<double, std::milli>(t3 - t2).count());
		}
		unsigned long long h = 1469598103934665603ull;
		for (size_t i = 0; i < n; ++i) { h = (h ^ (unsigned)pts[i].x) * 1099511628211ull; h = (h ^ (unsigned)pts[i].y) * 1099511628211ull; }
		printf("%s memory: pack %.3f ms  link %.3f ms  points %zu strings %zu  hash %016llx\n", pinned ? "pinned" : "plain", bestPack, bestLink, n, strings.size(), h);
	}
}
