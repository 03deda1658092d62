// multi_gpu_batch -- the batch split of BASELINE config 4 driven from C++ (north_star: "host side stays CompV C++"): ONE process, N devices,
// one compvhip context + plan + HIP stream per device.  The step's global batch is born on device 0 and split with ONE grouped
// ncclSend / ncclRecv (RCCL called directly: the copies to all peers run concurrently, each over its own xGMI link), every device runs
// Sobel -> Canny -> HoughSHT on its block through the C ABI (include/compv_hip.h, asynchronous step), the per-frame line counts are
// all-gathered.  Frames are independent units: no other exchange (replicas + batch split).
// After the timed steps every frame of the global batch is checked against tests/golden/golden_batch.json (edge-map MD5, edge count, line count,
// strength sum, line-set hash -- produced by the real CompV library).
//
//   multi_gpu_batch [--devices N] [--virtual V] [--frames-per-device F] [--steps K] [--width W --height H] [--golden file] [--no-verify]
//     --devices N   the first N visible devices (default: all)
//     --virtual V   V ranks = V contexts / plans / streams on device 0 (a one-GPU box): the split then uses device-to-device copies, RCCL
//                   refuses two ranks on one device
// Python is not involved; the SCALE entry point of the round driver stays `bench.py --gpus N`.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "../include/compv_hip.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define NCCLOK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, ncclGetErrorString(r_)); exit(2); } } while (0)
#define CVOK(ctx, x) do { int rc_ = (x); if (rc_ != COMPVHIP_OK) { fprintf(stderr, "%s:%d %s: code %d (%s)\n", __FILE__, __LINE__, #x, rc_, compvhip_last_error(ctx)); exit(2); } } while (0)

// ---- synthetic frames (bit-identical to oracle/compv_oracle.c::orc_synth_frame) --------------------------------------------
static void synthFrame(uint8_t* out, int W, int H, uint32_t seed)
{
	uint32_t s = seed;
	for (int j = 0; j < H; ++j)
		for (int i = 0; i < W; ++i) {
			s = s * 1664525u + 1013904223u;
			int v = 40 + (((i / 64 + j / 64) & 1) * 150) + (int)(s >> 28);
			if (((i + 2 * j) % 257) < 3) v = 255;
			out[(size_t)j * W + i] = (uint8_t)v;
		}
}

// ---- MD5 (RFC 1321), for the edge-map checksum the fixture holds -----------------------------------------------------------------------------
struct Md5 {
	uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
	uint64_t len = 0; uint8_t buf[64]; size_t fill = 0;
	static uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
	void block(const uint8_t* p)
	{
		static const uint32_t K[64] = {
			0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122,
			0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6,
			0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60,
			0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039,
			0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
		static const int S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
		                           4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
		uint32_t M[16];
		memcpy(M, p, 64);
		uint32_t A = a, B = b, C = c, D = d;
		for (int i = 0; i < 64; ++i) {
			uint32_t F; int g;
			if (i < 16) { F = (B & C) | (~B & D); g = i; }
			else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
			else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; }
			else { F = C ^ (B | ~D); g = (7 * i) & 15; }
			const uint32_t t = D; D = C; C = B;
			B = B + rol(A + F + K[i] + M[g], S[i]);
			A = t;
		}
		a += A; b += B; c += C; d += D;
	}
	void update(const uint8_t* p, size_t n)
	{
		len += n;
		while (n) {
			if (fill == 0 && n >= 64) { block(p); p += 64; n -= 64; continue; }
			const size_t k = (64 - fill < n) ? 64 - fill : n;
			memcpy(buf + fill, p, k); fill += k; p += k; n -= k;
			if (fill == 64) { block(buf); fill = 0; }
		}
	}
	std::string hex()
	{
		const uint64_t bits = len * 8;
		const uint8_t pad = 0x80; update(&pad, 1);
		const uint8_t z = 0; while (fill != 56) update(&z, 1);
		uint8_t l[8]; memcpy(l, &bits, 8);
		memcpy(buf + 56, l, 8); block(buf);
		uint32_t w[4] = { a, b, c, d };
		char out[33];
		for (int i = 0; i < 16; ++i) snprintf(out + 2 * i, 3, "%02x", (unsigned)((const uint8_t*)w)[i]);
		return std::string(out, 32);
	}
};

// ---- the fixture: tests/golden/golden_batch.json as written by json.dump(indent=0, sort_keys=True): one "key": value per line -------------------
struct Golden { std::string md5, lineHash; long long edges = -1, lines = -1, sum = -1; };
static bool loadGolden(const std::string& path, std::map<long long, Golden>& byseed, int& W, int& H)
{
	std::ifstream f(path);
	if (!f) return false;
	std::string line; Golden cur; long long seed = -1;
	auto strval = [](const std::string& l) { const size_t a = l.find('"', l.find(':')) + 1; return l.substr(a, l.find('"', a) - a); };
	auto numval = [](const std::string& l) { return atoll(l.c_str() + l.find(':') + 1); };
	bool inFrames = false;
	while (std::getline(f, line)) {
		if (line.find("\"frames\"") != std::string::npos) { inFrames = true; continue; }
		if (!inFrames) {
			if (line.find("\"W\"") != std::string::npos) W = (int)numval(line);
			if (line.find("\"H\"") != std::string::npos) H = (int)numval(line);
			continue;
		}
		if (line.find("\"canny_md5\"") != std::string::npos) cur.md5 = strval(line);
		else if (line.find("\"line_hash\"") != std::string::npos) cur.lineHash = strval(line);
		else if (line.find("\"edges\"") != std::string::npos) cur.edges = numval(line);
		else if (line.find("\"lines\"") != std::string::npos) cur.lines = numval(line);
		else if (line.find("\"sum_strength\"") != std::string::npos) cur.sum = numval(line);
		else if (line.find("\"seed\"") != std::string::npos) seed = numval(line);
		else if (line.find('}') != std::string::npos && seed >= 0) { byseed[seed] = cur; cur = Golden(); seed = -1; }
		else if (line.find(']') != std::string::npos) inFrames = false;   // keys after the frame list ("H", "W", ...: sort_keys puts some behind it)
		if (!inFrames) {
			if (line.find("\"W\"") != std::string::npos) W = (int)numval(line);
			if (line.find("\"H\"") != std::string::npos) H = (int)numval(line);
		}
	}
	return !byseed.empty();
}

struct Rank {
	int device = 0;
	hipStream_t stream = nullptr;
	compvhip_ctx* ctx = nullptr;
	compvhip_plan* plan = nullptr;
	uint8_t* in = nullptr;        // this rank's block of the step's batch (rank 0: a pointer into the global batch)
	uint8_t* edges = nullptr;
	compvhip_line* lines = nullptr;
	int32_t* counts = nullptr;
	int32_t* allCounts = nullptr; // [ranks][F]
	ncclComm_t comm = nullptr;
	int ticket = -1;
};

int main(int argc, char** argv)
{
	int nDev = 0, virt = 0, F = 32, steps = 4, W = 3840, H = 2160;
	bool verify = true;
	std::string golden;
	for (int i = 1; i < argc; ++i) {
		const std::string a = argv[i];
		auto next = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return std::string(argv[++i]); };
		if (a == "--devices") nDev = atoi(next().c_str());
		else if (a == "--virtual") virt = atoi(next().c_str());
		else if (a == "--frames-per-device") F = atoi(next().c_str());
		else if (a == "--steps") steps = atoi(next().c_str());
		else if (a == "--width") W = atoi(next().c_str());
		else if (a == "--height") H = atoi(next().c_str());
		else if (a == "--golden") golden = next();
		else if (a == "--no-verify") verify = false;
		else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
	}
	int visible = compvhip_device_count();
	if (visible <= 0) { fprintf(stderr, "no HIP device\n"); return 2; }
	if (nDev > visible) { fprintf(stderr, "multi_gpu_batch: --devices %d but only %d HIP device(s) are visible (a missing peer is an error, not a smaller run)\n", nDev, visible); return 3; }
	if (nDev <= 0) nDev = visible;
	const int R = virt > 0 ? virt : nDev;
	const bool useRccl = virt <= 0;
	if (F <= 0 || steps <= 0 || W < 3 || H < 3 || (W & 7)) { fprintf(stderr, "bad geometry (W %% 8 == 0: the plan's stride is the width here)\n"); return 1; }
	if (golden.empty()) {
		std::string exe = argv[0];
		const size_t slash = exe.rfind('/');
		golden = (slash == std::string::npos ? std::string(".") : exe.substr(0, slash)) + "/../../tests/golden/golden_batch.json";
	}
	const size_t frameBytes = (size_t)W * H, cap = 1 << 16;
	const float tLow = 59.f, tHigh = 119.f; const int threshold = 100;

	std::vector<Rank> rk(R);
	std::vector<int> devs(R);
	for (int r = 0; r < R; ++r) devs[r] = rk[r].device = (virt > 0) ? 0 : r;
	std::vector<ncclComm_t> comms(R, nullptr);
	if (useRccl) NCCLOK(ncclCommInitAll(comms.data(), R, devs.data()));   // one process, R devices: the communicators of all ranks
	uint8_t* globalIn = nullptr;
	for (int r = 0; r < R; ++r) {
		Rank& q = rk[r];
		q.comm = comms[r];
		HIPOK(hipSetDevice(q.device));
		HIPOK(hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking));
		if (compvhip_ctx_create(&q.ctx, q.device) != COMPVHIP_OK) { fprintf(stderr, "compvhip_ctx_create(%d) failed\n", q.device); return 2; }
		CVOK(q.ctx, compvhip_plan_create(q.ctx, W, H, W, F, 1.f, &q.plan));
		if (r == 0) { HIPOK(hipMalloc(&globalIn, frameBytes * F * R)); q.in = globalIn; }
		else HIPOK(hipMalloc(&q.in, frameBytes * F));
		HIPOK(hipMalloc(&q.edges, frameBytes * F));
		HIPOK(hipMalloc(&q.lines, sizeof(compvhip_line) * cap * F));
		HIPOK(hipMalloc(&q.counts, sizeof(int32_t) * F));
		HIPOK(hipMalloc(&q.allCounts, sizeof(int32_t) * F * R));
	}
	// the global batch is born on device 0: frame g of the batch has seed 12345 + (g mod 256) (the fixture's 256 frames)
	{
		std::vector<uint8_t> host(frameBytes);
		HIPOK(hipSetDevice(rk[0].device));
		for (int g = 0; g < R * F; ++g) {
			synthFrame(host.data(), W, H, 12345u + (uint32_t)(g % 256));
			HIPOK(hipMemcpy(globalIn + (size_t)g * frameBytes, host.data(), frameBytes, hipMemcpyHostToDevice));
		}
	}

	auto step = [&]() {
		// (1) batch split: one group = every peer's block in flight at once
		if (useRccl) {
			NCCLOK(ncclGroupStart());
			for (int r = 1; r < R; ++r) {
				NCCLOK(ncclSend(globalIn + (size_t)r * F * frameBytes, frameBytes * F, ncclUint8, r, rk[0].comm, rk[0].stream));
				NCCLOK(ncclRecv(rk[r].in, frameBytes * F, ncclUint8, 0, rk[r].comm, rk[r].stream));
			}
			NCCLOK(ncclGroupEnd());
		}
		else {
			for (int r = 1; r < R; ++r) HIPOK(hipMemcpyAsync(rk[r].in, globalIn + (size_t)r * F * frameBytes, frameBytes * F, hipMemcpyDeviceToDevice, rk[r].stream));
		}
		// (2) the hot path on every device, asynchronous: the host thread only enqueues
		for (int r = 0; r < R; ++r) {
			Rank& q = rk[r];
			CVOK(q.ctx, compvhip_plan_pipeline_async(q.plan, q.in, tLow, tHigh, threshold, 0, q.edges, q.lines, cap, q.counts, q.stream, &q.ticket));
		}
		// (3) every rank's step made final first (plan_wait replays a step whose speculative hysteresis rounds did not suffice and rewrites its counts) ...
		for (int r = 0; r < R; ++r) CVOK(rk[r].ctx, compvhip_plan_wait(rk[r].plan, rk[r].ticket));
		// ... then the only result exchange: per-frame line counts, all-gathered
		if (useRccl) {
			NCCLOK(ncclGroupStart());
			for (int r = 0; r < R; ++r) NCCLOK(ncclAllGather(rk[r].counts, rk[r].allCounts, F, ncclInt32, rk[r].comm, rk[r].stream));
			NCCLOK(ncclGroupEnd());
		}
		else {   // virtual ranks: gather with device copies once every rank's step is final
			for (int r = 0; r < R; ++r)
				for (int s = 0; s < R; ++s) HIPOK(hipMemcpyAsync(rk[r].allCounts + (size_t)s * F, rk[s].counts, sizeof(int32_t) * F, hipMemcpyDeviceToDevice, rk[r].stream));
		}
		for (int r = 0; r < R; ++r) { HIPOK(hipSetDevice(rk[r].device)); HIPOK(hipStreamSynchronize(rk[r].stream)); }
	};

	step();   // warm-up (allocations of the plans, RCCL channels)
	const auto t0 = std::chrono::steady_clock::now();
	for (int k = 0; k < steps; ++k) step();
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	const double mpix = (double)R * F * W * H * steps / sec / 1e6;

	// ---- verification: every frame of the global batch against the reference-derived fixture -----------------------------------------------
	std::map<long long, Golden> gold;
	int gW = 0, gH = 0;
	long long checked = 0, totalLines = 0;
	const bool haveGold = verify && loadGolden(golden, gold, gW, gH) && gW == W && gH == H;
	std::vector<uint8_t> e(frameBytes * F);
	std::vector<compvhip_line> ln(cap);
	std::vector<int32_t> cnt(F), all((size_t)F * R);
	bool ok = true;
	for (int r = 0; r < R && ok; ++r) {
		Rank& q = rk[r];
		HIPOK(hipSetDevice(q.device));
		HIPOK(hipMemcpy(cnt.data(), q.counts, sizeof(int32_t) * F, hipMemcpyDeviceToHost));
		HIPOK(hipMemcpy(all.data(), q.allCounts, sizeof(int32_t) * F * R, hipMemcpyDeviceToHost));
		HIPOK(hipMemcpy(e.data(), q.edges, frameBytes * F, hipMemcpyDeviceToHost));
		for (int f = 0; f < F; ++f) {
			if (all[(size_t)r * F + f] != cnt[f]) { fprintf(stderr, "rank %d: gathered count of its own frame %d differs\n", r, f); ok = false; break; }
			totalLines += cnt[f];
			if (!haveGold) continue;
			const long long seed = 12345 + ((long long)r * F + f) % 256;
			const Golden& g = gold[seed];
			long long edges = 0;
			const uint8_t* ef = e.data() + (size_t)f * frameBytes;
			for (size_t i = 0; i < frameBytes; ++i) edges += ef[i] != 0;
			Md5 md; md.update(ef, frameBytes);
			const size_t n = (size_t)cnt[f] < cap ? (size_t)cnt[f] : cap;
			HIPOK(hipMemcpy(ln.data(), q.lines + (size_t)f * cap, sizeof(compvhip_line) * n, hipMemcpyDeviceToHost));
			long long sum = 0; uint64_t hv = 0;
			for (size_t i = 0; i < n; ++i) {
				sum += ln[i].strength;
				hv += (uint64_t)((long long)(W + H) - ln[i].row + 32768) * 1000003ull + (uint64_t)ln[i].col * 7919ull + (uint64_t)ln[i].strength * 31337ull;
			}
			char hh[17]; snprintf(hh, sizeof(hh), "%016llx", (unsigned long long)hv);
			if (md.hex() != g.md5 || edges != g.edges || cnt[f] != g.lines || sum != g.sum || g.lineHash != hh) {
				fprintf(stderr, "rank %d frame %d (seed %lld) differs from the CompV reference: md5 %s/%s edges %lld/%lld lines %d/%lld sum %lld/%lld hash %s/%s\n", r, f, seed,
				        md.hex().c_str(), g.md5.c_str(), edges, g.edges, cnt[f], g.lines, sum, g.sum, hh, g.lineHash.c_str());
				ok = false; break;
			}
			++checked;
		}
	}
	// every rank must hold the same gathered table
	for (int r = 1; r < R && ok; ++r) {
		std::vector<int32_t> other((size_t)F * R);
		HIPOK(hipSetDevice(rk[r].device));
		HIPOK(hipMemcpy(other.data(), rk[r].allCounts, sizeof(int32_t) * F * R, hipMemcpyDeviceToHost));
		HIPOK(hipSetDevice(rk[0].device));
		HIPOK(hipMemcpy(all.data(), rk[0].allCounts, sizeof(int32_t) * F * R, hipMemcpyDeviceToHost));
		if (other != all) { fprintf(stderr, "rank %d: gathered line counts differ from rank 0's\n", r); ok = false; }
	}
	printf("{\"program\": \"multi_gpu_batch\", \"ranks\": %d, \"devices\": %d, \"transport\": \"%s\", \"frames_per_device\": %d, \"W\": %d, \"H\": %d, \"steps\": %d, "
	       "\"ms_per_step\": %.4f, \"Mpixels_per_s\": %.1f, \"lines_last_step\": %lld, \"frames_checked\": %lld, \"fixture\": \"%s\"}\n",
	       R, virt > 0 ? 1 : R, useRccl ? "rccl grouped send/recv + allgather" : "device copies (virtual ranks share one GPU)", F, W, H, steps, sec / steps * 1e3, mpix, totalLines,
	       checked, haveGold ? "tests/golden/golden_batch.json" : "none");
	for (int r = 0; r < R; ++r) {
		Rank& q = rk[r];
		HIPOK(hipSetDevice(q.device));
		compvhip_plan_destroy(q.plan);
		HIPOK(hipFree(q.edges)); HIPOK(hipFree(q.lines)); HIPOK(hipFree(q.counts)); HIPOK(hipFree(q.allCounts));
		if (r) HIPOK(hipFree(q.in));
		compvhip_ctx_destroy(q.ctx);
		HIPOK(hipStreamDestroy(q.stream));
		if (q.comm) NCCLOK(ncclCommDestroy(q.comm));
	}
	HIPOK(hipSetDevice(rk[0].device));
	HIPOK(hipFree(globalIn));
	if (!ok) return 3;
	if (verify && !haveGold) { printf("MULTI-GPU BATCH RAN (no fixture for this geometry: outputs not checked)\n"); return 0; }
	printf("MULTI-GPU BATCH OK\n");
	return 0;
}
