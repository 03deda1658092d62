#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, unsigned* out, int n) {
	int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
	unsigned d = 0xaabbccddu;
	asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(d) : "v"(in[i]));
	out[i] = d;
}
int main() {
	const int n = 4096; float h[n]; unsigned o[n];
	for (int i = 0; i < n; ++i) h[i] = (i - 64) * 0.0873f;           // -5.6 .. 352
	h[0] = NAN; h[1] = INFINITY; h[2] = -INFINITY; h[3] = 254.9999f; h[4] = 255.0f; h[5] = 255.5f; h[6] = 0.999f; h[7] = 1e9f; h[8] = -0.5f; h[9] = 127.5f; h[10] = 128.5f; h[11] = 2.5f;
	float* di; unsigned* d_o; hipMalloc(&di, sizeof(h)); hipMalloc(&d_o, sizeof(o)); hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
	k<<<n / 256, 256>>>(di, d_o, n); hipMemcpy(o, d_o, sizeof(o), hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < n; ++i) {
		float x = h[i]; unsigned exp8;
		if (!(x == x)) exp8 = 0; else if (x <= 0.f) exp8 = 0; else if (x >= 255.f) exp8 = 255; else exp8 = (unsigned)x;   // truncation + saturation
		unsigned got8 = (o[i] >> 8) & 0xff;
		if ((o[i] & 0xffff00ffu) != 0xaabb00ddu || got8 != exp8) { if (bad < 12) printf("x=%g got byte %u expected %u word %08x\n", x, got8, exp8, o[i]); ++bad; }
	}
	printf("mismatches vs truncate+saturate: %d of %d\n", bad, n);
	for (int i = 0; i < 12; ++i) printf("x=%g -> %u\n", h[i], (o[i] >> 8) & 0xff);
}
