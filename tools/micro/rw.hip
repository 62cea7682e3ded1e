// scattered read / write throughput in the go() kernel's access shape: every lane works inside its OWN region of `stride` bytes
// (one region per read in flight), touching `hot` bytes of it.  mode 0: dependent 4 B loads; 1: 4 B stores; 2: load + store to
// the same line; 3: load + store to another line; 4: 16 B stores; 5: whole 64 B (4 x 16 B) stores; 6: whole 128 B line stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(uint8_t* base, size_t stride, uint32_t hot, int steps, unsigned* out, int mode) {
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	uint8_t* r = base + t * stride;
	uint32_t off = (uint32_t)(t * 64) % hot, acc = 0;
	for(int i = 0; i < steps; i++) {
		uint32_t v = 0;
		if(mode == 0 || mode == 2 || mode == 3) v = *(const volatile uint32_t*)(r + off);
		if(mode == 1 || mode == 2) *(volatile uint32_t*)(r + off + 4) = acc + i;
		if(mode == 3) *(volatile uint32_t*)(r + (off + 2048) % hot) = acc + i;
		if(mode == 4) { uint4 w = {acc, (uint32_t)i, 2, 3}; *(uint4*)(r + (off & ~15u)) = w; }
		if(mode == 5) { uint4 w = {acc, (uint32_t)i, 2, 3}; uint4* p = (uint4*)(r + (off & ~63u)); p[0] = w; p[1] = w; p[2] = w; p[3] = w; }
		if(mode == 6) { uint4 w = {acc, (uint32_t)i, 2, 3}; uint4* p = (uint4*)(r + (off & ~127u)); for(int q = 0; q < 8; q++) p[q] = w; }
		off = (off + 4160 + v) % hot;
		acc += v;
	}
	out[t] = acc + off;
}
int main(int argc, char** argv) {
	const size_t lanes = argc > 1 ? (size_t)atol(argv[1]) : 262144;
	const size_t stride = 115 * 1024;
	const char* names[] = {"dependent 4 B loads", "4 B stores", "4 B load + store same line", "4 B load + store other line", "16 B stores", "64 B stores", "128 B stores"};
	uint8_t* buf; unsigned* out;
	if(hipMalloc(&buf, lanes * stride + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMalloc(&out, lanes * 4);
	hipMemset(buf, 0, lanes * stride + 4096);
	for(uint32_t hot : {16384u, 2048u}) for(int mode = 0; mode < 7; mode++) {
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		const int steps = 100;
		k<<<lanes / 256, 256>>>(buf, stride, hot, steps, out, mode);
		hipDeviceSynchronize();
		hipEventRecord(a);
		k<<<lanes / 256, 256>>>(buf, stride, hot, steps, out, mode);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("lanes %zu hot %5u  %-30s %8.3f ms  %6.1f G steps/s\n", lanes, hot, names[mode], ms, lanes * (double)steps / ms / 1e6);
	}
	return 0;
}
