// dependent-load latency under the go() kernel's access shape: every lane chases pointers inside its OWN region of
// `stride` bytes (one region per slot), `hot` bytes of which are touched; 262144 lanes in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void chase(const uint8_t* base, size_t stride, uint32_t hot, int steps, unsigned* out, int active) {
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	if((int)(threadIdx.x & 63) >= active) return;
	const uint8_t* r = base + t * stride;
	uint32_t off = (uint32_t)(t * 64) % hot, acc = 0;
	for(int i = 0; i < steps; i++) {
		const uint32_t v = *(const uint32_t*)(r + off);      // buffer is zero: v == 0, but the address depends on it
		off = (off + 4160 + v) % hot;
		acc += v;
	}
	out[t] = acc + off;
}
int main(int argc, char** argv) {
	const size_t lanes = 262144;
	for(int cfg = 0; cfg < 8; cfg++) {
		size_t stride = cfg < 4 ? 115 * 1024 : 4096;
		uint32_t hot = cfg % 4 == 0 ? 115 * 1024 : cfg % 4 == 1 ? 16384 : cfg % 4 == 2 ? 4096 : 1024;
		if(hot > stride) hot = (uint32_t)stride;
		uint8_t* buf; unsigned* out;
		if(hipMalloc(&buf, lanes * stride + 4096) != hipSuccess) { printf("alloc failed\n"); continue; }
		hipMalloc(&out, lanes * 4);
		hipMemset(buf, 0, lanes * stride + 4096);
		for(int active = 64; active >= 8; active /= 8) {
			hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
			const int steps = 200;
			chase<<<lanes / 256, 256>>>(buf, stride, hot, steps, out, active);
			hipDeviceSynchronize();
			hipEventRecord(a);
			chase<<<lanes / 256, 256>>>(buf, stride, hot, steps, out, active);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			printf("stride %7zu hot %6u active lanes/wave %2d: %.3f ms for %d dependent loads = %.0f ns per load (%.1f GB pool)\n", stride, hot, active, ms, steps, ms * 1e6 / steps, lanes * stride / 1e9);
		}
		hipFree(buf); hipFree(out);
	}
	return 0;
}
