// Measurement aid: which XCD does workgroup b of a large grid run on? Prints how often XCC_ID == b % 8 and the mapping of the first
// workgroups, for a grid shaped like the pose kernels' (256 threads, 20 KiB of LDS per workgroup).
// hipcc --offload-arch=gfx950 -O2 tools/xcd_probe.hip -o tools/xcd_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* out, unsigned spin)
{
	extern __shared__ unsigned char lds[];
	unsigned xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	// some work so that workgroups overlap like real ones
	unsigned v = threadIdx.x;
	for (unsigned i = 0; i < spin; ++i)
		v = v * 1664525u + 1013904223u;
	lds[threadIdx.x] = (unsigned char)v;
	__syncthreads();
	if (threadIdx.x == 0)
		out[blockIdx.x] = (xcc & 0xF) | (unsigned(lds[1]) << 31 >> 31 << 30 & 0);
}

int main()
{
	const unsigned blocks = 16384;
	unsigned* d = nullptr;
	hipMalloc(&d, blocks * sizeof(unsigned));
	for (unsigned spin : { 0u, 2000u })
	{
		hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 20480, 0, d, spin);
		hipDeviceSynchronize();
		std::vector<unsigned> h(blocks);
		hipMemcpy(h.data(), d, blocks * sizeof(unsigned), hipMemcpyDeviceToHost);
		unsigned match = 0, hist[16] = {};
		for (unsigned b = 0; b < blocks; ++b) { match += (h[b] == b % 8); hist[h[b] & 15]++; }
		printf("spin %u: XCC_ID == b %% 8 for %u of %u workgroups; per XCC:", spin, match, blocks);
		for (int x = 0; x < 8; ++x) printf(" %u", hist[x]);
		printf("\n  first 24:");
		for (unsigned b = 0; b < 24; ++b) printf(" %u", h[b]);
		printf("\n  b=8000..8023:");
		for (unsigned b = 8000; b < 8024; ++b) printf(" %u", h[b]);
		printf("\n");
	}
	return 0;
}
