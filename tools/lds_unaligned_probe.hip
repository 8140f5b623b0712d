// tools/lds_unaligned_probe.hip -- does gfx950 under ROCm serve unaligned 8 / 4 byte LDS reads? (measurement aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void probe(const uint8_t* __restrict__ source, uint64_t* __restrict__ out64, uint32_t* __restrict__ out32)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[1024];
	for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x)
		lds[i] = source[i];
	__syncthreads();
	const uint32_t offset = threadIdx.x * 3 + 1;		// every alignment
	uint64_t a;
	uint32_t b;
	__builtin_memcpy(&a, lds + offset, 8);
	__builtin_memcpy(&b, lds + offset + 5, 4);
	out64[threadIdx.x] = a;
	out32[threadIdx.x] = b;
}

int main()
{
	uint8_t host[1024];
	for (int i = 0; i < 1024; ++i) host[i] = uint8_t(i * 7 + 3);
	uint8_t* d_source; uint64_t* d_out64; uint32_t* d_out32;
	hipMalloc((void**)&d_source, 1024); hipMalloc((void**)&d_out64, 64 * 8); hipMalloc((void**)&d_out32, 64 * 4);
	hipMemcpy(d_source, host, 1024, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_source, d_out64, d_out32);
	const hipError_t status = hipDeviceSynchronize();
	printf("status %d (%s)\n", int(status), hipGetErrorString(status));
	uint64_t out64[64]; uint32_t out32[64];
	hipMemcpy(out64, d_out64, sizeof(out64), hipMemcpyDeviceToHost);
	hipMemcpy(out32, d_out32, sizeof(out32), hipMemcpyDeviceToHost);
	int bad = 0;
	for (int t = 0; t < 64; ++t)
	{
		uint64_t a; uint32_t b;
		memcpy(&a, host + t * 3 + 1, 8); memcpy(&b, host + t * 3 + 6, 4);
		bad += (a != out64[t]) + (b != out32[t]);
	}
	printf("mismatches %d\n", bad);
	return 0;
}
