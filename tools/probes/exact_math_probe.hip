#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float exact_sqrt(float x)
{
	const float s = __builtin_amdgcn_sqrtf(x);
	const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
	const float residual_down = __builtin_fmaf(-down, s, x), residual_up = __builtin_fmaf(-up, s, x);
	float result = residual_down <= 0.0f ? down : s;
	result = residual_up > 0.0f ? up : result;
	return result;
}
__device__ __forceinline__ float exact_rcp(float x)
{
	const float r0 = __builtin_amdgcn_rcpf(x);
	const float e0 = __builtin_fmaf(-x, r0, 1.0f);
	const float r1 = __builtin_fmaf(e0, r0, r0);
	const float e1 = __builtin_fmaf(-x, r1, 1.0f);
	return __builtin_fmaf(e1, r1, r1);
}
__global__ void check(unsigned long long* out, uint32_t first, uint32_t last)
{
	// every bit pattern in [first, last]
	const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
	unsigned long long bad_sqrt = 0, bad_rcp = 0;
	for (uint64_t bits = uint64_t(first) + blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; bits <= last; bits += stride)
	{
		const float x = __uint_as_float(uint32_t(bits));
		const float a = exact_sqrt(x), b = sqrtf(x);
		if (__float_as_uint(a) != __float_as_uint(b)) { if (bad_sqrt == 0) atomicMin(&out[2], (unsigned long long)bits); bad_sqrt++; }
		const float c = exact_rcp(x), d = 1.0f / x;
		if (__float_as_uint(c) != __float_as_uint(d)) { if (bad_rcp == 0) atomicMin(&out[3], (unsigned long long)bits); bad_rcp++; }
	}
	if (bad_sqrt) atomicAdd(&out[0], bad_sqrt);
	if (bad_rcp) atomicAdd(&out[1], bad_rcp);
}
int main()
{
	unsigned long long* d; hipMalloc(&d, 32);
	struct { const char* name; uint32_t first, last; } ranges[] = {
		{"zero", 0, 0}, {"denormals", 1, 0x007FFFFF}, {"normals below 2^-96", 0x00800000, 0x0F7FFFFF}, {"2^-96 .. 2^-20", 0x0F800000, 0x35800000},
		{"2^-20 .. 16", 0x35800000, 0x41800000}, {"16 .. 2^96", 0x41800000, 0x6F800000}, {"2^96 .. max", 0x6F800000, 0x7F7FFFFF}, {"inf", 0x7F800000, 0x7F800000}};
	for (auto& r : ranges)
	{
		unsigned long long h[4] = {0, 0, ~0ull, ~0ull};
		hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, d, r.first, r.last);
		hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
		printf("%-22s sqrt mismatches %llu (first 0x%08llx)  rcp mismatches %llu (first 0x%08llx)\n", r.name, h[0], h[2] & 0xFFFFFFFFull, h[1], h[3] & 0xFFFFFFFFull);
	}
	return 0;
}
