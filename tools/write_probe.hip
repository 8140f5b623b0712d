// tools/write_probe.hip -- measurement aid, not part of the product: how fast can gfx950 stream 16 byte per lane stores?
// hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o /tmp/write_probe && /tmp/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template<int kMode>
__global__ __launch_bounds__(256) void write_kernel(float4* __restrict__ dst, uint64_t num_quads, float seed)
{
	const uint64_t stride = uint64_t(gridDim.x) * 256;
	const float4 v = make_float4(seed, seed + 1, seed + 2, seed + 3);
	for (uint64_t q = uint64_t(blockIdx.x) * 256 + threadIdx.x; q < num_quads; q += stride)
	{
		if (kMode == 0) dst[q] = v;
		else if (kMode == 1) __builtin_nontemporal_store(v.x, &dst[q].x), __builtin_nontemporal_store(v.y, &dst[q].y), __builtin_nontemporal_store(v.z, &dst[q].z), __builtin_nontemporal_store(v.w, &dst[q].w);
	}
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template<int kMode>
__global__ __launch_bounds__(256) void write_kernel_vec(f32x4* __restrict__ dst, uint64_t num_quads, float seed)
{
	const uint64_t stride = uint64_t(gridDim.x) * 256;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	for (uint64_t q = uint64_t(blockIdx.x) * 256 + threadIdx.x; q < num_quads; q += stride)
	{
		if (kMode == 0) dst[q] = v;
		else __builtin_nontemporal_store(v, &dst[q]);
	}
}

// wave-per-4800-bytes pattern like the decode kernel: each wave writes 300 consecutive quads
__global__ __launch_bounds__(256) void write_kernel_pose(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads_per_pose, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pose >= num_poses) return;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(pose) * quads_per_pose;
	for (uint32_t q = lane; q < quads_per_pose; q += 64)
		p[q] = v;
}

int main()
{
	const uint64_t bytes = 65536ull * 4800ull;
	const uint64_t num_quads = bytes / 16;
	float4* d;
	hipMalloc((void**)&d, bytes);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	auto time_it = [&](const char* name, auto launch)
	{
		for (int i = 0; i < 3; ++i) launch(i);
		hipEventRecord(a);
		const int reps = 30;
		for (int i = 0; i < reps; ++i) launch(i);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("%-40s %8.2f us  %8.1f GB/s\n", name, ms / reps * 1000.0, bytes * reps / (ms * 1e-3) / 1e9);
	};
	for (int blocks : {2048, 4096, 8192, 16384, 65536})
	{
		char name[128];
		snprintf(name, sizeof(name), "plain float4 grid=%d", blocks);
		time_it(name, [&](int i) { hipLaunchKernelGGL(write_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, num_quads, float(i)); });
		snprintf(name, sizeof(name), "vec4 plain grid=%d", blocks);
		time_it(name, [&](int i) { hipLaunchKernelGGL(write_kernel_vec<0>, dim3(blocks), dim3(256), 0, 0, (f32x4*)d, num_quads, float(i)); });
		snprintf(name, sizeof(name), "vec4 nontemporal grid=%d", blocks);
		time_it(name, [&](int i) { hipLaunchKernelGGL(write_kernel_vec<1>, dim3(blocks), dim3(256), 0, 0, (f32x4*)d, num_quads, float(i)); });
	}
	time_it("pose pattern (wave per 4800 B) plain", [&](int i) { hipLaunchKernelGGL(write_kernel_pose, dim3(16384), dim3(256), 0, 0, (f32x4*)d, 65536u, 300u, float(i)); });
	time_it("hipMemsetAsync", [&](int i) { hipMemsetAsync(d, i, bytes, 0); });
	return 0;
}
