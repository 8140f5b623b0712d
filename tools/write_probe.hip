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

// the same bytes from persistent waves: grid = a few blocks per CU, every wave walks poses wave_id, wave_id + num_waves, ...
__global__ __launch_bounds__(256) void write_kernel_pose_persistent(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads_per_pose, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t num_waves = gridDim.x * 4;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	for (uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6); pose < num_poses; pose += num_waves)
	{
		f32x4* p = dst + uint64_t(pose) * quads_per_pose;
		for (uint32_t q = lane; q < quads_per_pose; q += 64)
			p[q] = v;
	}
}

// poses padded to a multiple of 128 bytes (stride in quads given separately)
__global__ __launch_bounds__(256) void write_kernel_pose_strided(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads_per_pose, uint32_t stride_quads, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pose >= num_poses) return;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(pose) * stride_quads;
	for (uint32_t q = lane; q < quads_per_pose; q += 64)
		p[q] = v;
}

// one wave per 1 KiB chunk of a pose (5 waves cover a 4800 byte pose; the last one writes 704 bytes)
__global__ __launch_bounds__(256) void write_kernel_pose_chunks(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads_per_pose, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t chunks_per_pose = (quads_per_pose + 63) / 64;
	const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t pose = wave / chunks_per_pose;
	const uint32_t chunk = wave - pose * chunks_per_pose;
	if (pose >= num_poses) return;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	const uint32_t q = chunk * 64 + lane;
	if (q < quads_per_pose)
		dst[uint64_t(pose) * quads_per_pose + q] = v;
}

// block cooperative: the 4 poses of a block are one contiguous region swept by all 256 threads
__global__ __launch_bounds__(256) void write_kernel_pose_block(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads_per_pose, float seed)
{
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	const uint64_t first = uint64_t(blockIdx.x) * 4 * quads_per_pose;
	const uint32_t count = min(4u, num_poses - blockIdx.x * 4) * quads_per_pose;
	for (uint32_t q = threadIdx.x; q < count; q += 256)
		dst[first + q] = v;
}

int main()
{
	uint64_t bytes = 65536ull * 4800ull;
	const uint64_t num_quads = bytes / 16;
	float4* d;
	hipMalloc((void**)&d, 65536ull * 14400ull + (1 << 20));
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	auto time_it = [&](const char* name, auto launch)
	{
		for (int i = 0; i < 100; ++i) launch(i);
		hipEventRecord(a);
		const int reps = 300;
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
	for (int blocks : {1024, 2048, 4096})
	{
		char name[128];
		snprintf(name, sizeof(name), "pose pattern persistent grid=%d", blocks);
		time_it(name, [&](int i) { hipLaunchKernelGGL(write_kernel_pose_persistent, dim3(blocks), dim3(256), 0, 0, (f32x4*)d, 65536u, 300u, float(i)); });
	}
	time_it("pose pattern, one wave per 1 KiB chunk", [&](int i) { hipLaunchKernelGGL(write_kernel_pose_chunks, dim3(65536 * 5 / 4), dim3(256), 0, 0, (f32x4*)d, 65536u, 300u, float(i)); });
	time_it("pose pattern, block cooperative sweep", [&](int i) { hipLaunchKernelGGL(write_kernel_pose_block, dim3(16384), dim3(256), 0, 0, (f32x4*)d, 65536u, 300u, float(i)); });
	bytes = 2 * 65536ull * 4800ull;
	time_it("pose pattern x2 instances (131072)", [&](int i) { hipLaunchKernelGGL(write_kernel_pose, dim3(32768), dim3(256), 0, 0, (f32x4*)d, 131072u, 300u, float(i)); });
	time_it("plain x2 bytes grid=131072", [&](int i) { hipLaunchKernelGGL(write_kernel_vec<0>, dim3(131072), dim3(256), 0, 0, (f32x4*)d, 2 * 65536ull * 300ull, float(i)); });
	bytes = 65536ull * 4800ull / 2;
	time_it("pose pattern x0.5 instances (32768)", [&](int i) { hipLaunchKernelGGL(write_kernel_pose, dim3(8192), dim3(256), 0, 0, (f32x4*)d, 32768u, 300u, float(i)); });
	time_it("plain x0.5 bytes grid=32768", [&](int i) { hipLaunchKernelGGL(write_kernel_vec<0>, dim3(32768), dim3(256), 0, 0, (f32x4*)d, 32768ull * 300ull, float(i)); });
	bytes = 65536ull * 4864ull;
	time_it("pose pattern, stride 4864 (128 B aligned), 4800 written", [&](int i) { hipLaunchKernelGGL(write_kernel_pose_strided, dim3(16384), dim3(256), 0, 0, (f32x4*)d, 65536u, 300u, 304u, float(i)); });
	time_it("pose pattern, 4864 B poses", [&](int i) { hipLaunchKernelGGL(write_kernel_pose, dim3(16384), dim3(256), 0, 0, (f32x4*)d, 65536u, 304u, float(i)); });
	bytes = 65536ull * 5120ull;
	time_it("pose pattern, 5120 B poses", [&](int i) { hipLaunchKernelGGL(write_kernel_pose, dim3(16384), dim3(256), 0, 0, (f32x4*)d, 65536u, 320u, float(i)); });
	bytes = 65536ull * 14400ull;
	time_it("pose pattern (wave per 14400 B)", [&](int i) { hipLaunchKernelGGL(write_kernel_pose, dim3(16384), dim3(256), 0, 0, (f32x4*)d, 65536u, 900u, float(i)); });
	time_it("pose pattern persistent 14400 B grid=2048", [&](int i) { hipLaunchKernelGGL(write_kernel_pose_persistent, dim3(2048), dim3(256), 0, 0, (f32x4*)d, 65536u, 900u, float(i)); });
	time_it("plain float4 944 MB grid=4096", [&](int i) { hipLaunchKernelGGL(write_kernel<0>, dim3(4096), dim3(256), 0, 0, d, 65536ull * 900ull, float(i)); });
	time_it("hipMemsetAsync 944 MB", [&](int i) { hipMemsetAsync(d, i, bytes, 0); });
	bytes = 65536ull * 4800ull;
	time_it("hipMemsetAsync", [&](int i) { hipMemsetAsync(d, i, bytes, 0); });
	return 0;
}
