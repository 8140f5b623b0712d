// tools/write_probe4.hip -- measurement aid: the pose write pattern (one wave per 4800 byte pose) with its stores unrolled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rolled loop (what tools/write_probe.hip measures)
__global__ __launch_bounds__(256) void pose_loop(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pose >= num_poses) return;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(pose) * quads;
	for (uint32_t q = lane; q < quads; q += 64)
		p[q] = v;
}

// 5 rows, one base address, immediate offsets; full rows unpredicated (wave uniform count), the partial row under an exec mask
__global__ __launch_bounds__(256) void pose_unrolled(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pose >= num_poses) return;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(pose) * quads + lane;
	const uint32_t full_rows = quads / 64;
	#pragma unroll
	for (uint32_t r = 0; r < 5; ++r)
	{
		if (r < full_rows)
			p[r * 64] = v;
		else if (r == full_rows && r * 64 + lane < quads)
			p[r * 64] = v;
	}
}

// the same but the 4800 bytes are written by 75 lanes x 64 bytes?? no: by 5 rows where row 0 is the PARTIAL one (tail first)
__global__ __launch_bounds__(256) void pose_unrolled_tail_first(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pose >= num_poses) return;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(pose) * quads + lane;
	const uint32_t full_rows = quads / 64;
	if (full_rows * 64 + lane < quads)
		p[full_rows * 64] = v;
	#pragma unroll
	for (uint32_t r = 0; r < 4; ++r)
		if (r < full_rows)
			p[r * 64] = v;
}

int main()
{
	f32x4* d;
	hipMalloc((void**)&d, 1ull << 30);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	const uint64_t bytes = 65536ull * 4800;
	auto time_it = [&](const char* name, auto launch)
	{
		for (int i = 0; i < 200; ++i) launch(i);
		hipEventRecord(a);
		const int reps = 500;
		for (int i = 0; i < reps; ++i) launch(i);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("%-56s %8.2f us  %8.1f GB/s\n", name, ms / reps * 1000.0, bytes * reps / (ms * 1e-3) / 1e9);
	};
	for (int round = 0; round < 2; ++round)
	{
		time_it("pose 4800 B, rolled loop", [&](int i) { hipLaunchKernelGGL(pose_loop, dim3(16384), dim3(256), 0, 0, d, 65536u, 300u, float(i)); });
		time_it("pose 4800 B, unrolled rows", [&](int i) { hipLaunchKernelGGL(pose_unrolled, dim3(16384), dim3(256), 0, 0, d, 65536u, 300u, float(i)); });
		time_it("pose 4800 B, unrolled, partial row first", [&](int i) { hipLaunchKernelGGL(pose_unrolled_tail_first, dim3(16384), dim3(256), 0, 0, d, 65536u, 300u, float(i)); });
		time_it("hipMemsetAsync", [&](int i) { hipMemsetAsync(d, i, bytes, 0); });
	}
	return 0;
}
