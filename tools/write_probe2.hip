// tools/write_probe2.hip -- measurement aid: which properties of a pose-shaped write stream cost HBM efficiency on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/write_probe2.hip -o tools/write_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per pose, back to back 1 KiB stores; `lds_bytes` of dynamic LDS per block throttles the waves in flight per CU
__global__ __launch_bounds__(256) void pose_kernel(f32x4* __restrict__ dst, uint32_t num_poses, uint32_t quads_per_pose, float seed)
{
	extern __shared__ uint8_t lds[];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t pose = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pose >= num_poses) return;
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(pose) * quads_per_pose;
	for (uint32_t q = lane; q < quads_per_pose; q += 64)
		p[q] = v;
}

// every block writes `chunks` consecutive 4 KiB chunks (aligned), one per iteration
__global__ __launch_bounds__(256) void block_chunks_kernel(f32x4* __restrict__ dst, uint32_t chunks, float seed)
{
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(blockIdx.x) * chunks * 256;
	for (uint32_t c = 0; c < chunks; ++c)
		p[c * 256 + threadIdx.x] = v;
}

// every WAVE writes `chunks` consecutive 1 KiB chunks (aligned)
__global__ __launch_bounds__(256) void wave_chunks_kernel(f32x4* __restrict__ dst, uint32_t chunks, float seed)
{
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
	f32x4* p = dst + uint64_t(wave) * chunks * 64;
	for (uint32_t c = 0; c < chunks; ++c)
		p[c * 64 + (threadIdx.x & 63)] = v;
}

int main()
{
	f32x4* d;
	hipMalloc((void**)&d, 1ull << 30);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	uint64_t bytes = 0;
	auto time_it = [&](const char* name, auto launch)
	{
		for (int i = 0; i < 100; ++i) launch(i);
		hipEventRecord(a);
		const int reps = 300;
		for (int i = 0; i < reps; ++i) launch(i);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("%-64s %8.2f us  %8.1f GB/s\n", name, ms / reps * 1000.0, bytes * reps / (ms * 1e-3) / 1e9);
	};
	char name[160];
	for (uint32_t poses : {65536u, 131072u})
		for (uint32_t lds_kb : {0u, 10u, 20u, 40u, 80u})		// 160 KB LDS per CU: 0 -> 8 blocks/CU (wave limit), 10 -> 16?, 20 -> 8, 40 -> 4, 80 -> 2
		{
			bytes = uint64_t(poses) * 4800;
			snprintf(name, sizeof(name), "pose 4800 B x %u, %u KB LDS per block", poses, lds_kb);
			time_it(name, [&](int i) { hipLaunchKernelGGL(pose_kernel, dim3(poses / 4), dim3(256), lds_kb * 1024, 0, d, poses, 300u, float(i)); });
		}
	for (uint32_t quads : {256u, 288u, 296u, 300u, 304u, 312u, 320u, 384u, 512u, 900u})
	{
		const uint32_t poses = uint32_t((65536ull * 300) / quads);
		bytes = uint64_t(poses) * quads * 16;
		snprintf(name, sizeof(name), "pose %u B x %u (same total bytes)", quads * 16, poses);
		time_it(name, [&](int i) { hipLaunchKernelGGL(pose_kernel, dim3((poses + 3) / 4), dim3(256), 0, 0, d, poses, quads, float(i)); });
	}
	for (uint32_t chunks : {1u, 2u, 5u, 8u})
	{
		const uint32_t blocks = uint32_t(65536ull * 4800 / (chunks * 4096));
		bytes = uint64_t(blocks) * chunks * 4096;
		snprintf(name, sizeof(name), "block writes %u consecutive 4 KiB chunks, grid %u", chunks, blocks);
		time_it(name, [&](int i) { hipLaunchKernelGGL(block_chunks_kernel, dim3(blocks), dim3(256), 0, 0, d, chunks, float(i)); });
		snprintf(name, sizeof(name), "wave writes %u consecutive 1 KiB chunks, grid %u", chunks, blocks);
		time_it(name, [&](int i) { hipLaunchKernelGGL(wave_chunks_kernel, dim3(blocks), dim3(256), 0, 0, d, chunks, float(i)); });
	}
	return 0;
}
