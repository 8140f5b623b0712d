// tools/write_probe3.hip -- measurement aid: stores per wave vs bytes per block vs block size, aligned 1 KiB chunks, same total bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// block of kWaves waves; every wave writes kStores consecutive 1 KiB chunks (immediate offsets, no loop)
template<int kWaves, int kStores>
__global__ __launch_bounds__(kWaves * 64) void wave_stores(f32x4* __restrict__ dst, float seed)
{
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	const uint32_t wave = blockIdx.x * kWaves + (threadIdx.x >> 6);
	f32x4* p = dst + uint64_t(wave) * kStores * 64 + (threadIdx.x & 63);
	#pragma unroll
	for (int c = 0; c < kStores; ++c)
		p[c * 64] = v;
}

// like above but the kStores chunks of a wave are interleaved with the other waves of the block (block sweeps its region)
template<int kWaves, int kStores>
__global__ __launch_bounds__(kWaves * 64) void block_sweep(f32x4* __restrict__ dst, float seed)
{
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(blockIdx.x) * kWaves * kStores * 64 + threadIdx.x;
	#pragma unroll
	for (int c = 0; c < kStores; ++c)
		p[c * kWaves * 64] = v;
}

// two chunks per wave, the second one half the buffer away
__global__ __launch_bounds__(256) void far_apart(f32x4* __restrict__ dst, uint64_t half_quads, float seed)
{
	const f32x4 v = { seed, seed + 1, seed + 2, seed + 3 };
	f32x4* p = dst + uint64_t(blockIdx.x) * 256 + threadIdx.x;
	p[0] = v;
	p[half_quads] = v;
}

int main()
{
	f32x4* d;
	hipMalloc((void**)&d, 1ull << 30);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	const uint64_t total_kib = 65536ull * 4800 / 1024 / 120 * 120;		// divisible by every chunk count below
	const uint64_t bytes = total_kib * 1024;
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
#define RUN(kernel, W, S) time_it(#kernel " waves/block=" #W " stores/wave=" #S, [&](int i) { hipLaunchKernelGGL((kernel<W, S>), dim3(uint32_t(total_kib / (W * S))), dim3(W * 64), 0, 0, d, float(i)); })
	RUN(wave_stores, 4, 1); RUN(wave_stores, 4, 2); RUN(wave_stores, 4, 3); RUN(wave_stores, 4, 5); RUN(wave_stores, 4, 8);
	RUN(wave_stores, 1, 1); RUN(wave_stores, 1, 2); RUN(wave_stores, 1, 5);
	RUN(wave_stores, 2, 1); RUN(wave_stores, 2, 2); RUN(wave_stores, 8, 1); RUN(wave_stores, 8, 2); RUN(wave_stores, 16, 1);
	RUN(block_sweep, 4, 2); RUN(block_sweep, 4, 5); RUN(block_sweep, 8, 2); RUN(block_sweep, 16, 2);
	time_it("two chunks per thread, half a buffer apart", [&](int i) { hipLaunchKernelGGL(far_apart, dim3(uint32_t(total_kib / 8)), dim3(256), 0, 0, d, total_kib * 64 / 2, float(i)); });
	RUN(wave_stores, 4, 1);
	return 0;
}
