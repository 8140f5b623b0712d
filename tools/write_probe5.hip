// tools/write_probe5.hip -- measurement aid (round 3): what the HBM write path sustains for the 300-bone rig's store pattern
// (196 608 windows of <= 4 992 bytes, three per 14 400 byte pose, one wave per window, streaming stores) as a function of
//   * resident waves per CU (dynamic LDS padding),
//   * WHEN a resident wave stores (at once / after a fixed delay / after a random delay): does the order in which windows reach
//     memory matter, or the number of waves that hold stores in flight?
//   * the address span of the windows in flight (work items scrambled inside blocks of 8 192),
//   * a chain of dependent scalar loads in front of the stores (the decode's seek).
// build: hipcc --offload-arch=gfx950 -O3 tools/write_probe5.hip -o tools/write_probe5.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_streaming(void* address, f32x4 value)
{
	asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" :: "v"(address), "v"(value) : "memory");
}

struct probe_params
{
	uint32_t num_items;
	uint32_t windows_per_pose;		// 3
	uint32_t window_quads;			// 312
	uint32_t pose_quads;			// 900
	uint32_t delay_mode;			// 0 none, 1 fixed, 2 random (hash of the item), 3 ordered: wait until item - slack has stored
	uint32_t delay_units;			// s_sleep units of 64 clocks
	uint32_t scramble;				// 1: bit-reverse the low 13 bits of the item index (the in-flight set covers the same addresses in another order)
	uint32_t chain_hops;			// dependent scalar loads in front of the stores
	uint32_t plain_stores;			// 1: ordinary stores instead of sc0 sc1 nt
};

__global__ __launch_bounds__(256) void rig_store_kernel(f32x4* __restrict__ dst, const uint32_t* __restrict__ chain, probe_params p, float seed)
{
	extern __shared__ uint8_t pad[];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint32_t item = blockIdx.x * 4 + wave;
	if (item >= p.num_items)
		return;
	if (p.scramble)
	{
		const uint32_t low = item & 8191u;
		item = (item & ~8191u) | (__builtin_bitreverse32(low) >> 19);
	}

	uint32_t hop = item & 4095u;
	for (uint32_t h = 0; h < p.chain_hops; ++h)
		hop = ((const __attribute__((address_space(4))) uint32_t*)chain)[hop & 4095u] + h;
	const uint32_t chained = hop;

	if (p.delay_mode == 1)
		for (uint32_t i = 0; i < p.delay_units; ++i)
			__builtin_amdgcn_s_sleep(1);
	else if (p.delay_mode == 2)
	{
		const uint32_t units = ((item * 2654435761u) >> 16) % (2u * p.delay_units + 1u);
		for (uint32_t i = 0; i < units; ++i)
			__builtin_amdgcn_s_sleep(1);
	}

	const uint32_t pose = item / p.windows_per_pose;
	const uint32_t window = item - pose * p.windows_per_pose;
	const uint32_t first_quad = window * p.window_quads;
	const uint32_t quads = min(p.pose_quads - first_quad, p.window_quads);
	const f32x4 v = { seed, seed + 1, float(chained), seed + 3 };
	f32x4* out = dst + uint64_t(pose) * p.pose_quads + first_quad + lane;
	const uint32_t full_rows = quads / 64;
	#pragma unroll
	for (uint32_t r = 0; r < 5; ++r)
	{
		if (r < full_rows || (r == full_rows && r * 64 + lane < quads))
		{
			if (p.plain_stores)
				out[r * 64] = v;
			else
				store_streaming(&out[r * 64], v);
		}
	}
	if (pad[0] == 255 && seed == -1.0f)
		out[0] = v;		// keeps the LDS allocation alive
}

int main()
{
	f32x4* d;
	uint32_t* chain;
	const uint32_t instances = 65536;
	const uint64_t bytes = uint64_t(instances) * 14400;
	hipMalloc((void**)&d, bytes + (1u << 20));
	hipMalloc((void**)&chain, 4096 * 4);
	{
		uint32_t host[4096];
		for (uint32_t i = 0; i < 4096; ++i)
			host[i] = (i * 1237u + 511u) & 4095u;
		hipMemcpy(chain, host, sizeof(host), hipMemcpyHostToDevice);
	}
	hipFuncSetAttribute((const void*)rig_store_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);

	auto run = [&](const char* name, uint32_t lds_bytes, probe_params p)
	{
		const uint32_t blocks = (p.num_items + 3) / 4;
		for (int i = 0; i < 50; ++i)
			hipLaunchKernelGGL(rig_store_kernel, dim3(blocks), dim3(256), lds_bytes, 0, d, chain, p, float(i));
		hipEventRecord(a);
		const int reps = 200;
		for (int i = 0; i < reps; ++i)
			hipLaunchKernelGGL(rig_store_kernel, dim3(blocks), dim3(256), lds_bytes, 0, d, chain, p, float(i));
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("%-72s %8.2f us  %7.1f GB/s\n", name, ms / reps * 1000.0, bytes * reps / (ms * 1e-3) / 1e9);
		fflush(stdout);
	};

	probe_params base = { instances * 3u, 3u, 312u, 900u, 0u, 0u, 0u, 0u, 0u };
	// LDS per workgroup of 4 waves -> workgroups per CU: 20 KB -> 8 (32 waves), 40 KB -> 4 (16), 52 KB -> 3 (12), 80 KB -> 2 (8), 159 KB -> 1 (4)
	const uint32_t lds_for_waves[5][2] = { { 32, 20 * 1024 }, { 16, 40 * 1024 }, { 12, 52 * 1024 }, { 8, 80 * 1024 }, { 4, 159 * 1024 } };
	char name[160];
	for (int round = 0; round < 2; ++round)
	{
		for (const auto& entry : lds_for_waves)
		{
			snprintf(name, sizeof(name), "stores only, %2u waves per CU", entry[0]);
			run(name, entry[1], base);
		}
		for (uint32_t hops : { 3u, 6u })
			for (const auto& entry : lds_for_waves)
			{
				if (entry[0] == 12 || entry[0] == 4) continue;
				probe_params p = base; p.chain_hops = hops;
				snprintf(name, sizeof(name), "%u dependent scalar loads, then stores, %2u waves per CU", hops, entry[0]);
				run(name, entry[1], p);
			}
		for (uint32_t units : { 32u, 96u, 192u })		// x 64 clocks: ~0.9 / 2.6 / 5.2 us
		{
			probe_params p = base; p.delay_mode = 1; p.delay_units = units;
			snprintf(name, sizeof(name), "32 waves per CU, fixed delay of %u x 64 clocks, then stores", units);
			run(name, 20 * 1024, p);
			p.delay_mode = 2;
			snprintf(name, sizeof(name), "32 waves per CU, random delay of 0 .. %u x 64 clocks, then stores", 2 * units);
			run(name, 20 * 1024, p);
		}
		{
			probe_params p = base; p.scramble = 1;
			run("stores only, 32 waves per CU, items scrambled inside blocks of 8192", 20 * 1024, p);
			run("stores only,  8 waves per CU, items scrambled inside blocks of 8192", 80 * 1024, p);
			p = base; p.plain_stores = 1;
			run("plain stores, 32 waves per CU", 20 * 1024, p);
			run("plain stores,  8 waves per CU", 80 * 1024, p);
		}
		hipEventRecord(a);
		for (int i = 0; i < 200; ++i) hipMemsetAsync(d, i, bytes, 0);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("%-72s %8.2f us  %7.1f GB/s\n", "hipMemsetAsync", ms / 200 * 1000.0, bytes * 200 / (ms * 1e-3) / 1e9);
	}
	return 0;
}
