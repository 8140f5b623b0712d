// tools/dispatch_probe.hip -- measurement aid (round 3): how fast the hardware launches and retires workgroups of the pose kernels' shape
// (4 waves, 20 KB of LDS) when they do (almost) nothing: the floor under any one-wave-per-window kernel.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/dispatch_probe.hip -o tools/dispatch_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(1024) void empty_kernel(uint32_t* sink, uint32_t sleep_units)
{
	extern __shared__ uint8_t pad[];
	for (uint32_t i = 0; i < sleep_units; ++i)
		__builtin_amdgcn_s_sleep(1);
	if (pad[threadIdx.x] == 255 && sink == nullptr)
		sink[0] = 1;
}

int main()
{
	uint32_t* sink;
	(void)hipMalloc((void**)&sink, 4096);
	(void)hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	auto run = [&](const char* name, uint32_t blocks, uint32_t threads, uint32_t lds, uint32_t sleep_units)
	{
		for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), lds, 0, sink, sleep_units);
		(void)hipEventRecord(a);
		const int reps = 100;
		for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), lds, 0, sink, sleep_units);
		(void)hipEventRecord(b); (void)hipEventSynchronize(b);
		float ms; (void)hipEventElapsedTime(&ms, a, b);
		const double us = ms / reps * 1000.0;
		printf("%-64s %8.2f us  %7.1f workgroups/us  %7.1f waves/us\n", name, us, blocks / us, double(blocks) * (threads / 64) / us);
		fflush(stdout);
	};
	for (int round = 0; round < 2; ++round)
	{
		run("49152 x 4 waves, 20 KB LDS, empty", 49152, 256, 20 * 1024, 0);
		run("16384 x 4 waves, 20 KB LDS, empty", 16384, 256, 20 * 1024, 0);
		run("49152 x 4 waves, no LDS, empty", 49152, 256, 0, 0);
		run("196608 x 1 wave, 5 KB LDS, empty", 196608, 64, 5 * 1024, 0);
		run("98304 x 2 waves, 10 KB LDS, empty", 98304, 128, 10 * 1024, 0);
		run("24576 x 8 waves, 40 KB LDS, empty", 24576, 512, 40 * 1024, 0);
		run("12288 x 16 waves, 80 KB LDS, empty", 12288, 1024, 80 * 1024, 0);
		run("49152 x 4 waves, 20 KB LDS, sleeps 64 x 64 clk (~1.7 us)", 49152, 256, 20 * 1024, 64);
		run("49152 x 4 waves, 20 KB LDS, sleeps 256 x 64 clk (~7 us)", 49152, 256, 20 * 1024, 256);
		run("24576 x 8 waves, 40 KB LDS, sleeps 256 x 64 clk (~7 us)", 24576, 512, 40 * 1024, 256);
		run("196608 x 1 wave, 5 KB LDS, sleeps 256 x 64 clk (~7 us)", 196608, 64, 5 * 1024, 256);
	}
	return 0;
}
