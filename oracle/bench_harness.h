// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
//
// The thread harness of the cpu_baseline leg (bench.py): runs one decode loop per host thread for a fixed wall-clock
// window and reports how many poses all threads completed inside it. Follows the shape of the reference's own benchmark
// (tools/acl_decompressor/sources/benchmark.cpp:94-101,232-281: one context per worker, warm data, time a fixed loop) with
// what a many-core box needs on top:
//   * every thread is PINNED to one CPU of the process's affinity mask (thread t -> allowed[t % allowed.size()]),
//   * threads wait on a condition variable (no spinning: a yield() loop of 256 waiting threads steals the CPUs the
//     late starters still need) and start the timed window together,
//   * the window is time based (>= `seconds`): every thread walks its share of the instance list again and again and
//     checks a stop flag every few poses, so no thread's start-up skew or a short share is part of the figure,
//   * throughput = poses completed by all threads / (stop time - start time).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__linux__)
	#include <pthread.h>
	#include <sched.h>
#endif

namespace bench_harness
{
	// CPUs this process may run on, in index order
	inline std::vector<int> allowed_cpus()
	{
		std::vector<int> cpus;
#if defined(__linux__)
		cpu_set_t set;
		CPU_ZERO(&set);
		if (sched_getaffinity(0, sizeof(set), &set) == 0)
			for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu)
				if (CPU_ISSET(cpu, &set))
					cpus.push_back(cpu);
#endif
		if (cpus.empty())
			cpus.push_back(0);
		return cpus;
	}

	inline void pin_current_thread(int cpu)
	{
#if defined(__linux__)
		cpu_set_t set;
		CPU_ZERO(&set);
		CPU_SET(cpu, &set);
		(void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
#else
		(void)cpu;
#endif
	}

	// `make_worker(thread_index)` returns a callable `uint64_t step()`: decodes a few poses (its warm-up call is untimed) and
	// returns how many. Returns poses per second over the common window; *out_total_poses (optional) the poses counted.
	template<class make_worker_type>
	double run_timed(uint32_t num_threads, double seconds, bool pin, make_worker_type&& make_worker, uint64_t* out_total_poses)
	{
		if (num_threads == 0)
			num_threads = 1;
		const std::vector<int> cpus = allowed_cpus();

		std::mutex mutex;
		std::condition_variable wake;
		uint32_t num_ready = 0;
		bool go = false;
		std::atomic<bool> stop(false);
		std::vector<uint64_t> counts(num_threads, 0);

		auto thread_main = [&](uint32_t thread_index)
		{
			if (pin)
				pin_current_thread(cpus[thread_index % cpus.size()]);
			auto step = make_worker(thread_index);
			(void)step();		// warm-up: binds the context, touches the clip and the output buffer
			{
				std::unique_lock<std::mutex> lock(mutex);
				num_ready++;
				wake.notify_all();
				wake.wait(lock, [&]() { return go; });
			}
			uint64_t count = 0;
			while (!stop.load(std::memory_order_relaxed))
				count += step();
			counts[thread_index] = count;
		};

		std::vector<std::thread> threads;
		threads.reserve(num_threads);
		for (uint32_t t = 0; t < num_threads; ++t)
			threads.emplace_back(thread_main, t);

		std::chrono::steady_clock::time_point start;
		{
			std::unique_lock<std::mutex> lock(mutex);
			wake.wait(lock, [&]() { return num_ready == num_threads; });
			go = true;
			start = std::chrono::steady_clock::now();
			wake.notify_all();
		}
		std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
		stop.store(true);
		const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
		for (std::thread& t : threads)
			t.join();

		uint64_t total = 0;
		for (uint64_t count : counts)
			total += count;
		if (out_total_poses != nullptr)
			*out_total_poses = total;
		return double(total) / elapsed;
	}
}
