// host_pool.h -- the thread pool of the host side of lock-step solves.  Header-only and free of HIP, so that
// it is built and run on the CPU under ThreadSanitizer / AddressSanitizer (tests/cpp/hostlm_stress.cpp,
// tests/test_host_sanitizers.py) exactly as it ships inside libebo_hip.so.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ebo
{
// A small thread pool for the host side of lock-step solves: the per-window LM state machines
// are independent, and with hundreds of windows their linear algebra is what bounds a round
// (measured: 256 windows of the reference configuration, 10 us per window-round).  The pool
// lives for ONE solve call (threads are created when the call has enough independent problems
// and joined before it returns): no thread of this library outlives an API call, so process
// exit, dlclose and profilers that wrap the process never meet a parked worker.
// A round's work is tens of microseconds per thread and rounds follow each other every ~100 us, so
// waking parked workers through a condition variable (30-50 us) was most of a round's host time
// (64 reference-default windows: 9.7 of 17.6 ms in the LM steps): workers and the caller SPIN on an
// atomic for EBO_HOST_SPIN_US microseconds (default 200) before they park.
class HostPool
{
   public:
	// problems: independent state machines of the call; perThread: how many make a thread worth it
	HostPool(size_t problems, size_t perThread)
	{
		unsigned hw = std::thread::hardware_concurrency();
		const char* v = std::getenv("EBO_HOST_THREADS");
		size_t want = v ? static_cast<size_t>(std::max(1, std::atoi(v))) : std::min<size_t>(hw ? hw : 1, 16);
		want = std::min(want, std::max<size_t>(1, problems / std::max<size_t>(1, perThread)));
		{
			const char* sv = std::getenv("EBO_HOST_SPIN_US");
			spinUs_ = (sv && *sv) ? std::max(0L, std::atol(sv)) : 200;
		}
		for (size_t i = 1; i < want; ++i)
		{
			workers_.emplace_back([this] { run(); });
		}
	}
	// fn(begin, end) over [0, n) in contiguous chunks; the caller works too.
	template <class F>
	void parallel_for(size_t n, size_t minPerThread, F&& fn)
	{
		const size_t maxT = workers_.size() + 1;
		size_t T = std::min(maxT, std::max<size_t>(1, n / std::max<size_t>(1, minPerThread)));
		if (T <= 1)
		{
			fn(static_cast<size_t>(0), n);
			return;
		}
		std::function<void(size_t, size_t)> f = fn;
		const size_t chunk = (n + T - 1) / T;
		{
			std::unique_lock<std::mutex> lk(mu_);
			job_ = &f;
			n_ = n;
			chunk_ = chunk;
			next_ = 1;  // chunk 0 is the caller's
			chunks_ = T;
			pending_.store(T - 1, std::memory_order_relaxed);
			++generation_;
			published_.store(generation_, std::memory_order_release);
		}
		cv_.notify_all();
		fn(static_cast<size_t>(0), std::min(n, chunk));
		spin_until([&] { return pending_.load(std::memory_order_acquire) == 0; });
		std::unique_lock<std::mutex> lk(mu_);
		done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
		job_ = nullptr;
	}

	HostPool(const HostPool&) = delete;
	HostPool& operator=(const HostPool&) = delete;
	~HostPool()
	{
		{
			std::unique_lock<std::mutex> lk(mu_);
			stop_ = true;
			published_.store(~static_cast<size_t>(0), std::memory_order_release);  // ends the spinning
		}
		cv_.notify_all();
		for (auto& t : workers_)
		{
			t.join();
		}
	}
   private:
	template <class P>
	void spin_until(P&& ready) const
	{
		if (spinUs_ <= 0 || ready())
		{
			return;
		}
		const auto t0 = std::chrono::steady_clock::now();
		for (;;)
		{
			for (int i = 0; i < 32; ++i)
			{
				if (ready())
				{
					return;
				}
#if defined(__x86_64__) || defined(__i386__)
				__builtin_ia32_pause();
#endif
			}
			if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spinUs_)
			{
				return;
			}
		}
	}
	void run()
	{
		size_t seen = 0;  // the last generation this worker has nothing more to do for
		for (;;)
		{
			spin_until([&] { return published_.load(std::memory_order_acquire) != seen; });
			std::function<void(size_t, size_t)>* job = nullptr;
			size_t b = 0, e = 0;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
				if (stop_)
				{
					return;
				}
				if (!job_ || next_ >= chunks_)
				{
					seen = generation_;  // every chunk of this generation has been taken
					continue;
				}
				const size_t k = next_++;
				if (next_ >= chunks_)
				{
					seen = generation_;
				}
				job = job_;
				b = k * chunk_;
				e = std::min(n_, b + chunk_);
			}
			if (b < e)
			{
				(*job)(b, e);
			}
			if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1)
			{
				std::unique_lock<std::mutex> lk(mu_);
				done_.notify_all();
			}
		}
	}
	std::vector<std::thread> workers_;
	std::mutex mu_;
	std::condition_variable cv_, done_;
	std::function<void(size_t, size_t)>* job_ = nullptr;
	size_t n_ = 0, chunk_ = 0, next_ = 0, chunks_ = 0, generation_ = 0;
	std::atomic<size_t> pending_{0}, published_{0};
	long spinUs_ = 200;
	bool stop_ = false;
};
}  // namespace ebo
