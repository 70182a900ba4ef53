// staging.hip -- pinned host staging owned by the library and the transfers through it (host arrays in, solution out)
// (one of the translation units solver.hip was split into in round 5: solver.hip the handle and its device memory,
// staging.hip pinned staging and uploads, sparse_setup.hip the analysis of the sparse block path, sparse_enqueue.hip its launches,
// capi.hip the C ABI of include/slampp_hip.h)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <pthread.h>
#include "solver.h"
#include "sparse_inverse.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <sys/mman.h>

using namespace slampp;

static size_t pinned_bytes(size_t n_doubles) // what Alloc_Pinned maps for that many doubles
{
	const size_t n_huge = size_t(2) << 20;
	return (std::max<size_t>(n_doubles, 1) * sizeof(double) + n_huge - 1) / n_huge * n_huge;
}

// worker threads that are joined on every way out of the scope that started them: a std::thread destroyed while
// joinable is std::terminate (a wordless abort), and that is what an exception thrown between two emplace_back calls --
// std::system_error when the process is out of threads -- would otherwise leave behind
struct CJoiningThreads {
	std::vector<std::thread> v;
	~CJoiningThreads() { Join(); }
	void Join()
	{
		for(size_t i = 0; i < v.size(); ++ i) {
			if(v[i].joinable())
				v[i].join();
		}
	}
};

static void Free_Pinned(double *p, bool b_registered, size_t n_doubles, bool b_deferred = false)
{
	if(!p)
		return;
	if(b_deferred && !b_registered) { // a mapping of ours the driver never saw
		(void)munmap(p, pinned_bytes(n_doubles));
		return;
	}
	if(b_registered) {
		// a mapping of its own, never the allocator's memory: pages the driver has pinned do not go back into a heap.  If
		// the driver will not let go of them, the mapping stays (a leak of address space, not a block that two owners use)
		const hipError_t e = hipHostUnregister(p);
		if(e == hipSuccess)
			(void)munmap(p, pinned_bytes(n_doubles));
		else {
			(void)hipGetLastError();
			static std::atomic<bool> b_said(false);
			if(!b_said.exchange(true)) {
				fprintf(stderr, "libslampp_hip: hipHostUnregister failed (%s): %zu bytes of pinned staging stay mapped "
					"(said once per process)\n", hipGetErrorString(e), pinned_bytes(n_doubles));
			}
		}
	} else
		(void)hipHostFree(p);
}

void slampp_hip_solver::Free_Staging()
{
	// registered memory is a mapping of ours that the driver pinned: unlike hipHostFree, unregistering does not wait for copies
	// that still read it (a handle destroyed right after an asynchronous call: memory access fault at a host address)
	if(copy_stream)
		(void)hipStreamSynchronize(copy_stream);
	if(stream)
		(void)hipStreamSynchronize(stream);
	Join_Staging_Registration();
	Free_Pinned(p_pin_values, b_pin_values_registered, n_pin_values, b_pin_values_deferred);
	Free_Pinned(p_pin_rhs, b_pin_rhs_registered, n_pin_rhs, b_pin_rhs_deferred);
	b_pin_values_deferred = b_pin_rhs_deferred = false;
	b_pin_values_registered = b_pin_rhs_registered = false;
	p_pin_values = p_pin_rhs = 0;
	n_pin_values = n_pin_rhs = 0;
	n_uploaded = 0;
}

// Pinned host memory, pinned for EVERY device of the process (the Portable flags): the members of a device group DMA
// their shards out of the front handle's staging, each over its own link (group.hip checks that they can, see
// group_check_staging).  hipHostMalloc pays 0.2 ms per MB (62 ms for the 336 MB of C4's values, measured): nearly all of
// it is the kernel handing out and clearing 4 kB pages one at a time.  The same memory as 2 MB pages (madvise, where
// transparent huge pages are on or on request), first touched by a few threads and then registered, costs 1 - 5 ms
// and moves at the same 54 GB/s; without huge pages it is still no slower than hipHostMalloc.
static double *Alloc_Pinned(size_t n_doubles, bool &r_b_registered, bool b_defer, bool &r_b_deferred) // throw(std::bad_alloc, CDeviceError)
{
	const size_t n_huge = size_t(2) << 20;
	const size_t n_bytes = pinned_bytes(n_doubles);
	r_b_registered = false;
	r_b_deferred = false;
	if(n_bytes >= 4 * n_huge || b_defer) {
		// an anonymous mapping aligned to the huge page size (mapped one huge page longer, the ends cut off)
		char *p = 0;
		{
			void *p_map = mmap(0, n_bytes + n_huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
			if(p_map != MAP_FAILED) {
				char *p_begin = (char*)p_map, *p_aligned = (char*)((uintptr_t(p_begin) + n_huge - 1) / n_huge * n_huge);
				if(p_aligned > p_begin)
					(void)munmap(p_begin, size_t(p_aligned - p_begin));
				if(p_aligned + n_bytes < p_begin + n_bytes + n_huge)
					(void)munmap(p_aligned + n_bytes, size_t((p_begin + n_bytes + n_huge) - (p_aligned + n_bytes)));
				p = p_aligned;
			}
		}
		if(p && b_defer) { // neither touched nor pinned now: slampp_hip_solver::Register_Staging_Later()
			(void)madvise(p, n_bytes, MADV_HUGEPAGE);
			r_b_deferred = true;
			return (double*)p;
		}
		if(p) {
			(void)madvise(p, n_bytes, MADV_HUGEPAGE);
			const size_t n_threads = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(8, std::thread::hardware_concurrency()), n_bytes / (8 * n_huge)));
			try {
				CJoiningThreads threads;
				for(size_t t = 0; t < n_threads; ++ t) {
					const size_t n_begin = n_bytes / n_huge * t / n_threads * n_huge, n_end = n_bytes / n_huge * (t + 1) / n_threads * n_huge;
					auto touch = [p, n_begin, n_end]() {
						for(size_t i = n_begin; i < n_end; i += 4096)
							((volatile char*)p)[i] = 0;
					};
					if(t + 1 < n_threads)
						threads.v.emplace_back(touch);
					else
						touch();
				}
				threads.Join();
			} catch(std::system_error&) {
				// no more threads to be had: the registration below touches the pages itself
			}
			if(hipHostRegister(p, n_bytes, hipHostRegisterPortable | hipHostRegisterMapped) == hipSuccess) {
				r_b_registered = true;
				return (double*)p;
			}
			(void)hipGetLastError();
			(void)munmap(p, n_bytes);
		}
	}
	double *p = 0;
	const hipError_t e = hipHostMalloc((void**)&p, n_bytes, hipHostMallocPortable | hipHostMallocMapped);
	if(e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
		(void)hipGetLastError();
		throw std::bad_alloc();
	}
	if(e != hipSuccess)
		throw CDeviceError(std::string("hipHostMalloc: ") + hipGetErrorString(e));
	return p;
}

static void Grow_Pinned(double *&r_p, size_t &r_n, bool &r_b_registered, size_t n_doubles, bool b_defer, bool &r_b_deferred) // throws
{
	if(r_n >= n_doubles && r_p)
		return;
	// a staging that has to grow grows to twice its size at least (the reference's workspaces do the same, e.g.
	// LinearSolver_CholMod.cpp:898-901): an incremental solver hands over systems a few blocks larger at every call, and
	// pinning is milliseconds each time
	const size_t n_new = (r_p && r_n)? std::max(n_doubles, 2 * r_n) : n_doubles;
	Free_Pinned(r_p, r_b_registered, r_n, r_b_deferred);
	r_p = 0;
	r_n = 0;
	r_b_registered = r_b_deferred = false;
	r_p = Alloc_Pinned(n_new, r_b_registered, b_defer, r_b_deferred);
	r_n = n_new;
}

static double staging_wall_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the deferred registration of a first staging (solver.h): pins the mapping where it is, pages and all
void slampp_hip_solver::Register_Staging_Later()
{
	if((!b_pin_values_deferred || b_pin_values_registered) && (!b_pin_rhs_deferred || b_pin_rhs_registered))
		return;
	Join_Staging_Registration();
	t_staging_registration = std::thread([this]() {
		if(hipSetDevice(n_device) != hipSuccess)
			return;
		// (a failure leaves the mapping what it was: transfers keep going through the pageable path)
		if(b_pin_values_deferred && !b_pin_values_registered && p_pin_values) {
			if(hipHostRegister(p_pin_values, pinned_bytes(n_pin_values), hipHostRegisterPortable | hipHostRegisterMapped) == hipSuccess)
				b_pin_values_registered = true;
			else
				(void)hipGetLastError();
		}
		if(b_pin_rhs_deferred && !b_pin_rhs_registered && p_pin_rhs) {
			if(hipHostRegister(p_pin_rhs, pinned_bytes(n_pin_rhs), hipHostRegisterPortable | hipHostRegisterMapped) == hipSuccess)
				b_pin_rhs_registered = true;
			else
				(void)hipGetLastError();
		}
	});
}

void slampp_hip_solver::Join_Staging_Registration()
{
	if(t_staging_registration.joinable())
		t_staging_registration.join();
}

void slampp_hip_solver::Require_Staging()
{
	Join_Staging_Registration();
	const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0 && (n_pin_values < size_t(n_values) || !p_pin_values);
	const double t0 = staging_wall_ms();
	if(!copy_stream)
		SLAMPP_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)); // (handles made by slampp_hip_create have one already)
	if(!copy_done)
		SLAMPP_HIP_CHECK(hipEventCreateWithFlags(&copy_done, hipEventDisableTiming));
	const double t_streams = staging_wall_ms();
	if(n_pin_values < size_t(n_values) || !p_pin_values)
		n_uploaded = 0;
	if((n_pin_values < size_t(n_values) && p_pin_values) || (n_pin_rhs < size_t(n_scalars) && p_pin_rhs)) {
		(void)hipStreamSynchronize(copy_stream); // a buffer is about to be replaced: no copy may still read it
		if(stream)
			(void)hipStreamSynchronize(stream);
	}
	// (deferral: the first staging of a one-device handle that was told host arrays are coming -- solver.h.  Built early in round 6,
	// when pinning 58 MB beside the analysis slowed the analysis by more than the first upload out of unregistered memory cost;
	// with the analysis' work arrays on huge pages and in the heap that turned around -- first call through the header at C3
	// 66-71 ms deferred, 54-55 registered beside the analysis; Venice-like 73-78 against 61-62 -- and it is a development
	// knob now, SLAMPP_HIP_DEV_STAGING_DEFERRAL)
	const bool b_defer = !b_staging_ever && !p_pin_values && !p_pin_rhs && n_staging_ahead && !b_group_active && group_devices.empty() &&
		dev_knob_set("SLAMPP_HIP_DEV_STAGING_DEFERRAL");
	Grow_Pinned(p_pin_values, n_pin_values, b_pin_values_registered, size_t(n_values), b_defer, b_pin_values_deferred);
	const double t1 = staging_wall_ms();
	Grow_Pinned(p_pin_rhs, n_pin_rhs, b_pin_rhs_registered, size_t(n_scalars), b_defer, b_pin_rhs_deferred);
	b_staging_ever = true;
	const double t2 = staging_wall_ms();
	if(!b_group_active) { // (with landmark shards the values go from the staging straight to the members' devices)
		d_A.Alloc(size_t(n_values));
		d_rhs.Alloc(size_t(n_scalars));
	}
	if(b_timing) {
		fprintf(stderr, "[staging] copy stream %.2f ms, values %.2f ms (%s), rhs %.2f ms, device arrays %.2f ms\n", t_streams - t0, t1 - t_streams,
			b_pin_values_deferred? "mapped, registration deferred" : b_pin_values_registered? "registered" : "hipHostMalloc", t2 - t1, staging_wall_ms() - t2);
	}
}

// Copy workers that outlive the call (round 4).  Staged_Upload() and Parallel_Copy() used to start eight threads per call
// and join them: 0.1 - 0.2 ms each way on a 2 ms solve (C3 from host arrays).  One pool per process, made at first use and
// never taken down (its threads sleep on a condition variable between calls; after a job they spin for a moment first,
// since the next copy of a solve loop is usually microseconds away).  One job at a time: a caller that finds the pool
// taken (the member threads of a device group upload side by side) gets false and starts threads of its own as before.
class CCopyPool {
	std::mutex m_mutex;
	std::condition_variable m_wake;
	std::function<void(int)> m_job;
	std::atomic<uint64_t> m_n_generation{0};
	std::atomic<int> m_n_running{0};
	std::atomic<bool> m_b_taken{false};
	int m_n_threads = 0;
public:
	static CCopyPool &r_Get()
	{
		// (leaked on purpose: no destructor runs against sleeping threads at exit.  A child of fork() inherits the object but
		// none of its threads -- a job given to them would never run --: the child starts with no pool and makes its own)
		static std::mutex t_make;
		static const int n_registered = pthread_atfork(0, 0, []() { p_Instance().store(0); });
		(void)n_registered;
		CCopyPool *p_pool = p_Instance().load(std::memory_order_acquire);
		if(!p_pool) {
			std::lock_guard<std::mutex> lock(t_make);
			p_pool = p_Instance().load(std::memory_order_acquire);
			if(!p_pool) {
				p_pool = new CCopyPool();
				p_Instance().store(p_pool, std::memory_order_release);
			}
		}
		return *p_pool;
	}
	int n_Threads() const { return m_n_threads; }
	// f(t) on every worker, t = 0 .. n_Threads() - 1; returns at once (Wait() joins), false if the pool is busy or has no threads
	bool Start(std::function<void(int)> f)
	{
		if(!m_n_threads || m_b_taken.exchange(true))
			return false;
		{
			std::lock_guard<std::mutex> lock(m_mutex);
			m_job = std::move(f);
			m_n_running.store(m_n_threads, std::memory_order_relaxed);
			m_n_generation.fetch_add(1, std::memory_order_release);
		}
		m_wake.notify_all();
		return true;
	}
	void Wait()
	{
		while(m_n_running.load(std::memory_order_acquire) > 0)
			std::this_thread::yield();
		m_b_taken.store(false, std::memory_order_release);
	}
private:
	static std::atomic<CCopyPool*> &p_Instance()
	{
		static std::atomic<CCopyPool*> p_instance(0);
		return p_instance;
	}
	CCopyPool()
	{
		const unsigned n_hw = std::thread::hardware_concurrency();
		const int n_want = int(std::min<unsigned>(8, std::max<unsigned>(n_hw, 1)));
		try {
			for(int t = 0; t < n_want; ++ t) {
				std::thread([this, t]() { Work(t); }).detach();
				++ m_n_threads;
			}
		} catch(std::system_error&) {
			// fewer threads, or none (Start() then says no)
		}
	}
	void Work(int t)
	{
		uint64_t n_seen = 0;
		for(;;) {
			// a moment of spinning (the next job of a solve loop), then sleep
			const auto t_spin_end = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
			while(m_n_generation.load(std::memory_order_acquire) == n_seen && std::chrono::steady_clock::now() < t_spin_end)
				std::this_thread::yield();
			if(m_n_generation.load(std::memory_order_acquire) == n_seen) {
				std::unique_lock<std::mutex> lock(m_mutex);
				m_wake.wait(lock, [&]() { return m_n_generation.load(std::memory_order_acquire) != n_seen; });
			}
			n_seen = m_n_generation.load(std::memory_order_acquire);
			if(t < m_n_threads) // (a thread made before a later one failed to start still counts: m_n_threads only grows in the constructor)
				m_job(t);
			m_n_running.fetch_sub(1, std::memory_order_release);
		}
	}
};

// the chunks of a staged transfer: small first (the bus waits for the first chunk's copy: C3's 58 MB at 54 GB/s are 1.07 ms
// on the bus, and a first chunk of 8 MB was 0.25 ms of memcpy before the first byte moved), doubling up to n_max
static std::vector<size_t> staged_chunk_ends(size_t n, size_t n_first, size_t n_max)
{
	std::vector<size_t> ends;
	size_t b = 0, n_chunk = n_first;
	while(b < n) {
		b = std::min(n, b + n_chunk);
		ends.push_back(b);
		n_chunk = std::min(n_max, n_chunk * 2);
	}
	return ends;
}

// A caller's array to the device through pinned staging, in chunks: the DMA engines cannot be pointed at pageable
// memory, and one thread's memcpy is slower than PCIe -- a few host threads copy chunk c + 1 while chunk c is on the bus.
static void Staged_Upload(double *p_dev, double *p_pin, const double *p_src, size_t n, hipStream_t copy_stream)
{
	if(n < (size_t(1) << 19)) { // (4 MB: one thread, one transfer)
		memcpy(p_pin, p_src, n * sizeof(double));
		SLAMPP_HIP_CHECK(hipMemcpyAsync(p_dev, p_pin, n * sizeof(double), hipMemcpyHostToDevice, copy_stream));
		return;
	}
	const std::vector<size_t> ends = staged_chunk_ends(n, size_t(1) << 17, size_t((n <= (size_t(16) << 20))? 1 : 4) << 20); // 1 MB first; 8 / 32 MB at most
	const size_t n_chunks = ends.size();
	std::vector<std::atomic<int> > done(n_chunks);
	for(size_t c = 0; c < n_chunks; ++ c)
		done[c].store(0);
	CCopyPool &r_pool = CCopyPool::r_Get();
	int n_threads = r_pool.n_Threads();
	auto copy_share = [=, &done, &ends](int t, int n_of) {
		for(size_t c = 0; c < n_chunks; ++ c) {
			const size_t b = c? ends[c - 1] : 0, e = ends[c], n_piece = (e - b + n_of - 1) / n_of;
			const size_t pb = std::min(e, b + t * n_piece), pe = std::min(e, pb + n_piece);
			if(pe > pb)
				memcpy(p_pin + pb, p_src + pb, (pe - pb) * sizeof(double));
			done[c].fetch_add(1, std::memory_order_release);
		}
	};
	CJoiningThreads workers; // (only if the pool is taken) joined before `done` goes, whichever way this scope is left
	const bool b_pool = r_pool.Start([=](int t) { copy_share(t, n_threads); });
	if(!b_pool) {
		n_threads = int(std::min<unsigned>(8, std::max<unsigned>(std::thread::hardware_concurrency(), 1)));
		for(int t = 0; t < n_threads; ++ t)
			workers.v.emplace_back([=]() { copy_share(t, n_threads); });
	}
	hipError_t n_err = hipSuccess;
	for(size_t c = 0; c < n_chunks; ++ c) {
		while(done[c].load(std::memory_order_acquire) < n_threads)
			std::this_thread::yield();
		const size_t b = c? ends[c - 1] : 0, e = ends[c];
		if(n_err == hipSuccess)
			n_err = hipMemcpyAsync(p_dev + b, p_pin + b, (e - b) * sizeof(double), hipMemcpyHostToDevice, copy_stream);
	}
	if(b_pool)
		r_pool.Wait();
	else
		workers.Join();
	SLAMPP_HIP_CHECK(n_err);
}

// the way back: DMA into the pinned staging (already enqueued and waited for by the caller), then out of it
void Parallel_Copy(double *p_dst, const double *p_src, size_t n)
{
	if(n < (size_t(1) << 17)) { // (1 MB)
		memcpy(p_dst, p_src, n * sizeof(double));
		return;
	}
	CCopyPool &r_pool = CCopyPool::r_Get();
	const int n_threads = r_pool.n_Threads();
	const size_t n_piece = n_threads? (n + n_threads - 1) / n_threads : n;
	if(r_pool.Start([=](int t) {
		const size_t b = std::min(n, t * n_piece), e = std::min(n, b + n_piece);
		if(e > b)
			memcpy(p_dst + b, p_src + b, (e - b) * sizeof(double));
	}))
		r_pool.Wait();
	else
		memcpy(p_dst, p_src, n * sizeof(double)); // (the pool is another caller's for the moment)
}

// Lambda's values to d_A.  From the library's own pinned staging (the header class gathers the blocks of a
// CUberBlockMatrix straight into it, and may have sent leading chunks already): one DMA transfer of what is left.
// From a caller's array: through the staging, see Staged_Upload().
void slampp_hip_solver::Upload_Values(const double *p_values)
{
	Require_Staging();
	const size_t n = size_t(n_values);
	if(p_values == p_pin_values) {
		if(size_t(n_uploaded) < n) {
			SLAMPP_HIP_CHECK(hipMemcpyAsync(d_A.p() + n_uploaded, p_pin_values + n_uploaded, (n - size_t(n_uploaded)) * sizeof(double),
				hipMemcpyHostToDevice, copy_stream));
		}
	} else
		Staged_Upload(d_A.p(), p_pin_values, p_values, n, copy_stream);
	n_uploaded = 0;
}

// the right-hand side to d_rhs (same two cases), then `stream` waits for everything the copy stream was given
void Upload_Rhs_And_Join(slampp_hip_solver &s, const double *p_rhs)
{
	s.Require_Staging();
	if(p_rhs == s.p_pin_rhs) {
		SLAMPP_HIP_CHECK(hipMemcpyAsync(s.d_rhs.p(), s.p_pin_rhs, size_t(s.n_scalars) * sizeof(double), hipMemcpyHostToDevice,
			s.copy_stream));
	} else
		Staged_Upload(s.d_rhs.p(), s.p_pin_rhs, p_rhs, size_t(s.n_scalars), s.copy_stream);
	SLAMPP_HIP_CHECK(hipEventRecord(s.copy_done, s.copy_stream));
	SLAMPP_HIP_CHECK(hipStreamWaitEvent(s.stream, s.copy_done, 0));
}

// the values alone (entry points without a right-hand side)
void Upload_Values_And_Join(slampp_hip_solver &s, const double *p_values)
{
	s.Upload_Values(p_values);
	SLAMPP_HIP_CHECK(hipEventRecord(s.copy_done, s.copy_stream));
	SLAMPP_HIP_CHECK(hipStreamWaitEvent(s.stream, s.copy_done, 0));
}

