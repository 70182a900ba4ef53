// solver.hip -- C ABI of libslampp_hip.so (include/slampp_hip.h): handle, device memory,
// orchestration of the sparse-Cholesky and Schur paths.  Host code only; kernels live in
// sparse_kernels.hip / schur.hip / dense_chol.hip.
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

slampp_hip_solver::slampp_hip_solver()
	:n_device(0), stream(0), n_dense_nb(64), b_shard_primary(1), n_shard_rank(-1), n_shard_world(0), n_marginals_dense(0), n_schur_sparse(-1), b_has_structure(false),
	b_analyzed(false), b_factored(false), n_mode(SLAMPP_HIP_MODE_SPARSE), n_matrix_cut(0),
	n_values(0), n_scalars(0), n_bottom_stages(1), n_dense_gaps(0), b_dense_tiles(false), b_dense_clean(false), n_dense_top_tiles(-1), n_dense_blks(0), n_dense_cols(0),
	n_dense_dim(0), n_dense_pad(0),
	p_host_flag(0), p_schur(0), p_allreduce(0), p_allreduce_context(0),
	b_profile(0), n_open_phase(-1)
{
	memset(&dplan, 0, sizeof(dplan));
	memset(&times, 0, sizeof(times));
}

slampp_hip_solver::~slampp_hip_solver()
{
	Free_Device();
	for(size_t i = 0; i < phase_pending.size(); ++ i) {
		(void)hipEventDestroy(phase_pending[i].start);
		(void)hipEventDestroy(phase_pending[i].stop);
	}
	for(size_t i = 0; i < event_pool.size(); ++ i)
		(void)hipEventDestroy(event_pool[i]);
	if(p_host_flag)
		(void)hipHostFree(p_host_flag);
	Free_Staging();
	if(copy_done)
		(void)hipEventDestroy(copy_done);
	if(copy_stream)
		(void)hipStreamDestroy(copy_stream);
	if(stream)
		(void)hipStreamDestroy(stream);
}

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

static void Free_Pinned(double *p, bool b_registered, size_t n_doubles)
{
	if(!p)
		return;
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
	Free_Pinned(p_pin_values, b_pin_values_registered, n_pin_values);
	Free_Pinned(p_pin_rhs, b_pin_rhs_registered, n_pin_rhs);
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
static double *Alloc_Pinned(size_t n_doubles, bool &r_b_registered) // throw(std::bad_alloc, CDeviceError)
{
	const size_t n_huge = size_t(2) << 20;
	const size_t n_bytes = pinned_bytes(n_doubles);
	r_b_registered = false;
	if(n_bytes >= 4 * n_huge) {
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

static void Grow_Pinned(double *&r_p, size_t &r_n, bool &r_b_registered, size_t n_doubles) // throws
{
	if(r_n >= n_doubles && r_p)
		return;
	Free_Pinned(r_p, r_b_registered, r_n);
	r_p = 0;
	r_n = 0;
	r_p = Alloc_Pinned(n_doubles, r_b_registered);
	r_n = n_doubles;
}

static double staging_wall_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void slampp_hip_solver::Require_Staging()
{
	const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0 && (n_pin_values < size_t(n_values) || !p_pin_values);
	const double t0 = staging_wall_ms();
	if(!copy_stream)
		SLAMPP_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
	if(!copy_done)
		SLAMPP_HIP_CHECK(hipEventCreateWithFlags(&copy_done, hipEventDisableTiming));
	if(n_pin_values < size_t(n_values) || !p_pin_values)
		n_uploaded = 0;
	if((n_pin_values < size_t(n_values) && p_pin_values) || (n_pin_rhs < size_t(n_scalars) && p_pin_rhs)) {
		(void)hipStreamSynchronize(copy_stream); // a buffer is about to be replaced: no copy may still read it
		if(stream)
			(void)hipStreamSynchronize(stream);
	}
	Grow_Pinned(p_pin_values, n_pin_values, b_pin_values_registered, size_t(n_values));
	const double t1 = staging_wall_ms();
	Grow_Pinned(p_pin_rhs, n_pin_rhs, b_pin_rhs_registered, size_t(n_scalars));
	const double t2 = staging_wall_ms();
	if(!b_group_active) { // (with landmark shards the values go from the staging straight to the members' devices)
		d_A.Alloc(size_t(n_values));
		d_rhs.Alloc(size_t(n_scalars));
	}
	if(b_timing) {
		fprintf(stderr, "[staging] values %.2f ms (%s), rhs %.2f ms, device arrays %.2f ms\n", t1 - t0,
			b_pin_values_registered? "registered" : "hipHostMalloc", t2 - t1, staging_wall_ms() - t2);
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
static void Parallel_Copy(double *p_dst, const double *p_src, size_t n)
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
static void Upload_Rhs_And_Join(slampp_hip_solver &s, const double *p_rhs)
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
static void Upload_Values_And_Join(slampp_hip_solver &s, const double *p_values)
{
	s.Upload_Values(p_values);
	SLAMPP_HIP_CHECK(hipEventRecord(s.copy_done, s.copy_stream));
	SLAMPP_HIP_CHECK(hipStreamWaitEvent(s.stream, s.copy_done, 0));
}

void slampp_hip_solver::Free_Device()
{
	d_cols.Free(); d_blks.Free(); d_rents.Free(); d_pairs.Free(); d_task_ptr.Free(); d_task_pkg.Free(); d_pkg.Free();
	d_simt_chunks.Free(); d_simt_prog.Free(); d_simt_rest.Free(); d_simt_tab.Free();
	d_simt_bwd_chunks.Free(); d_simt_bwd_prog.Free(); d_simt_bwd_tab.Free();
	b_leaf_linv_valid = true;
	d_panel_pkg.Free(); d_panel_off.Free(); d_panel_out_off.Free(); d_handup.Free(); d_panel_rest.Free(); d_panel_upd_slots.Free(); d_panel_upd_ents.Free();
	simt_chunk_ptr.clear(); simt_rest_ptr.clear();
	d_dense_blks.Free(); d_dense_blk_loff.Free(); d_dense.Free(); d_dense_invdiag.Free(); d_dense_z.Free(); d_dense_x.Free();
	n_dense_blks = n_dense_cols = n_dense_dim = n_dense_pad = 0;
	dense_tiles.Free();
	b_dense_tiles = false;
	d_dense_gaps.Free(); d_dense_unit.Free(); d_dense_dst.Free();
	n_dense_gaps = 0;
	d_A.Free(); d_rhs.Free(); d_L.Free(); d_Linv.Free(); d_w.Free(); d_flag.Free();
	d_cov.Free(); d_damp_off.Free(); d_timing.Free();
	d_refine_map.Free(); d_refined.Free();
	b_damp_valid = false;
	dplan.p_timing = 0;
	n_uploaded = 0;
	if(p_sinv) {
		sparse_inverse_destroy(p_sinv);
		p_sinv = 0;
	}
	b_sinv_tried = false;
	d_Z.Free(); d_diag_zoff.Free(); d_diag_dim.Free(); d_diag_out_off.Free(); d_Zd.Free(); d_Zd_work.Free();
	if(p_schur) {
		schur_destroy(p_schur);
		p_schur = 0;
	}
	b_analyzed = false;
	b_factored = false;
}

size_t slampp_hip_solver::n_Device_Bytes() const
{
	return d_dense_blks.n_Bytes() + d_dense_unit.n_Bytes() + d_dense_dst.n_Bytes() + d_dense.n_Bytes() + d_dense_invdiag.n_Bytes() + dense_tiles.n_Bytes() +
		d_dense_z.n_Bytes() + d_dense_x.n_Bytes() + d_cols.n_Bytes() + d_blks.n_Bytes() + d_rents.n_Bytes() +
		d_task_ptr.n_Bytes() + d_task_pkg.n_Bytes() + d_pkg.n_Bytes() + d_pairs.n_Bytes() + d_A.n_Bytes() +
		d_simt_chunks.n_Bytes() + d_simt_prog.n_Bytes() + d_simt_rest.n_Bytes() + d_simt_tab.n_Bytes() +
		d_simt_bwd_chunks.n_Bytes() + d_simt_bwd_prog.n_Bytes() + d_simt_bwd_tab.n_Bytes() +
		d_panel_pkg.n_Bytes() + d_panel_off.n_Bytes() + d_panel_out_off.n_Bytes() + d_handup.n_Bytes() + d_panel_rest.n_Bytes() + d_panel_upd_slots.n_Bytes() + d_panel_upd_ents.n_Bytes() +
		d_rhs.n_Bytes() + d_L.n_Bytes() + d_Linv.n_Bytes() + d_w.n_Bytes() + d_cov.n_Bytes() + d_flag.n_Bytes() +
		d_Z.n_Bytes() + d_diag_zoff.n_Bytes() + d_Zd.n_Bytes() + d_Zd_work.n_Bytes() + sparse_inverse_bytes(p_sinv) +
		(p_schur? schur_device_bytes(p_schur) : 0);
}

void slampp_hip_solver::Phase_Begin(const char *p_s_label)
{
	if(!b_profile)
		return;
	if(b_profile == 3) { // only the phases of the kernels that move a step's bytes / flops: one event pair in a timed region
		static const char *p_kept[] = {"factor_leaves", "schur_tiles", "schur_gather", "dense_chol"};
		bool b_kept = false;
		for(size_t i = 0; i < sizeof(p_kept) / sizeof(p_kept[0]); ++ i)
			b_kept = b_kept || !strcmp(p_s_label, p_kept[i]);
		if(!b_kept)
			return;
	}
	int n_label = -1;
	for(size_t i = 0; i < phase_names.size(); ++ i) {
		if(phase_names[i] == p_s_label)
			n_label = int(i);
	}
	if(n_label < 0) {
		n_label = int(phase_names.size());
		phase_names.push_back(p_s_label);
		phase_ms.push_back(0);
		phase_count.push_back(0);
	}
	TPhaseRecord r;
	r.n_label = n_label;
	for(int i = 0; i < 2; ++ i) {
		hipEvent_t e;
		if(!event_pool.empty()) {
			e = event_pool.back();
			event_pool.pop_back();
		} else
			SLAMPP_HIP_CHECK(hipEventCreate(&e));
		(i? r.stop : r.start) = e;
	}
	SLAMPP_HIP_CHECK(hipEventRecord(r.start, stream));
	phase_pending.push_back(r);
	n_open_phase = int(phase_pending.size()) - 1;
}

void slampp_hip_solver::Phase_End()
{
	if(!b_profile || n_open_phase < 0)
		return;
	SLAMPP_HIP_CHECK(hipEventRecord(phase_pending[n_open_phase].stop, stream));
	n_open_phase = -1;
}

void slampp_hip_solver::Phase_Collect()
{
	for(size_t i = 0; i < phase_pending.size(); ++ i) {
		float f_ms = 0;
		if(hipEventElapsedTime(&f_ms, phase_pending[i].start, phase_pending[i].stop) == hipSuccess) {
			phase_ms[phase_pending[i].n_label] += f_ms;
			++ phase_count[phase_pending[i].n_label];
		} else
			(void)hipGetLastError();
		event_pool.push_back(phase_pending[i].start);
		event_pool.push_back(phase_pending[i].stop);
	}
	phase_pending.clear();
}

// Cuts block columns wider than 8 into pieces (as equal as possible, at most 8 wide) and builds the map from the
// refined packed values to the caller's: block (r, c) becomes the pieces (r_i, c_j), a diagonal block the pieces with
// i <= j (the upper triangle, as everywhere).  Nothing to do -- and nothing allocated -- for the usual 3 / 6 / 7.
void slampp_hip_solver::Refine_Structure()
{
	const int64_t n = int64_t(cumsum.size()) - 1;
	b_refined = false;
	for(int64_t c = 0; c < n && !b_refined; ++ c)
		b_refined = cumsum[c + 1] - cumsum[c] > 8;
	if(!b_refined) {
		refined_cumsum.clear(); refined_bcol_ptr.clear(); refined_brow.clear();
		d_refine_map.Free(); d_refined.Free();
		n_refined_values = 0;
		return;
	}
	std::vector<int64_t> first_piece(size_t(n) + 1, 0); // pieces of block column c: [first_piece[c], first_piece[c + 1])
	refined_cumsum.assign(1, 0);
	for(int64_t c = 0; c < n; ++ c) {
		const int64_t w = cumsum[c + 1] - cumsum[c], n_pieces = (w + 7) / 8;
		for(int64_t i = 0; i < n_pieces; ++ i)
			refined_cumsum.push_back(cumsum[c] + w * (i + 1) / n_pieces);
		first_piece[c + 1] = first_piece[c] + n_pieces;
	}
	const int64_t n_refined = first_piece[n];
	refined_bcol_ptr.assign(size_t(n_refined) + 1, 0);
	refined_brow.clear();
	std::vector<int64_t> map;
	int64_t n_src_off = 0; // offset of the caller's block (r, c) in the packed values
	std::vector<int64_t> col_src_off; // per block of column c: its offset
	for(int64_t c = 0; c < n; ++ c) {
		const int64_t w = cumsum[c + 1] - cumsum[c];
		col_src_off.clear();
		for(int64_t k = bcol_ptr[c]; k < bcol_ptr[c + 1]; ++ k) {
			col_src_off.push_back(n_src_off);
			n_src_off += (cumsum[brow[k] + 1] - cumsum[brow[k]]) * w;
		}
		for(int64_t pj = first_piece[c]; pj < first_piece[c + 1]; ++ pj) { // refined column pj: rows ascend with the caller's blocks
			const int64_t n_col0 = refined_cumsum[pj] - cumsum[c], n_pw = refined_cumsum[pj + 1] - refined_cumsum[pj];
			for(int64_t k = bcol_ptr[c]; k < bcol_ptr[c + 1]; ++ k) {
				const int64_t r = brow[k], h = cumsum[r + 1] - cumsum[r];
				for(int64_t pi = first_piece[r]; pi < first_piece[r + 1]; ++ pi) {
					if(pi > pj)
						break; // below the diagonal of a diagonal block
					const int64_t n_row0 = refined_cumsum[pi] - cumsum[r], n_ph = refined_cumsum[pi + 1] - refined_cumsum[pi];
					refined_brow.push_back(int32_t(pi));
					for(int64_t b = 0; b < n_pw; ++ b) {
						for(int64_t a = 0; a < n_ph; ++ a)
							map.push_back(col_src_off[size_t(k - bcol_ptr[c])] + (n_row0 + a) + (n_col0 + b) * h);
					}
				}
			}
			refined_bcol_ptr[pj + 1] = int64_t(refined_brow.size());
		}
	}
	n_refined_values = int64_t(map.size());
	d_refine_map.Upload(map, stream);
	d_refined.Alloc(map.size());
	SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // map lives on this stack frame
}

void slampp_hip_solver::Analyze_Sparse()
{
	if(p_sinv) { // lists of the previous plan
		sparse_inverse_destroy(p_sinv);
		p_sinv = 0;
	}
	b_sinv_tried = false;
	const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
	double t_phase = wall_ms();
#define SETUP_PHASE(name) do { if(b_timing) { const double t_ = wall_ms(); \
	fprintf(stderr, "[setup] %-12s %8.2f ms\n", name, t_ - t_phase); t_phase = t_; } } while(0)
	Refine_Structure();
	{ // a tall task must fit the panel kernel: its columns and the blocks of its LDS image
		const std::vector<int64_t> &r_cs = b_refined? refined_cumsum : cumsum;
		const int n_dim0 = int(r_cs[1] - r_cs[0]);
		if(const char *p_s_wide = getenv("SLAMPP_HIP_WIDE_MIN")) // development aid: overrides the option "wide_min_tasks"
			n_wide_min_tasks = std::max(atoi(p_s_wide), 1);
		opt.task_wide_min = n_wide_min_tasks;
		opt.task_max_cols = int(PANEL_COLS);
		opt.task_max_blocks = panel_slot_cap(n_dim0);
		// the top of the tree as one task (option "panel_top"): what one workgroup's LDS holds next to the package and the staging
		const bool b_top = n_panel_top != 0 && n_panel != 0 && (n_dim0 == 3 || n_dim0 == 6 || n_dim0 == 7);
		opt.task_top_cols = b_top? int(PANEL_TOP_COLS) : 0;
		opt.task_top_blocks = b_top? panel_top_slot_cap(n_dim0) : 0;
	}
	std::string s_err = b_refined? build_plan(int64_t(refined_cumsum.size()) - 1, refined_cumsum.data(), refined_bcol_ptr.data(),
		refined_brow.data(), opt, plan) : build_plan(int64_t(cumsum.size()) - 1, cumsum.data(), bcol_ptr.data(), brow.data(), opt, plan);
	SETUP_PHASE("build_plan");
	if(!s_err.empty())
		throw std::invalid_argument(s_err);
	if(plan.max_dim > 8)
		throw std::logic_error("a block column wider than 8 survived the refinement");
	const Plan &P = plan;
	const int64_t n_lblocks = int64_t(P.lrow.size());
	// the bottom stage and the wide stages right above it (more tasks than the 8-wave kernel keeps
	// resident at 2 workgroups per CU) run one wave per task: there throughput beats single-column latency
	n_bottom_stages = 1;
	while(n_bottom_stages < int(P.stage_ptr.size()) - 1 &&
	   P.stage_ptr[n_bottom_stages + 1] - P.stage_ptr[n_bottom_stages] > n_wide_min_tasks)
		++ n_bottom_stages; // (tall tasks, Plan::col_sub, begin above these: the same threshold)
	// the shape grouping of the leaf kernel (13 ms of host work at 100 000 poses, plan in, tables out) runs beside the
	// records, packages and uploads below
	std::exception_ptr p_simt_error;
	struct TJoin { std::thread t; ~TJoin() { if(t.joinable()) t.join(); } } t_simt_thread;
	const double t_simt = wall_ms();
	t_simt_thread.t = std::thread([this, &p_simt_error]() {
		try {
			Build_Simt();
		} catch(...) {
			p_simt_error = std::current_exception();
		}
	});

	if(P.cs_new[P.n] >= INT32_MAX)
		throw std::domain_error("systems with 2^31 or more scalar unknowns are not supported by the sparse path");

	// packed device records (see sparse_kernels.h)
	const int32_t n_sched = int32_t(P.task_cols.size()); // all columns but those of the dense top
	std::vector<TColDesc> cols(n_sched); // in schedule order
	for(int32_t i = 0; i < n_sched; ++ i) {
		const int32_t j = P.task_cols[i];
		TColDesc &c = cols[i];
		memset(&c, 0, sizeof(c));
		c.k0 = P.lptr[j];
		c.nb = int32_t(P.lptr[j + 1] - P.lptr[j]);
		c.dj = P.dim[j];
		c.linv_off = P.linv_off[j];
		c.cs_new = P.cs_new[j];
		c.cs_src = P.cs_src[j];
		c.r0 = P.rptr[j];
		c.nr = int32_t(P.rptr[j + 1] - P.rptr[j]);
		c.p0 = P.pptr[P.lptr[j] + 1]; // pairs are stored block by block: those of the sub-diagonal blocks are contiguous
		const int64_t np = P.pptr[P.lptr[j + 1]] - c.p0;
		c.np = int32_t(std::min<int64_t>(np, INT32_MAX));
	}
	std::vector<TBlkDesc> blks(n_lblocks);
	for(int64_t k = 0; k < n_lblocks; ++ k) {
		TBlkDesc &b = blks[k];
		const int64_t np = P.pptr[k + 1] - P.pptr[k];
		if(np >= (int64_t(1) << 24))
			throw std::domain_error("a factor block has 2^24 or more updates: use the dense path");
		b.loff = P.loff[k];
		b.asrc = (P.asrc[k] < 0)? -1 : P.asrc[k] * 2 + P.atrans[k];
		if(k == P.lptr[P.blk_col[k]] && b.asrc >= 0)
			b.asrc |= 1; // diagonal blocks are read transposed: the lower triangle of the factor block then comes from the upper triangle of Lambda's block, the one the reference's solvers consume
		b.p0 = P.pptr[k];
		b.np_di = uint32_t(np) | (uint32_t(P.dim[P.lrow[k]]) << 24);
		b.xcs = int32_t(P.cs_new[P.lrow[k]]);
	}
	if(P.loff[n_lblocks] >= (int64_t(1) << 48))
		throw std::domain_error("the factor has 2^48 or more values");
	std::vector<longlong2> pairs(P.pa.size());
	for(int64_t k = 0; k < n_lblocks; ++ k) { // pairs are stored block by block
		const int64_t n_pos = std::min<int64_t>(k - P.lptr[P.blk_col[k]], 255); // position of the target block in its column
		for(int64_t e = P.pptr[k]; e < P.pptr[k + 1]; ++ e) {
			const int64_t dc = P.dim[P.blk_col[P.pa[e]]];
			pairs[e].x = P.loff[P.pa[e]] | (n_pos << 48) | (dc << 56);
			pairs[e].y = P.loff[P.pb[e]];
		}
	}
	std::vector<TRowEnt> rents(P.rblk.size());
	for(size_t e = 0; e < P.rblk.size(); ++ e) {
		const int32_t c = P.blk_col[P.rblk[e]];
		rents[e].off = P.loff[P.rblk[e]];
		rents[e].ycs = int32_t(P.cs_new[c]);
		rents[e].dc = P.dim[c];
	}
	// column packages for the upper stages (see sparse_kernels.h); the limits are those of factor_stage_kernel's staged path
	std::vector<longlong2> pkg;
	std::vector<int64_t> task_pkg(P.task_ptr.size() - 1, -1);
	if(P.uniform_dim && (P.max_dim == 3 || P.max_dim == 6 || P.max_dim == 7)) {
		const int n_stages = int(P.stage_ptr.size()) - 1;
		// (the wide stages above the leaves and the stages near the root run the same kernel with different capacities)
		const int n_first_stage = (n_stages > 1)? 1 : n_stages;
		for(int t = (n_first_stage < n_stages)? P.stage_ptr[n_first_stage] : int(task_pkg.size()); t < int(task_pkg.size()); ++ t) {
			const bool b_wide = t < P.stage_ptr[std::min(n_bottom_stages, n_stages)];
			const int PKG_CHUNK = b_wide? int(WIDE_CHUNK) : int(UP_CHUNK), PKG_NR = b_wide? int(WIDE_NR) : int(UP_NR),
				PKG_NP = b_wide? int(WIDE_NP) : int(UP_NP);
			task_pkg[t] = int64_t(pkg.size());
			for(int64_t i = P.task_ptr[t]; i < P.task_ptr[t + 1]; ++ i) {
				const TColDesc &c = cols[i];
				const size_t n_at = pkg.size();
				const bool b_fits = c.nb <= PKG_CHUNK && c.nr <= PKG_NR && c.np <= PKG_NP;
				const int ne = b_fits? c.nr + c.np : 0;
				pkg.resize(n_at + (b_fits? package_units(c.nb, ne) : 4), longlong2{0, 0});
				memcpy(&pkg[n_at], &c, sizeof(TColDesc));
				if(!b_fits)
					continue;
				memcpy(&pkg[n_at + 4], &blks[c.k0], size_t(c.nb) * sizeof(TBlkDesc));
				longlong2 *p_ent = &pkg[n_at + 4 + 2 * c.nb];
				int32_t *p_ycs = reinterpret_cast<int32_t*>(p_ent + ne);
				unsigned char *p_tag = reinterpret_cast<unsigned char*>(p_ent + ne + (ne + 3) / 4);
				for(int e = 0; e < c.nr; ++ e) { // row entries of the diagonal block: both operands are the block L(j,c)
					p_ent[e] = longlong2{rents[c.r0 + e].off, rents[c.r0 + e].off};
					p_ycs[e] = rents[c.r0 + e].ycs;
					p_tag[e] = 0;
				}
				for(int e = 0; e < c.np; ++ e) {
					const longlong2 pr = pairs[c.p0 + e];
					p_ent[c.nr + e] = longlong2{pr.x & ((int64_t(1) << 48) - 1), pr.y};
					p_tag[c.nr + e] = (unsigned char)((pr.x >> 48) & 0xff);
				}
			}
		}
		pkg.resize(pkg.size() + PKG_SPECULATIVE, longlong2{0, 0});
	}
	SETUP_PHASE("records");
	// dense top
	n_dense_dim = P.dense_dim;
	n_dense_pad = n_dense_dim? dense_padded_dim(n_dense_dim) : 0;
	std::vector<TDenseBlk> dense_blks;
	std::vector<TDenseCol> dense_cols;
	std::vector<int64_t> dense_blk_loff;
	if(n_dense_dim) {
		for(int32_t j = 0; j < P.n; ++ j) {
			if(P.dense_pos[j] < 0)
				continue;
			TDenseCol dc;
			dc.cs_new = P.cs_new[j]; dc.cs_src = P.cs_src[j]; dc.pos = P.dense_pos[j]; dc.dj = P.dim[j];
			dense_cols.push_back(dc);
			for(int64_t k = P.lptr[j]; k < P.lptr[j + 1]; ++ k) {
				const int32_t i = P.lrow[k];
				if(P.dense_pos[i] < 0)
					throw std::logic_error("dense top is not closed upwards");
				TDenseBlk b;
				memset(&b, 0, sizeof(b));
				b.asrc = (P.asrc[k] < 0)? -1 : P.asrc[k] * 2 + P.atrans[k];
				b.p0 = P.pptr[k];
				b.np = int32_t(P.pptr[k + 1] - P.pptr[k]);
				b.dst = int64_t(P.dense_pos[i]) + int64_t(P.dense_pos[j]) * n_dense_pad;
				b.di = P.dim[i]; b.dj = P.dim[j];
				if(k == P.lptr[j]) {
					b.r0 = P.rptr[j];
					b.nr = int32_t(P.rptr[j + 1] - P.rptr[j]);
					b.cs_src = P.cs_src[j];
					b.pos = P.dense_pos[j];
				} else
					b.nr = -1;
				dense_blks.push_back(b);
				dense_blk_loff.push_back(P.loff[k]);
			}
		}
		d_dense_blks.Upload(dense_blks, stream);
		d_dense_blk_loff.Upload(dense_blk_loff, stream);
		{
			std::vector<char> covered(n_dense_dim, 0);
			for(size_t k = 0; k < dense_cols.size(); ++ k)
				std::fill(covered.begin() + dense_cols[k].pos, covered.begin() + dense_cols[k].pos + dense_cols[k].dj, char(1));
			std::vector<int32_t> gaps;
			for(int32_t q = 0; q < n_dense_dim; ++ q) {
				if(!covered[q])
					gaps.push_back(q);
			}
			n_dense_gaps = int(gaps.size());
			d_dense_gaps.Upload(gaps, stream);
			// the same as a byte per position (with the padding behind the last column: tile_zero writes the identity there
			// while it zeroes the diagonal tiles), and where every entry of the dense system's x goes in the solver's vectors
			// (the last launch of the substitution stores there: no scatter launch)
			std::vector<uint8_t> unit(n_dense_pad, uint8_t(1));
			std::vector<longlong2> dst(n_dense_pad, longlong2{-1, -1});
			for(size_t k = 0; k < dense_cols.size(); ++ k) {
				for(int q = 0; q < dense_cols[k].dj; ++ q) {
					unit[dense_cols[k].pos + q] = 0;
					dst[dense_cols[k].pos + q] = longlong2{(long long)(dense_cols[k].cs_new + q), (long long)(dense_cols[k].cs_src + q)};
				}
			}
			d_dense_unit.Upload(unit, stream);
			d_dense_dst.Upload(dst, stream);
			SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // the vectors live in this scope
		}
		d_dense.Alloc(size_t(n_dense_pad) * n_dense_pad);
		b_dense_clean = false;
		d_dense_invdiag.Alloc(size_t(n_dense_pad / dense_NB) * dense_NB * dense_NB);
		d_dense_z.Alloc(n_dense_pad);
		d_dense_x.Alloc(n_dense_pad);
		// which 64 x 64 tiles of the dense top are structurally nonzero, and how long the dependent chain is if only
		// those are touched and independent tile columns are factored side by side
		b_dense_tiles = false;
		if(n_dense_top_tiles != 0) {
			std::vector<char> nonzero;
			const int T = dense_top_tile_pattern(P, nonzero);
			if(T != n_dense_pad / dense_NB)
				throw std::logic_error("dense top: tile count mismatch");
			if(dense_tiles.Build(T, nonzero, stream)) // two launches (26 us) per tile against three (34 us) per level, and fewer tiles touched
				b_dense_tiles = n_dense_top_tiles > 0 || 100 * dense_tiles.n_levels <= 85 * T;
			if(b_timing) {
				size_t n_nz = 0;
				for(size_t k = 0; k < nonzero.size(); ++ k)
					n_nz += nonzero[k];
				fprintf(stderr, "[setup] dense top: %d tiles per side, %zu of %d lower tiles nonzero before fill, %d levels, "
					"%d trsm tiles, %d update targets -> %s schedule\n", T, n_nz, T * (T + 1) / 2, dense_tiles.n_levels,
					dense_tiles.level_trsm_ptr.empty()? 0 : dense_tiles.level_trsm_ptr.back(),
					dense_tiles.level_tgt_ptr.empty()? 0 : dense_tiles.level_tgt_ptr.back(), b_dense_tiles? "tile" : "dense");
			}
		}
	}
	n_dense_blks = int(dense_blks.size());
	n_dense_cols = int(dense_cols.size());
	// panel packages for the separator stages (panel_kernel.hip): a task qualifies if its columns' blocks are one range of
	// the factor and everything fits the kernel's LDS; the updates it receives from earlier stages go to the lists of
	// panel_update_kernel, block by block
	std::vector<longlong2> panel_pkg;
	std::vector<int64_t> panel_off, panel_out_off; // (panel_out_off: per package the offset of its hand-up list, or -1)
	int64_t n_handup_doubles = 0;
	std::vector<int32_t> panel_rest;
	std::vector<TUpdSlot> upd_slots;
	std::vector<TUpdEnt> upd_ents;
	// (a second pass, without hand-ups, if a stage's hand-up list would take its workgroups past the LDS of a CU: the list
	// rides in the dynamic LDS request on top of the task's image, and nothing else bounds its length -- advisor, round 4)
	for(bool b_hand_up_allowed = n_panel_handup != 0;;) {
	panel_pkg.clear();
	panel_off.clear();
	panel_out_off.clear();
	n_handup_doubles = 0;
	panel_rest.clear();
	upd_slots.clear();
	upd_ents.clear();
	panel_ptr.clear();
	panel_rest_ptr.clear();
	panel_upd_ptr.clear();
	b_any_hand_up = false;
	if(n_panel && P.uniform_dim && (P.max_dim == 3 || P.max_dim == 6 || P.max_dim == 7)) {
		const int n_stages = int(P.stage_ptr.size()) - 1, D = P.max_dim;
		const int n_slot_cap = panel_slot_cap(D);
		// the leaf subtrees too, where they are so few that one round of workgroups takes them all: a small system's leaf
		// stage is all latency, and eight waves on a subtree of four columns beat one (37 -> 19 us on the reduced camera
		// system of C4; with 1 600 leaf tasks -- 10 000 poses -- the wave-per-task kernel wins again, 0.33 against 0.38 ms)
		const bool b_leaf_panels = n_stages > 0 && n_simt <= 0 && P.stage_ptr[1] - P.stage_ptr[0] <= 512; // (one round of workgroups)
		panel_ptr.assign(n_stages + 1, 0);
		panel_rest_ptr.assign(n_stages + 1, 0);
		panel_upd_ptr.assign(n_stages + 1, 0);
		std::vector<int32_t> col_local(size_t(P.n), -1), col_stage(size_t(P.n), -1);
		std::vector<int32_t> slot_of(size_t(n_lblocks), -1); // factor block -> slot of the task being packed (else -1)
		// round 4, hand-ups (TPanelOut): the slot every factor block has in the image of its own task, once that task's package
		// exists (-1: the task went to the column kernel), the package of every column's task, and per package what it hands up
		std::vector<int32_t> img_slot(size_t(n_lblocks), -1), col_package(size_t(P.n), -1), col_level(size_t(P.n), 0); // (col_level: which of its task's levels a column is in)
		struct THandUp { std::vector<TPanelOut> recs; std::vector<uint32_t> pairs; };
		std::vector<THandUp> hand_up; // indexed by package
		std::map<std::pair<int32_t, int64_t>, int32_t> out_of; // (source package, target factor block) -> record of that package
		const bool b_hand_up = b_hand_up_allowed;
		const int n_handup_max_tasks = dev_knob("SLAMPP_HIP_DEV_HANDUP_MAX_TASKS", 1 << 30); // (measured at C3: handing up from the 2 420-task stage as well 224 -> 208 us for the separator launches, from the narrow stages only 224 -> 214)
		std::vector<int64_t> order; // the task's columns (indices into cols) level by level
		for(int s = 0; s < n_stages; ++ s) {
			for(int64_t i = P.task_ptr[P.stage_ptr[s]]; i < P.task_ptr[P.stage_ptr[s + 1]]; ++ i)
				col_stage[P.task_cols[i]] = s;
		}
		std::vector<TPanelExt> fresh;
		std::vector<uint32_t> irow, ipair;
		std::vector<TPanelCol> pcols;
		std::vector<TPanelSlot> pslots;
		panel_ride.assign(n_stages + 1, 0);
		panel_cfg.assign(size_t(n_stages) + 1, TPanelLaunch{int32_t(PANEL_W), int32_t(64 * PANEL_W), 1, 1, 1, 0});
		const int n_ride_max_fresh = dev_knob("SLAMPP_HIP_DEV_PANEL_RIDE_FRESH", 96);
		for(int s = 0; s < n_stages; ++ s) {
			const bool b_panel_stage = s >= n_bottom_stages || (s == 0 && b_leaf_panels);
			// Do this stage's updates from further down ride in the launch of the stage below?  Only if that is a panel launch,
			// and only if what is then left to the tasks themselves -- the updates from the stage right below -- is little:
			// a task brings those in with its own eight waves, on the stage's critical path (a launch saved is about 4 us)
			// Waves per task: eight where the stage is a launch on the critical path, four where it holds more tasks than the
			// chip takes at once (more workgroups per CU: throughput), two where it holds them several times over.
			// (round 4: two where it holds them several times over -- C3's 2 151-task launch 91 -> 78 us, the step 0.330 -> 0.318 ms;
			// a million poses 2.185 -> 2.146; one wave per task is slower again, 169 against 147 us for C3's slice launches, and two
			// waves for the 303-task launch as well 153: the development knobs below moved the lines)
			const int n_w4_min_tasks = dev_knob("SLAMPP_HIP_DEV_PANEL_W4_MIN", 512);
			const int n_w2_min_tasks = dev_knob("SLAMPP_HIP_DEV_PANEL_W2_MIN", 1024);
			const int n_stage_waves = (b_panel_stage && P.stage_ptr[s + 1] - P.stage_ptr[s] > n_w2_min_tasks)? 2 :
				(b_panel_stage && P.stage_ptr[s + 1] - P.stage_ptr[s] > n_w4_min_tasks)? 4 : int(PANEL_W);
			// hand-ups from the stage below (development knob SLAMPP_HIP_DEV_HANDUP_MAX_TASKS: only from stages of at most that many tasks --
			// a stage that fills the chip several times over is bound by throughput, and what its tasks compute for the stage
			// above they compute instead of the next task's columns: C3's 2 420-task launch 70 -> 92 us; the stage above gains more)
			const bool b_hand_up_stage = b_hand_up && s > 0 && P.stage_ptr[s] - P.stage_ptr[s - 1] <= n_handup_max_tasks;
			panel_cfg[s].n_waves = n_stage_waves;
			panel_cfg[s].n_cap_units = 64 * n_stage_waves; // (one speculative unit per thread)
			// The first stage above a leaf stage that is not a panel launch: everything its tasks receive comes from that one
			// stage, nothing from further down -- the tasks bring it in themselves and no update launch is needed (if it fits
			// the packages: the tall tasks of a wide stage receive some fifty products each)
			// ... Or do the tasks bring in everything themselves (mode 2: they read Lambda and all their updates, no update role
			// has prepared their blocks)?  Where the launch below is no panel launch (the first stage above lane-per-task
			// leaves: everything comes from that one stage), and where it is so crowded -- more workgroups than the chip holds at
			// once -- that riders only make it longer (C3: 5 816 riders in the 2 420-task stage cost it 20 us; the 625 tasks
			// above them take their ~150 products each in 6) -- if it fits the packages.
			const bool b_first_above_leaves = b_panel_stage && s == 1 && panel_ptr[1] == panel_ptr[0];
			// (measured at C3 and not kept as the default: without its 5 816 riders the 2 420-task launch takes the same 67 us --
			// its own tasks fill the chip for that long --, and the stage above, bringing in ~150 products a task, 32 instead of 23)
			const bool b_below_crowded = dev_knob_set("SLAMPP_HIP_DEV_PANEL_SELF_ABOVE_CROWDED") && b_panel_stage && s > 0 && panel_ptr[s] - panel_ptr[s - 1] > 1024;
			if(b_panel_stage && s > 0 && (panel_ptr[s] > panel_ptr[s - 1] || b_first_above_leaves)) {
				int64_t n_max_fresh = 0, n_max_external = 0;
				for(int t = P.stage_ptr[s]; t < P.stage_ptr[s + 1]; ++ t) {
					int64_t n_fresh = 0, n_external = 0;
					for(int64_t i = P.task_ptr[t]; i < P.task_ptr[t + 1]; ++ i) {
						const TColDesc &c = cols[i];
						for(int64_t e = c.r0; e < c.r0 + c.nr; ++ e) {
							const bool b_up = b_hand_up_stage && img_slot[P.rblk[e]] >= 0 && col_stage[P.blk_col[P.rblk[e]]] == s - 1;
							n_fresh += !b_up && col_stage[P.blk_col[P.rblk[e]]] == s - 1;
							n_external += !b_up && col_stage[P.blk_col[P.rblk[e]]] < s;
						}
						for(int64_t e = P.pptr[c.k0 + 1]; e < P.pptr[c.k0 + c.nb]; ++ e) {
							const bool b_up = b_hand_up_stage && img_slot[P.pa[e]] >= 0 && col_stage[P.blk_col[P.pa[e]]] == s - 1;
							n_fresh += !b_up && col_stage[P.blk_col[P.pa[e]]] == s - 1;
							n_external += !b_up && col_stage[P.blk_col[P.pa[e]]] < s;
						}
					}
					n_max_fresh = std::max(n_max_fresh, n_fresh);
					n_max_external = std::max(n_max_external, n_external);
				}
				if((b_first_above_leaves || b_below_crowded) && n_max_external <= 320)
					panel_ride[s] = 2;
				else if(panel_ptr[s] > panel_ptr[s - 1])
					panel_ride[s] = n_max_fresh <= n_ride_max_fresh;
				panel_cfg[s].b_from_lambda = panel_ride[s] == 2;
				if(b_timing)
					fprintf(stderr, "[setup] stage %d: %d tasks, at most %lld updates from the stage below, %lld in all: %s\n", s,
						P.stage_ptr[s + 1] - P.stage_ptr[s], (long long)n_max_fresh, (long long)n_max_external,
						(panel_ride[s] == 2)? "the tasks bring them in" : panel_ride[s]? "ride" : "own launch");
			}
			int64_t n_stage_max_slots = 0, n_stage_max_units = 0, n_stage_rest = 0; // (for the development print below)
			for(int t = P.stage_ptr[s]; b_panel_stage && t < P.stage_ptr[s + 1]; ++ t) {
				const int64_t c_begin = P.task_ptr[t], c_end = P.task_ptr[t + 1];
				const int n_cols = int(c_end - c_begin);
				const bool b_top_task = n_cols > int(PANEL_COLS); // (only the merged top of the tree outgrows a slice: Plan, task_top_cols)
				bool b_fits = n_cols >= 1 && n_cols <= (opt.task_top_cols? std::max(int(PANEL_COLS), opt.task_top_cols) : int(PANEL_COLS));
				// the package lists the task's columns level by level (a tall task: Plan::col_sub; a chain: one column per
				// level, in order), the slots of the LDS image are their blocks in that order
				order.clear();
				for(int64_t i = c_begin; i < c_end; ++ i)
					order.push_back(i);
				bool b_tall = false;
				for(int64_t i = c_begin; i < c_end; ++ i)
					b_tall = b_tall || P.col_sub[P.task_cols[i]] != 0;
				if(b_tall) {
					std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
						return P.col_sub[P.task_cols[a]] < P.col_sub[P.task_cols[b]]; });
				}
				int64_t n_slots = 0, n_int_rows = 0, n_int_pairs = 0;
				for(size_t o = 0; b_fits && o < order.size(); ++ o)
					n_slots += cols[order[o]].nb;
				b_fits = b_fits && n_slots <= (b_top_task? std::max(n_slot_cap, opt.task_top_blocks) : n_slot_cap);
				if(b_fits) {
					int32_t n_slot = 0;
					for(size_t o = 0; o < order.size(); ++ o) {
						const TColDesc &c = cols[order[o]];
						for(int64_t k = c.k0; k < c.k0 + c.nb; ++ k)
							slot_of[k] = n_slot ++;
					}
				}
				auto Release_Slots = [&]() {
					for(size_t o = 0; o < order.size(); ++ o) {
						const TColDesc &c = cols[order[o]];
						for(int64_t k = c.k0; k < c.k0 + c.nb; ++ k)
							slot_of[k] = -1;
					}
				};
				// the updates from stages further down are applied inside the launch of the stage below, if that is a panel
				// launch: then what the stage right below contributes ("fresh") is left to the task itself
				const bool b_ride = panel_ride[s] != 0, b_self = panel_ride[s] == 2;
				int64_t n_fresh = 0;
				// an update whose operands a task of the stage right below keeps in its image is handed up by that task (one
				// ready-made block per source task and target block) instead of fetched and multiplied here
				auto Handed_Up = [&](int64_t n_operand_blk) {
					return b_hand_up_stage && !b_self && img_slot[n_operand_blk] >= 0 && col_stage[P.blk_col[n_operand_blk]] == s - 1;
				};
				std::vector<std::pair<int32_t, int64_t> > up_keys; // (source package, target block) of this task's hand-ups, in order of first use
				auto Count_Up = [&](int64_t n_operand_blk, int64_t n_target_blk) {
					const std::pair<int32_t, int64_t> key(col_package[P.blk_col[n_operand_blk]], n_target_blk);
					if(std::find(up_keys.begin(), up_keys.end(), key) == up_keys.end())
						up_keys.push_back(key);
				};
				for(int64_t i = c_begin; b_fits && i < c_end; ++ i) { // size of the package
					const TColDesc &c = cols[i];
					for(int64_t e = c.r0; e < c.r0 + c.nr; ++ e) {
						const bool b_int = slot_of[P.rblk[e]] >= 0;
						n_int_rows += b_int;
						if(!b_int && Handed_Up(P.rblk[e]))
							Count_Up(P.rblk[e], c.k0);
						else
							n_fresh += !b_int && b_ride && (b_self || col_stage[P.blk_col[P.rblk[e]]] == s - 1);
					}
					for(int64_t k = c.k0 + 1; k < c.k0 + c.nb; ++ k) {
						for(int64_t e = P.pptr[k]; e < P.pptr[k + 1]; ++ e) {
							const bool b_int = slot_of[P.pa[e]] >= 0;
							n_int_pairs += b_int;
							if(!b_int && Handed_Up(P.pa[e]))
								Count_Up(P.pa[e], k);
							else
								n_fresh += !b_int && b_ride && (b_self || col_stage[P.blk_col[P.pa[e]]] == s - 1);
						}
					}
				}
				n_fresh += int64_t(up_keys.size());
				const size_t n_units = 4 + 3 * size_t(n_cols) + 2 * size_t(n_slots) + size_t(n_int_rows + 3) / 4 + size_t(n_int_pairs + 3) / 4 +
					2 * size_t(n_fresh);
				b_fits = b_fits && n_units <= size_t(b_top_task? PANEL_TOP_UNITS : PANEL_UNITS);
				if(b_fits && b_top_task) { // (the top task's LDS is sized by the task itself: does it fit a CU's, with room for its hand-up list?)
					TPanelLaunch t_cfg = {int32_t(PANEL_W), int32_t(std::max<size_t>(n_units, 64 * PANEL_W)), int32_t(n_slots), n_cols, n_cols, 0, 0};
					b_fits = size_t(panel_lds(D, true, t_cfg).TOTAL) * sizeof(double) <= 150 * 1024;
				}
				n_stage_max_slots = std::max(n_stage_max_slots, n_slots);
				n_stage_max_units = std::max(n_stage_max_units, int64_t(n_units));
				n_stage_rest += !b_fits;
				if(!b_fits) {
					if(n_slots <= (b_top_task? std::max(n_slot_cap, opt.task_top_blocks) : n_slot_cap) && n_cols >= 1 &&
					   n_cols <= (opt.task_top_cols? std::max(int(PANEL_COLS), opt.task_top_cols) : int(PANEL_COLS)))
						Release_Slots();
					panel_rest.push_back(t);
					continue;
				}
				irow.clear(); ipair.clear(); pcols.clear(); pslots.clear(); fresh.clear();
				for(size_t o = 0; o < order.size(); ++ o)
					col_local[P.task_cols[order[o]]] = int32_t(o);
				// one more operand pair for the block the source task hands up for target block n_target (a new record there, and
				// the entry here that subtracts it, when it is the first)
				auto Hand_Up = [&](int64_t ka, int64_t kb, int64_t n_target, int32_t n_col_here, int32_t n_slot_here, bool b_diag) {
					const int32_t n_src = col_package[P.blk_col[ka]];
					const std::pair<int32_t, int64_t> key(n_src, n_target);
					std::map<std::pair<int32_t, int64_t>, int32_t>::iterator it = out_of.find(key);
					THandUp &r_up = hand_up[size_t(n_src)];
					if(it == out_of.end()) {
						TPanelOut rec;
						rec.op0 = -1; // (the pairs of a record are collected apart and laid out when the list is written)
						rec.onp = 0;
						rec.dst = n_handup_doubles | (int64_t(b_diag) << 62);
						it = out_of.insert(std::make_pair(key, int32_t(r_up.recs.size()))).first;
						r_up.recs.push_back(rec);
						TPanelExt en;
						memset(&en, 0, sizeof(en));
						en.a_off = n_handup_doubles;
						en.slot = uint16_t(n_slot_here);
						en.kind = b_diag? 3 : 2;
						en.col = n_col_here;
						fresh.push_back(en);
						n_handup_doubles += P.max_dim * P.max_dim + 8;
					}
					// (until the list is written, onp holds the last of the source task's levels the record's operands come from)
					r_up.recs[size_t(it->second)].onp = std::max(r_up.recs[size_t(it->second)].onp, col_level[P.blk_col[ka]]);
					r_up.pairs.push_back(uint32_t(it->second));
					r_up.pairs.push_back(b_diag? (uint32_t(img_slot[ka]) | (uint32_t(col_local[P.blk_col[ka]]) << 16)) :
						(uint32_t(img_slot[ka]) | (uint32_t(img_slot[kb]) << 16)));
				};
				for(size_t o = 0; o < order.size(); ++ o) {
					const int64_t i = order[o];
					const TColDesc &c = cols[i];
					TPanelCol pc;
					memset(&pc, 0, sizeof(pc));
					pc.linv_off = c.linv_off;
					pc.cs_new = c.cs_new;
					pc.cs_src = c.cs_src;
					pc.slot0 = slot_of[c.k0];
					pc.nb = c.nb;
					pc.sub = b_tall? P.col_sub[P.task_cols[i]] : int32_t(o); // (a chain: every column a level of its own)
					pc.ir0 = int32_t(irow.size());
					TUpdSlot us;
					memset(&us, 0, sizeof(us));
					us.loff = blks[c.k0].loff;
					us.asrc = blks[c.k0].asrc;
					us.e0 = int64_t(upd_ents.size());
					us.kind = 1;
					us.cs_src = c.cs_src;
					us.cs_new = c.cs_new;
					for(int64_t e = c.r0; e < c.r0 + c.nr; ++ e) { // row entries of the diagonal block: blocks L(j,c)
						const int64_t k = P.rblk[e];
						if(slot_of[k] >= 0)
							irow.push_back(uint32_t(slot_of[k]) | (uint32_t(col_local[P.blk_col[k]]) << 16));
						else if(Handed_Up(k))
							Hand_Up(k, k, c.k0, int32_t(o), pc.slot0, true);
						else if(b_ride && (b_self || col_stage[P.blk_col[k]] == s - 1)) {
							TPanelExt en;
							memset(&en, 0, sizeof(en));
							en.a_off = en.b_off = rents[e].off;
							en.ycs = rents[e].ycs;
							en.slot = uint16_t(pc.slot0);
							en.kind = 1;
							en.col = int32_t(o);
							fresh.push_back(en);
						} else
							upd_ents.push_back(TUpdEnt{rents[e].off, int64_t(rents[e].ycs)});
					}
					us.ne = int32_t(int64_t(upd_ents.size()) - us.e0);
					upd_slots.push_back(us);
					pc.inr = int32_t(irow.size()) - pc.ir0;
					pcols.push_back(pc);
					for(int64_t k = c.k0; k < c.k0 + c.nb; ++ k) {
						TPanelSlot ps;
						memset(&ps, 0, sizeof(ps));
						ps.loff = blks[k].loff;
						ps.asrc = blks[k].asrc;
						ps.ip0 = int32_t(ipair.size());
						if(k > c.k0) { // (the diagonal block's updates are its row entries)
							memset(&us, 0, sizeof(us));
							us.loff = blks[k].loff;
							us.asrc = blks[k].asrc;
							us.e0 = int64_t(upd_ents.size());
							for(int64_t e = P.pptr[k]; e < P.pptr[k + 1]; ++ e) {
								const int64_t ka = P.pa[e], kb = P.pb[e];
								if(slot_of[ka] >= 0)
									ipair.push_back(uint32_t(slot_of[ka]) | (uint32_t(slot_of[kb]) << 16));
								else if(Handed_Up(ka))
									Hand_Up(ka, kb, k, 0, slot_of[k], false);
								else if(b_ride && (b_self || col_stage[P.blk_col[ka]] == s - 1)) {
									TPanelExt en;
									memset(&en, 0, sizeof(en));
									en.a_off = P.loff[ka];
									en.b_off = P.loff[kb];
									en.slot = uint16_t(slot_of[k]);
									fresh.push_back(en);
								} else
									upd_ents.push_back(TUpdEnt{P.loff[ka], P.loff[kb]});
							}
							us.ne = int32_t(int64_t(upd_ents.size()) - us.e0);
							upd_slots.push_back(us);
						}
						ps.inp = int32_t(ipair.size()) - ps.ip0;
						pslots.push_back(ps);
					}
				}
				TPanelHead hd;
				memset(&hd, 0, sizeof(hd));
				hd.n_cols = n_cols;
				hd.n_slots = int32_t(n_slots);
				hd.n_units = int32_t(n_units);
				hd.n_int_rows = int32_t(irow.size());
				// fresh entries by the wave that owns their slot, inside a wave by slot, inside a slot in list order
				// (of a wave's entries the handed-up blocks first: the kernel takes them eight at a time)
				std::stable_sort(fresh.begin(), fresh.end(), [n_stage_waves](const TPanelExt &x, const TPanelExt &y) {
					const int wx = x.slot % n_stage_waves, wy = y.slot % n_stage_waves, ux = x.kind < 2, uy = y.kind < 2;
					return wx < wy || (wx == wy && (ux < uy || (ux == uy && x.slot < y.slot))); });
				for(size_t e = 0; e < fresh.size(); ++ e)
					++ hd.ext_ptr[fresh[e].slot % n_stage_waves + 1];
				for(int v = 0; v < n_stage_waves; ++ v)
					hd.ext_ptr[v + 1] += hd.ext_ptr[v];
				{ // what the stage's launch must hold
					TPanelLaunch &r_cfg = panel_cfg[s];
					r_cfg.n_cap_units = std::max(r_cfg.n_cap_units, int32_t(n_units));
					r_cfg.n_cap_blk = std::max(r_cfg.n_cap_blk, int32_t(n_slots));
					r_cfg.n_cap_cols = std::max(r_cfg.n_cap_cols, int32_t(n_cols));
					int n_level_cols = 0, n_level = -1;
					for(size_t o = 0; o < pcols.size(); ++ o) {
						n_level_cols = (pcols[o].sub == n_level)? n_level_cols + 1 : 1;
						n_level = pcols[o].sub;
						r_cfg.n_cap_lvl = std::max(r_cfg.n_cap_lvl, int32_t(n_level_cols));
					}
				}
				if(int64_t(fresh.size()) != n_fresh)
					throw std::logic_error("panel package: fresh entries miscounted");
				const size_t n_at = panel_pkg.size();
				panel_pkg.resize(n_at + n_units, longlong2{0, 0});
				char *p_dst = reinterpret_cast<char*>(&panel_pkg[n_at]);
				memcpy(p_dst, &hd, sizeof(hd));
				p_dst += 64;
				memcpy(p_dst, pcols.data(), pcols.size() * sizeof(TPanelCol));
				p_dst += pcols.size() * sizeof(TPanelCol);
				memcpy(p_dst, pslots.data(), pslots.size() * sizeof(TPanelSlot));
				p_dst += pslots.size() * sizeof(TPanelSlot);
				if(!irow.empty())
					memcpy(p_dst, irow.data(), irow.size() * sizeof(uint32_t));
				p_dst += (irow.size() + 3) / 4 * 16;
				if(!ipair.empty())
					memcpy(p_dst, ipair.data(), ipair.size() * sizeof(uint32_t));
				p_dst += (ipair.size() + 3) / 4 * 16;
				if(!fresh.empty())
					memcpy(p_dst, fresh.data(), fresh.size() * sizeof(TPanelExt));
				for(size_t o = 0, n_level = 0; o < order.size(); ++ o) {
					const TColDesc &c = cols[order[o]];
					for(int64_t k = c.k0; k < c.k0 + c.nb; ++ k)
						img_slot[k] = slot_of[k];
					col_package[P.task_cols[order[o]]] = int32_t(panel_off.size());
					if(o > 0 && pcols[o].sub != pcols[o - 1].sub)
						++ n_level;
					col_level[P.task_cols[order[o]]] = int32_t(n_level);
				}
				panel_off.push_back(int64_t(n_at));
				panel_out_off.push_back(-1);
				hand_up.push_back(THandUp());
				Release_Slots();
			}
			panel_ptr[s + 1] = int32_t(panel_off.size());
			// the hand-up lists of the stage below (its packages exist already: the lists go behind this stage's, the heads are told)
			for(int32_t n_pkg = (s > 0)? panel_ptr[s - 1] : 0; s > 0 && n_pkg < panel_ptr[s]; ++ n_pkg) {
				THandUp &r_up = hand_up[size_t(n_pkg)];
				if(r_up.recs.empty())
					continue;
				// the list: [12 x int32: records whose operands are final after level 0, 1, ...][records, in that order][their pairs] --
				// the waves a level's column work leaves idle take the records that are ready, the rest is done at the end
				const size_t n_out = r_up.recs.size(), n_pairs = r_up.pairs.size() / 2;
				enum { OUT_LEVELS = 12 };
				std::vector<int32_t> rec_order(n_out), rec_new(n_out), level_end(OUT_LEVELS, 0);
				for(size_t o = 0; o < n_out; ++ o)
					rec_order[o] = int32_t(o);
				std::stable_sort(rec_order.begin(), rec_order.end(), [&](int32_t a, int32_t b) { return r_up.recs[size_t(a)].onp < r_up.recs[size_t(b)].onp; });
				for(size_t o = 0; o < n_out; ++ o) {
					rec_new[size_t(rec_order[o])] = int32_t(o);
					for(int l = std::min(r_up.recs[size_t(rec_order[o])].onp, int32_t(OUT_LEVELS) - 1); l < int(OUT_LEVELS); ++ l)
						++ level_end[size_t(l)];
				}
				std::vector<TPanelOut> recs_sorted(n_out);
				for(size_t o = 0; o < n_out; ++ o)
					recs_sorted[o] = r_up.recs[size_t(rec_order[o])];
				std::vector<uint32_t> sorted(n_pairs);
				{
					std::vector<int32_t> count(n_out + 1, 0);
					for(size_t e = 0; e < n_pairs; ++ e)
						++ count[size_t(rec_new[r_up.pairs[2 * e]]) + 1];
					for(size_t o = 0; o < n_out; ++ o) {
						recs_sorted[o].op0 = count[o];
						recs_sorted[o].onp = count[o + 1];
						count[o + 1] += count[o];
					}
					std::vector<int32_t> fill(count.begin(), count.end() - 1);
					for(size_t e = 0; e < n_pairs; ++ e) // (stable: the pairs of a record keep their order)
						sorted[size_t(fill[size_t(rec_new[r_up.pairs[2 * e]])] ++)] = r_up.pairs[2 * e + 1];
				}
				const size_t n_units = 3 + n_out + (n_pairs + 3) / 4;
				const size_t n_at = panel_pkg.size();
				panel_pkg.resize(n_at + n_units, longlong2{0, 0});
				memcpy(&panel_pkg[n_at], level_end.data(), OUT_LEVELS * sizeof(int32_t));
				memcpy(&panel_pkg[n_at + 3], recs_sorted.data(), n_out * sizeof(TPanelOut));
				memcpy(&panel_pkg[n_at + 3 + n_out], sorted.data(), n_pairs * sizeof(uint32_t));
				panel_out_off[size_t(n_pkg)] = int64_t(n_at);
				TPanelHead *p_head = reinterpret_cast<TPanelHead*>(&panel_pkg[size_t(panel_off[size_t(n_pkg)])]);
				p_head->ext_ptr[10] = int32_t(n_out);
				p_head->ext_ptr[11] = int32_t(n_units);
				panel_cfg[s - 1].n_cap_out = std::max(panel_cfg[s - 1].n_cap_out, int32_t(n_units));
				b_any_hand_up = true;
				{ THandUp t_empty; std::swap(r_up, t_empty); }
			}
			out_of.clear();
			if(b_timing && b_panel_stage)
				fprintf(stderr, "[setup] stage %d panels: at most %lld blocks and %lld package units per task, %lld tasks left to the column kernel\n",
					s, (long long)n_stage_max_slots, (long long)n_stage_max_units, (long long)n_stage_rest);
			panel_rest_ptr[s + 1] = int32_t(panel_rest.size());
			panel_upd_ptr[s + 1] = int32_t(upd_slots.size());
		}
		static_assert(sizeof(TPanelOut) == 16 && sizeof(TPanelHead) == 64 && sizeof(TPanelCol) == 48 && sizeof(TPanelSlot) == 32 && sizeof(TPanelExt) == 32 && sizeof(TUpdSlot) == 64 &&
			sizeof(TUpdEnt) == 16, "record sizes");
		if(panel_off.empty()) {
			panel_ptr.clear();
			panel_rest_ptr.clear();
			panel_upd_ptr.clear();
		} else
			panel_pkg.resize(panel_pkg.size() + 64 * PANEL_W, longlong2{0, 0}); // speculative reads past the last package
	}
	bool b_lds_fits = true;
	for(size_t i = 0; i < panel_cfg.size() && !panel_off.empty(); ++ i)
		b_lds_fits = b_lds_fits && size_t(panel_lds(P.max_dim, true, panel_cfg[i]).TOTAL) * sizeof(double) <= PANEL_LDS_BUDGET;
	if(b_lds_fits || !b_hand_up_allowed)
		break;
	b_hand_up_allowed = false;
	}
	SETUP_PHASE("packages");
	d_panel_upd_slots.Upload(upd_slots, stream);
	d_panel_upd_ents.Upload(upd_ents, stream);
	d_panel_pkg.Upload(panel_pkg, stream);
	d_panel_off.Upload(panel_off, stream);
	d_panel_out_off.Upload(panel_out_off, stream);
	d_handup.Alloc(size_t(std::max<int64_t>(n_handup_doubles, int64_t(P.max_dim) * P.max_dim + 8))); // (every wave of a fused panel launch prefetches one block + 8 from offset 0, hand-ups or not)
	d_panel_rest.Upload(panel_rest, stream);
	d_cols.Upload(cols, stream);
	d_blks.Upload(blks, stream);
	d_pairs.Upload(pairs, stream);
	d_rents.Upload(rents, stream);
	d_task_ptr.Upload(P.task_ptr, stream);
	if(!pkg.empty()) {
		d_pkg.Upload(pkg, stream);
		d_task_pkg.Upload(task_pkg, stream);
	} else {
		d_pkg.Free();
		d_task_pkg.Free();
	}
	SETUP_PHASE("uploads");
	d_L.Alloc(size_t(P.loff[n_lblocks]));
	d_Linv.Alloc(size_t(P.linv_off[P.n]));
	d_w.Alloc(size_t(P.cs_new[P.n]));
	d_flag.Alloc(1);
	SLAMPP_HIP_CHECK(hipMemsetAsync(d_flag.p(), 0, sizeof(int), stream)); // sync() before the first factorization reads it
	SETUP_PHASE("allocs");
	SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // the staging vectors above die here
	SETUP_PHASE("sync");
#undef SETUP_PHASE

	dplan.cols = d_cols.p(); dplan.blks = d_blks.p(); dplan.pairs = d_pairs.p(); dplan.rents = d_rents.p();
	dplan.task_ptr = d_task_ptr.p();
	dplan.uniform_dim = P.uniform_dim? P.max_dim : 0;
	dplan.pkg = d_pkg.p();
	dplan.task_pkg = d_pkg.p()? d_task_pkg.p() : 0;
	dplan.n_blks = n_lblocks;
	dplan.n_pairs = int64_t(pairs.size());
	dplan.n_rents = int64_t(rents.size());
	dplan.p_timing = 0;
	dplan.task_map = 0;
	{
		const double t_wait = wall_ms();
		t_simt_thread.t.join();
		if(p_simt_error)
			std::rethrow_exception(p_simt_error);
		Upload_Simt();
		if(b_timing)
			fprintf(stderr, "[setup] %-12s %8.2f ms since it was started, %.2f ms of them waited for\n", "shapes", wall_ms() - t_simt, wall_ms() - t_wait);
	}
	if(getenv("SLAMPP_HIP_STAGE_TIMING")) { // development aid: clock samples of the upper-stage kernel, printed at sync
		d_timing.Alloc(1 + 32 * 4096);
		SLAMPP_HIP_CHECK(hipMemsetAsync(d_timing.p(), 0, (1 + 32 * 4096) * sizeof(long long), stream));
		dplan.p_timing = d_timing.p();
	}
}

// Sorts the tasks of the wide bottom stages by shape for the lane-per-task kernel (simt_kernel.hip; the formats are
// described in sparse_kernels.h).  A shape is the task's whole program -- counts and operand indices, the operands
// numbered in order of first use -- so two tasks of one shape differ in nothing but where their blocks live.
// host part of the lane-per-task tables (no HIP call: runs on a thread of its own next to the rest of the analysis);
// Upload_Simt() sends what it built
void slampp_hip_solver::Build_Simt()
{
	simt_chunk_ptr.clear();
	simt_rest_ptr.clear();
	simt_lds_bytes.clear();
	simt_host_chunks.clear(); simt_host_prog.clear(); simt_host_tab.clear(); simt_host_rest.clear();
	simt_bwd_lds_bytes.clear();
	simt_host_bwd_chunks.clear(); simt_host_bwd_prog.clear(); simt_host_bwd_tab.clear();
	const Plan &P = plan;
	if(!n_simt || !P.uniform_dim || (P.max_dim != 3 && P.max_dim != 6 && P.max_dim != 7))
		return;
	// one lane per leaf task pays when there are enough tasks to fill waves with them: a small system (the reduced camera
	// system of 1000 cameras has 250 leaf tasks) is faster with a wave per task (0.49 -> 0.42 ms there)
	if(n_simt < 0 && P.stage_ptr.size() > 1 && P.stage_ptr[1] - P.stage_ptr[0] < 2048)
		return;
	enum { MIN_GROUP = 1, MAX_PROG = 4096, MAX_TABLE_BYTES = 40960 }; // (rare shapes run with few busy lanes, beside the others: cheaper than a launch of their own)
	const int n_stages = int(P.stage_ptr.size()) - 1;
	std::vector<TSimtChunk> &chunks = simt_host_chunks;
	std::vector<int32_t> &prog_all = simt_host_prog, &rest = simt_host_rest;
	std::vector<int64_t> &tab = simt_host_tab;
	struct TTask { int32_t n_task; std::vector<int32_t> ops; std::vector<int32_t> ys; };
	std::vector<int32_t> op_index(P.lrow.size(), -1), y_index(size_t(P.n), -1);
	simt_chunk_ptr.push_back(0);
	simt_rest_ptr.push_back(0);
	const size_t W = size_t(n_simt_width);
	for(int s = 0; s < n_bottom_stages && s < n_stages && s < n_simt_stages; ++ s) {
		std::map<std::vector<int32_t>, std::vector<TTask> > groups;
		int32_t n_stage_lds = 0, n_stage_bwd_lds = 0;
		for(int32_t t = P.stage_ptr[s]; t < P.stage_ptr[s + 1]; ++ t) {
			std::vector<int32_t> prog(4, 0);
			TTask tt;
			tt.n_task = t;
			int32_t n_blocks = 0;
			bool b_fits = true;
			auto op_of = [&](int32_t n_blk) {
				if(op_index[n_blk] < 0) {
					op_index[n_blk] = int32_t(tt.ops.size());
					tt.ops.push_back(n_blk);
				}
				return op_index[n_blk];
			};
			for(int64_t i = P.task_ptr[t]; i < P.task_ptr[t + 1] && b_fits; ++ i) {
				const int32_t j = P.task_cols[i];
				const int32_t nb = int32_t(P.lptr[j + 1] - P.lptr[j]), nr = int32_t(P.rptr[j + 1] - P.rptr[j]);
				prog.push_back(nb);
				prog.push_back(nr);
				const size_t n_touch_at = prog.size();
				prog.push_back(0); // number of distinct operands of the column, then their indices
				n_blocks += nb;
				std::vector<int32_t> touch, body;
				auto touch_op = [&](int32_t n_op) {
					if(std::find(touch.begin(), touch.end(), n_op) == touch.end())
						touch.push_back(n_op);
					return n_op;
				};
				for(int64_t e = P.rptr[j]; e < P.rptr[j + 1]; ++ e) {
					const int32_t n_blk = P.rblk[e], c = P.blk_col[n_blk];
					if(y_index[c] < 0) {
						y_index[c] = int32_t(tt.ys.size());
						tt.ys.push_back(c);
					}
					body.push_back(touch_op(op_of(n_blk)));
					body.push_back(y_index[c]);
				}
				for(int64_t k = P.lptr[j] + 1; k < P.lptr[j + 1]; ++ k) {
					body.push_back(int32_t(P.pptr[k + 1] - P.pptr[k]));
					for(int64_t e = P.pptr[k]; e < P.pptr[k + 1]; ++ e) {
						body.push_back(touch_op(op_of(P.pa[e])));
						body.push_back(touch_op(op_of(P.pb[e])));
					}
				}
				prog[n_touch_at] = int32_t(touch.size());
				prog.insert(prog.end(), touch.begin(), touch.end());
				prog.insert(prog.end(), body.begin(), body.end());
				b_fits = prog.size() <= MAX_PROG;
			}
			for(size_t k = 0; k < tt.ops.size(); ++ k)
				op_index[tt.ops[k]] = -1;
			for(size_t k = 0; k < tt.ys.size(); ++ k)
				y_index[tt.ys[k]] = -1;
			const int32_t n_cols = int32_t(P.task_ptr[t + 1] - P.task_ptr[t]);
			prog[0] = n_cols;
			prog[1] = n_blocks;
			prog[2] = int32_t(tt.ops.size());
			prog[3] = int32_t(tt.ys.size());
			if(!b_fits || size_t(4 * n_cols + n_blocks) + tt.ops.size() + tt.ys.size() > MAX_TABLE_BYTES / (8 * W)) // (the table is staged in LDS)
				rest.push_back(t);
			else
				groups[prog].push_back(std::move(tt));
		}
		for(auto &r_group : groups) {
			const std::vector<int32_t> &prog = r_group.first;
			std::vector<TTask> &tasks = r_group.second;
			if(tasks.size() < MIN_GROUP) {
				for(const TTask &tt : tasks)
					rest.push_back(tt.n_task);
				continue;
			}
			const int32_t n_prog_off = int32_t(prog_all.size());
			prog_all.insert(prog_all.end(), prog.begin(), prog.end());
			const int n_cols = prog[0], n_blocks = prog[1], n_ops = prog[2], n_ys = prog[3];
			const int n_fields = 4 * n_cols + n_blocks + n_ops + n_ys;
			n_stage_lds = std::max(n_stage_lds, int32_t(n_fields * W * 8));
			// the shape's backward program: n_cols, blocks below the diagonals, nb per column
			const int32_t n_bwd_prog_off = int32_t(simt_host_bwd_prog.size());
			const int n_bwd_fields = 3 * n_cols + (n_blocks - n_cols);
			n_stage_bwd_lds = std::max(n_stage_bwd_lds, int32_t(n_bwd_fields * W * 8));
			simt_host_bwd_prog.push_back(n_cols);
			simt_host_bwd_prog.push_back(n_blocks - n_cols);
			{
				const TTask &tt = tasks[0];
				for(int64_t i = P.task_ptr[tt.n_task]; i < P.task_ptr[tt.n_task + 1]; ++ i)
					simt_host_bwd_prog.push_back(int32_t(P.lptr[P.task_cols[i] + 1] - P.lptr[P.task_cols[i]]));
			}
			for(size_t n_first = 0; n_first < tasks.size(); n_first += W) {
				const size_t n_in_chunk = std::min<size_t>(W, tasks.size() - n_first);
				TSimtChunk ch;
				ch.prog_off = n_prog_off;
				ch.n_tasks = int32_t(n_in_chunk);
				ch.tab_off = int64_t(tab.size());
				chunks.push_back(ch);
				tab.resize(tab.size() + size_t(n_fields) * W);
				int64_t *p_tab = &tab[size_t(ch.tab_off)];
				for(int n_lane = 0; n_lane < int(W); ++ n_lane) {
					const TTask &tt = tasks[n_first + std::min<size_t>(n_lane, n_in_chunk - 1)]; // spare lanes repeat the last task
					int f = 0;
					for(int64_t i = P.task_ptr[tt.n_task]; i < P.task_ptr[tt.n_task + 1]; ++ i) {
						const int32_t j = P.task_cols[i];
						p_tab[W * (f ++) + n_lane] = P.loff[P.lptr[j]];
						p_tab[W * (f ++) + n_lane] = P.linv_off[j];
						p_tab[W * (f ++) + n_lane] = P.cs_new[j];
						p_tab[W * (f ++) + n_lane] = P.cs_src[j];
					}
					for(int64_t i = P.task_ptr[tt.n_task]; i < P.task_ptr[tt.n_task + 1]; ++ i) {
						const int32_t j = P.task_cols[i];
						for(int64_t k = P.lptr[j]; k < P.lptr[j + 1]; ++ k)
							p_tab[W * (f ++) + n_lane] = (P.asrc[k] < 0)? -1 : P.asrc[k] * 2 + P.atrans[k];
					}
					for(int32_t n_blk : tt.ops)
						p_tab[W * (f ++) + n_lane] = P.loff[n_blk];
					for(int32_t c : tt.ys)
						p_tab[W * (f ++) + n_lane] = P.cs_new[c];
					if(f != n_fields)
						throw std::logic_error("lane-per-task tables: field count mismatch");
				}
				TSimtChunk ch_bwd;
				ch_bwd.prog_off = n_bwd_prog_off;
				ch_bwd.n_tasks = int32_t(n_in_chunk);
				ch_bwd.tab_off = int64_t(simt_host_bwd_tab.size());
				simt_host_bwd_chunks.push_back(ch_bwd);
				simt_host_bwd_tab.resize(simt_host_bwd_tab.size() + size_t(n_bwd_fields) * W);
				int64_t *p_bwd = &simt_host_bwd_tab[size_t(ch_bwd.tab_off)];
				for(int n_lane = 0; n_lane < int(W); ++ n_lane) {
					const TTask &tt = tasks[n_first + std::min<size_t>(n_lane, n_in_chunk - 1)];
					int f = 0;
					for(int64_t i = P.task_ptr[tt.n_task]; i < P.task_ptr[tt.n_task + 1]; ++ i) {
						const int32_t j = P.task_cols[i];
						p_bwd[W * (f ++) + n_lane] = P.loff[P.lptr[j]];
						p_bwd[W * (f ++) + n_lane] = P.cs_new[j];
						p_bwd[W * (f ++) + n_lane] = P.cs_src[j];
					}
					for(int64_t i = P.task_ptr[tt.n_task]; i < P.task_ptr[tt.n_task + 1]; ++ i) {
						const int32_t j = P.task_cols[i];
						for(int64_t k = P.lptr[j] + 1; k < P.lptr[j + 1]; ++ k) {
							if(P.loff[k] != P.loff[P.lptr[j]] + (k - P.lptr[j]) * int64_t(P.max_dim) * P.max_dim)
								throw std::logic_error("lane-per-task tables: the blocks of a column are not contiguous");
							p_bwd[W * (f ++) + n_lane] = P.cs_new[P.lrow[k]];
						}
					}
					if(f != n_bwd_fields)
						throw std::logic_error("lane-per-task tables: backward field count mismatch");
				}
			}
		}
		std::sort(rest.begin() + simt_rest_ptr.back(), rest.end());
		simt_chunk_ptr.push_back(int32_t(chunks.size()));
		simt_rest_ptr.push_back(int32_t(rest.size()));
		simt_lds_bytes.push_back(n_stage_lds);
		simt_bwd_lds_bytes.push_back(n_stage_bwd_lds);
	}
	if(chunks.empty()) {
		simt_chunk_ptr.clear();
		simt_rest_ptr.clear();
		return;
	}
}

// inv(L_jj) of the columns of the lane-per-task stages, where the factorization left them out: computed from the factor, once
// per factorization, and stored by every factorization from now on
void slampp_hip_solver::Ensure_Leaf_Inverses()
{
	b_leaf_linv_wanted = true;
	if(b_leaf_linv_valid || simt_chunk_ptr.empty())
		return;
	const Plan &P = plan;
	const int n_simt_stages_used = int(simt_chunk_ptr.size()) - 1;
	const int64_t n_col_end = P.task_ptr[size_t(P.stage_ptr[size_t(n_simt_stages_used)])];
	launch_invert_diagonals(dplan, 0, n_col_end, d_L.p(), d_Linv.p(), stream);
	b_leaf_linv_valid = true;
}

void slampp_hip_solver::Upload_Simt()
{
	const Plan &P = plan;
	if(simt_host_chunks.empty())
		return;
	d_simt_chunks.Upload(simt_host_chunks, stream);
	d_simt_prog.Upload(simt_host_prog, stream);
	d_simt_tab.Upload(simt_host_tab, stream);
	d_simt_rest.Upload(simt_host_rest, stream);
	d_simt_bwd_chunks.Upload(simt_host_bwd_chunks, stream);
	d_simt_bwd_prog.Upload(simt_host_bwd_prog, stream);
	d_simt_bwd_tab.Upload(simt_host_bwd_tab, stream);
	SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // (the host copies are no longer needed)
	{ std::vector<TSimtChunk> e; simt_host_bwd_chunks.swap(e); }
	{ std::vector<int32_t> e; simt_host_bwd_prog.swap(e); }
	{ std::vector<int64_t> e; simt_host_bwd_tab.swap(e); }
	{ std::vector<TSimtChunk> e; simt_host_chunks.swap(e); }
	{ std::vector<int32_t> e0, e1; simt_host_prog.swap(e0); simt_host_rest.swap(e1); }
	{ std::vector<int64_t> e; simt_host_tab.swap(e); }
	if(getenv("SLAMPP_HIP_PLAN_TIMING")) {
		for(size_t s = 0; s + 1 < simt_chunk_ptr.size(); ++ s) {
			fprintf(stderr, "[setup] stage %zu: %d tasks -> %d chunks of 64 lanes, %d tasks left to the wave-per-task kernel\n", s,
				P.stage_ptr[s + 1] - P.stage_ptr[s], simt_chunk_ptr[s + 1] - simt_chunk_ptr[s], simt_rest_ptr[s + 1] - simt_rest_ptr[s]);
		}
	}
}

void slampp_hip_solver::Enqueue_Sparse(const double *p_values_dev, double *p_rhs_dev, bool b_factor, bool b_factor_only)
{
	const Plan &P = plan;
	const int n_stages = int(P.stage_ptr.size()) - 1;
	int *p_flag = p_flag_shared? p_flag_shared : d_flag.p(); // (the inner solver of a Schur solve reports into the outer one's flag)
	if(b_factor && b_refined) { // wide block columns were cut into pieces: the values regrouped accordingly
		launch_gather_values(d_refine_map.p(), n_refined_values, p_values_dev, d_refined.p(), stream);
		p_values_dev = d_refined.p();
	}
	// (the backward kernel of the lane-per-task stages writes x with 16-byte stores where the block dimension is even)
	// (option simt_backward, -1 = by size: with the leaf subtrees of 20 000 poses the lane-per-task backward kernel costs 14 us of
	// 179, at 100 000 -- 15 928 subtrees -- the step is 0.318 -> 0.313 ms, at 300 000 0.853 -> 0.805, at a million 2.11 -> 1.96)
	const bool b_simt_backward_wanted = (n_simt_backward < 0)? P.stage_ptr.size() > 1 && P.stage_ptr[1] - P.stage_ptr[0] >= 12288 : n_simt_backward != 0;
	const bool b_simt_bwd = b_simt_backward_wanted && !simt_chunk_ptr.empty() && d_simt_bwd_chunks.p() &&
		(P.max_dim % 2 != 0 || (reinterpret_cast<uintptr_t>(p_rhs_dev) & 15) == 0);
	if(b_factor)
		b_leaf_linv_valid = true; // (every factor kernel but the lane-per-task one stores its inverses; that one answers below)
	else
		Ensure_Leaf_Inverses(); // another right-hand side: the forward kernel multiplies by inv(L_jj)
	if(b_factor) {
		// numeric factorization with the forward substitution fused in
		// (the flag is zero here: set to zero when it was allocated and again by every slampp_hip_sync() that found it raised.
		// A memset per solve erased an earlier solve's failure before slampp_hip_sync() could report it: the call answers for
		// everything enqueued since the last one)
		// the lane-per-task kernel reads blocks and vectors with 16-byte loads where the block dimension is even
		const bool b_simt = !simt_chunk_ptr.empty() && (P.max_dim % 2 != 0 ||
			((reinterpret_cast<uintptr_t>(p_values_dev) | reinterpret_cast<uintptr_t>(p_rhs_dev)) & 15) == 0);
		// phases: the leaf subtrees (stage 0), the wide stages right above them, the separators further up
		const int n_wide_end = std::min(n_bottom_stages, n_stages);
		// A stage of panel tasks: the updates its blocks receive from stages further down were applied inside the launch of the
		// stage below if that was a panel launch too (nothing there depends on them: they ride as extra workgroups), by a
		// launch of their own otherwise; what the stage right below contributed is brought in by the tasks themselves.
		bool b_panel_fused = false;
		for(size_t i = 0; i < panel_ride.size(); ++ i)
			b_panel_fused = b_panel_fused || panel_ride[i] != 0;
		b_panel_fused = b_panel_fused || b_any_hand_up; // (the handed-up blocks come in through the fresh entries' loop)
		auto Launch_Panels = [&](int s, bool b_bottom) {
			const int n_panels = panel_ptr[s + 1] - panel_ptr[s];
			const bool b_rode = panel_ride[s] != 0;
			if(!b_rode)
				launch_panel_update(P.max_dim, d_panel_upd_slots.p() + panel_upd_ptr[s], panel_upd_ptr[s + 1] - panel_upd_ptr[s],
					d_panel_upd_ents.p(), p_values_dev, d_L.p(), p_rhs_dev, d_w.p(), stream);
			const int n_next = (s + 1 < n_stages && panel_ride[s + 1] == 1)? panel_upd_ptr[s + 2] - panel_upd_ptr[s + 1] : 0;
			if(!launch_factor_panel(P.max_dim, b_panel_fused, (n_panel_rows < 0)? P.max_dim >= 6 : n_panel_rows != 0, panel_cfg[s], d_panel_pkg.p(), d_panel_off.p() + panel_ptr[s],
				d_panel_out_off.p() + panel_ptr[s], n_panels,
				d_panel_upd_slots.p() + ((n_next > 0)? panel_upd_ptr[s + 1] : 0), n_next, d_panel_upd_ents.p(), p_values_dev, p_rhs_dev,
				d_L.p(), d_Linv.p(), d_w.p(), d_handup.p(), p_flag, stream, dplan.p_timing))
				throw CDeviceError("panel launch refused: block size or LDS request outside what the analysis planned for");
			if(panel_rest_ptr[s + 1] > panel_rest_ptr[s]) {
				TDevPlan t_rest = dplan;
				t_rest.task_map = d_panel_rest.p();
				launch_factor_stage(t_rest, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), panel_rest_ptr[s],
					panel_rest_ptr[s + 1] - panel_rest_ptr[s], b_bottom, p_flag, stream);
			}
		};
		for(int s = 0; s < n_stages; ++ s) {
			if(s == 0)
				Phase_Begin("factor_leaves");
			else if(s == 1 && (b_profile >= 2 || n_wide_end <= 1))
				Phase_Begin((s < n_wide_end)? "factor_wide" : "factor_upper");
			else if(s == 1)
				Phase_Begin("factor_rest"); // the wide stages and the separators as one phase
			else if(s == n_wide_end && b_profile >= 2)
				Phase_Begin("factor_upper");
			if(b_simt && s + 1 < int(simt_chunk_ptr.size())) {
				const int n_chunks = simt_chunk_ptr[s + 1] - simt_chunk_ptr[s], n_rest = simt_rest_ptr[s + 1] - simt_rest_ptr[s];
				const bool b_store_linv = b_leaf_linv_wanted || !b_simt_backward_wanted; // (the wave-per-task backward kernel reads the inverses)
				launch_factor_simt(d_simt_chunks.p() + simt_chunk_ptr[s], n_chunks, n_simt_width, simt_lds_bytes[s], d_simt_prog.p(), d_simt_tab.p(), P.max_dim,
					p_values_dev, d_L.p(), b_store_linv? d_Linv.p() : 0, p_rhs_dev, d_w.p(), p_flag, stream, dplan.p_timing);
				b_leaf_linv_valid = b_leaf_linv_valid && b_store_linv;
				if(n_rest > 0) {
					TDevPlan t_rest = dplan;
					t_rest.task_map = d_simt_rest.p();
					launch_factor_stage(t_rest, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), simt_rest_ptr[s], n_rest,
						true, p_flag, stream);
				}
			} else if(s == 0 && !panel_ptr.empty() && panel_ptr[1] > panel_ptr[0]) {
				Launch_Panels(s, true); // few leaf subtrees: as panels (they receive no updates: the update just copies Lambda's blocks over)
			} else if(s > 0 && s < n_bottom_stages && dplan.task_pkg)
				launch_factor_wide(dplan, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), P.stage_ptr[s],
					P.stage_ptr[s + 1] - P.stage_ptr[s], p_flag, stream);
			else if(s >= n_bottom_stages && !panel_ptr.empty()) {
				Launch_Panels(s, false); // separators: as panels in LDS where they fit, column by column otherwise
			} else
			launch_factor_stage(dplan, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), P.stage_ptr[s],
				P.stage_ptr[s + 1] - P.stage_ptr[s], s < n_bottom_stages, p_flag, stream);
			if(s == 0 || (s == n_wide_end - 1 && b_profile >= 2) || s == n_stages - 1)
				Phase_End();
		}
	} else {
		Phase_Begin("forward");
		for(int s = 0; s < n_stages; ++ s) {
			launch_forward_stage(dplan, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), P.stage_ptr[s],
				P.stage_ptr[s + 1] - P.stage_ptr[s], stream);
		}
		Phase_End();
	}
	if(b_factor_only && !n_dense_dim) { // (the caller wants every column of L: here they all are)
		SLAMPP_HIP_CHECK(hipGetLastError());
		return;
	}
	if(n_dense_dim) {
		// dense top: Schur complement onto the big separators, dense MFMA Cholesky, both substitutions
		const int ld = n_dense_pad;
		if(b_factor) {
			Phase_Begin("dense_assemble");
			if(b_dense_tiles && b_dense_clean) // (334 MB at the Venice-like C4's reduced system, 40 % of it in the schedule)
				tile_zero(dense_tiles, d_dense.p(), ld, stream, d_dense_unit.p()); // with the identity of padding and gaps
			else {
				SLAMPP_HIP_CHECK(hipMemsetAsync(d_dense.p(), 0, size_t(ld) * ld * sizeof(double), stream));
				b_dense_clean = b_dense_tiles;
				dense_prepare_padding(d_dense.p(), ld, n_dense_dim, stream);
				dense_prepare_gaps(d_dense.p(), ld, d_dense_gaps.p(), n_dense_gaps, stream);
			}
			launch_dense_assemble(dplan, d_dense_blks.p(), n_dense_blks, p_values_dev, d_L.p(), p_rhs_dev, d_w.p(),
				d_dense.p(), ld, false, stream);
			Phase_End();
			Phase_Begin("dense_chol");
			if(b_dense_tiles)
				tile_cholesky(dense_tiles, d_dense.p(), ld, n_dense_dim, d_dense_invdiag.p(), p_flag, stream);
			else
				dense_cholesky(d_dense.p(), ld, n_dense_dim, d_dense_invdiag.p(), p_flag, stream);
			Phase_End();
			if(b_factor_only) { // the dense top's columns back into the factor's block layout, and no substitutions
				launch_dense_gather_factor(d_dense_blks.p(), d_dense_blk_loff.p(), n_dense_blks, d_dense.p(), ld, d_L.p(), stream);
				SLAMPP_HIP_CHECK(hipGetLastError());
				return;
			}
		} else {
			Phase_Begin("dense_forward");
			launch_dense_assemble(dplan, d_dense_blks.p(), n_dense_blks, 0, d_L.p(), p_rhs_dev, d_w.p(),
				d_dense.p(), ld, true, stream);
			dense_forwardsolve(d_dense.p(), ld, d_dense_invdiag.p(), stream);
			Phase_End();
		}
		Phase_Begin("dense_solve");
		if(b_dense_tiles) // by the levels of the tile schedule, reading its nonzero tiles only
			tile_backsolve(dense_tiles, d_dense.p(), ld, n_dense_dim, d_dense_invdiag.p(), d_dense_z.p(), d_dense_x.p(), stream,
				d_dense_dst.p(), d_w.p(), p_rhs_dev);
		else
		dense_backsolve(d_dense.p(), ld, n_dense_dim, d_dense_invdiag.p(), d_dense_z.p(), d_dense_x.p(), stream,
			d_dense_dst.p(), d_w.p(), p_rhs_dev); // (x goes to w and to the caller's vector as each panel publishes it)
		Phase_End();
	}
	Phase_Begin("backward");
	for(int s = n_stages; s > 0; -- s) {
		if(b_simt_bwd && s < int(simt_chunk_ptr.size())) {
			// a lane-per-task stage: its chunks by backward_simt_kernel (no inverses read), the tasks of rare shapes by the
			// wave-per-task kernel (their factor kernel stored the inverses)
			const int n_chunks = simt_chunk_ptr[s] - simt_chunk_ptr[s - 1], n_rest = simt_rest_ptr[s] - simt_rest_ptr[s - 1];
			launch_backward_simt(d_simt_bwd_chunks.p() + simt_chunk_ptr[s - 1], n_chunks, n_simt_width, simt_bwd_lds_bytes[s - 1],
				d_simt_bwd_prog.p(), d_simt_bwd_tab.p(), P.max_dim, d_L.p(), d_w.p(), p_rhs_dev, stream);
			if(n_rest > 0) {
				TDevPlan t_rest = dplan;
				t_rest.task_map = d_simt_rest.p();
				launch_backward_stage(t_rest, d_L.p(), d_Linv.p(), d_w.p(), p_rhs_dev, simt_rest_ptr[s - 1], n_rest, stream);
			}
			continue;
		}
		if(s < int(simt_chunk_ptr.size()))
			Ensure_Leaf_Inverses(); // (the wave-per-task kernel on a lane-per-task stage: unaligned caller vector)
		launch_backward_stage(dplan, d_L.p(), d_Linv.p(), d_w.p(), p_rhs_dev, P.stage_ptr[s - 1],
			P.stage_ptr[s] - P.stage_ptr[s - 1], stream);
	}
	Phase_End();
	SLAMPP_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------

namespace {

// runs f, maps exceptions to status codes, records the message
template <class F>
int guarded(slampp_hip_solver *p, F f)
{
	if(!p)
		return SLAMPP_HIP_ERR_INVALID;
	try {
		if(hipSetDevice(p->n_device) != hipSuccess)
			throw CDeviceError("hipSetDevice failed");
		return f();
	} catch(std::bad_alloc&) {
		p->s_error = "out of memory";
		return SLAMPP_HIP_ERR_ALLOC;
	} catch(CDeviceError &e) {
		p->s_error = e.what();
		return SLAMPP_HIP_ERR_DEVICE;
	} catch(std::domain_error &e) {
		p->s_error = e.what();
		return SLAMPP_HIP_ERR_UNSUPPORTED;
	} catch(std::exception &e) {
		p->s_error = e.what();
		return SLAMPP_HIP_ERR_INVALID;
	}
}

int fail(slampp_hip_solver *p, int n_code, const char *p_s_msg)
{
	p->s_error = p_s_msg;
	return n_code;
}

} // anonymous namespace

extern "C" {

// development aid (SLAMPP_HIP_ABORT_TRACE=1): where an abort() came from, for the ones that say nothing
static struct sigaction g_abort_previous; // whoever had SIGABRT before us (pytest's faulthandler, torch): called after the trace

static void abort_trace_handler(int n_signal)
{
	void *p_frames[64];
	const int n_frames = backtrace(p_frames, 64);
	static const char p_s_head[] = "[slampp_hip] abort: backtrace follows\n";
	(void)!write(2, p_s_head, sizeof(p_s_head) - 1);
	backtrace_symbols_fd(p_frames, n_frames, 2);
	(void)sigaction(n_signal, &g_abort_previous, 0); // hand the signal back: the host's handler (or the default) runs next
	raise(n_signal);
}

static void abort_trace_install() // strictly opt-in, once per process
{
	void *p_frames[4];
	(void)backtrace(p_frames, 4); // the first call loads libgcc's unwinder and may allocate: not something to do inside the handler
	struct sigaction t_action;
	memset(&t_action, 0, sizeof(t_action));
	t_action.sa_handler = abort_trace_handler;
	sigemptyset(&t_action.sa_mask);
	memset(&g_abort_previous, 0, sizeof(g_abort_previous));
	g_abort_previous.sa_handler = SIG_DFL;
	(void)sigaction(SIGABRT, &t_action, &g_abort_previous);
}

int slampp_hip_create(slampp_hip_solver **pp_solver, int device_id)
{
	if(!pp_solver)
		return SLAMPP_HIP_ERR_INVALID;
	*pp_solver = 0;
	static const bool b_trace = [] { if(getenv("SLAMPP_HIP_ABORT_TRACE")) { abort_trace_install(); return true; } return false; }();
	(void)b_trace;
	int n_count = 0;
	if(hipGetDeviceCount(&n_count) != hipSuccess || n_count <= 0 || device_id < 0 || device_id >= n_count)
		return SLAMPP_HIP_ERR_DEVICE; // no silent CPU fallback: without a GPU there is no solver
	slampp_hip_solver *p = new(std::nothrow) slampp_hip_solver();
	if(!p)
		return SLAMPP_HIP_ERR_ALLOC;
	p->n_device = device_id;
	if(hipSetDevice(device_id) != hipSuccess ||
	   hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess ||
	   hipHostMalloc((void**)&p->p_host_flag, sizeof(int), hipHostMallocDefault) != hipSuccess) {
		delete p;
		return SLAMPP_HIP_ERR_DEVICE;
	}
	*pp_solver = p;
	return SLAMPP_HIP_OK;
}

int slampp_hip_create_multi(slampp_hip_solver **pp_solver, const int *p_device_ids, int n_devices)
{
	if(!pp_solver || !p_device_ids || n_devices < 1)
		return SLAMPP_HIP_ERR_INVALID;
	const int n_result = slampp_hip_create(pp_solver, p_device_ids[0]);
	if(n_result != SLAMPP_HIP_OK || n_devices == 1)
		return n_result;
	slampp_hip_solver *p_front = *pp_solver;
	int n_count = 0;
	(void)hipGetDeviceCount(&n_count);
	bool b_valid = n_devices <= 16;
	for(int i = 0; i < n_devices && b_valid; ++ i)
		b_valid = p_device_ids[i] >= 0 && p_device_ids[i] < n_count;
	if(!b_valid) {
		slampp_hip_destroy(p_front);
		*pp_solver = 0;
		return SLAMPP_HIP_ERR_INVALID;
	}
	// the members (a solver, a stream and a host thread per device) come up with the first Schur-mode analysis: a
	// handle that only ever sees pose graphs stays a plain solver on the first device
	p_front->group_devices.assign(p_device_ids, p_device_ids + n_devices);
	return SLAMPP_HIP_OK;
}

int slampp_hip_group_info(const slampp_hip_solver *p_solver, int *p_member_num, int64_t *p_point_bounds, int n_max_members,
	const char **pp_s_exchange)
{
	if(!p_solver)
		return SLAMPP_HIP_ERR_INVALID;
	const slampp_hip_solver &s = *p_solver;
	const int n_members = (s.p_group && s.b_group_active && s.b_analyzed)? group_member_num(*s.p_group) : 0;
	if(p_member_num)
		*p_member_num = n_members;
	if(pp_s_exchange)
		*pp_s_exchange = n_members? group_exchange_name(*s.p_group) : "none";
	if(p_point_bounds && n_members) {
		if(n_max_members < n_members)
			return SLAMPP_HIP_ERR_INVALID;
		try {
			std::vector<int64_t> bounds;
			shard_bounds(int64_t(s.cumsum.size()) - 1, s.n_matrix_cut, &s.bcol_ptr[0], n_members, bounds);
			std::copy(bounds.begin(), bounds.end(), p_point_bounds);
		} catch(std::bad_alloc&) {
			return SLAMPP_HIP_ERR_ALLOC;
		}
	}
	return SLAMPP_HIP_OK;
}

void slampp_hip_destroy(slampp_hip_solver *p_solver)
{
	if(p_solver) {
		if(p_solver->p_group) {
			group_destroy(p_solver->p_group);
			p_solver->p_group = 0;
		}
		(void)hipSetDevice(p_solver->n_device);
		(void)hipStreamSynchronize(p_solver->stream);
		for(slampp_hip_assembly *p_assembly : p_solver->assemblies) { // orphaned, not freed: the caller owns the handles
			assembly_destroy(p_assembly->p_state);
			p_assembly->p_state = 0;
			p_assembly->p_solver = 0;
		}
		delete p_solver;
	}
}

int slampp_hip_free_memory(slampp_hip_solver *p_solver)
{
	return guarded(p_solver, [&]() -> int {
		if(p_solver->p_group) {
			const int n_group_result = group_free_memory(*p_solver->p_group);
			p_solver->b_group_active = false;
			p_solver->b_analyzed = p_solver->b_analyzed && p_solver->n_mode == SLAMPP_HIP_MODE_SPARSE;
			SLAMPP_HIP_CHECK(hipSetDevice(p_solver->n_device));
			if(n_group_result != SLAMPP_HIP_OK)
				return fail(p_solver, n_group_result, "device group: a member could not free its memory");
		}
		SLAMPP_HIP_CHECK(hipStreamSynchronize(p_solver->stream));
		p_solver->Free_Device();
		if(p_solver->copy_stream)
			SLAMPP_HIP_CHECK(hipStreamSynchronize(p_solver->copy_stream));
		p_solver->Free_Staging();
		p_solver->plan = Plan();
		return SLAMPP_HIP_OK;
	});
}

const char *slampp_hip_last_error(const slampp_hip_solver *p_solver)
{
	return p_solver? p_solver->s_error.c_str() : "null solver handle";
}

// the option itself, on this handle (and on the members of its device group, if they exist)
static int set_option_checked(slampp_hip_solver *p_solver, const char *p_s_name, int64_t n_value)
{
	if(!p_solver || !p_s_name)
		return SLAMPP_HIP_ERR_INVALID;
	const std::string s(p_s_name);
	if(p_solver->p_group) { // the members take the same options (the front handle keeps them for the sparse mode)
		const int n_group_result = group_set_option(*p_solver->p_group, p_s_name, n_value);
		(void)hipSetDevice(p_solver->n_device);
		if(n_group_result != SLAMPP_HIP_OK)
			return fail(p_solver, n_group_result, "unknown option or value out of range");
	}
	if(s == "group_exchange" && n_value >= 0 && n_value <= 2) {
		// (without a device list there is nothing to exchange: accepted, so that one configuration serves both)
	} else if(s == "group_fail_member" && n_value >= 0 && n_value <= 16)
		return SLAMPP_HIP_OK; // test hook of the device group (group.hip): nothing on a single-device handle
	else if(s == "leaf_size" && n_value >= 1)
		p_solver->opt.leaf_size = int(n_value);
	else if(s == "subtree_size" && n_value >= 1)
		p_solver->opt.subtree_size = int(n_value);
	else if(s == "task_height" && n_value >= 1 && n_value <= 8)
		p_solver->opt.task_height = int(n_value);
	else if(s == "natural_order")
		p_solver->opt.natural_order = (n_value != 0);
	else if(s == "nd_balance" && n_value >= 1 && n_value <= 49)
		p_solver->opt.nd_balance_pct = int(n_value);
	else if(s == "dense_nb" && (n_value == 32 || n_value == 64 || n_value == 128))
		p_solver->n_dense_nb = int(n_value);
	else if(s == "dense_top_nb" && n_value >= 0) {
		p_solver->opt.dense_top_nb = int(n_value);
		p_solver->opt.dense_top_auto = false; // the caller's threshold, as is
	}
	else if(s == "dense_top_max_dim" && n_value >= 0)
		p_solver->opt.dense_top_max_dim = int(n_value);
	else if(s == "dense_top_min_dim" && n_value >= 0)
		p_solver->opt.dense_top_min_dim = int(n_value);
	else if(s == "shard_primary")
		p_solver->b_shard_primary = (n_value != 0);
	else if(s == "shard_rank" && n_value >= 0) {
		p_solver->n_shard_rank = int(n_value);
		return SLAMPP_HIP_OK; // read when the ranks agree on their block list: does not invalidate the analysis
	} else if(s == "shard_world" && n_value >= 0) {
		p_solver->n_shard_world = int(n_value);
		return SLAMPP_HIP_OK;
	}
	else if(s == "assembly_groups" && n_value >= 0) {
		p_solver->n_assembly_groups = int(std::min(n_value, int64_t(1 << 20)));
		return SLAMPP_HIP_OK; // read by slampp_hip_assembly_create: does not invalidate the analysis
	}
	else if(s == "marginals_dense" && n_value >= 0 && n_value <= 1) {
		p_solver->n_marginals_dense = int(n_value);
		return SLAMPP_HIP_OK; // read by schur_marginals: does not invalidate the analysis
	}
	else if(s == "schur_sparse" && n_value >= -1 && n_value <= 1)
		p_solver->n_schur_sparse = int(n_value);
	else if(s == "staging_ahead" && n_value >= 0 && n_value <= 1) {
		p_solver->n_staging_ahead = int(n_value);
		return SLAMPP_HIP_OK; // read by analyze: does not invalidate anything
	}
	else if(s == "schur_fallback" && n_value >= 0 && n_value <= 1)
		p_solver->n_schur_fallback_option = int(n_value);
	else if(s == "schur_distributed" && n_value >= 0 && n_value <= 1) {
		p_solver->n_schur_distributed = int(n_value);
		return SLAMPP_HIP_OK; // read at every solve
	}
	else if(s == "schur_tiles" && n_value >= -1 && n_value <= 3)
		p_solver->n_schur_tiles = int(n_value);
	else if(s == "schur_incremental" && n_value >= 0 && n_value <= 2)
		p_solver->n_schur_incremental = int(n_value);
	else if(s == "dense_top_tiles" && n_value >= -1 && n_value <= 1) {
		p_solver->n_dense_top_tiles = int(n_value);
		p_solver->opt.dense_top_align = n_value? 64 : 0; // the alignment padding only serves the tile schedule
	}
	else if(s == "simt" && n_value >= -1 && n_value <= 1)
		p_solver->n_simt = int(n_value);
	else if(s == "wide_min_tasks" && n_value >= 1)
		p_solver->n_wide_min_tasks = int(n_value);
	else if(s == "simt_width" && (n_value == 16 || n_value == 32 || n_value == 64))
		p_solver->n_simt_width = int(n_value);
	else if(s == "panel" && n_value >= -1 && n_value <= 1)
		p_solver->n_panel = int(n_value);
	else if(s == "panel_handup" && n_value >= 0 && n_value <= 1)
		p_solver->n_panel_handup = int(n_value);
	else if(s == "panel_top" && n_value >= 0 && n_value <= 1)
		p_solver->n_panel_top = int(n_value);
	else if(s == "panel_rows" && n_value >= -1 && n_value <= 1) {
		p_solver->n_panel_rows = int(n_value);
		return SLAMPP_HIP_OK; // read at every launch
	}
	else if(s == "simt_stages" && n_value >= 0)
		p_solver->n_simt_stages = int(n_value);
	else if(s == "simt_backward" && n_value >= -1 && n_value <= 1) {
		p_solver->n_simt_backward = int(n_value);
		return SLAMPP_HIP_OK; // read at every solve
	}
	else if(s == "profile") {
		p_solver->b_profile = int(n_value); // 0 = off, 1 = phases, 2 = the factorization split further (every event pair costs microseconds), 3 = only the phase of the dominant kernel
		return SLAMPP_HIP_OK; // does not invalidate the analysis
	}
	else
		return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "unknown option or value out of range");
	p_solver->b_analyzed = false; // options take effect at the next analyze
	return SLAMPP_HIP_OK;
}

int slampp_hip_set_option(slampp_hip_solver *p_solver, const char *p_s_name, int64_t n_value)
{
	const int n_result = set_option_checked(p_solver, p_s_name, n_value);
	// recorded for members that do not exist yet -- only once the handle has accepted it: a refused option that was
	// recorded anyway would be replayed into the group at the first Schur-mode analysis and fail every analysis after it
	if(n_result == SLAMPP_HIP_OK && !p_solver->group_devices.empty()) {
		const std::string s(p_s_name);
		size_t i = 0;
		while(i < p_solver->group_options.size() && p_solver->group_options[i].first != s)
			++ i;
		if(i == p_solver->group_options.size())
			p_solver->group_options.push_back(std::make_pair(s, n_value));
		else
			p_solver->group_options[i].second = n_value;
	}
	return n_result;
}

int slampp_hip_group_exchange_count(const slampp_hip_solver *p_solver, int64_t *p_n_enqueued)
{
	if(!p_solver || !p_n_enqueued)
		return SLAMPP_HIP_ERR_INVALID;
	*p_n_enqueued = p_solver->p_group? group_exchange_count(*p_solver->p_group) : 0;
	return SLAMPP_HIP_OK;
}

int slampp_hip_set_structure(slampp_hip_solver *p_solver, int64_t n_bcols, const int64_t *p_bcol_cumsum,
	const int64_t *p_bcol_ptr, const int32_t *p_brow_idx)
{
	return guarded(p_solver, [&]() -> int {
		if(n_bcols <= 0 || !p_bcol_cumsum || !p_bcol_ptr || (p_bcol_ptr[n_bcols] > 0 && !p_brow_idx))
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_structure: null or empty structure");
		if(p_bcol_cumsum[0] != 0 || p_bcol_ptr[0] != 0)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_structure: cumsum / pointer arrays must start at 0");
		slampp_hip_solver &s = *p_solver;
		const bool b_same = s.b_has_structure && int64_t(s.cumsum.size()) == n_bcols + 1 &&
			std::equal(s.cumsum.begin(), s.cumsum.end(), p_bcol_cumsum) &&
			std::equal(s.bcol_ptr.begin(), s.bcol_ptr.end(), p_bcol_ptr) &&
			std::equal(s.brow.begin(), s.brow.end(), p_brow_idx);
		if(!b_same) {
			for(slampp_hip_assembly *p_assembly : s.assemblies)
				p_assembly->b_stale = true; // their block offsets belong to the previous structure
		}
		s.cumsum.assign(p_bcol_cumsum, p_bcol_cumsum + n_bcols + 1);
		s.bcol_ptr.assign(p_bcol_ptr, p_bcol_ptr + n_bcols + 1);
		s.brow.assign(p_brow_idx, p_brow_idx + p_bcol_ptr[n_bcols]);
		int64_t n_values = 0;
		for(int64_t c = 0; c < n_bcols; ++ c) {
			const int64_t w = s.cumsum[c + 1] - s.cumsum[c];
			if(w <= 0 || s.bcol_ptr[c + 1] < s.bcol_ptr[c])
				return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_structure: malformed cumsum / pointer arrays");
			for(int64_t k = s.bcol_ptr[c]; k < s.bcol_ptr[c + 1]; ++ k) {
				const int32_t r = s.brow[k];
				if(r < 0 || r > c)
					return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_structure: block outside the upper triangle");
				n_values += (s.cumsum[r + 1] - s.cumsum[r]) * w;
			}
		}
		s.n_values = n_values;
		s.n_scalars = s.cumsum[n_bcols];
		s.b_has_structure = true;
		s.b_analyzed = false;
		s.b_factored = false;
		s.b_damp_valid = false;
		s.n_uploaded = 0;
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_apply_damping_device_async(slampp_hip_solver *p_solver, double *p_values_dev, double f_alpha,
	int64_t n_first_vertex, int64_t n_last_vertex)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_has_structure)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "apply_damping: set_structure was not called");
		const int64_t n = int64_t(s.cumsum.size()) - 1;
		if(!p_values_dev || n_first_vertex < 0 || n_first_vertex > n_last_vertex || n_last_vertex > n)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "apply_damping: null pointer or bad vertex range");
		if(!s.b_damp_valid) {
			std::vector<int64_t> off_dim(size_t(2 * n));
			int64_t n_off = 0;
			for(int64_t c = 0; c < n; ++ c) {
				const int64_t w = s.cumsum[c + 1] - s.cumsum[c];
				if(s.bcol_ptr[c + 1] == s.bcol_ptr[c] || s.brow[s.bcol_ptr[c + 1] - 1] != c)
					return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "apply_damping: a block column has no diagonal block");
				for(int64_t k = s.bcol_ptr[c]; k + 1 < s.bcol_ptr[c + 1]; ++ k)
					n_off += (s.cumsum[s.brow[k] + 1] - s.cumsum[s.brow[k]]) * w;
				off_dim[2 * c] = n_off; // the diagonal block is the last of its column
				off_dim[2 * c + 1] = w;
				n_off += w * w;
			}
			s.d_damp_off.Upload(off_dim, s.stream);
			SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream)); // off_dim lives on this stack frame
			s.b_damp_valid = true;
		}
		damping_enqueue(s.d_damp_off.p(), n_first_vertex, n_last_vertex, f_alpha, p_values_dev, s.stream);
		SLAMPP_HIP_CHECK(hipGetLastError());
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_analyze(slampp_hip_solver *p_solver, int n_mode, int64_t n_matrix_cut)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_has_structure)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "analyze: set_structure was not called");
		if(n_mode != SLAMPP_HIP_MODE_SPARSE && n_mode != SLAMPP_HIP_MODE_SCHUR)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "analyze: unknown mode");
		SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
		if(s.copy_stream)
			SLAMPP_HIP_CHECK(hipStreamSynchronize(s.copy_stream));
		s.Free_Device();
		s.b_group_active = false;
		s.b_schur_fallback = false;
		memset(&s.times, 0, sizeof(s.times));
		s.n_mode = n_mode;
		s.n_matrix_cut = n_matrix_cut;
		const double t0 = wall_ms();
		// option "staging_ahead" (callers that will hand over host arrays: the header class, the host entry points): the
		// pinned staging for Lambda's values -- 10 ms of page faults and registration at C3's 58 MB, more at C4's 336 MB --
		// comes up on a thread of its own while this one orders and analyzes
		std::exception_ptr p_staging_error;
		struct TJoin { std::thread t; ~TJoin() { if(t.joinable()) t.join(); } } t_staging_thread;
		if(s.n_staging_ahead && s.group_devices.empty() && !getenv("SLAMPP_HIP_NO_STAGING_AHEAD")) { // (the variable: a development aid)
			t_staging_thread.t = std::thread([&s, &p_staging_error]() {
				try {
					SLAMPP_HIP_CHECK(hipSetDevice(s.n_device));
					s.Require_Staging();
				} catch(...) {
					p_staging_error = std::current_exception();
				}
			});
		}
		if(n_mode == SLAMPP_HIP_MODE_SPARSE) {
			s.Analyze_Sparse();
			s.times.order_ms = s.plan.order_ms;
			s.times.symbolic_ms = wall_ms() - t0 - s.plan.order_ms;
		} else {
			const int64_t n = int64_t(s.cumsum.size()) - 1;
			const bool b_no_landmarks = n_matrix_cut <= 0 || n_matrix_cut >= n;
			if(b_no_landmarks && !s.n_schur_fallback_option)
				return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "analyze: n_matrix_cut must split the block columns");
			try {
				if(b_no_landmarks)
					throw std::domain_error("no landmark part");
				if(!s.group_devices.empty() && !s.p_group) {
					s.p_group = group_create(&s.group_devices[0], int(s.group_devices.size()));
					for(size_t i = 0; i < s.group_options.size(); ++ i) {
						if(group_set_option(*s.p_group, s.group_options[i].first.c_str(), s.group_options[i].second) != SLAMPP_HIP_OK)
							throw std::invalid_argument("device group: a member refused an option this handle had accepted");
					}
				}
				if(s.p_group) {
					group_analyze(s, n_matrix_cut); // landmark shards on the listed devices; this handle keeps structure and staging
					s.b_group_active = true;
					SLAMPP_HIP_CHECK(hipSetDevice(s.n_device));
				} else
					s.p_schur = schur_analyze(s);
			} catch(std::domain_error&) {
				// a structure the Schur kernels do not take, which the reference nevertheless solves (LinearSolver_Schur.h:1635-1638,
				// 1721-1726): the sparse block path on the whole of Lambda gives the same solution
				if(!s.n_schur_fallback_option)
					throw;
				(void)hipSetDevice(s.n_device);
				if(t_staging_thread.t.joinable())
					t_staging_thread.t.join(); // (it allocates the device arrays Free_Device() is about to free)
				s.Free_Device();
				s.b_group_active = false;
				s.b_schur_fallback = true;
				s.n_mode = SLAMPP_HIP_MODE_SPARSE; // from here on this is a sparse-mode handle that remembers why
				s.Analyze_Sparse();
				s.times.order_ms = s.plan.order_ms;
			}
			s.times.symbolic_ms = wall_ms() - t0;
		}
		if(t_staging_thread.t.joinable()) {
			t_staging_thread.t.join();
			if(p_staging_error)
				std::rethrow_exception(p_staging_error);
		}
		s.b_analyzed = true;
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_factor_solve_device_async(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_rhs_inout_dev)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve: analyze was not called");
		if(s.b_group_active)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve_device: this handle solves with landmark shards on several devices: host entry points only");
		if(!p_values_dev || !p_rhs_inout_dev)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve: null pointer");
		if(s.n_mode == SLAMPP_HIP_MODE_SPARSE)
			s.Enqueue_Sparse(p_values_dev, p_rhs_inout_dev, true);
		else
			schur_enqueue(s, p_values_dev, p_rhs_inout_dev);
		s.b_factored = true;
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_sync(slampp_hip_solver *p_solver)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(s.b_group_active)
			return SLAMPP_HIP_OK; // the host entry points of a sharded handle return with everything finished
		*s.p_host_flag = 0;
		if(s.d_flag.p())
			SLAMPP_HIP_CHECK(hipMemcpyAsync(s.p_host_flag, s.d_flag.p(), sizeof(int), hipMemcpyDeviceToHost, s.stream));
		SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
		s.Phase_Collect();
		if(s.dplan.p_timing && s.d_timing.p()) { // development aid: the last launches' clock samples (100 MHz ticks)
			std::vector<long long> tm(1 + 32 * 4096);
			SLAMPP_HIP_CHECK(hipMemcpy(tm.data(), s.d_timing.p(), tm.size() * sizeof(long long), hipMemcpyDeviceToHost));
			const long long n_launches = std::min<long long>(tm[0], 4096);
			for(long long i = std::max<long long>(0, n_launches - 40); i < n_launches; ++ i) {
				fprintf(stderr, "stage_timing launch %lld:", i);
				for(int k = 1; k < 32 && tm[1 + 32 * i + k]; ++ k)
					fprintf(stderr, " %.2f", double(tm[1 + 32 * i + k] - tm[1 + 32 * i + k - 1]) * 0.01);
				fprintf(stderr, " us\n");
			}
			SLAMPP_HIP_CHECK(hipMemset(s.d_timing.p(), 0, tm.size() * sizeof(long long)));
		}
		if(*s.p_host_flag) {
			SLAMPP_HIP_CHECK(hipMemsetAsync(s.d_flag.p(), 0, sizeof(int), s.stream)); // (what was enqueued since the last sync has been answered for)
			s.b_factored = false;
			schur_invalidate_previous(s.p_schur); // nothing to update from
			return fail(p_solver, SLAMPP_HIP_NOT_POSDEF, "matrix is not positive definite");
		}
		return SLAMPP_HIP_OK;
	});
}

void *slampp_hip_stream(slampp_hip_solver *p_solver)
{
	return p_solver? (void*)p_solver->stream : 0;
}

int slampp_hip_factor_solve_device(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_rhs_inout_dev, slampp_hip_times *p_times)
{
	if(!p_solver)
		return SLAMPP_HIP_ERR_INVALID;
	const double t0 = wall_ms();
	int n_result = slampp_hip_factor_solve_device_async(p_solver, p_values_dev, p_rhs_inout_dev);
	if(n_result == SLAMPP_HIP_OK)
		n_result = slampp_hip_sync(p_solver);
	p_solver->times.total_ms = wall_ms() - t0;
	if(p_times)
		*p_times = p_solver->times;
	return n_result;
}

int slampp_hip_factor_solve(slampp_hip_solver *p_solver, const double *p_values, double *p_rhs_inout,
	slampp_hip_times *p_times)
{
	const double t0 = wall_ms();
	int n_result = guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve: analyze was not called");
		if(!p_values || !p_rhs_inout)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve: null pointer");
		if(s.b_group_active)
			return SLAMPP_HIP_OK;
		s.Upload_Values(p_values);
		Upload_Rhs_And_Join(s, p_rhs_inout);
		return SLAMPP_HIP_OK;
	});
	if(n_result != SLAMPP_HIP_OK)
		return n_result;
	slampp_hip_solver &s = *p_solver;
	if(s.b_group_active) {
		n_result = guarded(p_solver, [&]() -> int { return group_factor_solve(s, p_values, p_rhs_inout); });
		s.n_uploaded = 0;
		if(p_times)
			*p_times = s.times;
		return n_result;
	}
	const double t1 = wall_ms(); // (the last chunks may still be on the bus: the solve is enqueued behind them)
	s.times.upload_ms = t1 - t0;
	n_result = slampp_hip_factor_solve_device_async(p_solver, s.d_A.p(), s.d_rhs.p());
	if(n_result == SLAMPP_HIP_OK) {
		n_result = guarded(p_solver, [&]() -> int { // the solution comes back behind the solve, one synchronization for both
			SLAMPP_HIP_CHECK(hipMemcpyAsync(s.p_pin_rhs, s.d_rhs.p(), size_t(s.n_scalars) * sizeof(double), hipMemcpyDeviceToHost, s.stream));
			return SLAMPP_HIP_OK;
		});
	}
	if(n_result == SLAMPP_HIP_OK)
		n_result = slampp_hip_sync(p_solver);
	const double t2 = wall_ms();
	if(s.n_mode == SLAMPP_HIP_MODE_SPARSE)
		s.times.factor_ms = t2 - t1; // factor + both substitutions (one stream, no sync between them)
	else
		s.times.schur_ms = t2 - t1;
	if(n_result == SLAMPP_HIP_OK && p_rhs_inout != s.p_pin_rhs)
		Parallel_Copy(p_rhs_inout, s.p_pin_rhs, size_t(s.n_scalars));
	s.times.download_ms = wall_ms() - t2;
	s.times.total_ms = wall_ms() - t0;
	if(p_times)
		*p_times = s.times;
	return n_result;
}

int slampp_hip_host_staging(slampp_hip_solver *p_solver, double **pp_values, double **pp_rhs)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_has_structure)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "host_staging: set_structure was not called");
		s.Require_Staging();
		if(s.b_group_active && s.p_group) { // every member must be able to DMA from it: checked once per allocation
			if(const int n_check = group_check_staging(s, s.p_pin_values, s.p_pin_rhs))
				return n_check;
		}
		if(pp_values)
			*pp_values = s.p_pin_values;
		if(pp_rhs)
			*pp_rhs = s.p_pin_rhs;
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_upload_values_async(slampp_hip_solver *p_solver, int64_t n_first, int64_t n_count)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(s.b_group_active) {
			if(!s.p_pin_values)
				return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "upload_values: host_staging was not called");
			return SLAMPP_HIP_OK; // every member fetches its own columns from the staging when the solve is called
		}
		if(!s.p_pin_values || !s.d_A.p())
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "upload_values: host_staging was not called");
		if(n_first == 0)
			s.n_uploaded = 0; // a new pass over the values (what an abandoned pass has sent is simply sent again)
		if(n_first != s.n_uploaded || n_count < 0 || n_first + n_count > s.n_values)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "upload_values: chunks must follow each other from 0 and stay inside the values");
		if(n_count) {
			SLAMPP_HIP_CHECK(hipMemcpyAsync(s.d_A.p() + n_first, s.p_pin_values + n_first, size_t(n_count) * sizeof(double),
				hipMemcpyHostToDevice, s.copy_stream));
		}
		s.n_uploaded = n_first + n_count;
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_schur_set_changed_points(slampp_hip_solver *p_solver, const int64_t *p_points, int64_t n_points)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if((s.b_group_active || s.b_schur_fallback) && s.b_analyzed)
			return SLAMPP_HIP_OK; // landmark shards rebuild the reduced system (as the header comment says): the list is a hint
		if(!s.b_analyzed || s.n_mode != SLAMPP_HIP_MODE_SCHUR || !s.p_schur)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_set_changed_points: analyze (Schur mode) was not called");
		if(!s.n_schur_incremental)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_set_changed_points: set the option schur_incremental first");
		if(n_points < 0 || (n_points > 0 && !p_points))
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_set_changed_points: null list");
		SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream)); // (the previous list may still be read)
		schur_set_changed_points(s, p_points, n_points);
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_solve_marginal_poses_device_async(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_rhs_inout_dev)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "solve_marginal_poses: analyze was not called");
		if(s.n_mode != SLAMPP_HIP_MODE_SCHUR)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "solve_marginal_poses: needs the Schur mode (cameras and landmarks)");
		if(s.b_group_active)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "solve_marginal_poses_device: this handle solves with landmark shards on several devices: host entry points only");
		if(!p_values_dev || !p_rhs_inout_dev)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "solve_marginal_poses: null pointer");
		schur_enqueue_marginal_poses(s, p_values_dev, p_rhs_inout_dev);
		s.b_factored = false; // no factor of the reduced system comes out of this
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_marginals_device_async(slampp_hip_solver *p_solver, const double *p_values_dev, double *p_block_diag_dev)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "marginals: analyze was not called");
		if(s.n_mode != SLAMPP_HIP_MODE_SPARSE)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "marginals: sparse mode only (Schur mode: slampp_hip_schur_marginals)");
		if(s.b_refined)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "marginals: block columns wider than 8 are factored in pieces: no covariance blocks in the caller's layout");
		if(!p_values_dev || !p_block_diag_dev)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "marginals: null pointer");
		const Plan &P = s.plan;
		if(!s.b_sinv_tried) {
			s.b_sinv_tried = true;
			s.p_sinv = sparse_inverse_setup(P, s.stream, true);
			if(s.p_sinv) {
				std::vector<int64_t> zoff(size_t(P.n));
				for(int32_t c = 0; c < P.n; ++ c) {
					const int32_t j = P.pinv[c];
					zoff[c] = (P.dense_dim && P.dense_pos[j] >= 0)? -int64_t(P.dense_pos[j]) - 1 : P.loff[P.lptr[j]];
				}
				s.d_diag_zoff.Upload(zoff, s.stream);
				if(!P.uniform_dim) { // mixed block sizes: where every caller's column's block goes, and how big it is
					std::vector<int32_t> dims(size_t(P.n));
					std::vector<int64_t> out_off(size_t(P.n));
					int64_t n_at = 0;
					for(int32_t c = 0; c < P.n; ++ c) {
						dims[c] = int32_t(s.cumsum[c + 1] - s.cumsum[c]);
						out_off[c] = n_at;
						n_at += int64_t(dims[c]) * dims[c];
					}
					s.d_diag_dim.Upload(dims, s.stream);
					s.d_diag_out_off.Upload(out_off, s.stream);
				}
				s.d_Z.Alloc(size_t(P.loff.back()));
				if(s.n_dense_dim) {
					s.d_Zd.Alloc(size_t(s.n_dense_pad) * s.n_dense_pad);
					s.d_Zd_work.Alloc(size_t(s.n_dense_pad) * s.n_dense_pad);
				}
				SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream)); // zoff lives on this stack frame
			}
		}
		if(!s.p_sinv)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "marginals: mixed block sizes are taken without a dense top only (set the option dense_top_nb to 0), block sizes above 8 not at all");
		// the fused forward substitution reads a right-hand side, and with a dense top it rides through that factorization
		// as a row of the matrix: zeros (a NaN there would spread through 0 x NaN in the tile products)
		s.d_rhs.Alloc(size_t(s.n_scalars));
		SLAMPP_HIP_CHECK(hipMemsetAsync(s.d_rhs.p(), 0, size_t(s.n_scalars) * sizeof(double), s.stream));
		// (with a dense top the whole factor + solve runs: the top is factored on the way; opens its own phases)
		s.b_leaf_linv_wanted = true; // (the inverse subset multiplies by inv(L_jj) of every column)
		s.Enqueue_Sparse(p_values_dev, s.d_rhs.p(), true, s.n_dense_dim == 0);
		s.Ensure_Leaf_Inverses();
		s.Phase_Begin("marginals_inverse");
		if(s.n_dense_dim) { // the top's inverse from a copy of its factor (the factor itself stays for solve_again)
			SLAMPP_HIP_CHECK(hipMemcpyAsync(s.d_Zd_work.p(), s.d_dense.p(), size_t(s.n_dense_pad) * s.n_dense_pad * sizeof(double),
				hipMemcpyDeviceToDevice, s.stream));
			dense_top_clear_rhs_row(s.d_Zd_work.p(), s.n_dense_pad, s.stream);
			dense_inverse_from_factor(s.d_Zd_work.p(), s.n_dense_pad, s.d_dense_invdiag.p(), s.d_Zd.p(), s.stream);
		}
		sparse_inverse_enqueue(*s.p_sinv, P, s.d_L.p(), s.d_Linv.p(), s.d_Z.p(), s.stream, s.d_Zd.p(), s.n_dense_pad);
		s.Phase_End();
		if(P.uniform_dim)
			inverse_diag_blocks_launch(P.n, P.max_dim, s.d_diag_zoff.p(), s.d_Z.p(), s.d_Zd.p(), s.n_dense_pad, p_block_diag_dev, s.stream);
		else
			inverse_diag_blocks_any_launch(P.n, s.d_diag_dim.p(), s.d_diag_zoff.p(), s.d_diag_out_off.p(), s.d_Z.p(), p_block_diag_dev, s.stream);
		SLAMPP_HIP_CHECK(hipGetLastError());
		s.b_factored = true; // the factor of these values is in place
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_marginals(slampp_hip_solver *p_solver, const double *p_values, double *p_block_diag)
{
	size_t n_out = 0;
	int n_result = guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "marginals: analyze was not called");
		if(!p_values || !p_block_diag)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "marginals: null pointer");
		for(size_t c = 0; c + 1 < s.cumsum.size(); ++ c)
			n_out += size_t((s.cumsum[c + 1] - s.cumsum[c]) * (s.cumsum[c + 1] - s.cumsum[c]));
		s.d_A.Alloc(size_t(s.n_values));
		s.d_cov.Alloc(n_out);
		Upload_Values_And_Join(s, p_values);
		return SLAMPP_HIP_OK;
	});
	if(n_result != SLAMPP_HIP_OK)
		return n_result;
	slampp_hip_solver &s = *p_solver;
	n_result = slampp_hip_marginals_device_async(p_solver, s.d_A.p(), s.d_cov.p());
	if(n_result == SLAMPP_HIP_OK)
		n_result = slampp_hip_sync(p_solver);
	if(n_result == SLAMPP_HIP_OK) {
		n_result = guarded(p_solver, [&]() -> int {
			SLAMPP_HIP_CHECK(hipMemcpyAsync(p_block_diag, s.d_cov.p(), n_out * sizeof(double), hipMemcpyDeviceToHost, s.stream));
			SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
			return SLAMPP_HIP_OK;
		});
	}
	return n_result;
}

int slampp_hip_schur_marginals_device_async(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_cam_cov_dev, double *p_point_cov_dev)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_marginals: analyze was not called");
		if(s.n_mode != SLAMPP_HIP_MODE_SCHUR)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "schur_marginals: needs the Schur mode (cameras and landmarks)");
		if(s.b_group_active)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_marginals_device: this handle solves with landmark shards on several devices: host entry points only");
		if(!p_values_dev || (!p_cam_cov_dev && !p_point_cov_dev))
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_marginals: null pointer");
		schur_enqueue_marginals(s, p_values_dev, p_cam_cov_dev, p_point_cov_dev);
		s.b_factored = false; // C^-1 and W were recomputed from these values: a kept factor may no longer match them
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_schur_marginals(slampp_hip_solver *p_solver, const double *p_values, double *p_cam_cov, double *p_point_cov)
{
	size_t n_cam_doubles = 0, n_point_doubles = 0;
	int n_result = guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_marginals: analyze was not called");
		if(s.n_mode != SLAMPP_HIP_MODE_SCHUR)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "schur_marginals: needs the Schur mode (cameras and landmarks)");
		if(!p_values || (!p_cam_cov && !p_point_cov))
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "schur_marginals: null pointer");
		if(s.b_group_active)
			return group_schur_marginals(s, p_values, p_cam_cov, p_point_cov);
		const int64_t nc = s.n_matrix_cut, np = int64_t(s.cumsum.size()) - 1 - nc;
		const int64_t dc = s.cumsum[1] - s.cumsum[0], dp = s.cumsum[nc + 1] - s.cumsum[nc];
		n_cam_doubles = size_t(nc * dc * dc);
		n_point_doubles = size_t(np * dp * dp);
		s.d_A.Alloc(size_t(s.n_values));
		s.d_cov.Alloc(n_cam_doubles + n_point_doubles);
		Upload_Values_And_Join(s, p_values);
		return SLAMPP_HIP_OK;
	});
	if(n_result != SLAMPP_HIP_OK || p_solver->b_group_active)
		return n_result;
	slampp_hip_solver &s = *p_solver;
	n_result = slampp_hip_schur_marginals_device_async(p_solver, s.d_A.p(), p_cam_cov? s.d_cov.p() : 0,
		p_point_cov? s.d_cov.p() + n_cam_doubles : 0);
	if(n_result == SLAMPP_HIP_OK)
		n_result = slampp_hip_sync(p_solver);
	if(n_result == SLAMPP_HIP_OK) {
		n_result = guarded(p_solver, [&]() -> int {
			if(p_cam_cov)
				SLAMPP_HIP_CHECK(hipMemcpyAsync(p_cam_cov, s.d_cov.p(), n_cam_doubles * sizeof(double), hipMemcpyDeviceToHost, s.stream));
			if(p_point_cov)
				SLAMPP_HIP_CHECK(hipMemcpyAsync(p_point_cov, s.d_cov.p() + n_cam_doubles, n_point_doubles * sizeof(double),
					hipMemcpyDeviceToHost, s.stream));
			SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
			return SLAMPP_HIP_OK;
		});
	}
	return n_result;
}

int slampp_hip_solve_marginal_poses(slampp_hip_solver *p_solver, const double *p_values, double *p_rhs_inout)
{
	int n_result = guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "solve_marginal_poses: analyze was not called");
		if(!p_values || !p_rhs_inout)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "solve_marginal_poses: null pointer");
		if(s.b_group_active)
			return group_solve_marginal_poses(s, p_values, p_rhs_inout);
		s.d_A.Alloc(size_t(s.n_values));
		s.d_rhs.Alloc(size_t(s.n_scalars));
		s.Upload_Values(p_values);
		Upload_Rhs_And_Join(s, p_rhs_inout);
		return SLAMPP_HIP_OK;
	});
	if(n_result != SLAMPP_HIP_OK || p_solver->b_group_active)
		return n_result;
	slampp_hip_solver &s = *p_solver;
	n_result = slampp_hip_solve_marginal_poses_device_async(p_solver, s.d_A.p(), s.d_rhs.p());
	if(n_result == SLAMPP_HIP_OK)
		n_result = slampp_hip_sync(p_solver);
	if(n_result == SLAMPP_HIP_OK) {
		n_result = guarded(p_solver, [&]() -> int {
			SLAMPP_HIP_CHECK(hipMemcpyAsync(p_rhs_inout, s.d_rhs.p(), size_t(s.n_scalars) * sizeof(double), hipMemcpyDeviceToHost, s.stream));
			SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
			return SLAMPP_HIP_OK;
		});
	}
	return n_result;
}

// The factor's block structure in the CALLER's block columns (what slampp_hip_factorize fills).  Without wide columns that is
// the plan's own; where block columns wider than 8 were cut into pieces (Refine_Structure) the pieces are put together
// again: block (I, J) of the caller's columns exists where any of its pieces does.  Needs the pieces of a column next to
// each other and in order, which the caller's own order (option natural_order; what Factorize_PosDef_Blocky asks for:
// the matrix comes pre-ordered, LinearSolver_CholMod.cpp:362-544) guarantees.
namespace {

struct TCoarseFactor {
	std::vector<int32_t> perm, dim, lrow;
	std::vector<int64_t> lptr, loff; // loff[l_blocks] = number of values
	std::vector<int32_t> piece_col, piece_off; // refined column -> caller's column, scalar offset inside it
};

bool coarse_factor_structure(const slampp_hip_solver &s, TCoarseFactor &r_out, std::string &r_s_why)
{
	const Plan &P = s.plan;
	const int64_t n = int64_t(s.cumsum.size()) - 1, n_refined = int64_t(s.refined_cumsum.size()) - 1;
	r_out.piece_col.assign(size_t(n_refined), 0);
	r_out.piece_off.assign(size_t(n_refined), 0);
	{
		int64_t c = 0;
		for(int64_t p = 0; p < n_refined; ++ p) {
			while(s.refined_cumsum[p] >= s.cumsum[c + 1])
				++ c;
			r_out.piece_col[p] = int32_t(c);
			r_out.piece_off[p] = int32_t(s.refined_cumsum[p] - s.cumsum[c]);
		}
	}
	for(int64_t p = 0; p < n_refined; ++ p) {
		if(P.perm[p] != p) {
			r_s_why = "factorize: block columns wider than 8 are factored in pieces: the factor has the caller's block layout only in the caller's own order (option natural_order = 1)";
			return false;
		}
	}
	r_out.perm.resize(size_t(n));
	r_out.dim.resize(size_t(n));
	for(int64_t c = 0; c < n; ++ c) {
		r_out.perm[c] = int32_t(c);
		r_out.dim[c] = int32_t(s.cumsum[c + 1] - s.cumsum[c]);
	}
	r_out.lptr.assign(1, 0);
	r_out.lrow.clear();
	r_out.loff.clear();
	std::vector<int32_t> rows;
	int64_t n_off = 0, p = 0;
	for(int64_t c = 0; c < n; ++ c) {
		rows.clear();
		for(; p < n_refined && r_out.piece_col[p] == c; ++ p) {
			for(int64_t k = P.lptr[p]; k < P.lptr[p + 1]; ++ k)
				rows.push_back(r_out.piece_col[P.lrow[k]]);
		}
		std::sort(rows.begin(), rows.end());
		rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
		for(size_t i = 0; i < rows.size(); ++ i) { // (ascending: the diagonal block first)
			r_out.lrow.push_back(rows[i]);
			r_out.loff.push_back(n_off);
			n_off += int64_t(r_out.dim[rows[i]]) * r_out.dim[c];
		}
		r_out.lptr.push_back(int64_t(r_out.lrow.size()));
	}
	r_out.loff.push_back(n_off);
	return true;
}

} // anonymous namespace

int slampp_hip_factor_structure(const slampp_hip_solver *p_solver, int64_t *p_n_bcols, int64_t *p_l_blocks, int64_t *p_l_values,
	int32_t *p_perm, int32_t *p_dim, int64_t *p_lptr, int32_t *p_lrow, int64_t *p_loff)
{
	if(!p_solver || !p_solver->b_analyzed || p_solver->n_mode != SLAMPP_HIP_MODE_SPARSE)
		return SLAMPP_HIP_ERR_INVALID;
	const slampp_hip_solver &s = *p_solver;
	const Plan &P = s.plan;
	try {
		if(!s.b_refined) {
			if(p_n_bcols) *p_n_bcols = P.n;
			if(p_l_blocks) *p_l_blocks = int64_t(P.lrow.size());
			if(p_l_values) *p_l_values = P.loff[P.lrow.size()];
			if(p_perm) std::copy(P.perm.begin(), P.perm.end(), p_perm);
			if(p_dim) std::copy(P.dim.begin(), P.dim.end(), p_dim);
			if(p_lptr) std::copy(P.lptr.begin(), P.lptr.end(), p_lptr);
			if(p_lrow) std::copy(P.lrow.begin(), P.lrow.end(), p_lrow);
			if(p_loff) std::copy(P.loff.begin(), P.loff.begin() + P.lrow.size(), p_loff);
			return SLAMPP_HIP_OK;
		}
		TCoarseFactor t;
		std::string s_why;
		if(!coarse_factor_structure(s, t, s_why)) {
			const_cast<slampp_hip_solver*>(p_solver)->s_error = s_why;
			return SLAMPP_HIP_ERR_UNSUPPORTED;
		}
		if(p_n_bcols) *p_n_bcols = int64_t(t.dim.size());
		if(p_l_blocks) *p_l_blocks = int64_t(t.lrow.size());
		if(p_l_values) *p_l_values = t.loff.back();
		if(p_perm) std::copy(t.perm.begin(), t.perm.end(), p_perm);
		if(p_dim) std::copy(t.dim.begin(), t.dim.end(), p_dim);
		if(p_lptr) std::copy(t.lptr.begin(), t.lptr.end(), p_lptr);
		if(p_lrow) std::copy(t.lrow.begin(), t.lrow.end(), p_lrow);
		if(p_loff) std::copy(t.loff.begin(), t.loff.end() - 1, p_loff);
		return SLAMPP_HIP_OK;
	} catch(std::bad_alloc&) {
		return SLAMPP_HIP_ERR_ALLOC;
	}
}

int slampp_hip_factorize(slampp_hip_solver *p_solver, const double *p_values, double *p_factor_out)
{
	int n_result = guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factorize: analyze was not called");
		if(s.n_mode != SLAMPP_HIP_MODE_SPARSE)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "factorize: the sparse mode only");
		if(!p_values || !p_factor_out)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factorize: null pointer");
		if(s.b_refined) {
			for(size_t p = 0; p < s.plan.perm.size(); ++ p) {
				if(s.plan.perm[p] != int32_t(p))
					return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "factorize: block columns wider than 8 are factored in pieces: the factor has the caller's block layout only in the caller's own order (option natural_order = 1)");
			}
		}
		s.d_A.Alloc(size_t(s.n_values));
		s.d_rhs.Alloc(size_t(s.n_scalars));
		Upload_Values_And_Join(s, p_values);
		SLAMPP_HIP_CHECK(hipMemsetAsync(s.d_rhs.p(), 0, size_t(s.n_scalars) * sizeof(double), s.stream)); // the fused forward substitution runs on zeros
		s.Enqueue_Sparse(s.d_A.p(), s.d_rhs.p(), true, true); // (a dense top factors its columns on the matrix cores and hands them back into the block layout)
		s.b_factored = s.n_dense_dim == 0; // (with a dense top the substitutions' vectors were not brought along: no solve_again from this)
		return SLAMPP_HIP_OK;
	});
	if(n_result != SLAMPP_HIP_OK)
		return n_result;
	n_result = slampp_hip_sync(p_solver);
	if(n_result != SLAMPP_HIP_OK)
		return n_result;
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		const Plan &P = s.plan;
		const size_t n_l_values = size_t(P.loff[P.lrow.size()]);
		if(!s.b_refined) {
			SLAMPP_HIP_CHECK(hipMemcpyAsync(p_factor_out, s.d_L.p(), n_l_values * sizeof(double), hipMemcpyDeviceToHost, s.stream));
			SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
			return SLAMPP_HIP_OK;
		}
		// the pieces of the wide columns put together again: piece block (pi, pj) is a sub-block of the caller's block (I, J)
		TCoarseFactor t;
		std::string s_why;
		if(!coarse_factor_structure(s, t, s_why))
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, s_why.c_str());
		std::vector<double> pieces(n_l_values);
		SLAMPP_HIP_CHECK(hipMemcpyAsync(pieces.data(), s.d_L.p(), n_l_values * sizeof(double), hipMemcpyDeviceToHost, s.stream));
		SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
		std::fill(p_factor_out, p_factor_out + t.loff.back(), 0.0);
		for(int64_t pj = 0; pj < int64_t(P.n); ++ pj) {
			const int32_t J = t.piece_col[pj];
			const int n_col0 = t.piece_off[pj], w = P.dim[pj];
			for(int64_t k = P.lptr[pj]; k < P.lptr[pj + 1]; ++ k) {
				const int32_t pi = P.lrow[k], I = t.piece_col[pi];
				const int n_row0 = t.piece_off[pi], h = P.dim[pi], H = t.dim[I];
				const int32_t *p_first = &t.lrow[size_t(t.lptr[J])], *p_last = &t.lrow[size_t(t.lptr[J + 1])];
				const int64_t n_blk = t.lptr[J] + (std::lower_bound(p_first, p_last, I) - p_first);
				double *p_dst = p_factor_out + t.loff[size_t(n_blk)];
				const double *p_src = &pieces[size_t(P.loff[k])];
				for(int b = 0; b < w; ++ b) {
					for(int a = 0; a < h; ++ a)
						p_dst[(n_row0 + a) + size_t(n_col0 + b) * H] = p_src[a + size_t(b) * h];
				}
			}
		}
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_solve_again(slampp_hip_solver *p_solver, double *p_rhs_inout)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_factored)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "solve_again: no valid factorization");
		if(s.n_mode != SLAMPP_HIP_MODE_SPARSE)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "solve_again: only the sparse path keeps its factor");
		if(!p_rhs_inout)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "solve_again: null pointer");
		s.d_rhs.Alloc(size_t(s.n_scalars));
		SLAMPP_HIP_CHECK(hipMemcpyAsync(s.d_rhs.p(), p_rhs_inout, size_t(s.n_scalars) * sizeof(double), hipMemcpyHostToDevice, s.stream));
		s.Enqueue_Sparse(0, s.d_rhs.p(), false);
		SLAMPP_HIP_CHECK(hipMemcpyAsync(p_rhs_inout, s.d_rhs.p(), size_t(s.n_scalars) * sizeof(double), hipMemcpyDeviceToHost, s.stream));
		SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_get_stats(const slampp_hip_solver *p_solver, slampp_hip_stats *p_stats)
{
	if(!p_solver || !p_stats)
		return SLAMPP_HIP_ERR_INVALID;
	const slampp_hip_solver &s = *p_solver;
	memset(p_stats, 0, sizeof(*p_stats));
	if(!s.b_has_structure)
		return SLAMPP_HIP_ERR_INVALID;
	p_stats->n_bcols = int64_t(s.cumsum.size()) - 1;
	p_stats->n_blocks_upper = int64_t(s.brow.size());
	p_stats->n_scalars = s.n_scalars;
	if(s.b_analyzed && s.n_mode == SLAMPP_HIP_MODE_SPARSE) {
		const Plan &P = s.plan;
		p_stats->nnz_upper = P.nnz_upper;
		p_stats->l_blocks = int64_t(P.lrow.size());
		p_stats->l_nnz = P.l_nnz;
		p_stats->factor_flops = P.factor_flops;
		p_stats->solve_flops = 4.0 * double(P.l_nnz);
		p_stats->n_stages = int64_t(P.stage_ptr.size()) - 1;
		p_stats->n_tasks = int64_t(P.task_ptr.size()) - 1;
		p_stats->etree_height = P.etree_height;
		p_stats->n_update_pairs = int64_t(P.pa.size());
		p_stats->n_bottom_stages = s.n_bottom_stages;
		p_stats->schur_dim = P.dense_dim; // sparse path: dimension of the dense top (0 = none)
	} else if(s.b_analyzed && s.p_schur)
		schur_fill_stats(s.p_schur, *p_stats);
	p_stats->device_bytes = int64_t(s.n_Device_Bytes());
	if(s.b_analyzed && s.b_group_active)
		group_fill_stats(*s.p_group, *p_stats); // the members' landmark shards, summed
	return SLAMPP_HIP_OK;
}

int slampp_hip_get_reduced_stats(const slampp_hip_solver *p_solver, slampp_hip_stats *p_stats)
{
	if(!p_solver || !p_stats)
		return SLAMPP_HIP_ERR_INVALID;
	memset(p_stats, 0, sizeof(*p_stats));
	if(p_solver->b_group_active && p_solver->p_group)
		return slampp_hip_get_reduced_stats(group_member(*p_solver->p_group, 0), p_stats);
	if(!p_solver->b_analyzed || p_solver->n_mode != SLAMPP_HIP_MODE_SCHUR)
		return SLAMPP_HIP_ERR_INVALID;
	if(!schur_reduced_stats(p_solver->p_schur, *p_stats))
		memset(p_stats, 0, sizeof(*p_stats)); // dense reduced system (or none yet): all zero
	return SLAMPP_HIP_OK;
}

int slampp_hip_get_profile(slampp_hip_solver *p_solver, slampp_hip_phase_time *p_phases, int n_max_phases,
	int *p_phase_num, int b_reset)
{
	if(!p_solver || !p_phase_num)
		return SLAMPP_HIP_ERR_INVALID;
	if(p_solver->b_group_active) // the phases of member 0 (the primary: the one that also adds the camera blocks)
		return slampp_hip_get_profile(group_member(*p_solver->p_group, 0), p_phases, n_max_phases, p_phase_num, b_reset);
	slampp_hip_solver &s = *p_solver;
	*p_phase_num = int(s.phase_names.size());
	for(int i = 0; i < *p_phase_num && i < n_max_phases && p_phases; ++ i) {
		memset(p_phases[i].name, 0, sizeof(p_phases[i].name));
		strncpy(p_phases[i].name, s.phase_names[i].c_str(), sizeof(p_phases[i].name) - 1);
		p_phases[i].n_count = s.phase_count[i];
		p_phases[i].f_total_ms = s.phase_ms[i];
	}
	if(b_reset) {
		std::fill(s.phase_ms.begin(), s.phase_ms.end(), 0.0);
		std::fill(s.phase_count.begin(), s.phase_count.end(), int64_t(0));
	}
	return SLAMPP_HIP_OK;
}

int slampp_hip_set_allreduce(slampp_hip_solver *p_solver, slampp_hip_allreduce_fn p_fn, void *p_context)
{
	if(!p_solver)
		return SLAMPP_HIP_ERR_INVALID;
	if(!p_solver->group_devices.empty() && p_fn)
		return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_allreduce: a handle over several devices exchanges inside the library");
	p_solver->p_allreduce = p_fn;
	p_solver->p_allreduce_context = p_context;
	return SLAMPP_HIP_OK;
}

static void Fill_PlanView(const Plan &P, slampp_hip_plan_view *v)
{
	v->n_bcols = P.n;
	v->l_blocks = int64_t(P.lrow.size());
	v->n_pairs = int64_t(P.pa.size());
	v->n_row_entries = int64_t(P.rblk.size());
	v->n_stages = int64_t(P.stage_ptr.size()) - 1;
	v->n_tasks = int64_t(P.task_ptr.size()) - 1;
	v->n_task_cols = int64_t(P.task_cols.size());
	v->l_values = P.loff.back();
#define COPY_OUT(dst, src) do { if(dst) memcpy(dst, (src).data(), (src).size() * sizeof((src)[0])); } while(0)
	COPY_OUT(v->p_perm, P.perm);
	COPY_OUT(v->p_dim, P.dim);
	COPY_OUT(v->p_lptr, P.lptr);
	COPY_OUT(v->p_lrow, P.lrow);
	if(v->p_loff)
		memcpy(v->p_loff, P.loff.data(), P.lrow.size() * sizeof(int64_t));
	COPY_OUT(v->p_asrc, P.asrc);
	COPY_OUT(v->p_atrans, P.atrans);
	COPY_OUT(v->p_pptr, P.pptr);
	COPY_OUT(v->p_pa, P.pa);
	COPY_OUT(v->p_pb, P.pb);
	COPY_OUT(v->p_rptr, P.rptr);
	COPY_OUT(v->p_rblk, P.rblk);
	COPY_OUT(v->p_stage_ptr, P.stage_ptr);
	COPY_OUT(v->p_task_ptr, P.task_ptr);
	COPY_OUT(v->p_task_cols, P.task_cols);
	COPY_OUT(v->p_dense_pos, P.dense_pos);
	v->dense_dim = P.dense_dim;
#undef COPY_OUT
}

int slampp_hip_assembly_create(slampp_hip_solver *p_solver, slampp_hip_assembly **pp_assembly, int64_t n_edges,
	const int64_t *p_vertex0, const int64_t *p_vertex1, int n_residual_dim)
{
	if(!pp_assembly)
		return SLAMPP_HIP_ERR_INVALID;
	*pp_assembly = 0;
	return guarded(p_solver, [&]() -> int {
		CAssemblyState *p_state = assembly_setup(*p_solver, n_edges, p_vertex0, p_vertex1, n_residual_dim);
		slampp_hip_assembly *p = new(std::nothrow) slampp_hip_assembly;
		if(!p) {
			assembly_destroy(p_state);
			throw std::bad_alloc();
		}
		p->p_solver = p_solver;
		p->p_state = p_state;
		p->b_stale = false;
		try {
			p_solver->assemblies.push_back(p);
		} catch(...) {
			assembly_destroy(p_state);
			delete p;
			throw;
		}
		*pp_assembly = p;
		return SLAMPP_HIP_OK;
	});
}

void slampp_hip_assembly_destroy(slampp_hip_assembly *p_assembly)
{
	if(!p_assembly)
		return;
	if(slampp_hip_solver *p_solver = p_assembly->p_solver) {
		(void)hipSetDevice(p_solver->n_device);
		(void)hipStreamSynchronize(p_solver->stream);
		assembly_destroy(p_assembly->p_state);
		std::vector<slampp_hip_assembly*> &r_list = p_solver->assemblies;
		r_list.erase(std::remove(r_list.begin(), r_list.end(), p_assembly), r_list.end());
	}
	delete p_assembly;
}

int slampp_hip_assemble_device_async(slampp_hip_assembly *p_assembly, const double *p_J0_dev, const double *p_J1_dev,
	const double *p_sigma_inv_dev, const double *p_error_dev, const double *p_weight_dev, int64_t n_unary_vertex,
	const double *p_unary_factor, const double *p_unary_error, double *p_values_dev, double *p_eta_dev, int b_accumulate)
{
	if(!p_assembly || !p_assembly->p_solver)
		return SLAMPP_HIP_ERR_INVALID; // the solver it was created from is gone
	return guarded(p_assembly->p_solver, [&]() -> int {
		if(p_assembly->b_stale)
			throw std::invalid_argument("assemble: set_structure was called after this assembly was created");
		assembly_enqueue(*p_assembly->p_state, p_J0_dev, p_J1_dev, p_sigma_inv_dev, p_error_dev, p_weight_dev,
			n_unary_vertex, p_unary_factor, p_unary_error, p_values_dev, p_eta_dev, b_accumulate);
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_assemble_sets_device_async(const slampp_hip_edge_set *p_sets, int n_sets, int64_t n_unary_vertex,
	const double *p_unary_factor, const double *p_unary_error, double *p_values_dev, double *p_eta_dev, int b_accumulate)
{
	if(!p_sets || n_sets < 1 || !p_sets[0].p_assembly || !p_sets[0].p_assembly->p_solver)
		return SLAMPP_HIP_ERR_INVALID;
	slampp_hip_solver *p_solver = p_sets[0].p_assembly->p_solver;
	return guarded(p_solver, [&]() -> int {
		for(int i = 0; i < n_sets; ++ i) {
			if(!p_sets[i].p_assembly || p_sets[i].p_assembly->p_solver != p_solver)
				throw std::invalid_argument("assemble_sets: the edge sets belong to different solvers (or one was destroyed)");
			if(p_sets[i].p_assembly->b_stale)
				throw std::invalid_argument("assemble_sets: set_structure was called after an assembly was created");
		}
		if(!p_values_dev || !p_eta_dev)
			throw std::invalid_argument("assemble_sets: null device pointer");
		slampp_hip_solver &s = *p_solver;
		if(!b_accumulate) { // a block of Lambda may receive edges of one type only: everything starts from zero, every set adds
			SLAMPP_HIP_CHECK(hipMemsetAsync(p_values_dev, 0, size_t(s.n_values) * sizeof(double), s.stream));
			SLAMPP_HIP_CHECK(hipMemsetAsync(p_eta_dev, 0, size_t(s.n_scalars) * sizeof(double), s.stream));
		}
		for(int i = 0; i < n_sets; ++ i) {
			assembly_enqueue(*p_sets[i].p_assembly->p_state, p_sets[i].p_J0_dev, p_sets[i].p_J1_dev, p_sets[i].p_sigma_inv_dev,
				p_sets[i].p_error_dev, p_sets[i].p_weight_dev, i? -1 : n_unary_vertex, i? 0 : p_unary_factor, i? 0 : p_unary_error,
				p_values_dev, p_eta_dev, 1);
		}
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_get_plan(const slampp_hip_solver *p_solver, slampp_hip_plan_view *p_view)
{
	if(!p_solver || !p_view || !p_solver->b_analyzed || p_solver->n_mode != SLAMPP_HIP_MODE_SPARSE)
		return SLAMPP_HIP_ERR_INVALID;
	Fill_PlanView(p_solver->plan, p_view);
	return SLAMPP_HIP_OK;
}

struct slampp_hip_plan {
	Plan plan;
};

int slampp_hip_plan_create(slampp_hip_plan **pp_plan, int64_t n_bcols, const int64_t *p_bcol_cumsum,
	const int64_t *p_bcol_ptr, const int32_t *p_brow_idx, int n_leaf_size, int n_subtree_size, int n_dense_top_nb)
{
	if(!pp_plan || !p_bcol_cumsum || !p_bcol_ptr || !p_brow_idx)
		return SLAMPP_HIP_ERR_INVALID;
	*pp_plan = 0;
	try {
		slampp_hip_plan *p = new slampp_hip_plan();
		PlanOptions opt;
		if(n_leaf_size > 0)
			opt.leaf_size = n_leaf_size;
		if(n_subtree_size > 0)
			opt.subtree_size = n_subtree_size;
		if(n_dense_top_nb >= 0) {
			opt.dense_top_nb = n_dense_top_nb;
			opt.dense_top_auto = false;
		}
		if(!build_plan(n_bcols, p_bcol_cumsum, p_bcol_ptr, p_brow_idx, opt, p->plan).empty()) {
			delete p;
			return SLAMPP_HIP_ERR_INVALID;
		}
		*pp_plan = p;
		return SLAMPP_HIP_OK;
	} catch(std::bad_alloc&) {
		return SLAMPP_HIP_ERR_ALLOC;
	}
}

int slampp_hip_plan_get(const slampp_hip_plan *p_plan, slampp_hip_plan_view *p_view, slampp_hip_stats *p_stats)
{
	if(!p_plan || !p_view)
		return SLAMPP_HIP_ERR_INVALID;
	Fill_PlanView(p_plan->plan, p_view);
	if(p_stats) {
		const Plan &P = p_plan->plan;
		memset(p_stats, 0, sizeof(*p_stats));
		p_stats->n_bcols = P.n;
		p_stats->nnz_upper = P.nnz_upper;
		p_stats->l_blocks = int64_t(P.lrow.size());
		p_stats->l_nnz = P.l_nnz;
		p_stats->factor_flops = P.factor_flops;
		p_stats->solve_flops = 4.0 * double(P.l_nnz);
		p_stats->n_stages = int64_t(P.stage_ptr.size()) - 1;
		p_stats->n_tasks = int64_t(P.task_ptr.size()) - 1;
		p_stats->etree_height = P.etree_height;
		p_stats->n_update_pairs = int64_t(P.pa.size());
		p_stats->schur_dim = P.dense_dim;
	}
	return SLAMPP_HIP_OK;
}

void slampp_hip_plan_destroy(slampp_hip_plan *p_plan)
{
	delete p_plan;
}

} // extern "C"
