// capi.hip -- C ABI of libslampp_hip.so (include/slampp_hip.h): argument checks, error mapping, orchestration
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

#include "preload.h"
#include <atomic>

using namespace slampp;

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------

namespace {

// runs f, maps exceptions to status codes, records the message
template <class F>
int guarded(slampp_hip_solver *p, F f, bool b_join_bringup = true /* false: the entry point waits for the handle's streams itself, or needs none */)
{
	if(!p)
		return SLAMPP_HIP_ERR_INVALID;
	try {
		if(b_join_bringup)
			p->Join_Bringup();
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

// What slampp_hip_create used to do before it returned (round 6; solver.h: t_bringup), and what the first solve of a process
// used to find out it had to wait for: the second stream and the first pinned copies on a thread of their own beside the
// first stream, the first pageable copy and the code objects of the solve paths.
static void device_bringup(slampp_hip_solver *p)
{
	if(hipSetDevice(p->n_device) != hipSuccess) {
		p->n_bringup_status = SLAMPP_HIP_ERR_DEVICE;
		return;
	}
	static std::atomic<uint64_t> n_devices_warm(0); // a process pays the first uses once per device
	const uint64_t n_device_bit = uint64_t(1) << (p->n_device & 63);
	const bool b_first = !(n_devices_warm.fetch_or(n_device_bit) & n_device_bit) && !dev_knob_set("SLAMPP_HIP_DEV_NO_WARMUP"); // (development aid, plan.h)
	int n_copy_status = SLAMPP_HIP_OK;
	auto Copy_Side = [p, b_first, &n_copy_status]() {
		if(hipSetDevice(p->n_device) != hipSuccess ||
		   hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) != hipSuccess ||
		   hipEventCreateWithFlags(&p->copy_done, hipEventDisableTiming) != hipSuccess ||
		   hipHostMalloc((void**)&p->p_host_flag, sizeof(int), hipHostMallocDefault) != hipSuccess) {
			n_copy_status = SLAMPP_HIP_ERR_DEVICE;
			return;
		}
		if(b_first) { // the first copies out of and into pinned memory (best effort: a failure here is the first real copy's to report)
			void *p_dev = 0;
			if(hipMalloc(&p_dev, 256) == hipSuccess) {
				*p->p_host_flag = 0;
				(void)hipMemcpyAsync(p_dev, p->p_host_flag, sizeof(int), hipMemcpyHostToDevice, p->copy_stream);
				(void)hipMemcpyAsync(p->p_host_flag, p_dev, sizeof(int), hipMemcpyDeviceToHost, p->copy_stream);
				(void)hipStreamSynchronize(p->copy_stream);
				(void)hipFree(p_dev);
			}
		}
	};
	std::thread t_copy_side;
	try {
		t_copy_side = std::thread(Copy_Side);
	} catch(std::exception&) {
		Copy_Side();
	}
	const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
	const double t0 = wall_ms();
	int n_status = (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess)? SLAMPP_HIP_OK : SLAMPP_HIP_ERR_DEVICE;
	const double t1 = wall_ms();
	if(n_status == SLAMPP_HIP_OK && b_first) {
		void *p_dev = 0;
		if(hipMalloc(&p_dev, 65536) == hipSuccess) {
			std::vector<char> pageable(65536, 0); // the analysis uploads its tables out of pageable vectors
			(void)hipMemcpyAsync(p_dev, pageable.data(), pageable.size(), hipMemcpyHostToDevice, p->stream);
			(void)hipStreamSynchronize(p->stream);
			preload_device_code(p->stream);
			(void)hipStreamSynchronize(p->stream);
			(void)hipFree(p_dev);
		}
	}
	const double t2 = wall_ms();
	if(t_copy_side.joinable())
		t_copy_side.join();
	if(b_timing)
		fprintf(stderr, "[bring-up] stream %.2f ms, first pageable copy + code objects %.2f ms, copy stream and pinned copies waited for %.2f ms%s\n",
			t1 - t0, t2 - t1, wall_ms() - t2, b_first? "" : " (not the process's first handle on the device)");
	p->n_bringup_status = (n_status != SLAMPP_HIP_OK)? n_status : n_copy_status;
}

int slampp_hip_create(slampp_hip_solver **pp_solver, int device_id)
{
	(void)dev_knobs_refresh(); // (the development switch, plan.h: read on the cold path, remembered for the warm one)
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
	{
		// The C library decides by size whether a block comes out of its heap or is a mapping of its own, and it moves that size
		// up (and with it the amount of free heap it keeps) when the application frees a large mapped block -- up to 32 MB.  A
		// process that has never done so gets every work array of the analysis above 128 KB as a fresh mapping, page-faulted in
		// and torn down again: set_structure + analyze at 100k poses takes 38 - 44 ms in a plain C process and 29 - 33 ms in one
		// whose allocator has seen such a block (Python after numpy; MALLOC_MMAP_THRESHOLD_ / _TRIM_THRESHOLD_ / _TOP_PAD_
		// in the environment: tools/micro/analyze_c.c).  The first handle of a process frees one such block: the allocator's own
		// adaptation, triggered once, nothing set behind the application's back (SLAMPP_HIP_DEV_NO_MALLOC_NUDGE: off).
		static std::atomic<bool> b_nudged(false);
		if(!b_nudged.exchange(true) && !dev_knob_set("SLAMPP_HIP_DEV_NO_MALLOC_NUDGE")) {
			void *p_block = malloc((size_t(32) << 20) - 65536);
			if(p_block)
				*(volatile char*)p_block = 0; // (the pair must not be optimized away)
			free(p_block);
			// ... and the heap the analysis' remaining vectors (the planner's, the header's) will come out of now grows by 4 KB
			// page faults: with glibc >= 2.35 GLIBC_TUNABLES=glibc.malloc.hugetlb=1 makes it ask for huge pages and takes the first
			// call through the header at 100k poses from 50.7 to 42.6 - 44.6 ms, but that is read at process start.  The same for the
			// part of the heap this process is about to use: two blocks just under the (now 32 MB) mapping threshold come out of the
			// heap, their range is offered to the kernel as huge pages, and they are freed again -- 60 MB of untouched heap that
			// stays (the allocator gives free heap back from 64 MB on), on huge pages once something is written there.
			if(!dev_knob_set("SLAMPP_HIP_DEV_NO_HEAP_HUGE_PAGES")) {
				const size_t n_block = (size_t(30) << 20);
				void *p_a = malloc(n_block), *p_b = malloc(n_block);
				void *p_blocks[2] = {p_a, p_b};
				for(int i = 0; i < 2; ++ i) {
					if(!p_blocks[i])
						continue;
					const uintptr_t n_huge = uintptr_t(2) << 20;
					const uintptr_t n_begin = (uintptr_t(p_blocks[i]) + n_huge - 1) / n_huge * n_huge, n_end = (uintptr_t(p_blocks[i]) + n_block) / n_huge * n_huge;
					if(n_end > n_begin)
						(void)madvise((void*)n_begin, size_t(n_end - n_begin), MADV_HUGEPAGE); // (a hint on memory that is ours right now; refused or ignored where huge pages are off)
				}
				free(p_b);
				free(p_a);
			}
		}
	}
	// the streams, and what a process pays at the first use of each of the runtime's parts, on a thread beside the caller's
	// next steps (solver.h: t_bringup); a failure is reported by the first entry point that needs the streams
	if(dev_knob_set("SLAMPP_HIP_DEV_NO_BRINGUP_THREAD")) // (development aid, plan.h: everything before slampp_hip_create returns)
		device_bringup(p);
	else {
		try {
			p->t_bringup = std::thread(device_bringup, p);
		} catch(std::exception&) {
			device_bringup(p);
		}
	}
	*pp_solver = p;
	return SLAMPP_HIP_OK;
}

int slampp_hip_create_multi(slampp_hip_solver **pp_solver, const int *p_device_ids, int n_devices)
{
	(void)dev_knobs_refresh(); // (the development switch, plan.h: read on the cold path, remembered for the warm one)
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
		if(p_solver->n_Join_Bringup() == SLAMPP_HIP_OK)
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
	{
		// The options of include/slampp_hip.h are the interface.  The rest are development options -- alternatives the
		// defaults were measured against, test hooks --: they keep kernels and plan branches reachable for A/B timing and
		// for the parity tests of those branches, and are refused unless the process runs with SLAMPP_HIP_DEV=1 (plan.h).
		static const char *p_dev_options[] = {"nd_balance", "dense_nb", "dense_top_tiles", "simt", "simt_width", "simt_stages",
			"simt_backward", "wide_min_tasks", "panel", "panel_handup", "panel_rows", "group_fail_member", "schur_distributed"};
		for(const char *p_s_dev : p_dev_options) {
			if(s == p_s_dev && !dev_knobs_on())
				return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "development option: set SLAMPP_HIP_DEV=1 in the environment to use it");
		}
	}
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
	(void)dev_knobs_refresh(); // (the development switch, plan.h: read on the cold path, remembered for the warm one)
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
	(void)dev_knobs_refresh();
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
		s.b_has_structure = false; // (until the new one has passed its checks: a refused structure leaves the handle without one)
		s.b_analyzed = false;
		s.cumsum.assign(p_bcol_cumsum, p_bcol_cumsum + n_bcols + 1);
		s.bcol_ptr.assign(p_bcol_ptr, p_bcol_ptr + n_bcols + 1);
		// (the pointer array is checked before anything is read through it)
		if(p_bcol_ptr[n_bcols] < 0)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_structure: malformed cumsum / pointer arrays");
		// the copy of the block rows and the check of every block: on a few threads from 65 536 block columns on (round 6: one
		// thread took 5.6 ms at the Venice-like C4 and 8.2 ms at C5 -- a sixth of those analyses)
		const int64_t n_blocks_all = p_bcol_ptr[n_bcols];
		s.brow.resize(size_t(n_blocks_all));
		const int n_check_threads = (n_bcols >= 65536)? 4 : 1;
		std::vector<int64_t> values_of(size_t(n_check_threads), 0);
		std::vector<int> error_of(size_t(n_check_threads), 0); // 1: malformed arrays, 2: a block outside the upper triangle
		auto Check = [&](int t) {
			const int64_t c0 = n_bcols * t / n_check_threads, c1 = n_bcols * (t + 1) / n_check_threads;
			int64_t n_sum = 0;
			for(int64_t c = c0; c < c1; ++ c) {
				const int64_t w = s.cumsum[c + 1] - s.cumsum[c], k0 = s.bcol_ptr[c], k1 = s.bcol_ptr[c + 1];
				if(w <= 0 || k1 < k0 || k0 < 0 || k1 > n_blocks_all) {
					error_of[size_t(t)] = 1;
					return;
				}
				for(int64_t k = k0; k < k1; ++ k) {
					const int32_t r = p_brow_idx[k];
					s.brow[size_t(k)] = r;
					if(r < 0 || r > c) {
						error_of[size_t(t)] = 2;
						return;
					}
					n_sum += (s.cumsum[r + 1] - s.cumsum[r]) * w;
				}
			}
			values_of[size_t(t)] = n_sum;
		};
		{
			std::vector<std::thread> threads;
			for(int t = 1; t < n_check_threads; ++ t)
				threads.emplace_back(Check, t);
			Check(0);
			for(size_t t = 0; t < threads.size(); ++ t)
				threads[t].join();
		}
		int64_t n_values = 0;
		for(int t = 0; t < n_check_threads; ++ t) {
			if(error_of[size_t(t)] == 1)
				return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_structure: malformed cumsum / pointer arrays");
			if(error_of[size_t(t)] == 2)
				return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "set_structure: block outside the upper triangle");
			n_values += values_of[size_t(t)];
		}
		s.n_values = n_values;
		s.n_scalars = s.cumsum[n_bcols];
		s.b_has_structure = true;
		s.b_analyzed = false;
		s.b_factored = false;
		s.b_damp_valid = false;
		s.n_uploaded = 0;
		return SLAMPP_HIP_OK;
	}, false); // (host arrays only)
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
	(void)dev_knobs_refresh();
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_has_structure)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "analyze: set_structure was not called");
		if(n_mode != SLAMPP_HIP_MODE_SPARSE && n_mode != SLAMPP_HIP_MODE_SCHUR)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "analyze: unknown mode");
		if(s.b_analyzed || s.b_factored || s.n_uploaded || s.b_staging_ever) // (not a handle fresh from slampp_hip_create: its streams may still be coming up, and there is nothing on them or on the device)
			s.Join_Bringup();
		const bool b_fresh = s.t_bringup.joinable(); // (only this thread and the threads it starts below join)
		if(!b_fresh) {
			SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
			if(s.copy_stream)
				SLAMPP_HIP_CHECK(hipStreamSynchronize(s.copy_stream));
			CKeepDeviceMemory t_keep; // (a re-analysis: the arrays cease to exist, their memory waits for the new plan's)
			s.Free_Device();
		}
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
		// (a thread only where the staging is worth one: below 8 MB of values it is pinned on this thread, by whoever asks for it)
		if(s.n_staging_ahead && s.group_devices.empty() && s.n_values >= (int64_t(1) << 20) && !dev_knob_set("SLAMPP_HIP_DEV_NO_STAGING_AHEAD")) { // (the variable: a development aid, plan.h)
			t_staging_thread.t = std::thread([&s, &p_staging_error]() {
				try {
					s.Join_Bringup();
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
				if(!s.group_devices.empty())
					s.Join_Bringup(); // (the members exchange with this handle's stream)
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
		s.Join_Bringup(); // (every path above has: the handle's state after analyze does not depend on which)
		if(!s.t_discard.joinable())
			host_pool_release(); // (no thread is freeing this analysis' arrays: what its work arrays left mapped goes back now -- solver.h)
		s.b_analyzed = true;
		return SLAMPP_HIP_OK;
	}, false);
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

// K value sets of the analyzed structure, factored and solved by the same launches (blockIdx.y = member): the chain of
// dependent launches is as long as for one system, every launch K times as wide.  Stands where the reference's LM loop
// re-damps and re-solves one value after the other (NonlinearSolver_Lambda_LM.h:967-1001, 1660-1676) and for SURVEY.md
// section 8(e)'s "replicas only (multiple independent problems / damping values per GPU)" of the pose-graph path.
int slampp_hip_factor_solve_batch_device_async(slampp_hip_solver *p_solver, int n_batch, const double *p_values_dev,
	int64_t n_values_stride, double *p_rhs_inout_dev, int64_t n_rhs_stride)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!s.b_analyzed)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve_batch: analyze was not called");
		if(s.b_group_active || s.n_mode != SLAMPP_HIP_MODE_SPARSE)
			return fail(p_solver, SLAMPP_HIP_ERR_UNSUPPORTED, "factor_solve_batch: the sparse block path of a one-device handle only");
		if(!p_values_dev || !p_rhs_inout_dev || n_batch < 1 || n_batch > SLAMPP_HIP_MAX_BATCH)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve_batch: null pointer, or a batch of fewer than 1 / more than SLAMPP_HIP_MAX_BATCH members");
		if(n_batch > 1 && (n_values_stride < s.n_values || n_rhs_stride < s.n_scalars))
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "factor_solve_batch: the members' values / vectors overlap (stride below their length)");
		const Plan &P = s.plan;
		if(!s.d_batch_flag.p() || s.d_batch_flag.n() < size_t(SLAMPP_HIP_MAX_BATCH)) {
			s.d_batch_flag.Alloc(SLAMPP_HIP_MAX_BATCH);
			SLAMPP_HIP_CHECK(hipMemsetAsync(s.d_batch_flag.p(), 0, SLAMPP_HIP_MAX_BATCH * sizeof(int), s.stream));
		}
		if(!s.p_host_batch_flag)
			SLAMPP_HIP_CHECK(hipHostMalloc((void**)&s.p_host_batch_flag, SLAMPP_HIP_MAX_BATCH * sizeof(int), hipHostMallocDefault));
		s.n_batch_pending = std::max(s.n_batch_pending, n_batch);
		int *p_flag_before = s.p_flag_shared;
		const bool b_linv_valid_before = s.b_leaf_linv_valid;
		// the wide launches need what the kernels of one member assume of their bases (16-byte loads where the block dimension
		// is even) to hold for every member, and take neither a dense top (one dense matrix per handle) nor regrouped values
		const bool b_aligned = ((reinterpret_cast<uintptr_t>(p_values_dev) | reinterpret_cast<uintptr_t>(p_rhs_inout_dev)) & 15) == 0 &&
			(n_values_stride | n_rhs_stride) % 2 == 0;
		if(n_batch == 1 || s.n_dense_dim || s.b_refined || !b_aligned) {
			// one member after the other, each answering into its own flag (the factor of a member is overwritten by the next:
			// its solution has been written by then, the stream orders them)
			for(int k = 0; k < n_batch; ++ k) {
				s.p_flag_shared = s.d_batch_flag.p() + k;
				s.Enqueue_Sparse(p_values_dev + int64_t(k) * n_values_stride, p_rhs_inout_dev + int64_t(k) * n_rhs_stride, true);
			}
			s.p_flag_shared = p_flag_before;
			// The members went through the handle's own factor arrays (and its dense top), one after the other.  A batch of one is
			// an ordinary factorization that answers through the batch flags: its factor is the handle's, valid unless sync_batch
			// finds the member not positive definite.  After more than one member the handle holds the last member's factor where
			// the header promises "not touched": it is declared gone -- slampp_hip_solve_again and the covariance calls refuse
			// until the next factorization -- rather than passed off as the handle's own (advisor, round 5).
			s.b_factored = n_batch == 1;
			s.n_batch_owner_member = (n_batch == 1)? 0 : -1;
			if(n_batch > 1)
				schur_invalidate_previous(s.p_schur);
			return SLAMPP_HIP_OK;
		}
		auto Round = [](size_t n) { return (n + 31) / 32 * 32; }; // (256 bytes: every member's base is aligned like the first)
		TBatch t = {n_batch, n_values_stride, int64_t(Round(s.d_L.n())), int64_t(Round(s.d_Linv.n())), n_rhs_stride, int64_t(Round(s.d_w.n())),
			int64_t(Round(s.d_handup.n()))};
		s.d_batch_L.Alloc(size_t(t.l) * n_batch);
		s.d_batch_Linv.Alloc(size_t(t.linv) * n_batch);
		s.d_batch_w.Alloc(size_t(t.w) * n_batch);
		s.d_batch_handup.Alloc(size_t(t.h) * n_batch);
		(void)P;
		// the members' arrays stand in for the handle's own while the launches are enqueued (the kernels get pointers, not
		// the arrays); the handle's own factor -- what solve_again and the covariances work from -- stays what it was
		s.d_L.Swap(s.d_batch_L); s.d_Linv.Swap(s.d_batch_Linv); s.d_w.Swap(s.d_batch_w); s.d_handup.Swap(s.d_batch_handup);
		s.p_flag_shared = s.d_batch_flag.p();
		s.t_batch = t;
		try {
			s.Enqueue_Sparse(p_values_dev, p_rhs_inout_dev, true);
		} catch(...) {
			s.t_batch = t_No_Batch();
			s.p_flag_shared = p_flag_before;
			s.d_L.Swap(s.d_batch_L); s.d_Linv.Swap(s.d_batch_Linv); s.d_w.Swap(s.d_batch_w); s.d_handup.Swap(s.d_batch_handup);
			s.b_leaf_linv_valid = b_linv_valid_before;
			throw;
		}
		s.t_batch = t_No_Batch();
		s.p_flag_shared = p_flag_before;
		s.d_L.Swap(s.d_batch_L); s.d_Linv.Swap(s.d_batch_Linv); s.d_w.Swap(s.d_batch_w); s.d_handup.Swap(s.d_batch_handup);
		s.b_leaf_linv_valid = b_linv_valid_before;
		return SLAMPP_HIP_OK;
	});
}

int slampp_hip_sync_batch(slampp_hip_solver *p_solver, int *p_status, int n_batch)
{
	return guarded(p_solver, [&]() -> int {
		slampp_hip_solver &s = *p_solver;
		if(!p_status || n_batch < 1 || n_batch > SLAMPP_HIP_MAX_BATCH)
			return fail(p_solver, SLAMPP_HIP_ERR_INVALID, "sync_batch: null pointer or bad member count");
		const int n = std::max(n_batch, s.n_batch_pending);
		if(s.d_batch_flag.p() && s.p_host_batch_flag)
			SLAMPP_HIP_CHECK(hipMemcpyAsync(s.p_host_batch_flag, s.d_batch_flag.p(), size_t(n) * sizeof(int), hipMemcpyDeviceToHost, s.stream));
		SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
		s.Phase_Collect();
		bool b_any = false;
		for(int k = 0; k < n_batch; ++ k) {
			p_status[k] = (s.p_host_batch_flag && s.d_batch_flag.p() && s.p_host_batch_flag[k])? SLAMPP_HIP_NOT_POSDEF : SLAMPP_HIP_OK;
			b_any = b_any || p_status[k] != SLAMPP_HIP_OK;
		}
		if(s.n_batch_owner_member >= 0 && s.p_host_batch_flag && s.d_batch_flag.p() && s.p_host_batch_flag[s.n_batch_owner_member]) {
			s.b_factored = false; // the member whose factor the handle kept failed: there is nothing to solve again from
			schur_invalidate_previous(s.p_schur);
		}
		s.n_batch_owner_member = -1;
		if(b_any || s.n_batch_pending > n_batch) // (answered for: the next batch starts clean)
			SLAMPP_HIP_CHECK(hipMemsetAsync(s.d_batch_flag.p(), 0, SLAMPP_HIP_MAX_BATCH * sizeof(int), s.stream));
		s.n_batch_pending = 0;
		return SLAMPP_HIP_OK;
	});
}

void *slampp_hip_stream(slampp_hip_solver *p_solver)
{
	return (p_solver && p_solver->n_Join_Bringup() == SLAMPP_HIP_OK)? (void*)p_solver->stream : 0;
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
	if(s.b_pin_values_deferred || s.b_pin_rhs_deferred) {
		try {
			s.Register_Staging_Later(); // the first answer is out: the staging it went through is pinned behind it
		} catch(std::exception&) {
			// (no thread to be had: the staging stays pageable)
		}
	}
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
		s.Join_Staging_Registration(); // (a first staging being pinned behind the first answer: solver.h)
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

int slampp_hip_get_profile_reference_names(slampp_hip_solver *p_solver, slampp_hip_phase_time *p_phases, int n_max_phases,
	int *p_phase_num)
{
	if(!p_solver || !p_phase_num)
		return SLAMPP_HIP_ERR_INVALID;
	if(p_solver->b_group_active)
		return slampp_hip_get_profile_reference_names(group_member(*p_solver->p_group, 0), p_phases, n_max_phases, p_phase_num);
	const slampp_hip_solver &s = *p_solver;
	auto Sum = [&s](std::initializer_list<const char*> names, int64_t &r_n_count) {
		double f_ms = 0;
		r_n_count = 0;
		for(const char *p_s_name : names) {
			for(size_t i = 0; i < s.phase_names.size(); ++ i) {
				if(s.phase_names[i] == p_s_name) {
					f_ms += s.phase_ms[i];
					r_n_count = std::max(r_n_count, s.phase_count[i]);
				}
			}
		}
		return f_ms;
	};
	struct TLine { const char *p_s_name; double f_ms; int64_t n_count; };
	std::vector<TLine> lines;
	int64_t n = 0;
	if(s.n_mode == SLAMPP_HIP_MODE_SCHUR && !s.b_schur_fallback) {
		lines.push_back(TLine{"reperm", 0, 0});
		lines.push_back(TLine{"slice", 0, 0});
		lines.push_back(TLine{"transpose", 0, 0});
		double f = Sum({"schur_init", "schur_tiles", "schur_points", "schur_gather"}, n);
		lines.push_back(TLine{"inverse + multiply + add", f, n});
		f = Sum({"schur_rhs"}, n);
		lines.push_back(TLine{"RHS prep", f, n});
		f = Sum({"reduced_sparse", "dense_chol", "dense_solve"}, n);
		lines.push_back(TLine{"cholsol", f, n});
		f = Sum({"backsubst"}, n);
		lines.push_back(TLine{"dy solve", f, n});
	} else {
		double f = Sum({"factor_leaves", "factor_rest", "factor_wide", "factor_upper", "forward"}, n);
		lines.push_back(TLine{"factorize", f, n});
		f = Sum({"backward"}, n);
		lines.push_back(TLine{"solve", f, n});
	}
	*p_phase_num = int(lines.size());
	for(int i = 0; i < *p_phase_num && i < n_max_phases && p_phases; ++ i) {
		memset(p_phases[i].name, 0, sizeof(p_phases[i].name));
		strncpy(p_phases[i].name, lines[size_t(i)].p_s_name, sizeof(p_phases[i].name) - 1);
		p_phases[i].n_count = lines[size_t(i)].n_count;
		p_phases[i].f_total_ms = lines[size_t(i)].f_ms;
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
	(void)dev_knobs_refresh();
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

