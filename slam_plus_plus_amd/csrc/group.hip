// group.hip -- one BA system on several GPUs of ONE process: the landmark shards of SURVEY.md section 8e behind a single
// handle (slampp_hip_create_multi), so that the reference's nonlinear solvers -- one thread of one process,
// /root/reference/include/slam/NonlinearSolver_Base.h:344-346,400,438; NonlinearSolver_Lambda_LM.h:1543-1552 -- reach
// every device of the node through the same CLinearSolver_* calls as they reach one.
//
//   * the shard splitter (slampp_hip_landmark_shard, host code, no GPU): every member keeps all cameras and a
//     contiguous range of landmarks balanced by observation count -- the partition the reference's own GPU code
//     relies on (C is block diagonal: src/slam/LinearSolver_Schur_GPU.cpp:2417, LinearSolver_Schur.h:1721);
//   * one internal solver handle per device, driven by one host thread per device (HIP's current device is a
//     per-thread setting): a member receives its own landmark columns of Lambda straight from the caller's pinned
//     staging over its own PCIe link, member 0 ("shard_primary") also the camera blocks and the camera part of eta;
//   * the one exchange step, a sum all-reduce of the packed blocks of S and the reduced right-hand side, through
//     the members' all-reduce callbacks: RCCL bound at run time (dlopen: ncclCommInitAll over the device list, one
//     communicator per member, ncclAllReduce on the member's own stream -- RCCL rings run over xGMI) or, where RCCL
//     is absent or two members share a device (the 1-GPU test boxes), a direct exchange through peer pointers: member
//     r sums slice r of all members' buffers, then collects the other slices -- on the xGMI mesh every link carries
//     1/P of the buffer in each phase; sums run in member order, every member ends with the same bits;
//   * dx is computed redundantly, dl shard-locally, each member writes its landmarks' part of the solution straight
//     into the caller's vector.
#include "solver.h"
#include "dense_chol.h"

#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

namespace slampp {

enum { GROUP_MAX_MEMBERS = 16 };
enum { EXCHANGE_AUTO = 0, EXCHANGE_RCCL = 1, EXCHANGE_PEER = 2 };

// ---- the shard splitter -------------------------------------------------------------------------------------------

// landmark ranges [b[r], b[r + 1]) balanced by observation count (slam_plus_plus_amd/sharding.py: shard_bounds is the
// same rule; tests/test_sharding_host.py holds the two against each other); no range is empty while there are at
// least as many landmarks as shards
void shard_bounds(int64_t n_bcols, int64_t n_cut, const int64_t *p_bcol_ptr, int n_world, std::vector<int64_t> &r_bounds)
{
	const int64_t n_pts = n_bcols - n_cut;
	std::vector<int64_t> cum(size_t(n_pts) + 1, 0);
	for(int64_t p = 0; p < n_pts; ++ p)
		cum[p + 1] = cum[p] + (p_bcol_ptr[n_cut + p + 1] - p_bcol_ptr[n_cut + p] - 1); // blocks of the column but its diagonal one
	r_bounds.assign(size_t(n_world) + 1, 0);
	r_bounds[n_world] = n_pts;
	for(int r = 1; r < n_world; ++ r) {
		const double f_target = double(cum[n_pts]) * double(r) / double(n_world);
		const int64_t n_pos = int64_t(std::lower_bound(cum.begin(), cum.end(), f_target,
			[](int64_t n_value, double f) { return double(n_value) < f; }) - cum.begin());
		r_bounds[r] = std::max(r_bounds[r - 1], std::min(n_pos, n_pts));
	}
	if(n_pts >= n_world) {
		for(int r = 1; r < n_world; ++ r)
			r_bounds[r] = std::max(r_bounds[r], r_bounds[r - 1] + 1);
		for(int r = n_world - 1; r >= 1; -- r)
			r_bounds[r] = std::min(r_bounds[r], r_bounds[r + 1] - 1);
	}
}

struct TShardStructure {
	std::vector<int64_t> cumsum, bcol_ptr;
	std::vector<int32_t> brow;
	int64_t n_camera_values;  // packed values of the camera block columns [0, n_cut)
	int64_t n_value_begin, n_value_end;   // this shard's landmark columns in the full packed values
	int64_t n_camera_scalars; // dimension of the camera part
	int64_t n_scalar_begin, n_scalar_end; // this shard's landmarks in the full right-hand side
	int64_t n_point_begin, n_point_end;
};

void landmark_shard(int64_t n_bcols, const int64_t *p_cumsum, const int64_t *p_bcol_ptr, const int32_t *p_brow, int64_t n_cut,
	int n_rank, int n_world, TShardStructure &r_out) // throw(std::bad_alloc, std::invalid_argument)
{
	if(n_cut <= 0 || n_cut >= n_bcols || n_world < 1 || n_rank < 0 || n_rank >= n_world)
		throw std::invalid_argument("landmark_shard: n_matrix_cut must split the block columns, rank must be below world");
	std::vector<int64_t> bounds;
	shard_bounds(n_bcols, n_cut, p_bcol_ptr, n_world, bounds);
	const int64_t p0 = bounds[n_rank], p1 = bounds[n_rank + 1];
	auto value_offset_of_column = [&](int64_t n_column_begin, int64_t n_column_end) {
		int64_t n_values = 0;
		for(int64_t c = n_column_begin; c < n_column_end; ++ c) {
			int64_t n_height = 0;
			for(int64_t k = p_bcol_ptr[c]; k < p_bcol_ptr[c + 1]; ++ k)
				n_height += p_cumsum[p_brow[k] + 1] - p_cumsum[p_brow[k]];
			n_values += n_height * (p_cumsum[c + 1] - p_cumsum[c]);
		}
		return n_values;
	};
	r_out.n_camera_values = value_offset_of_column(0, n_cut);
	r_out.n_value_begin = r_out.n_camera_values + value_offset_of_column(n_cut, n_cut + p0);
	r_out.n_value_end = r_out.n_value_begin + value_offset_of_column(n_cut + p0, n_cut + p1);
	r_out.n_camera_scalars = p_cumsum[n_cut];
	r_out.n_scalar_begin = p_cumsum[n_cut + p0];
	r_out.n_scalar_end = p_cumsum[n_cut + p1];
	r_out.n_point_begin = p0;
	r_out.n_point_end = p1;
	const int64_t n_own = p1 - p0, n_a_blocks = p_bcol_ptr[n_cut], k0 = p_bcol_ptr[n_cut + p0], k1 = p_bcol_ptr[n_cut + p1];
	r_out.cumsum.resize(size_t(n_cut + n_own) + 1);
	r_out.bcol_ptr.resize(size_t(n_cut + n_own) + 1);
	for(int64_t c = 0; c <= n_cut; ++ c) {
		r_out.cumsum[c] = p_cumsum[c];
		r_out.bcol_ptr[c] = p_bcol_ptr[c];
	}
	for(int64_t p = 0; p < n_own; ++ p) {
		r_out.cumsum[n_cut + p + 1] = p_cumsum[n_cut + p0 + p + 1] - p_cumsum[n_cut + p0] + p_cumsum[n_cut];
		r_out.bcol_ptr[n_cut + p + 1] = p_bcol_ptr[n_cut + p0 + p + 1] - k0 + n_a_blocks;
	}
	r_out.brow.resize(size_t(n_a_blocks + (k1 - k0)));
	std::copy(p_brow, p_brow + n_a_blocks, r_out.brow.begin());
	for(int64_t p = 0; p < n_own; ++ p) {
		for(int64_t k = p_bcol_ptr[n_cut + p0 + p]; k < p_bcol_ptr[n_cut + p0 + p + 1]; ++ k) {
			const int32_t n_row = p_brow[k]; // the landmarks' own (diagonal) blocks carry global row numbers
			if(n_row >= n_cut && n_row != n_cut + p0 + p)
				throw std::domain_error("landmark-landmark blocks present (C is not block diagonal): landmarks are not independent units");
			r_out.brow[size_t(n_a_blocks + (k - k0))] = (n_row >= n_cut)? int32_t(n_row - p0) : n_row;
		}
	}
}

// ---- RCCL, bound at run time --------------------------------------------------------------------------------------

struct CRccl {
	typedef int (*TCommInitAll)(void **pp_comms, int n_devices, const int *p_device_list);
	typedef int (*TCommDestroy)(void *p_comm);
	typedef int (*TCommAbort)(void *p_comm);
	typedef int (*TAllReduce)(const void *p_send, void *p_recv, size_t n_count, int n_data_type, int n_op, void *p_comm, hipStream_t stream);
	typedef const char *(*TGetErrorString)(int n_result);
	enum { nccl_Float64 = 8, nccl_Sum = 0 }; // rccl.h: ncclDataType_t, ncclRedOp_t

	void *p_library;
	TCommInitAll CommInitAll;
	TCommDestroy CommDestroy;
	TCommAbort CommAbort; // (may be null: an old library)
	TAllReduce AllReduce;
	TGetErrorString GetErrorString;
	std::string s_where;

	// process-wide: a loaded library, nothing else.  A Python caller has torch's RCCL in the process already (bound to the
	// same HIP runtime this library binds to): that copy first; then the system's.
	static CRccl *p_Get()
	{
		static CRccl *p_instance = [] () -> CRccl* {
			std::vector<std::pair<std::string, int> > candidates;
			if(const char *p_s_env = getenv("SLAMPP_HIP_RCCL_LIB"))
				candidates.push_back(std::make_pair(std::string(p_s_env), RTLD_NOW | RTLD_LOCAL));
			candidates.push_back(std::make_pair(std::string("librccl.so"), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD));
			candidates.push_back(std::make_pair(std::string("librccl.so.1"), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD));
			candidates.push_back(std::make_pair(std::string("librccl.so.1"), RTLD_NOW | RTLD_LOCAL));
			candidates.push_back(std::make_pair(std::string("librccl.so"), RTLD_NOW | RTLD_LOCAL));
			candidates.push_back(std::make_pair(std::string("/opt/rocm/lib/librccl.so.1"), RTLD_NOW | RTLD_LOCAL));
			for(size_t i = 0; i < candidates.size(); ++ i) {
				void *p_lib = dlopen(candidates[i].first.c_str(), candidates[i].second);
				if(!p_lib)
					continue;
				CRccl *p = new CRccl;
				p->p_library = p_lib;
				p->CommInitAll = (TCommInitAll)dlsym(p_lib, "ncclCommInitAll");
				p->CommDestroy = (TCommDestroy)dlsym(p_lib, "ncclCommDestroy");
				p->CommAbort = (TCommAbort)dlsym(p_lib, "ncclCommAbort");
				p->AllReduce = (TAllReduce)dlsym(p_lib, "ncclAllReduce");
				p->GetErrorString = (TGetErrorString)dlsym(p_lib, "ncclGetErrorString");
				p->s_where = candidates[i].first;
				if(p->CommInitAll && p->CommDestroy && p->AllReduce && p->GetErrorString)
					return p;
				delete p;
				(void)dlclose(p_lib);
			}
			return 0;
		}();
		return p_instance;
	}
};

// ---- host threads, one per member ---------------------------------------------------------------------------------

class CMemberThreads {
	std::vector<std::thread> m_threads;
	std::mutex m_mutex;
	std::condition_variable m_job_ready, m_job_done;
	std::function<int(int)> m_job;
	std::vector<int> m_results;
	std::vector<uint64_t> m_seen;
	uint64_t m_n_generation;
	int m_n_pending;
	bool m_b_quit;

public:
	CMemberThreads(const std::vector<int> &r_devices) // throw(std::bad_alloc, std::system_error)
		:m_results(r_devices.size(), 0), m_seen(r_devices.size(), 0), m_n_generation(0), m_n_pending(0), m_b_quit(false)
	{
		try {
			for(size_t i = 0; i < r_devices.size(); ++ i) {
				const int n_device = r_devices[i], n_member = int(i);
				m_threads.emplace_back([this, n_device, n_member]() {
					(void)hipSetDevice(n_device);
					for(;;) {
						std::function<int(int)> job;
						{
							std::unique_lock<std::mutex> lock(m_mutex);
							m_job_ready.wait(lock, [&]() { return m_b_quit || m_seen[n_member] != m_n_generation; });
							if(m_b_quit)
								return;
							m_seen[n_member] = m_n_generation;
							job = m_job;
						}
						int n_result;
						try {
							n_result = job(n_member);
						} catch(std::bad_alloc&) {
							n_result = SLAMPP_HIP_ERR_ALLOC;
						} catch(std::exception&) {
							n_result = SLAMPP_HIP_ERR_DEVICE;
						}
						{
							std::lock_guard<std::mutex> lock(m_mutex);
							m_results[n_member] = n_result;
							if(!-- m_n_pending)
								m_job_done.notify_all();
						}
					}
				});
			}
		} catch(...) {
			Quit();
			throw;
		}
	}

	~CMemberThreads()
	{
		Quit();
	}

	void Quit()
	{
		{
			std::lock_guard<std::mutex> lock(m_mutex);
			m_b_quit = true;
		}
		m_job_ready.notify_all();
		for(size_t i = 0; i < m_threads.size(); ++ i) {
			if(m_threads[i].joinable())
				m_threads[i].join();
		}
		m_threads.clear();
	}

	// runs job(member) on every member's thread at once, returns when all are back; the first result that is not OK
	// (an error before "not positive definite")
	int n_Run(std::function<int(int)> job)
	{
		std::unique_lock<std::mutex> lock(m_mutex);
		m_job = job;
		m_n_pending = int(m_threads.size());
		++ m_n_generation;
		m_job_ready.notify_all();
		m_job_done.wait(lock, [&]() { return m_n_pending == 0; });
		int n_result = SLAMPP_HIP_OK;
		for(size_t i = 0; i < m_results.size(); ++ i) {
			if(m_results[i] < 0 && n_result >= 0)
				n_result = m_results[i];
			else if(m_results[i] > 0 && n_result == SLAMPP_HIP_OK)
				n_result = m_results[i];
		}
		return n_result;
	}

	const std::vector<int> &r_Results() const
	{
		return m_results;
	}
};

// a barrier the members' threads meet at inside the peer exchange; a member that fails on its way there calls Abort()
// so that the others do not wait for it forever
class CAbortableBarrier {
	std::mutex m_mutex;
	std::condition_variable m_all_here;
	int m_n_members, m_n_waiting;
	uint64_t m_n_generation;
	bool m_b_aborted;

public:
	CAbortableBarrier()
		:m_n_members(1), m_n_waiting(0), m_n_generation(0), m_b_aborted(false)
	{}

	void Reset(int n_members)
	{
		std::lock_guard<std::mutex> lock(m_mutex);
		m_n_members = n_members;
		m_n_waiting = 0;
		m_b_aborted = false;
	}

	bool b_Wait() // false: aborted
	{
		std::unique_lock<std::mutex> lock(m_mutex);
		if(m_b_aborted)
			return false;
		const uint64_t n_generation = m_n_generation;
		if(++ m_n_waiting == m_n_members) {
			m_n_waiting = 0;
			++ m_n_generation;
			m_all_here.notify_all();
			return true;
		}
		m_all_here.wait(lock, [&]() { return m_b_aborted || m_n_generation != n_generation; });
		return !m_b_aborted;
	}

	void Abort()
	{
		{
			std::lock_guard<std::mutex> lock(m_mutex);
			m_b_aborted = true;
		}
		m_all_here.notify_all();
	}

	bool b_Aborted()
	{
		std::lock_guard<std::mutex> lock(m_mutex);
		return m_b_aborted;
	}
};

// ---- the direct exchange through peer pointers --------------------------------------------------------------------

struct TPeerBuffers {
	double *p[GROUP_MAX_MEMBERS];
};

// buffer[me][i] = sum over the members, in member order, for i in this member's slice
__global__ void group_reduce_slice_kernel(TPeerBuffers t_buffers, int n_members, int n_me, size_t n_begin, size_t n_end)
{
	for(size_t i = n_begin + size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_end; i += size_t(gridDim.x) * blockDim.x) {
		double f_sum = 0;
		for(int k = 0; k < n_members; ++ k)
			f_sum += t_buffers.p[k][i];
		t_buffers.p[n_me][i] = f_sum;
	}
}

// the other members' finished slices into this member's buffer
__global__ void group_gather_slices_kernel(TPeerBuffers t_buffers, int n_members, int n_me, size_t n_count)
{
	for(int k = 0; k < n_members; ++ k) {
		if(k == n_me)
			continue;
		const size_t n_begin = n_count * size_t(k) / size_t(n_members), n_end = n_count * size_t(k + 1) / size_t(n_members);
		const double *__restrict__ p_src = t_buffers.p[k];
		double *__restrict__ p_dst = t_buffers.p[n_me];
		for(size_t i = n_begin + size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_end; i += size_t(gridDim.x) * blockDim.x)
			p_dst[i] = p_src[i];
	}
}

// ---- the dense reduced camera system factored by all members together (option "schur_distributed") ------------------
// SURVEY.md section 8f rank 2, first half; the reference factors S once, serially (src/slam/LinearSolver_Schur.cpp:2314-2331)
// and falls back to a sparse solver when it does not fit (include/slam/LinearSolver_Schur.h:1844-1853).
//
// The matrix is cut into outer panels of 256 columns (4 tiles), panel b belongs to member b mod P.
//   1. reduce-scatter by panels: every member sums, for its own panels, the members' partial S -- peer reads, in member
//      order, into its own buffer;
//   2. right-looking factorization, owner computes: the owner of panel b factors it (diagonal tiles, panel solve) and
//      writes the finished columns and the inverses of their diagonal tiles into every other member's buffer; each
//      member then applies panel b to the panels it owns (K = 256 updates on the matrix cores), the owner of panel
//      b + 1 to that one first, which it factors and sends before it goes on with the others (look-ahead);
//   3. every member then holds the whole factor (its own panels and the received ones): the substitutions run as
//      before, redundantly.
// Members meet through events: "my partial S is assembled" (one per member), "panel b is in your buffer" (one per
// panel, recorded on the owner's stream, waited for on the others'); a host-side sequence number makes sure a wait is
// only enqueued after the record it refers to.  Summation and elimination orders are fixed: every member gets the same bits.

// columns of the panels this member owns: the sum over the members' buffers, rows from the column's diagonal tile down
__global__ void group_reduce_columns_kernel(TPeerBuffers t_S, int n_members, int n_me, int n_pad)
{
	const int j = blockIdx.x, n_panel = (j / int(dense_NB)) / int(dense_OUTER_TILES);
	if(n_panel % n_members != n_me)
		return;
	const size_t n_col = size_t(j) * n_pad;
	for(int i = (j / int(dense_NB)) * int(dense_NB) + threadIdx.x; i < n_pad; i += blockDim.x) {
		double f_sum = 0;
		for(int k = 0; k < n_members; ++ k)
			f_sum += t_S.p[k][n_col + i];
		t_S.p[n_me][n_col + i] = f_sum;
	}
}

// the finished tile columns [t0, t1) and the inverses of their diagonal tiles into the other members' buffers
__global__ void group_send_panel_kernel(TPeerBuffers t_S, TPeerBuffers t_inv, int n_members, int n_me, int n_pad, int t0, int t1)
{
	const int n_cols = (t1 - t0) * int(dense_NB);
	if(int(blockIdx.x) < n_cols) {
		const int j = t0 * int(dense_NB) + blockIdx.x;
		const size_t n_col = size_t(j) * n_pad;
		for(int i = t0 * int(dense_NB) + threadIdx.x; i < n_pad; i += blockDim.x) {
			const double f = t_S.p[n_me][n_col + i];
			for(int k = 0; k < n_members; ++ k) {
				if(k != n_me)
					t_S.p[k][n_col + i] = f;
			}
		}
	} else {
		const size_t n_tile = size_t(t0 + (int(blockIdx.x) - n_cols)) * dense_NB * dense_NB;
		for(int e = threadIdx.x; e < int(dense_NB) * int(dense_NB); e += blockDim.x) {
			const double f = t_inv.p[n_me][n_tile + e];
			for(int k = 0; k < n_members; ++ k) {
				if(k != n_me)
					t_inv.p[k][n_tile + e] = f;
			}
		}
	}
}

// ---- the group ----------------------------------------------------------------------------------------------------

struct CDeviceGroup;

struct TMemberContext {
	CDeviceGroup *p_group;
	int n_member;
	unsigned n_exchange_calls = 0; // this member's all-reduce calls so far (their parity picks the slot of its count)
};

struct TMemberShard {
	TShardStructure t_structure;
	double *p_values_dev, *p_rhs_dev, *p_cov_dev; // on the member's device
	size_t n_values_dev, n_rhs_dev, n_cov_dev;
	TMemberShard() :p_values_dev(0), p_rhs_dev(0), p_cov_dev(0), n_values_dev(0), n_rhs_dev(0), n_cov_dev(0) {}
};

struct CDeviceGroup {
	std::vector<int> devices;
	std::vector<slampp_hip_solver*> members;
	std::vector<TMemberContext> contexts;
	std::vector<TMemberShard> shards;
	CMemberThreads *p_threads;
	int n_active; // members that hold landmarks (all of them unless the system has fewer landmarks than the list has devices)
	const double *p_checked_staging = 0, *p_checked_staging_rhs = 0; // the front's staging as group_check_staging last saw it fit
	int n_exchange_option, n_exchange; // EXCHANGE_*: as asked for, as resolved
	std::vector<void*> comms; // RCCL communicators, one per active member
	CAbortableBarrier barrier;
	TPeerBuffers t_peer_buffers;
	size_t peer_counts[2][GROUP_MAX_MEMBERS]; // [parity of the call][member]: a fast member writing the next call's count must not overwrite what a slow one is still comparing
	// distributed dense factorization: the members' S and inverse-tile buffers, events, and how far the owners have got
	TPeerBuffers t_factor_S, t_factor_inv;
	int factor_dims[GROUP_MAX_MEMBERS][2];
	hipEvent_t ev_assembled[GROUP_MAX_MEMBERS];
	std::vector<hipEvent_t> ev_panel; // [n_outer], each created on its owner's device
	std::mutex ev_mutex;
	std::atomic<int64_t> n_panel_sequence; // solve number * 65536 + panels recorded so far in that solve
	int64_t n_factor_calls[GROUP_MAX_MEMBERS];
	std::string s_exchange_name;
	bool b_peer_access; // every member can read and write every other member's device memory (peer access enabled, or one device)
	bool b_comms_aborted = false; // the handles in comms were given to ncclCommAbort (group_break_exchange): not to be destroyed again
	std::atomic<bool> b_exchange_broken; // a collective failed after the members had agreed to enqueue it: the communicators are aborted and made anew
	std::atomic<int64_t> n_collectives_enqueued; // exchanges the members went into (all of them, or none: see group_members_agree)
	int n_fail_member; // test hook (option "group_fail_member"): this member + 1 fails on its way to the exchange

	CDeviceGroup() :p_threads(0), n_active(0), n_exchange_option(EXCHANGE_AUTO), n_exchange(EXCHANGE_PEER), n_panel_sequence(0),
		b_peer_access(false), b_exchange_broken(false), n_collectives_enqueued(0), n_fail_member(0)
	{
		memset(ev_assembled, 0, sizeof(ev_assembled));
		memset(n_factor_calls, 0, sizeof(n_factor_calls));
	}
};

static void grow_device(double *&r_p, size_t &r_n, size_t n_doubles) // on the calling thread's device; throws
{
	if(r_p && r_n >= n_doubles)
		return;
	if(r_p)
		(void)hipFree(r_p);
	r_p = 0;
	r_n = 0;
	const hipError_t e = hipMalloc((void**)&r_p, std::max<size_t>(n_doubles, 1) * sizeof(double));
	if(e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
		(void)hipGetLastError();
		r_p = 0;
		throw std::bad_alloc();
	}
	if(e != hipSuccess) {
		r_p = 0;
		throw CDeviceError(std::string("hipMalloc: ") + hipGetErrorString(e));
	}
	r_n = n_doubles;
}

// Failure agreement, the same for both exchanges: before anything that another member would wait for is enqueued (an
// ncclAllReduce parks the stream until every rank has joined; a peer reduction reads the others' buffers) the members
// meet at the host barrier.  A member that failed on its way here (allocation, upload, device error) has called
// barrier.Abort() instead: the barrier lets everybody through with `false`, and NOBODY enqueues.  group_finish() opens the
// barrier again for the next solve.
static bool group_members_agree(CDeviceGroup &g)
{
	return g.barrier.b_Wait();
}

// a collective that failed after the agreement: the other members' streams may be parked in it.  Abort every communicator
// (that is what releases them; ncclCommAbort is the call RCCL offers for use from another thread) and have the next solve
// make new ones.  The handles stay in g.comms: other members may be about to read theirs -- a call on an aborted
// communicator returns an error, a call on a null one would not return --; group_repair_exchange, where every member
// thread has joined, destroys and replaces them.
static void group_break_exchange(CDeviceGroup &g)
{
	bool b_expected = false;
	if(!g.b_exchange_broken.compare_exchange_strong(b_expected, true))
		return;
	CRccl *p_rccl = CRccl::p_Get();
	if(p_rccl && p_rccl->CommAbort) {
		for(size_t i = 0; i < g.comms.size(); ++ i) {
			if(g.comms[i])
				(void)p_rccl->CommAbort(g.comms[i]);
		}
		g.b_comms_aborted = true; // (aborted handles are released already: nothing left to destroy)
	}
	g.barrier.Abort();
}

static int group_allreduce_callback(void *p_context, double *p_dev, size_t n_count, void *p_hip_stream)
{
	TMemberContext &t = *(TMemberContext*)p_context;
	CDeviceGroup &g = *t.p_group;
	const int r = t.n_member, n_members = g.n_active;
	hipStream_t stream = (hipStream_t)p_hip_stream;
	const unsigned n_slot = (t.n_exchange_calls ++) & 1; // (the members call in step: one cannot be two calls ahead of another -- the barrier below)
	if(g.n_exchange == EXCHANGE_RCCL) {
		g.peer_counts[n_slot][r] = n_count;
		if(!group_members_agree(g) || g.b_exchange_broken) // one member is not coming: nobody enqueues
			return 1;
		for(int k = 0; k < n_members; ++ k) {
			if(g.peer_counts[n_slot][k] != n_count) {
				g.barrier.Abort();
				return 1; // the members do not agree on what they exchange
			}
		}
		CRccl *p_rccl = CRccl::p_Get();
		if(r == 0)
			++ g.n_collectives_enqueued;
		const int n_result = p_rccl->AllReduce(p_dev, p_dev, n_count, CRccl::nccl_Float64, CRccl::nccl_Sum, g.comms[r], stream);
		if(n_result != 0) {
			fprintf(stderr, "libslampp_hip: ncclAllReduce failed on member %d: %s\n", r, p_rccl->GetErrorString(n_result));
			group_break_exchange(g);
			return 1;
		}
		return 0;
	}
	// peer pointers: everybody's partial buffer complete and announced; slice r summed here; the other slices collected;
	// nobody moves on (and writes its buffer again) before everybody has read what it needs
	g.t_peer_buffers.p[r] = p_dev;
	g.peer_counts[n_slot][r] = n_count;
	if(hipStreamSynchronize(stream) != hipSuccess) {
		g.barrier.Abort();
		return 1;
	}
	if(!group_members_agree(g))
		return 1;
	for(int k = 0; k < n_members; ++ k) {
		if(g.peer_counts[n_slot][k] != n_count) {
			g.barrier.Abort();
			return 1; // the members do not agree on what they exchange
		}
	}
	if(r == 0)
		++ g.n_collectives_enqueued;
	const TPeerBuffers t_buffers = g.t_peer_buffers;
	const size_t n_begin = n_count * size_t(r) / size_t(n_members), n_end = n_count * size_t(r + 1) / size_t(n_members);
	if(n_end > n_begin) {
		const unsigned n_grid = unsigned(std::min<size_t>((n_end - n_begin + 255) / 256, 2048));
		hipLaunchKernelGGL(group_reduce_slice_kernel, dim3(n_grid), dim3(256), 0, stream, t_buffers, n_members, r, n_begin, n_end);
	}
	if(hipStreamSynchronize(stream) != hipSuccess) {
		g.barrier.Abort();
		return 1;
	}
	if(!g.barrier.b_Wait())
		return 1;
	{
		const unsigned n_grid = unsigned(std::min<size_t>((n_count / size_t(n_members) + 255) / 256 + 1, 2048));
		hipLaunchKernelGGL(group_gather_slices_kernel, dim3(n_grid), dim3(256), 0, stream, t_buffers, n_members, r, n_count);
	}
	if(hipStreamSynchronize(stream) != hipSuccess) {
		g.barrier.Abort();
		return 1;
	}
	return g.barrier.b_Wait()? 0 : 1;
}

// all-reduce + redundant factorization replaced: see the comment above group_reduce_columns_kernel
static int group_dense_factor_callback(void *p_context, double *p_S, int n_pad, int n, double *p_invdiag, int *p_flag, void *p_hip_stream)
{
	TMemberContext &t = *(TMemberContext*)p_context;
	CDeviceGroup &g = *t.p_group;
	const int r = t.n_member, P = g.n_active;
	hipStream_t stream = (hipStream_t)p_hip_stream;
	const int n_tiles = n_pad / int(dense_NB), n_outer = (n_tiles + int(dense_OUTER_TILES) - 1) / int(dense_OUTER_TILES);
	const int64_t n_solve = ++ g.n_factor_calls[r]; // (the members call in step: the same number on each of them)
	auto fail = [&g]() { g.barrier.Abort(); return 1; };
	// events: this member's "assembled" event and the events of the panels it owns (created on its own device)
	if(!g.ev_assembled[r] && hipEventCreateWithFlags(&g.ev_assembled[r], hipEventDisableTiming) != hipSuccess)
		return fail();
	{
		std::lock_guard<std::mutex> lock(g.ev_mutex); // (the vector may grow under another member's hands)
		if(int(g.ev_panel.size()) < n_outer)
			g.ev_panel.resize(size_t(n_outer), (hipEvent_t)0);
		for(int b = r; b < n_outer; b += P) {
			if(!g.ev_panel[b] && hipEventCreateWithFlags(&g.ev_panel[b], hipEventDisableTiming) != hipSuccess)
				return fail();
		}
	}
	g.t_factor_S.p[r] = p_S;
	g.t_factor_inv.p[r] = p_invdiag;
	g.factor_dims[r][0] = n_pad;
	g.factor_dims[r][1] = n;
	if(hipEventRecord(g.ev_assembled[r], stream) != hipSuccess)
		return fail();
	if(!g.barrier.b_Wait()) // everybody's pointers are in the tables, everybody's "assembled" event is recorded
		return 1;
	for(int k = 0; k < P; ++ k) {
		if(g.factor_dims[k][0] != n_pad || g.factor_dims[k][1] != n)
			return fail(); // the members do not agree on the reduced system
		if(k != r && hipStreamWaitEvent(stream, g.ev_assembled[k], 0) != hipSuccess)
			return fail();
	}
	const TPeerBuffers t_S = g.t_factor_S, t_inv = g.t_factor_inv;
	hipLaunchKernelGGL(group_reduce_columns_kernel, dim3(unsigned(n_pad)), dim3(256), 0, stream, t_S, P, r, n_pad);
	auto t0_of = [&](int b) { return b * int(dense_OUTER_TILES); };
	auto t1_of = [&](int b) { return std::min((b + 1) * int(dense_OUTER_TILES), n_tiles); };
	auto factor_and_send = [&](int b) -> bool {
		dense_factor_panel(p_S, n_pad, n, t0_of(b), t1_of(b), p_invdiag, p_flag, stream);
		if(P > 1) {
			const int n_grid = (t1_of(b) - t0_of(b)) * int(dense_NB) + (t1_of(b) - t0_of(b));
			hipLaunchKernelGGL(group_send_panel_kernel, dim3(unsigned(n_grid)), dim3(256), 0, stream, t_S, t_inv, P, r, n_pad, t0_of(b), t1_of(b));
		}
		if(hipEventRecord(g.ev_panel[b], stream) != hipSuccess)
			return false;
		g.n_panel_sequence.store(n_solve * 65536 + b + 1); // (panels are recorded in order, one owner after the other)
		return true;
	};
	bool b_ahead = false; // this member has factored the next panel already (look-ahead)
	for(int b = 0; b < n_outer; ++ b) {
		if(b % P == r) {
			if(!b_ahead && !factor_and_send(b))
				return fail();
			b_ahead = false;
		} else {
			const double f_wait_t0 = wall_ms();
			for(int n_spin = 0; g.n_panel_sequence.load() < n_solve * 65536 + b + 1; ++ n_spin) { // the owner has recorded the event: now it can be waited for
				if(g.barrier.b_Aborted())
					return 1;
				if(!(n_spin & 1023) && wall_ms() - f_wait_t0 > 120e3) // two minutes for one panel: the owner is not coming
					return fail();
				std::this_thread::yield();
			}
			if(hipStreamWaitEvent(stream, g.ev_panel[b], 0) != hipSuccess)
				return fail();
		}
		for(int c = b + 1 + ((r - (b + 1)) % P + P) % P; c < n_outer; c += P) { // the panels this member owns, beyond b
			dense_update_panels(p_S, n_pad, t0_of(b), t1_of(b), t0_of(c), t1_of(c), stream);
			if(c == b + 1) { // the next panel is this member's: out with it before the others are brought up to date
				if(!factor_and_send(c))
					return fail();
				b_ahead = true;
			}
		}
	}
	if(hipGetLastError() != hipSuccess)
		return fail();
	// nobody's buffers may be taken apart (the next solve's assembly) while a member still reads them: every member has
	// waited for every panel, which its owner sent after it had read what it needed; the last readers are the owners'
	// reduce-scatter kernels, which precede their first panel
	return 0;
}

CDeviceGroup *group_create(const int *p_device_ids, int n_devices) // throw(std::bad_alloc, std::invalid_argument, CDeviceError)
{
	if(!p_device_ids || n_devices < 1 || n_devices > GROUP_MAX_MEMBERS)
		throw std::invalid_argument("a device group takes 1 to 16 devices");
	int n_count = 0;
	if(hipGetDeviceCount(&n_count) != hipSuccess || n_count <= 0)
		throw CDeviceError("no HIP device");
	CDeviceGroup *p_group = new CDeviceGroup;
	CDeviceGroup &g = *p_group;
	try {
		g.devices.assign(p_device_ids, p_device_ids + n_devices);
		for(int i = 0; i < n_devices; ++ i) {
			if(g.devices[i] < 0 || g.devices[i] >= n_count)
				throw std::invalid_argument("device group: no such device");
		}
		g.members.assign(size_t(n_devices), (slampp_hip_solver*)0);
		g.shards.resize(size_t(n_devices));
		g.contexts.resize(size_t(n_devices));
		for(int i = 0; i < n_devices; ++ i) {
			g.contexts[i].p_group = p_group;
			g.contexts[i].n_member = i;
			const int n_result = slampp_hip_create(&g.members[i], g.devices[i]);
			if(n_result == SLAMPP_HIP_ERR_ALLOC)
				throw std::bad_alloc();
			if(n_result != SLAMPP_HIP_OK)
				throw CDeviceError("device group: cannot create a member");
		}
		for(int i = 0; i < n_devices; ++ i) { // (the members' streams came up side by side: solver.h, t_bringup; the group uses them directly)
			if(g.members[i]->n_Join_Bringup() != SLAMPP_HIP_OK)
				throw CDeviceError("device group: cannot create a member's streams");
		}
		g.p_threads = new CMemberThreads(g.devices);
	} catch(...) {
		group_destroy(p_group);
		throw;
	}
	return p_group;
}

static void group_release_exchange(CDeviceGroup &g)
{
	if(!g.comms.empty()) {
		if(CRccl *p_rccl = CRccl::p_Get()) {
			for(size_t i = 0; i < g.comms.size(); ++ i) {
				if(g.comms[i] && !g.b_comms_aborted) // (ncclCommAbort has freed an aborted one)
					(void)p_rccl->CommDestroy(g.comms[i]);
			}
		}
		g.comms.clear();
		g.b_comms_aborted = false;
	}
}

void group_destroy(CDeviceGroup *p_group)
{
	if(!p_group)
		return;
	CDeviceGroup &g = *p_group;
	if(g.p_threads) {
		g.p_threads->n_Run([&g](int r) -> int { // every member's memory goes on its own thread (its own device)
			TMemberShard &t = g.shards[r];
			if(t.p_values_dev) (void)hipFree(t.p_values_dev);
			if(t.p_rhs_dev) (void)hipFree(t.p_rhs_dev);
			if(t.p_cov_dev) (void)hipFree(t.p_cov_dev);
			t.p_values_dev = t.p_rhs_dev = t.p_cov_dev = 0;
			return SLAMPP_HIP_OK;
		});
	}
	group_release_exchange(g);
	for(int r = 0; r < GROUP_MAX_MEMBERS; ++ r) {
		if(g.ev_assembled[r])
			(void)hipEventDestroy(g.ev_assembled[r]);
	}
	for(size_t b = 0; b < g.ev_panel.size(); ++ b) {
		if(g.ev_panel[b])
			(void)hipEventDestroy(g.ev_panel[b]);
	}
	delete g.p_threads;
	g.p_threads = 0;
	for(size_t i = 0; i < g.members.size(); ++ i)
		slampp_hip_destroy(g.members[i]);
	delete p_group;
}

int group_set_option(CDeviceGroup &g, const char *p_s_name, int64_t n_value)
{
	if(!strcmp(p_s_name, "group_exchange")) {
		if(n_value < EXCHANGE_AUTO || n_value > EXCHANGE_PEER)
			return SLAMPP_HIP_ERR_INVALID;
		g.n_exchange_option = int(n_value);
		return SLAMPP_HIP_OK;
	}
	if(!strcmp(p_s_name, "group_fail_member")) { // test hook: member n_value - 1 fails on its way to the exchange (0: nobody)
		if(n_value < 0 || n_value > int64_t(g.members.size()))
			return SLAMPP_HIP_ERR_INVALID;
		g.n_fail_member = int(n_value);
		return SLAMPP_HIP_OK;
	}
	if(!strcmp(p_s_name, "shard_primary") || !strcmp(p_s_name, "shard_rank") || !strcmp(p_s_name, "shard_world") ||
	   !strcmp(p_s_name, "staging_ahead"))
		return SLAMPP_HIP_OK; // the group decides those for its members (which never take host arrays: no staging of their own)
	int n_result = SLAMPP_HIP_OK;
	for(size_t i = 0; i < g.members.size() && n_result == SLAMPP_HIP_OK; ++ i)
		n_result = slampp_hip_set_option(g.members[i], p_s_name, n_value);
	return n_result;
}

const char *group_exchange_name(const CDeviceGroup &g)
{
	return g.s_exchange_name.c_str();
}

int64_t group_exchange_count(const CDeviceGroup &g)
{
	return g.n_collectives_enqueued.load();
}

int group_member_num(const CDeviceGroup &g)
{
	return g.n_active;
}

slampp_hip_solver *group_member(CDeviceGroup &g, int n_member)
{
	return (n_member >= 0 && n_member < int(g.members.size()))? g.members[n_member] : 0;
}

// peer access between every pair of distinct member devices -- whatever exchange is chosen: the distributed factorization
// reads and writes the other members' buffers through raw pointers also when the all-reduce is RCCL's (relying on whatever
// mappings RCCL made internally would be a memory fault waiting to happen).  false: some pair cannot.
static bool group_enable_peer_access(CDeviceGroup &g, std::string &r_s_why)
{
	const int n = g.n_active;
	for(int i = 0; i < n; ++ i) {
		for(int j = 0; j < n; ++ j) {
			if(g.devices[i] == g.devices[j])
				continue;
			int b_can = 0;
			if(hipDeviceCanAccessPeer(&b_can, g.devices[i], g.devices[j]) != hipSuccess || !b_can) {
				(void)hipGetLastError();
				r_s_why = "device " + std::to_string(g.devices[i]) + " cannot access device " + std::to_string(g.devices[j]);
				return false;
			}
			(void)hipSetDevice(g.devices[i]);
			const hipError_t e = hipDeviceEnablePeerAccess(g.devices[j], 0);
			(void)hipGetLastError();
			if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
				r_s_why = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e);
				return false;
			}
		}
	}
	return true;
}

static void group_resolve_exchange(CDeviceGroup &g) // throw(CDeviceError)
{
	group_release_exchange(g);
	g.b_exchange_broken = false;
	const int n = g.n_active;
	bool b_distinct = true;
	for(int i = 0; i < n; ++ i) {
		for(int j = 0; j < i; ++ j)
			b_distinct = b_distinct && g.devices[i] != g.devices[j];
	}
	std::string s_no_peer;
	g.b_peer_access = group_enable_peer_access(g, s_no_peer);
	int n_want = g.n_exchange_option;
	if(const char *p_s_env = getenv("SLAMPP_HIP_GROUP_EXCHANGE")) {
		if(!strcmp(p_s_env, "rccl"))
			n_want = EXCHANGE_RCCL;
		else if(!strcmp(p_s_env, "peer"))
			n_want = EXCHANGE_PEER;
	}
	const bool b_one_member_rccl = n == 1 && n_want == EXCHANGE_RCCL; // (the exchange self-test: the RCCL calls themselves, on a 1-GPU box)
	CRccl *p_rccl = (n_want != EXCHANGE_PEER && (n > 1 || b_one_member_rccl))? CRccl::p_Get() : 0;
	if(n_want == EXCHANGE_RCCL) {
		if(!p_rccl)
			throw CDeviceError("device group: RCCL was asked for (group_exchange = 1) and librccl.so cannot be loaded");
		if(!b_distinct)
			throw CDeviceError("device group: RCCL needs distinct devices (group_exchange = 1 with a device listed twice)");
	}
	if((n > 1 || b_one_member_rccl) && p_rccl && b_distinct) {
		g.comms.assign(size_t(n), (void*)0);
		const int n_result = p_rccl->CommInitAll(&g.comms[0], n, &g.devices[0]);
		if(n_result != 0) {
			g.comms.clear();
			if(n_want == EXCHANGE_RCCL)
				throw CDeviceError(std::string("device group: ncclCommInitAll failed: ") + p_rccl->GetErrorString(n_result));
			fprintf(stderr, "libslampp_hip: ncclCommInitAll failed (%s): the members exchange through peer pointers instead\n",
				p_rccl->GetErrorString(n_result));
		} else {
			g.n_exchange = EXCHANGE_RCCL;
			g.s_exchange_name = "rccl (" + p_rccl->s_where + ")";
			return;
		}
	}
	if(!g.b_peer_access)
		throw CDeviceError("device group: the devices cannot read each other's memory (" + s_no_peer + ") and RCCL is not available");
	g.n_exchange = EXCHANGE_PEER;
	g.s_exchange_name = (n > 1)? "peer" : "none (one member)";
}

// splits the front handle's structure into landmark shards and analyzes every member (Schur mode) on its own thread
void group_analyze(slampp_hip_solver &r_front, int64_t n_cut) // throws
{
	CDeviceGroup &g = *r_front.p_group;
	const int64_t n_bcols = int64_t(r_front.cumsum.size()) - 1, n_points = n_bcols - n_cut;
	const int n_active = int(std::min<int64_t>(int64_t(g.members.size()), n_points));
	g.n_active = n_active;
	for(int r = 0; r < n_active; ++ r) {
		landmark_shard(n_bcols, &r_front.cumsum[0], &r_front.bcol_ptr[0], r_front.brow.empty()? 0 : &r_front.brow[0], n_cut,
			r, n_active, g.shards[r].t_structure);
	}
	group_resolve_exchange(g);
	(void)hipSetDevice(r_front.n_device);
	g.barrier.Reset(n_active);
	std::vector<std::string> errors(g.members.size());
	const int n_result = g.p_threads->n_Run([&](int r) -> int {
		slampp_hip_solver *p_member = g.members[r];
		if(r >= n_active)
			return slampp_hip_free_memory(p_member);
		const TShardStructure &t = g.shards[r].t_structure;
		int n_status = slampp_hip_set_option(p_member, "shard_primary", r == 0);
		if(n_status == SLAMPP_HIP_OK)
			n_status = slampp_hip_set_option(p_member, "shard_rank", r);
		if(n_status == SLAMPP_HIP_OK)
			n_status = slampp_hip_set_option(p_member, "shard_world", n_active);
		if(n_status == SLAMPP_HIP_OK) {
			n_status = slampp_hip_set_structure(p_member, int64_t(t.cumsum.size()) - 1, &t.cumsum[0], &t.bcol_ptr[0],
				t.brow.empty()? 0 : &t.brow[0]);
		}
		if(n_status == SLAMPP_HIP_OK)
			n_status = slampp_hip_analyze(p_member, SLAMPP_HIP_MODE_SCHUR, n_cut);
		if(n_status == SLAMPP_HIP_OK) {
			n_status = slampp_hip_set_allreduce(p_member, (n_active > 1)? group_allreduce_callback : (slampp_hip_allreduce_fn)0,
				&g.contexts[r]);
			// (the distributed factorization works through peer pointers whatever the all-reduce is: without peer access the
			// option schur_distributed is ignored and every member factors the summed system itself)
			p_member->p_dense_factor = (n_active > 1 && g.b_peer_access)? group_dense_factor_callback : (slampp_hip_solver::TDenseFactorFn)0;
			p_member->p_dense_factor_context = &g.contexts[r];
		}
		if(n_status == SLAMPP_HIP_OK) {
			TMemberShard &t_shard = g.shards[r];
			grow_device(t_shard.p_values_dev, t_shard.n_values_dev, size_t(t.n_camera_values + (t.n_value_end - t.n_value_begin)));
			grow_device(t_shard.p_rhs_dev, t_shard.n_rhs_dev, size_t(t.n_camera_scalars + (t.n_scalar_end - t.n_scalar_begin)));
			if(r != 0) { // a member that is not the primary never reads the camera blocks or the camera part of eta: zeros, once
				if(hipMemset(t_shard.p_values_dev, 0, size_t(t.n_camera_values) * sizeof(double)) != hipSuccess ||
				   hipMemset(t_shard.p_rhs_dev, 0, size_t(t.n_camera_scalars) * sizeof(double)) != hipSuccess)
					n_status = SLAMPP_HIP_ERR_DEVICE;
			}
		}
		if(n_status != SLAMPP_HIP_OK)
			errors[r] = slampp_hip_last_error(p_member);
		return n_status;
	});
	if(n_result == SLAMPP_HIP_ERR_ALLOC)
		throw std::bad_alloc();
	if(n_result != SLAMPP_HIP_OK) {
		std::string s_what = "device group: analysis of a member failed";
		for(size_t r = 0; r < errors.size(); ++ r) {
			if(!errors[r].empty())
				s_what += " (member " + std::to_string(r) + ": " + errors[r] + ")";
		}
		if(n_result == SLAMPP_HIP_ERR_UNSUPPORTED)
			throw std::domain_error(s_what);
		if(n_result == SLAMPP_HIP_ERR_INVALID)
			throw std::invalid_argument(s_what);
		throw CDeviceError(s_what);
	}
}

static int group_finish(slampp_hip_solver &r_front, CDeviceGroup &g, int n_result, const std::vector<std::string> &r_errors)
{
	(void)hipSetDevice(r_front.n_device);
	if(n_result == SLAMPP_HIP_OK)
		return n_result;
	if(n_result == SLAMPP_HIP_NOT_POSDEF) {
		r_front.s_error = "matrix is not positive definite";
		return n_result;
	}
	r_front.s_error = "device group:";
	for(size_t r = 0; r < r_errors.size(); ++ r) {
		if(!r_errors[r].empty())
			r_front.s_error += " member " + std::to_string(r) + ": " + r_errors[r] + ";";
	}
	g.barrier.Reset(g.n_active); // (an aborted exchange leaves the barrier closed)
	// the members count their calls of the distributed factorization and order their event waits by that count: one that
	// failed before it got there is a call behind the others.  Nobody is inside a solve now: everybody back to zero.
	memset(g.n_factor_calls, 0, sizeof(g.n_factor_calls));
	g.n_panel_sequence = 0;
	for(size_t i = 0; i < g.contexts.size(); ++ i)
		g.contexts[i].n_exchange_calls = 0; // (the same for the slots of the exchanged counts)
	return n_result;
}

// before a solve: a collective that failed in the last one took the communicators with it
static int group_repair_exchange(slampp_hip_solver &r_front, CDeviceGroup &g)
{
	if(!g.b_exchange_broken)
		return SLAMPP_HIP_OK;
	try {
		group_resolve_exchange(g);
		g.barrier.Reset(g.n_active);
		(void)hipSetDevice(r_front.n_device);
	} catch(std::exception &r_exc) {
		r_front.s_error = std::string("device group: the exchange could not be set up again: ") + r_exc.what();
		return SLAMPP_HIP_ERR_DEVICE;
	}
	return SLAMPP_HIP_OK;
}

enum { GROUP_SOLVE = 0, GROUP_MARGINAL_POSES = 1 };

// p_values / p_rhs_inout: the FULL system's packed values and right-hand side on the host (pinned staging or any array)
static int group_solve_kind(slampp_hip_solver &r_front, const double *p_values, double *p_rhs_inout, int n_kind)
{
	CDeviceGroup &g = *r_front.p_group;
	const int n_active = g.n_active;
	if(const int n_repair = group_repair_exchange(r_front, g))
		return n_repair;
	std::vector<std::string> errors(g.members.size());
	const double f_t0 = wall_ms();
	std::vector<double> upload_ms(g.members.size(), 0.0), solve_ms(g.members.size(), 0.0);
	const int n_result = g.p_threads->n_Run([&](int r) -> int {
		if(r >= n_active)
			return SLAMPP_HIP_OK;
		slampp_hip_solver *p_member = g.members[r];
		TMemberShard &t_shard = g.shards[r];
		const TShardStructure &t = t_shard.t_structure;
		hipStream_t stream = p_member->stream;
		const size_t n_own_values = size_t(t.n_value_end - t.n_value_begin), n_own_scalars = size_t(t.n_scalar_end - t.n_scalar_begin);
		int n_status = SLAMPP_HIP_OK;
		const double f_m0 = wall_ms();
		hipError_t e = hipSuccess;
		if(g.n_fail_member == r + 1) { // test hook: what an allocation or device error on the way to the exchange does
			errors[r] = "injected failure (option group_fail_member)";
			g.barrier.Abort();
			return SLAMPP_HIP_ERR_DEVICE;
		}
		if(r == 0 && n_kind == GROUP_SOLVE) { // the primary adds A and the camera part of eta
			e = hipMemcpyAsync(t_shard.p_values_dev, p_values, size_t(t.n_camera_values) * sizeof(double), hipMemcpyHostToDevice, stream);
			if(e == hipSuccess)
				e = hipMemcpyAsync(t_shard.p_rhs_dev, p_rhs_inout, size_t(t.n_camera_scalars) * sizeof(double), hipMemcpyHostToDevice, stream);
		}
		if(e == hipSuccess)
			e = hipMemcpyAsync(t_shard.p_values_dev + t.n_camera_values, p_values + t.n_value_begin, n_own_values * sizeof(double),
				hipMemcpyHostToDevice, stream);
		if(e == hipSuccess)
			e = hipMemcpyAsync(t_shard.p_rhs_dev + t.n_camera_scalars, p_rhs_inout + t.n_scalar_begin, n_own_scalars * sizeof(double),
				hipMemcpyHostToDevice, stream);
		if(e != hipSuccess) {
			errors[r] = std::string("upload: ") + hipGetErrorString(e);
			g.barrier.Abort();
			return SLAMPP_HIP_ERR_DEVICE;
		}
		upload_ms[r] = wall_ms() - f_m0;
		n_status = (n_kind == GROUP_SOLVE)? slampp_hip_factor_solve_device_async(p_member, t_shard.p_values_dev, t_shard.p_rhs_dev) :
			slampp_hip_solve_marginal_poses_device_async(p_member, t_shard.p_values_dev, t_shard.p_rhs_dev);
		if(n_status == SLAMPP_HIP_OK) {
			// the solution comes back behind the solve: every member its own landmarks, the primary the cameras as well
			e = hipMemcpyAsync(p_rhs_inout + t.n_scalar_begin, t_shard.p_rhs_dev + t.n_camera_scalars, n_own_scalars * sizeof(double),
				hipMemcpyDeviceToHost, stream);
			if(e == hipSuccess && r == 0)
				e = hipMemcpyAsync(p_rhs_inout, t_shard.p_rhs_dev, size_t(t.n_camera_scalars) * sizeof(double), hipMemcpyDeviceToHost, stream);
			if(e != hipSuccess) {
				errors[r] = std::string("download: ") + hipGetErrorString(e);
				n_status = SLAMPP_HIP_ERR_DEVICE;
			}
		}
		if(n_status == SLAMPP_HIP_OK)
			n_status = slampp_hip_sync(p_member);
		else
			g.barrier.Abort(); // the others may be waiting for this member in the exchange
		if(n_status < 0 && errors[r].empty())
			errors[r] = slampp_hip_last_error(p_member);
		solve_ms[r] = wall_ms() - f_m0;
		return n_status;
	});
	r_front.times.upload_ms = *std::max_element(upload_ms.begin(), upload_ms.end());
	r_front.times.schur_ms = *std::max_element(solve_ms.begin(), solve_ms.end());
	r_front.times.total_ms = wall_ms() - f_t0;
	return group_finish(r_front, g, n_result, errors);
}

// The members read the front handle's staging by DMA from their own devices.  Memory pinned without the Portable flag is
// pinned for the device that was current when it was pinned; another device's hipMemcpyAsync would then treat it as
// pageable (a staged copy on the calling thread: still correct, a fraction of the rate, and no longer asynchronous) or,
// on some configurations, refuse it.  Checked once per staging allocation, on every member's own thread and device:
// the pointer must be registered host memory there, and one double must make the round trip.
int group_check_staging(slampp_hip_solver &r_front, double *p_values, double *p_rhs)
{
	CDeviceGroup &g = *r_front.p_group;
	if(g.p_checked_staging == p_values && g.p_checked_staging_rhs == p_rhs)
		return SLAMPP_HIP_OK;
	if(!p_values || !p_rhs)
		return SLAMPP_HIP_OK; // (nothing allocated yet)
	std::vector<std::string> errors(g.members.size());
	const double f_probe = p_rhs[0];
	const int n_result = g.p_threads->n_Run([&](int r) -> int {
		if(r >= g.n_active)
			return SLAMPP_HIP_OK;
		hipPointerAttribute_t t_attr;
		hipError_t e = hipPointerGetAttributes(&t_attr, p_values);
		if(e != hipSuccess || t_attr.type != hipMemoryTypeHost) {
			(void)hipGetLastError();
			errors[r] = "device " + std::to_string(g.devices[r]) + " does not see the host staging as pinned memory (" +
				((e != hipSuccess)? std::string(hipGetErrorString(e)) : std::string("memory type ") + std::to_string(int(t_attr.type))) + ")";
			return SLAMPP_HIP_ERR_DEVICE;
		}
		double *p_dev = 0, f_back = 0;
		hipStream_t stream = g.members[r]->stream;
		e = hipMalloc((void**)&p_dev, sizeof(double));
		if(e == hipSuccess)
			e = hipMemcpyAsync(p_dev, p_rhs, sizeof(double), hipMemcpyHostToDevice, stream);
		if(e == hipSuccess)
			e = hipMemcpyAsync(&f_back, p_dev, sizeof(double), hipMemcpyDeviceToHost, stream);
		if(e == hipSuccess)
			e = hipStreamSynchronize(stream);
		if(p_dev)
			(void)hipFree(p_dev);
		if(e != hipSuccess || memcmp(&f_back, &f_probe, sizeof(double))) {
			(void)hipGetLastError();
			errors[r] = "device " + std::to_string(g.devices[r]) + " cannot copy from the host staging (" +
				((e != hipSuccess)? std::string(hipGetErrorString(e)) : std::string("the value read back differs")) + ")";
			return SLAMPP_HIP_ERR_DEVICE;
		}
		return SLAMPP_HIP_OK;
	});
	if(n_result != SLAMPP_HIP_OK) {
		r_front.s_error = "device group: ";
		for(size_t r = 0; r < errors.size(); ++ r) {
			if(!errors[r].empty())
				r_front.s_error += errors[r] + "; ";
		}
		r_front.s_error += "the members upload their shards from that memory";
		return SLAMPP_HIP_ERR_DEVICE;
	}
	g.p_checked_staging = p_values;
	g.p_checked_staging_rhs = p_rhs;
	return SLAMPP_HIP_OK;
}

int group_factor_solve(slampp_hip_solver &r_front, const double *p_values, double *p_rhs_inout)
{
	return group_solve_kind(r_front, p_values, p_rhs_inout, GROUP_SOLVE);
}

int group_solve_marginal_poses(slampp_hip_solver &r_front, const double *p_values, double *p_rhs_inout)
{
	const int n_result = group_solve_kind(r_front, p_values, p_rhs_inout, GROUP_MARGINAL_POSES);
	if(n_result == SLAMPP_HIP_OK) // (no member was given the camera part: it comes back as the zeros the call defines)
		memset(p_rhs_inout, 0, size_t(r_front.cumsum[size_t(r_front.n_matrix_cut)]) * sizeof(double));
	return n_result;
}

// block-diagonal covariances: every member its own landmarks, the primary the cameras
int group_schur_marginals(slampp_hip_solver &r_front, const double *p_values, double *p_cam_cov, double *p_point_cov)
{
	CDeviceGroup &g = *r_front.p_group;
	const int n_active = g.n_active;
	const int64_t n_cut = r_front.n_matrix_cut;
	const int64_t dc = r_front.cumsum[1] - r_front.cumsum[0], dp = r_front.cumsum[size_t(n_cut) + 1] - r_front.cumsum[size_t(n_cut)];
	const size_t n_cam_doubles = size_t(n_cut * dc * dc);
	if(const int n_repair = group_repair_exchange(r_front, g))
		return n_repair;
	std::vector<std::string> errors(g.members.size());
	const int n_result = g.p_threads->n_Run([&](int r) -> int {
		if(r >= n_active)
			return SLAMPP_HIP_OK;
		slampp_hip_solver *p_member = g.members[r];
		TMemberShard &t_shard = g.shards[r];
		const TShardStructure &t = t_shard.t_structure;
		hipStream_t stream = p_member->stream;
		const size_t n_own_values = size_t(t.n_value_end - t.n_value_begin);
		const size_t n_own_doubles = size_t((t.n_point_end - t.n_point_begin) * dp * dp);
		const bool b_cams = r == 0 && p_cam_cov != 0;
		try {
			grow_device(t_shard.p_cov_dev, t_shard.n_cov_dev, n_cam_doubles + n_own_doubles);
		} catch(std::exception &r_exc) {
			errors[r] = r_exc.what();
			g.barrier.Abort();
			return SLAMPP_HIP_ERR_ALLOC;
		}
		hipError_t e = hipSuccess;
		if(r == 0)
			e = hipMemcpyAsync(t_shard.p_values_dev, p_values, size_t(t.n_camera_values) * sizeof(double), hipMemcpyHostToDevice, stream);
		if(e == hipSuccess)
			e = hipMemcpyAsync(t_shard.p_values_dev + t.n_camera_values, p_values + t.n_value_begin, n_own_values * sizeof(double),
				hipMemcpyHostToDevice, stream);
		if(e != hipSuccess) {
			errors[r] = std::string("upload: ") + hipGetErrorString(e);
			g.barrier.Abort();
			return SLAMPP_HIP_ERR_DEVICE;
		}
		int n_status = slampp_hip_schur_marginals_device_async(p_member, t_shard.p_values_dev, b_cams? t_shard.p_cov_dev : 0,
			(p_point_cov || !b_cams)? t_shard.p_cov_dev + n_cam_doubles : 0);
		if(n_status == SLAMPP_HIP_OK)
			n_status = slampp_hip_sync(p_member);
		else
			g.barrier.Abort();
		if(n_status == SLAMPP_HIP_OK) {
			if(b_cams)
				e = hipMemcpyAsync(p_cam_cov, t_shard.p_cov_dev, n_cam_doubles * sizeof(double), hipMemcpyDeviceToHost, stream);
			if(e == hipSuccess && p_point_cov)
				e = hipMemcpyAsync(p_point_cov + size_t(t.n_point_begin * dp * dp), t_shard.p_cov_dev + n_cam_doubles,
					n_own_doubles * sizeof(double), hipMemcpyDeviceToHost, stream);
			if(e == hipSuccess)
				e = hipStreamSynchronize(stream);
			if(e != hipSuccess) {
				errors[r] = std::string("download: ") + hipGetErrorString(e);
				n_status = SLAMPP_HIP_ERR_DEVICE;
			}
		}
		if(n_status < 0 && errors[r].empty())
			errors[r] = slampp_hip_last_error(p_member);
		return n_status;
	});
	return group_finish(r_front, g, n_result, errors);
}

int group_free_memory(CDeviceGroup &g)
{
	group_release_exchange(g);
	return g.p_threads->n_Run([&g](int r) -> int {
		TMemberShard &t = g.shards[r];
		if(t.p_values_dev) (void)hipFree(t.p_values_dev);
		if(t.p_rhs_dev) (void)hipFree(t.p_rhs_dev);
		if(t.p_cov_dev) (void)hipFree(t.p_cov_dev);
		t.p_values_dev = t.p_rhs_dev = t.p_cov_dev = 0;
		t.n_values_dev = t.n_rhs_dev = t.n_cov_dev = 0;
		return slampp_hip_free_memory(g.members[r]);
	});
}

void group_fill_stats(CDeviceGroup &g, slampp_hip_stats &r_stats)
{
	slampp_hip_stats t_sum;
	memset(&t_sum, 0, sizeof(t_sum));
	for(int r = 0; r < g.n_active; ++ r) {
		slampp_hip_stats t;
		if(slampp_hip_get_stats(g.members[r], &t) != SLAMPP_HIP_OK)
			continue;
		if(r == 0) {
			r_stats.n_cams = t.n_cams;
			r_stats.schur_dim = t.schur_dim;
			r_stats.l_blocks = t.l_blocks;
			r_stats.factor_flops = t.factor_flops;
			r_stats.solve_flops = t.solve_flops;
		}
		t_sum.n_points += t.n_points;
		t_sum.n_observations += t.n_observations;
		t_sum.n_update_pairs += t.n_update_pairs;
		t_sum.device_bytes += t.device_bytes;
	}
	r_stats.n_points = t_sum.n_points;
	r_stats.n_observations = t_sum.n_observations;
	r_stats.n_update_pairs = t_sum.n_update_pairs;
	r_stats.device_bytes += t_sum.device_bytes;
	for(int r = 0; r < g.n_active; ++ r) {
		const TMemberShard &t = g.shards[r];
		r_stats.device_bytes += int64_t((t.n_values_dev + t.n_rhs_dev + t.n_cov_dev) * sizeof(double));
	}
}

// the exchange by itself: every member fills a buffer with a pattern of its own, the buffers are summed through the
// members' callback, every member checks every entry
int group_exchange_selftest(const int *p_device_ids, int n_devices, int n_exchange, size_t n_count, std::string &r_s_what)
{
	CDeviceGroup *p_group = group_create(p_device_ids, n_devices);
	CDeviceGroup &g = *p_group;
	int n_result = SLAMPP_HIP_OK;
	try {
		g.n_active = n_devices;
		g.n_exchange_option = n_exchange;
		group_resolve_exchange(g);
		g.barrier.Reset(n_devices);
		r_s_what = g.s_exchange_name;
		std::vector<double> expect(n_count);
		for(size_t i = 0; i < n_count; ++ i) {
			double f_sum = 0;
			for(int r = 0; r < n_devices; ++ r)
				f_sum += double(r + 1) * double(i % 7 + 1) + 0.25 * double(r);
			expect[i] = f_sum;
		}
		n_result = g.p_threads->n_Run([&](int r) -> int {
			std::vector<double> mine(n_count);
			for(size_t i = 0; i < n_count; ++ i)
				mine[i] = double(r + 1) * double(i % 7 + 1) + 0.25 * double(r);
			hipStream_t stream = g.members[r]->stream;
			double *p_dev = 0;
			if(hipMalloc((void**)&p_dev, std::max<size_t>(n_count, 1) * sizeof(double)) != hipSuccess) {
				g.barrier.Abort();
				return SLAMPP_HIP_ERR_ALLOC;
			}
			int n_status = SLAMPP_HIP_OK;
			if(hipMemcpyAsync(p_dev, mine.data(), n_count * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) {
				g.barrier.Abort();
				n_status = SLAMPP_HIP_ERR_DEVICE;
			}
			for(int n_pass = 0; n_pass < 2 && n_status == SLAMPP_HIP_OK; ++ n_pass) { // twice: the second pass sums the sums
				if(group_allreduce_callback(&g.contexts[r], p_dev, n_count, (void*)stream) != 0)
					n_status = SLAMPP_HIP_ERR_DEVICE;
			}
			if(n_status == SLAMPP_HIP_OK && (hipMemcpyAsync(mine.data(), p_dev, n_count * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
			   hipStreamSynchronize(stream) != hipSuccess))
				n_status = SLAMPP_HIP_ERR_DEVICE;
			(void)hipFree(p_dev);
			for(size_t i = 0; i < n_count && n_status == SLAMPP_HIP_OK; ++ i) {
				if(mine[i] != expect[i] * double(n_devices)) // (small integers and quarters: exact in any order)
					n_status = SLAMPP_HIP_ERR_DEVICE;
			}
			return n_status;
		});
	} catch(...) {
		group_destroy(p_group);
		throw;
	}
	group_destroy(p_group);
	return n_result;
}

} // namespace slampp

using namespace slampp;

extern "C" {

int slampp_hip_group_exchange_selftest(const int *p_device_ids, int n_devices, int n_exchange, int64_t n_count,
	char *p_s_exchange_out, int n_max_chars)
{
	if(!p_device_ids || n_devices < 1 || n_count < 0 || n_exchange < EXCHANGE_AUTO || n_exchange > EXCHANGE_PEER)
		return SLAMPP_HIP_ERR_INVALID;
	std::string s_what;
	int n_result;
	try {
		n_result = group_exchange_selftest(p_device_ids, n_devices, n_exchange, size_t(n_count), s_what);
	} catch(std::bad_alloc&) {
		return SLAMPP_HIP_ERR_ALLOC;
	} catch(std::invalid_argument&) {
		return SLAMPP_HIP_ERR_INVALID;
	} catch(std::exception &r_exc) {
		s_what = r_exc.what();
		n_result = SLAMPP_HIP_ERR_DEVICE;
	}
	if(p_s_exchange_out && n_max_chars > 0) {
		strncpy(p_s_exchange_out, s_what.c_str(), size_t(n_max_chars) - 1);
		p_s_exchange_out[n_max_chars - 1] = 0;
	}
	return n_result;
}

int slampp_hip_landmark_shard(int64_t n_bcols, const int64_t *p_bcol_cumsum, const int64_t *p_bcol_ptr,
	const int32_t *p_brow_idx, int64_t n_matrix_cut, int n_rank, int n_world, slampp_hip_shard_view *p_view)
{
	if(!p_bcol_cumsum || !p_bcol_ptr || !p_view || n_bcols <= 0 || (p_bcol_ptr[n_bcols] > 0 && !p_brow_idx))
		return SLAMPP_HIP_ERR_INVALID;
	try {
		TShardStructure t;
		landmark_shard(n_bcols, p_bcol_cumsum, p_bcol_ptr, p_brow_idx, n_matrix_cut, n_rank, n_world, t);
		p_view->n_bcols = int64_t(t.cumsum.size()) - 1;
		p_view->n_blocks = int64_t(t.brow.size());
		p_view->n_camera_values = t.n_camera_values;
		p_view->n_value_begin = t.n_value_begin;
		p_view->n_value_end = t.n_value_end;
		p_view->n_camera_scalars = t.n_camera_scalars;
		p_view->n_scalar_begin = t.n_scalar_begin;
		p_view->n_scalar_end = t.n_scalar_end;
		p_view->n_point_begin = t.n_point_begin;
		p_view->n_point_end = t.n_point_end;
		if(p_view->p_bcol_cumsum)
			std::copy(t.cumsum.begin(), t.cumsum.end(), p_view->p_bcol_cumsum);
		if(p_view->p_bcol_ptr)
			std::copy(t.bcol_ptr.begin(), t.bcol_ptr.end(), p_view->p_bcol_ptr);
		if(p_view->p_brow_idx)
			std::copy(t.brow.begin(), t.brow.end(), p_view->p_brow_idx);
		return SLAMPP_HIP_OK;
	} catch(std::bad_alloc&) {
		return SLAMPP_HIP_ERR_ALLOC;
	} catch(std::domain_error&) {
		return SLAMPP_HIP_ERR_UNSUPPORTED;
	} catch(std::exception&) {
		return SLAMPP_HIP_ERR_INVALID;
	}
}

} // extern "C"
