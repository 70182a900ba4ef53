// solver.h -- the object behind the C ABI (include/slampp_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include <thread>
#include <memory>
#include <mutex>

#include "../../include/slampp_hip.h"
#include "plan.h"
#include <utility>
#include "sparse_kernels.h"
#include "dense_chol.h"

namespace slampp {

struct CDeviceError : public std::runtime_error {
	explicit CDeviceError(const std::string &s) :std::runtime_error(s) {}
};

#define SLAMPP_HIP_CHECK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { \
	throw slampp::CDeviceError(std::string(#call) + ": " + hipGetErrorString(e_)); } } while(0)

inline double wall_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// owning device array
// CDevArray::Free() keeps the memory while this is set (thread-local: a handle is used from one thread at a time, the members
// of a device group each from their own)
extern thread_local bool g_b_keep_device_memory;

template <class T>
class CDevArray {
	T *m_p;
	size_t m_n, m_cap; // elements in use (0: the array does not exist for its users), elements allocated
public:
	CDevArray() :m_p(0), m_n(0), m_cap(0) {}
	~CDevArray() { Release(); }
	CDevArray(const CDevArray&) = delete;
	CDevArray &operator =(const CDevArray&) = delete;
	// gives the memory back
	void Release() { if(m_p) (void)hipFree(m_p); m_p = 0; m_n = 0; m_cap = 0; }
	// The array ceases to exist for its users -- p() is null, n() zero -- but keeps its memory for the next Alloc().  A
	// re-analysis frees and allocates some fifty arrays; hipFree synchronizes the device and hipMalloc is no cheaper: at
	// the small systems FastL hands Factorize_PosDef_Blocky (a new part of R at almost every call) that was most of an
	// analysis (round 5: DESIGN.md section 10).  slampp_hip_free_memory() and the destructor Release().
	void Free() { m_n = 0; if(!g_b_keep_device_memory) Release(); }
	void Alloc(size_t n) // throw(std::bad_alloc, CDeviceError)
	{
		if(!n)
			n = 1;
		if(n <= m_cap && m_p) {
			m_n = std::max(m_n, n); // (growing in place; a user that asks for less keeps what it has, as before)
			return;
		}
		Release();
		hipError_t e = hipMalloc((void**)&m_p, n * sizeof(T));
		if(e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
			(void)hipGetLastError();
			m_p = 0;
			throw std::bad_alloc();
		}
		if(e != hipSuccess) {
			m_p = 0;
			throw CDeviceError(std::string("hipMalloc: ") + hipGetErrorString(e));
		}
		m_n = m_cap = n;
	}
	template <class CAlloc>
	void Upload(const std::vector<T, CAlloc> &v, hipStream_t s)
	{
		Alloc(v.size());
		if(!v.empty())
			SLAMPP_HIP_CHECK(hipMemcpyAsync(m_p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
	}
	T *p() const { return m_n? m_p : 0; }
	size_t n() const { return m_n; }
	size_t n_Bytes() const { return m_p? m_n * sizeof(T) : 0; }
	void Swap(CDevArray &r_other) { std::swap(m_p, r_other.m_p); std::swap(m_n, r_other.m_n); std::swap(m_cap, r_other.m_cap); }
};

// allocator for the big work arrays of the analysis that are written in full before they are read: std::vector<T>(n)
// zero-fills -- 32 MB on one thread, a page fault every 4 KB: 5 ms, and C5's analysis made six of them --; with this
// allocator the elements are left as they are and the pages are first touched by the (threaded) loops that fill them
// ... and from one megabyte on their memory is mapped by the library itself, aligned to 2 MB and offered to the kernel as
// transparent huge pages (round 6, solver.hip: host_pool_*).  First touches are what an analysis of a large system waits for
// (tools/micro/page_touch.cpp on the box: 256 MB of fresh 4 KB pages take 38 ms to touch on one thread and 15 - 17 ms on
// eight -- the threads queue for the process's memory map --, and 30 - 39 ms to unmap; as huge pages 9.7 / 1.7 - 2.2 ms and
// 12 - 19 ms).  The C library's allocator hands out addresses 16 bytes behind a page boundary, which madvise() refuses: the
// one attempt at huge pages of the round's first hours did nothing for that reason.  A dropped array's mapping is kept for the
// next array of about its size and everything goes back to the system when the analysis is over (host_pool_release, called by
// the thread that frees the analysis' arrays).
void *host_pool_alloc(size_t n_bytes); // throws std::bad_alloc
void host_pool_free(void *p) noexcept;
void host_pool_release() noexcept;     // unmaps the blocks nobody holds
enum { host_pool_min_bytes = 1 << 20 };

template <class T>
struct CNoInitAlloc : std::allocator<T> {
	template <class U> struct rebind { typedef CNoInitAlloc<U> other; };
	CNoInitAlloc() {}
	template <class U> CNoInitAlloc(const CNoInitAlloc<U>&) {}
	T *allocate(size_t n)
	{
		if(n > size_t(-1) / sizeof(T))
			throw std::bad_alloc();
		return (n * sizeof(T) >= size_t(host_pool_min_bytes))? static_cast<T*>(host_pool_alloc(n * sizeof(T))) : static_cast<T*>(::operator new(n * sizeof(T)));
	}
	void deallocate(T *p, size_t n) noexcept
	{
		if(n * sizeof(T) >= size_t(host_pool_min_bytes))
			host_pool_free(p);
		else
			::operator delete(p);
	}
	template <class U> void construct(U *p) { ::new((void*)p) U; } // default-initialization: nothing for arithmetic types
	template <class U, class... CArgs> void construct(U *p, CArgs&&... args) { ::new((void*)p) U(std::forward<CArgs>(args)...); }
};
template <class T> using raw_vector = std::vector<T, CNoInitAlloc<T> >;

// work arrays of the analysis that nobody reads any more, kept until a thread that has nothing urgent left frees them:
// giving 150 MB back to the system (C5: hashes, sort items, orders of two million landmarks) is 9 - 15 ms of page-table
// work, and it used to happen on the analysis' own thread at the end of a scope
struct TTrash { virtual ~TTrash() {} };
template <class T> struct TTrashOf : TTrash { T t; explicit TTrashOf(T &r) { t.swap(r); } };
typedef std::vector<std::unique_ptr<TTrash> > CTrashList;
template <class T> inline void Discard_Later(CTrashList &r_trash, T &r_v) { r_trash.emplace_back(new TTrashOf<T>(r_v)); }

// while one of these lives on a thread, the device arrays freed on that thread keep their memory (a re-analysis)
struct CKeepDeviceMemory {
	bool b_before;
	CKeepDeviceMemory() :b_before(g_b_keep_device_memory) { g_b_keep_device_memory = true; }
	~CKeepDeviceMemory() { g_b_keep_device_memory = b_before; }
};

struct CSchurState; // schur.hip
struct CDeviceGroup; // group.hip
struct CSparseInverse; // sparse_inverse.hip
struct CAssemblyState; // assembly.hip

} // namespace slampp

struct slampp_hip_assembly {
	slampp_hip_solver *p_solver;     // null once the solver is gone
	slampp::CAssemblyState *p_state;
	bool b_stale;                    // set_structure() was called since: the block offsets no longer apply
};

struct slampp_hip_solver {
	int n_device;
	hipStream_t stream;
	std::string s_error;
	slampp::PlanOptions opt;
	int n_dense_nb;
	int b_shard_primary;
	int n_shard_rank, n_shard_world; // optional (-1, 0 = unknown): lets the ranks exchange block lists instead of an nc^2 indicator
	int n_marginals_dense; // option marginals_dense: 1 = the covariances always through the dense inverse of the reduced system
	int n_assembly_groups = 1 << 20; // option "assembly_groups": most vertices a group of the Lambda assembly takes (0: no groups, the one-wave kernels for everything)
	int n_schur_tiles = -1; // option "schur_tiles": landmark-major assembly of S: -1 = where it pays, 0 = never, 1 = wherever possible
	int n_schur_incremental = 0; // option "schur_incremental": keep the assembled reduced system for slampp_hip_schur_set_changed_points
	int n_schur_sparse; // reduced camera system: -1 = sparse path when few of its blocks are nonzero, 0 = always dense, 1 = always sparse

	// Lambda structure as given
	bool b_has_structure, b_analyzed, b_factored;
	int n_mode;
	int64_t n_matrix_cut;
	std::vector<int64_t> cumsum, bcol_ptr;
	std::vector<int32_t> brow;
	int64_t n_values, n_scalars;

	// sparse path
	// Block columns wider than the kernels take (8) are cut into pieces of at most 8: the scalar matrix is the same, the
	// right-hand side and the solution are untouched, and the packed values are regrouped on the device (one gather
	// through d_refine_map) in front of every factorization.  The reference's solvers take any block size.
	bool b_refined = false;
	int64_t n_refined_values = 0;
	std::vector<int64_t> refined_cumsum, refined_bcol_ptr;
	std::vector<int32_t> refined_brow;
	slampp::CDevArray<int64_t> d_refine_map;
	slampp::CDevArray<double> d_refined;
	void Refine_Structure(); // throws
	// Schur mode asked for a structure the Schur kernels do not take (landmark-landmark blocks: C not block diagonal, the
	// reference's InverseOf_Symmteric_FBS branch, LinearSolver_Schur.h:1721-1726; block sizes other than (6,3), (7,3), (3,2);
	// no landmark part at all, :1635-1638): the same system goes through the sparse block path, which solves it all the same
	bool b_schur_fallback = false;
	int n_staging_ahead = 0; // option "staging_ahead": slampp_hip_analyze brings up the pinned host staging on a thread of its own
	int n_schur_fallback_option = 1; // option "schur_fallback": 0 = report SLAMPP_HIP_ERR_UNSUPPORTED instead
	slampp::Plan plan;
	int n_bottom_stages; // leading stages launched with one wave per task
	slampp::TDevPlan dplan;
	slampp::CDevArray<slampp::TColDesc> d_cols;
	slampp::CDevArray<slampp::TBlkDesc> d_blks;
	slampp::CDevArray<slampp::TRowEnt> d_rents;
	slampp::CDevArray<longlong2> d_pairs;
	slampp::CDevArray<int64_t> d_task_ptr, d_task_pkg;
	slampp::CDevArray<longlong2> d_pkg;
	slampp::CDevArray<long long> d_timing; // development aid, see TDevPlan::p_timing
	// lane-per-task kernel of the wide bottom stages (simt_kernel.hip): chunks of 64 same-shaped tasks; the tasks of a
	// stage whose shape is too rare stay with the wave-per-task kernel (d_simt_rest lists them)
	slampp::CDevArray<slampp::TSimtChunk> d_simt_chunks;
	slampp::CDevArray<int32_t> d_simt_prog, d_simt_rest;
	// separator tasks that run as panels in LDS (panel_kernel.hip): their packages, per stage the offsets of the packages and
	// the tasks left to factor_stage_kernel
	int n_panel = -1; // option "panel": -1 / 1 = where a task fits (default), 0 = never
	int n_panel_rows = -1; // option "panel_rows": 1 = the panel tasks factor a block column as rows (round 4), 0 = block by block, -1 = rows where the
	                       // blocks are 6 x 6 or 7 x 7 (with the hand-ups on: C3 206 -> 196 us of separator launches, band reduced system 216 -> 208;
	                       // 3 x 3 blocks -- C1 -- 124 -> 140: their block-wise levels are cheaper than two passes over the image)
	slampp::CDevArray<longlong2> d_panel_pkg;
	slampp::CDevArray<int64_t> d_panel_off, d_panel_out_off;
	slampp::CDevArray<double> d_handup; // the blocks the panel tasks hand up to the next stage's (TPanelOut)
	bool b_any_hand_up = false;
	int n_panel_handup = 1; // option "panel_handup": 1 = a panel task computes what it owes the next stage's tasks out of its own image (round 4), 0 = they fetch the operands
	slampp::CDevArray<int32_t> d_panel_rest;
	slampp::CDevArray<slampp::TUpdSlot> d_panel_upd_slots; // the factor blocks of the panel tasks, stage by stage, and the
	slampp::CDevArray<slampp::TUpdEnt> d_panel_upd_ents;   // updates they receive from earlier stages (panel_update_kernel)
	std::vector<int32_t> panel_ptr, panel_rest_ptr, panel_upd_ptr; // [n_stages + 1] ranges of the lists (empty: no panels)
	std::vector<slampp::TPanelLaunch> panel_cfg; // [n_stages] waves per task and LDS capacities of the stage's panel launch
	std::vector<char> panel_ride; // [n_stages + 1] the stage's updates from further down are applied inside the launch of the stage below
	slampp::CDevArray<int64_t> d_simt_tab;
	// the same chunks for the backward substitution (backward_simt_kernel): per shape [n_cols, blocks below the diagonals, nb per column],
	// per lane (offset of the column's first factor block, scalar offset in the workspace, in the caller's vector) per column and
	// the workspace offset of every sub-diagonal block's row
	slampp::CDevArray<slampp::TSimtChunk> d_simt_bwd_chunks;
	slampp::CDevArray<int32_t> d_simt_bwd_prog;
	slampp::CDevArray<int64_t> d_simt_bwd_tab;
	std::vector<int32_t> simt_bwd_lds_bytes;
	std::vector<slampp::TSimtChunk> simt_host_bwd_chunks;
	std::vector<int32_t> simt_host_bwd_prog;
	slampp::raw_vector<int64_t> simt_host_bwd_tab; // (raw_vector: written in full by the table pass, never zero-filled; its own mapping on huge pages)
	// inv(L_jj) of the columns the lane-per-task kernel factors is not on the solve's path any more (its backward kernel solves with
	// L_jj^T): stored only once something has asked for it (another right-hand side, covariances) -- from then on always
	bool b_leaf_linv_wanted = false, b_leaf_linv_valid = true;
	// K value sets in the same launches (slampp_hip_factor_solve_batch_device_async): the members' factors, inverses,
	// workspaces and hand-up buffers side by side (swapped in for d_L .. d_handup while the batch is enqueued), a flag per
	// member, and what Enqueue_Sparse hands to every launch
	slampp::TBatch t_batch = slampp::t_No_Batch();
	slampp::CDevArray<double> d_batch_L, d_batch_Linv, d_batch_w, d_batch_handup;
	slampp::CDevArray<int> d_batch_flag;
	int n_batch_owner_member = -1; // the batch member whose factor became the handle's own (a batch of one), -1 = none; answered for at sync_batch
	int *p_host_batch_flag = 0; // pinned, SLAMPP_HIP_MAX_BATCH ints
	int n_batch_pending = 0;    // members of the batches enqueued since the last slampp_hip_sync_batch (the largest)
	int n_simt_backward = -1; // option "simt_backward": 1 = the leaf subtrees' backward substitution a lane per task as well and no inv(L_jj) stored for them; 0 = a wave per task; -1 (default) = by the number of leaf subtrees (round 4, after the new ordering: slower below ~12 000 of them, 1.6 % faster at C3, 7 % at a million poses)
	void Ensure_Leaf_Inverses();
	std::vector<int32_t> simt_chunk_ptr, simt_rest_ptr; // [n_bottom_stages + 1] each; empty = not in use
	std::vector<int32_t> simt_lds_bytes; // per stage: the largest chunk table (it is staged in LDS)
	int n_simt = -1; // option "simt": 1 / -1 = use it where it applies (default), 0 = never
	int n_wide_min_tasks = 8192; // option "wide_min_tasks": stages with more tasks than this run one wave per task, one tree level per stage (throughput); below, tasks are slices of the tree in LDS (C3: its 7 513-task stage 0.433 -> 0.404 ms as slices; a million poses: 2.31 ms with the 75 000-task stages wide, 2.36 as slices)
	int n_simt_width = 32; // option "simt_width": tasks per wave (16, 32, 64)
	int n_simt_stages = 1; // option "simt_stages": how many of the bottom stages it takes (the stages above the leaves hold
	                       // single separator columns whose operands other waves wrote: no gain there, measured)
	void Build_Simt(); // throws; host work only
	void Upload_Simt(); // throws
	std::vector<slampp::TSimtChunk> simt_host_chunks; // what Build_Simt() made, until Upload_Simt() has sent it
	std::vector<int32_t> simt_host_prog, simt_host_rest;
	slampp::raw_vector<int64_t> simt_host_tab;
	// dense top of the sparse path (plan.h): assembled Schur complement + dense factor workspaces
	slampp::CDevArray<slampp::TDenseBlk> d_dense_blks;
	slampp::CDevArray<int64_t> d_dense_blk_loff; // where each of those blocks lives in the factor's block layout (slampp_hip_factorize)
	slampp::CDevArray<double> d_dense, d_dense_invdiag, d_dense_z, d_dense_x;
	slampp::CDevArray<int32_t> d_dense_gaps; // positions inside the dense top that no column maps to (alignment padding)
	int n_dense_gaps;
	slampp::CDevArray<uint8_t> d_dense_unit;   // per position of the padded dense top: 1 = padding or gap (identity on the diagonal)
	slampp::CDevArray<longlong2> d_dense_dst;  // per position: where x goes (.x in w, .y in the caller's vector; < 0: nowhere)
	slampp::CTileSchedule dense_tiles; // level schedule over the nonzero tiles of the dense top (dense_chol.h)
	bool b_dense_tiles;                // use it (its dependent chain is clearly shorter than the tile count)
	bool b_dense_clean;                // the tiles of d_dense outside the schedule are zero (a full memset has run since it was allocated and only the schedule's tiles have been written): a step zeroes the schedule's tiles only
	int n_dense_top_tiles;             // option: -1 = decide per structure, 0 = always the dense schedule, 1 = always the tile schedule
	int n_dense_blks, n_dense_cols, n_dense_dim, n_dense_pad;
	slampp::CDevArray<double> d_A, d_rhs, d_L, d_Linv, d_w, d_cov;
	slampp::CSparseInverse *p_sinv = 0;    // sparse path: lists of the sparse inverse subset (slampp_hip_marginals), built on first use
	bool b_sinv_tried = false;
	slampp::CDevArray<double> d_Z;         // laid out like d_L
	slampp::CDevArray<double> d_Zd, d_Zd_work; // inverse of the dense top's Schur complement, and the copy of its factor that gets inverted
	slampp::CDevArray<int64_t> d_diag_zoff; // offset of every block column's diagonal block in it, original order
	slampp::CDevArray<int32_t> d_diag_dim;  // mixed block sizes: every caller's column's dimension
	slampp::CDevArray<int64_t> d_diag_out_off; // ... and where its block goes in the output
	slampp::CDevArray<int64_t> d_damp_off; // (offset of the diagonal block's first element, dimension) per block column: apply_damping
	bool b_damp_valid = false;
	slampp::CDevArray<int> d_flag;
	int *p_flag_shared = 0; // the reduced camera system's solver: the not-positive-definite flag of the solver it serves (not zeroed here)
	int *p_host_flag; // pinned

	// host entry points: pinned staging for Lambda's values and the right-hand side, a copy stream for the uploads
	// (slampp_hip_host_staging / slampp_hip_upload_values_async; callers' own arrays are moved through it in chunks)
	double *p_pin_values = 0, *p_pin_rhs = 0;
	size_t n_pin_values = 0, n_pin_rhs = 0;
	bool b_pin_values_registered = false, b_pin_rhs_registered = false; // malloc + hipHostRegister rather than hipHostMalloc
	// Round 6, a development knob now (SLAMPP_HIP_DEV_STAGING_DEFERRAL; staging.hip says what turned it around): the FIRST staging
	// of a handle whose caller hands over host arrays (option "staging_ahead") as a plain mapping -- not touched, not pinned: the
	// first call's gather first-touches it and its transfers go through the runtime's pageable path -- registered in place on a
	// thread of its own once the first answer is out (Register_Staging_Later / Join_Staging_Registration).  By default the
	// staging is pinned beside the analysis (slampp_hip_analyze's staging thread), as in round 5.
	bool b_pin_values_deferred = false, b_pin_rhs_deferred = false; // mapped, not registered (yet)
	bool b_staging_ever = false;       // this handle has had a staging before: no deferral again
	std::thread t_staging_registration; // joined wherever the staging is used, replaced or freed
	void Register_Staging_Later();      // staging.hip
	void Join_Staging_Registration();
	// The handle's streams come up on a thread of their own (round 6, capi.hip: device_bringup).  In a process that has used
	// the device for nothing else, hipStreamCreate takes 8.7 ms for each of the first three streams and 3.3 ms after that, the
	// first copy out of pinned memory 7.7 ms, the first launch out of a code object 0.4 ms and up (tools/micro/
	// first_launch_cost.hip) -- 30 ms that a caller's first solve spent waiting, all of it beside nothing: the analysis that
	// follows a handle's creation is tens of milliseconds of host work.  Whoever touches the streams first joins.
	// (round 6) the sparse analysis' record vectors -- 70 MB at C3 -- are freed on a thread behind the analysis (slampp::TTrash),
	// joined by the next analysis and by the destructor
	slampp::CTrashList analysis_trash;
	std::thread t_discard;
	void Join_Discard() { if(t_discard.joinable()) t_discard.join(); }
	std::thread t_bringup;
	std::mutex m_bringup;
	int n_bringup_status = SLAMPP_HIP_OK;
	int n_Join_Bringup() // thread-safe; the status of the bring-up (SLAMPP_HIP_OK or SLAMPP_HIP_ERR_DEVICE)
	{
		std::lock_guard<std::mutex> t_lock(m_bringup);
		if(t_bringup.joinable()) {
			const double t0 = slampp::wall_ms();
			t_bringup.join();
			if(getenv("SLAMPP_HIP_PLAN_TIMING"))
				fprintf(stderr, "[bring-up] waited for %.2f ms\n", slampp::wall_ms() - t0);
		}
		return n_bringup_status;
	}
	void Join_Bringup() // for the analysis: throws what guarded() reports as a device error
	{
		if(n_Join_Bringup() != SLAMPP_HIP_OK)
			throw slampp::CDeviceError("the handle's streams could not be created");
	}
	hipStream_t copy_stream = 0;
	hipEvent_t copy_done = 0;
	int64_t n_uploaded = 0; // values [0, n_uploaded) of the staging are already on their way to d_A
	void Require_Staging();  // throws
	void Free_Staging();
	void Upload_Values(const double *p_values); // throws; leaves `stream` waiting for the copy

	slampp::CSchurState *p_schur;
	// slampp_hip_create_multi: the members on the listed devices (group.hip); b_group_active = the current analysis is a
	// sharded one (Schur mode), and this handle itself holds the structure and the pinned staging only
	slampp::CDeviceGroup *p_group = 0;
	bool b_group_active = false;
	std::vector<int> group_devices; // the device list (empty: a single-device handle); the group comes up with the first Schur-mode analysis
	std::vector<std::pair<std::string, int64_t> > group_options; // options set so far, for members created later
	std::vector<slampp_hip_assembly*> assemblies; // live Lambda assemblies created from this solver

	slampp_hip_allreduce_fn p_allreduce;
	void *p_allreduce_context;
	// members of a device group (group.hip), option "schur_distributed": the dense reduced camera system is not summed on
	// every member and factored by each of them, but reduce-scattered by outer panels and factored by all of them together
	// (panel b by member b mod P, finished panels sent to the other members' copies, which then hold the whole factor);
	// in place on p_S, same contract as dense_cholesky() on the summed matrix.  Internal: not part of the C ABI.
	typedef int (*TDenseFactorFn)(void *p_context, double *p_S, int n_pad, int n, double *p_invdiag, int *p_flag, void *p_hip_stream);
	TDenseFactorFn p_dense_factor = 0;
	void *p_dense_factor_context = 0;
	int n_schur_distributed = 0; // option "schur_distributed"

	slampp_hip_times times;

	// optional per-phase device timing with HIP events on `stream` (option "profile")
	struct TPhaseRecord { int n_label; hipEvent_t start, stop; };
	int b_profile;
	std::vector<std::string> phase_names;
	std::vector<double> phase_ms;
	std::vector<int64_t> phase_count;
	std::vector<TPhaseRecord> phase_pending;
	std::vector<hipEvent_t> event_pool;
	int n_open_phase;
	void Phase_Begin(const char *p_s_label);
	void Phase_End();
	void Phase_Collect(); // after a stream synchronisation

	slampp_hip_solver();
	~slampp_hip_solver();
	void Free_Device();
	size_t n_Device_Bytes() const;
	void Analyze_Sparse();
	void Enqueue_Sparse(const double *p_values_dev, double *p_rhs_dev, bool b_factor, bool b_factor_only = false);
};


namespace slampp {

// Schur path entry points (schur.hip)
void schur_destroy(CSchurState *p);
CSchurState *schur_analyze(slampp_hip_solver &s); // throws
void schur_enqueue(slampp_hip_solver &s, const double *p_values_dev, double *p_rhs_dev); // throws
void schur_enqueue_marginal_poses(slampp_hip_solver &s, const double *p_values_dev, double *p_rhs_dev); // throws
void schur_marginals_sparse_launch(int DC, int DP, int64_t nc, int64_t np, const int64_t *ptr, const int64_t *cam_zoff,
	const int64_t *pair_ptr, const int64_t *pair_tab, const double *W, const double *Cinv, const double *Z, double *cam_cov,
	double *point_cov, hipStream_t stream); // schur_marginals.hip
void damping_enqueue(const int64_t *p_off_dim_dev, int64_t n_first, int64_t n_last, double f_alpha, double *p_values_dev,
	hipStream_t stream); // assembly.hip
void schur_enqueue_marginals(slampp_hip_solver &s, const double *p_values_dev, double *p_cam_cov_dev, double *p_point_cov_dev); // throws
size_t schur_device_bytes(const CSchurState *p);
void schur_invalidate_previous(CSchurState *p); // the kept reduced system no longer matches what the caller last solved
void schur_set_changed_points(slampp_hip_solver &s, const int64_t *p_points, int64_t n_points); // throws
void schur_fill_stats(const CSchurState *p, slampp_hip_stats &st);
bool schur_reduced_stats(const CSchurState *p, slampp_hip_stats &st);

// several devices behind one handle (group.hip)
CDeviceGroup *group_create(const int *p_device_ids, int n_devices); // throws
void group_destroy(CDeviceGroup *p_group);
int group_set_option(CDeviceGroup &g, const char *p_s_name, int64_t n_value);
void group_analyze(slampp_hip_solver &r_front, int64_t n_cut); // throws
int group_factor_solve(slampp_hip_solver &r_front, const double *p_values, double *p_rhs_inout);
// the front handle's pinned staging as every member's device sees it: registered host memory, and one double makes the
// trip to the member's device and back; SLAMPP_HIP_ERR_DEVICE with the member and the reason in the front's error otherwise
int group_check_staging(slampp_hip_solver &r_front, double *p_values, double *p_rhs);
int group_solve_marginal_poses(slampp_hip_solver &r_front, const double *p_values, double *p_rhs_inout);
int group_schur_marginals(slampp_hip_solver &r_front, const double *p_values, double *p_cam_cov, double *p_point_cov);
int group_free_memory(CDeviceGroup &g);
void group_fill_stats(CDeviceGroup &g, slampp_hip_stats &r_stats);
const char *group_exchange_name(const CDeviceGroup &g);
int group_member_num(const CDeviceGroup &g);
int64_t group_exchange_count(const CDeviceGroup &g);
slampp_hip_solver *group_member(CDeviceGroup &g, int n_member);
void shard_bounds(int64_t n_bcols, int64_t n_cut, const int64_t *p_bcol_ptr, int n_world, std::vector<int64_t> &r_bounds);

// Lambda assembly (assembly.hip)
CAssemblyState *assembly_setup(slampp_hip_solver &s, int64_t n_edges, const int64_t *v0, const int64_t *v1, int rd); // throws
void assembly_destroy(CAssemblyState *p);
void assembly_enqueue(CAssemblyState &a, const double *J0, const double *J1, const double *Si, const double *err,
	const double *wgt, int64_t n_unary_vertex, const double *p_unary_factor, const double *p_unary_error,
	double *values_out, double *eta_out, int b_accumulate); // throws
size_t assembly_device_bytes(const CAssemblyState *p);

} // namespace slampp

// staging.hip: the transfers the host entry points of capi.hip are made of
void Parallel_Copy(double *p_dst, const double *p_src, size_t n);             // a host copy on a few threads (the solution back into the caller's array)
void Upload_Rhs_And_Join(slampp_hip_solver &s, const double *p_rhs);          // eta through the staging; the main stream then waits for the copy stream
void Upload_Values_And_Join(slampp_hip_solver &s, const double *p_values);    // Lambda's packed values likewise
