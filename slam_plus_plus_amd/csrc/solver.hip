// solver.hip -- the handle: construction, device memory, phase timing.  Host code only; kernels live in
// sparse_kernels.hip / schur.hip / dense_chol.hip.
// (one of the translation units solver.hip was split into in round 5: solver.hip the handle and its device memory,
// staging.hip pinned staging and uploads, sparse_setup.hip the analysis of the sparse block path, sparse_enqueue.hip its launches,
// capi.hip the C ABI of include/slampp_hip.h)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <pthread.h>
#include "solver.h"
#include <unordered_map>
#include <sys/mman.h>
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

namespace slampp {
thread_local bool g_b_keep_device_memory = false;
}

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

// ---- the mappings behind raw_vector (solver.h) ----
namespace slampp {

namespace {

struct THostBlock { void *p_map; size_t n_map_bytes, n_bytes; }; // the mapping as mmap() gave it, and the aligned part handed out

struct THostPool {
	std::mutex t_mutex;
	std::vector<std::pair<void*, THostBlock> > free_blocks; // (aligned address, block) of the mappings nobody holds
	std::unordered_map<void*, THostBlock> held;            // aligned address -> block
};

THostPool &r_Host_Pool()
{
	static THostPool *p_pool = new THostPool(); // (never destroyed: containers of other static objects may be freed after it would be)
	return *p_pool;
}

} // anonymous namespace

void *host_pool_alloc(size_t n_bytes)
{
	const size_t n_huge = size_t(2) << 20;
	const size_t n_need = (n_bytes + n_huge - 1) / n_huge * n_huge;
	THostPool &r_pool = r_Host_Pool();
	{
		std::lock_guard<std::mutex> t_lock(r_pool.t_mutex);
		size_t n_best = size_t(-1);
		for(size_t i = 0; i < r_pool.free_blocks.size(); ++ i) { // the smallest block that holds it and is not more than twice as large
			const size_t n_size = r_pool.free_blocks[i].second.n_bytes;
			if(n_size >= n_need && n_size <= 2 * n_need && (n_best == size_t(-1) || n_size < r_pool.free_blocks[n_best].second.n_bytes))
				n_best = i;
		}
		if(n_best != size_t(-1)) {
			const std::pair<void*, THostBlock> t_block = r_pool.free_blocks[n_best];
			r_pool.free_blocks[n_best] = r_pool.free_blocks.back();
			r_pool.free_blocks.pop_back();
			r_pool.held[t_block.first] = t_block.second;
			return t_block.first;
		}
	}
	THostBlock t_block;
	t_block.n_map_bytes = n_need + n_huge;
	t_block.n_bytes = n_need;
	t_block.p_map = mmap(0, t_block.n_map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if(t_block.p_map == MAP_FAILED)
		throw std::bad_alloc();
	void *p = (void*)((uintptr_t(t_block.p_map) + n_huge - 1) / n_huge * n_huge);
	if(!dev_knob_set("SLAMPP_HIP_DEV_NO_HUGE_PAGES")) // (development aid, plan.h)
		(void)madvise(p, n_need, MADV_HUGEPAGE); // (refused or ignored where the system has them off: 4 KB pages then, as before)
	try {
		std::lock_guard<std::mutex> t_lock(r_pool.t_mutex);
		r_pool.held[p] = t_block;
	} catch(std::bad_alloc&) {
		(void)munmap(t_block.p_map, t_block.n_map_bytes);
		throw;
	}
	return p;
}

void host_pool_free(void *p) noexcept
{
	if(!p)
		return;
	THostPool &r_pool = r_Host_Pool();
	std::lock_guard<std::mutex> t_lock(r_pool.t_mutex);
	std::unordered_map<void*, THostBlock>::iterator p_it = r_pool.held.find(p);
	if(p_it == r_pool.held.end())
		return; // (not ours: cannot happen -- CNoInitAlloc decides by the same size on both ways)
	try {
		r_pool.free_blocks.push_back(std::make_pair(p, p_it->second));
	} catch(std::bad_alloc&) {
		(void)munmap(p_it->second.p_map, p_it->second.n_map_bytes);
	}
	r_pool.held.erase(p_it);
}

void host_pool_release() noexcept
{
	std::vector<std::pair<void*, THostBlock> > blocks;
	{
		THostPool &r_pool = r_Host_Pool();
		std::lock_guard<std::mutex> t_lock(r_pool.t_mutex);
		blocks.swap(r_pool.free_blocks);
	}
	for(size_t i = 0; i < blocks.size(); ++ i)
		(void)munmap(blocks[i].second.p_map, blocks[i].second.n_map_bytes);
}

} // ~slampp

slampp_hip_solver::~slampp_hip_solver()
{
	(void)n_Join_Bringup();
	Join_Discard();
	host_pool_release();
	Free_Device();
	for(size_t i = 0; i < phase_pending.size(); ++ i) {
		(void)hipEventDestroy(phase_pending[i].start);
		(void)hipEventDestroy(phase_pending[i].stop);
	}
	for(size_t i = 0; i < event_pool.size(); ++ i)
		(void)hipEventDestroy(event_pool[i]);
	if(p_host_flag)
		(void)hipHostFree(p_host_flag);
	if(p_host_batch_flag)
		(void)hipHostFree(p_host_batch_flag);
	Free_Staging();
	if(copy_done)
		(void)hipEventDestroy(copy_done);
	if(copy_stream)
		(void)hipStreamDestroy(copy_stream);
	if(stream)
		(void)hipStreamDestroy(stream);
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
