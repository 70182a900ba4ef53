// sparse_setup.hip -- analysis of the sparse block path: plan, task packages of the kernels, uploads (cold path)
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
#include <unordered_map>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <sys/mman.h>

using namespace slampp;

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
	Join_Bringup(); // (solver.h: a fresh handle's streams)
	d_refine_map.Upload(map, stream);
	d_refined.Alloc(map.size());
	SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // map lives on this stack frame
}

// index ranges on a few threads (the record loops of the cold path: every entry written once, from the plan alone)
template <class F>
static void Parallel_Ranges(int64_t n, int64_t n_min_per_thread, F f, int n_max_threads = 4)
{
	n_max_threads = std::min(n_max_threads, std::max(dev_knob("SLAMPP_HIP_DEV_SETUP_THREADS", n_max_threads), 1)); // (development knob, plan.h)
	const int n_threads = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_max_threads, std::max(1u, std::thread::hardware_concurrency())), n / std::max<int64_t>(n_min_per_thread, 1))));
	if(n_threads <= 1) {
		f(int64_t(0), n);
		return;
	}
	std::vector<std::thread> threads;
	std::exception_ptr p_error;
	std::mutex t_mutex;
	for(int t = 0; t < n_threads; ++ t) {
		const int64_t b = n * t / n_threads, e = n * (t + 1) / n_threads;
		auto job = [&, b, e]() {
			try {
				f(b, e);
			} catch(...) {
				std::lock_guard<std::mutex> t_lock(t_mutex);
				p_error = std::current_exception();
			}
		};
		if(t + 1 < n_threads)
			threads.emplace_back(job);
		else
			job();
	}
	for(size_t t = 0; t < threads.size(); ++ t)
		threads[t].join();
	if(p_error)
		std::rethrow_exception(p_error);
}

void slampp_hip_solver::Analyze_Sparse()
{
	Join_Discard(); // (the previous analysis' arrays)
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
		opt.task_wide_min = std::max(dev_knob("SLAMPP_HIP_DEV_WIDE_MIN", n_wide_min_tasks), 1); // (development aid: overrides the option "wide_min_tasks")
		opt.task_max_cols = int(PANEL_COLS);
		opt.task_max_blocks = panel_slot_cap(n_dim0);
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
	const bool b_small = P.n < 8192; // (small systems -- FastL's parts of R, reduced camera systems -- do their host work on this thread: a thread is ~0.1 ms to start, and threads that spin under a CPU quota cost a scheduler period now and then)
	if(b_small)
		Build_Simt();
	else {
		t_simt_thread.t = std::thread([this, &p_simt_error]() {
			try {
				Build_Simt();
			} catch(...) {
				p_simt_error = std::current_exception();
			}
		});
	}

	if(P.cs_new[P.n] >= INT32_MAX)
		throw std::domain_error("systems with 2^31 or more scalar unknowns are not supported by the sparse path");

	// packed device records (see sparse_kernels.h)
	const int32_t n_sched = int32_t(P.task_cols.size()); // all columns but those of the dense top
	raw_vector<TColDesc> cols(n_sched); // in schedule order (raw_vector: not zero-filled -- solver.h; every record is written in full below)
	raw_vector<TBlkDesc> blks(n_lblocks);
	raw_vector<longlong2> pairs(P.pa.size());
	raw_vector<TRowEnt> rents(P.rblk.size());
	if(P.loff[n_lblocks] >= (int64_t(1) << 48))
		throw std::domain_error("the factor has 2^48 or more values");
	// everything of one column: its blocks, their update pairs (stored block by block), the row entries of its diagonal block
	auto Fill_Column = [&](int32_t j) {
		for(int64_t k = P.lptr[j]; k < P.lptr[j + 1]; ++ k) {
			TBlkDesc &b = blks[k];
			const int64_t np = P.pptr[k + 1] - P.pptr[k];
			if(np >= (int64_t(1) << 24))
				throw std::domain_error("a factor block has 2^24 or more updates: use the dense path");
			b.loff = P.loff[k];
			b.asrc = (P.asrc[k] < 0)? -1 : P.asrc[k] * 2 + P.atrans[k];
			if(k == P.lptr[j] && b.asrc >= 0)
				b.asrc |= 1; // diagonal blocks are read transposed: the lower triangle of the factor block then comes from the upper triangle of Lambda's block, the one the reference's solvers consume
			b.p0 = P.pptr[k];
			b.np_di = uint32_t(np) | (uint32_t(P.dim[P.lrow[k]]) << 24);
			b.xcs = int32_t(P.cs_new[P.lrow[k]]);
			const int64_t n_pos = std::min<int64_t>(k - P.lptr[j], 255); // position of the target block in its column
			for(int64_t e = P.pptr[k]; e < P.pptr[k + 1]; ++ e) {
				const int64_t dc = P.dim[P.blk_col[P.pa[e]]];
				pairs[e].x = P.loff[P.pa[e]] | (n_pos << 48) | (dc << 56);
				pairs[e].y = P.loff[P.pb[e]];
			}
		}
		for(int64_t e = P.rptr[j]; e < P.rptr[j + 1]; ++ e) {
			const int32_t c = P.blk_col[P.rblk[e]];
			rents[e].off = P.loff[P.rblk[e]];
			rents[e].ycs = int32_t(P.cs_new[c]);
			rents[e].dc = P.dim[c];
		}
	};
	auto Fill_Scheduled = [&](int64_t i_begin, int64_t i_end) {
		for(int64_t i = i_begin; i < i_end; ++ i) {
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
			Fill_Column(j);
		}
	};
	// Round 6: the columns of the separator stages first (a tenth of the records at C3) -- the packages of their tasks, the longest
	// single-threaded piece of the analysis, are built from those on a thread of their own while this one writes the rest
	const int64_t n_upper_begin = (b_small || P.stage_ptr.size() < 2 || P.stage_ptr[1] - P.stage_ptr[0] <= 512)? 0 : // (few leaf tasks: they may get panel packages too)
		 P.task_ptr[size_t(P.stage_ptr[size_t(std::min(n_bottom_stages, int(P.stage_ptr.size()) - 1))])];
	Fill_Scheduled(n_upper_begin, n_sched);
	auto Fill_Rest = [&]() {
		Parallel_Ranges(n_upper_begin, 4096, Fill_Scheduled, 8);
		for(int32_t j = 0; j < P.n; ++ j) { // (the blocks of the dense top's columns are not scheduled; their records are read all the same)
			if(P.dense_pos[j] >= 0)
				Fill_Column(j);
		}
	};
	// panel packages for the separator stages (panel_kernel.hip): a task qualifies if its columns' blocks are one range of
	// the factor and everything fits the kernel's LDS; the updates it receives from earlier stages go to the lists of
	// panel_update_kernel, block by block
	raw_vector<longlong2> panel_pkg; // (raw_vector: the big arrays of the analysis live in mappings of the library's own, on huge pages -- solver.h)
	std::vector<int64_t> panel_off, panel_out_off; // (panel_out_off: per package the offset of its hand-up list, or -1)
	std::vector<int32_t> panel_units; // per package its size in 16-byte units: what the launch order of a stage goes by
	int64_t n_handup_doubles = 0;
	std::vector<int32_t> panel_rest;
	raw_vector<TUpdSlot> upd_slots;
	raw_vector<TUpdEnt> upd_ents;
	// (a second pass, without hand-ups, if a stage's hand-up list would take its workgroups past the LDS of a CU: the list
	// rides in the dynamic LDS request on top of the task's image, and nothing else bounds its length -- advisor, round 4)
	{
		// room for everything up front (address space only: pages come when they are written): the lists used to grow by
		// doubling, each step an mmap, a copy and a munmap of megabytes -- and a munmap interrupts every thread of the process
		// (the TLB shootdown), of which the analysis runs a dozen at this point (round 6)
		const size_t n_upper_tasks = (P.stage_ptr.size() > 1)? size_t(P.stage_ptr.back() - P.stage_ptr[size_t(std::min(n_bottom_stages, int(P.stage_ptr.size()) - 1))]) : 0;
		const size_t n_tasks_cap = (n_upper_begin == 0)? P.task_ptr.size() : n_upper_tasks;
		size_t n_upper_blocks = 0, n_upper_pairs = 0;
		for(int64_t i = n_upper_begin; i < n_sched; ++ i) {
			n_upper_blocks += size_t(cols[i].nb);
			n_upper_pairs += size_t(cols[i].np) + size_t(cols[i].nr);
		}
		panel_pkg.reserve(n_tasks_cap * 64 + 3 * n_upper_blocks + n_upper_pairs + 64 * PANEL_W + 4096);
		upd_slots.reserve(n_upper_blocks + 16);
		upd_ents.reserve(n_upper_pairs + 16);
		panel_off.reserve(n_tasks_cap + 16);
		panel_out_off.reserve(n_tasks_cap + 16);
	}
	auto Build_Panel_Packages = [&]() {
	for(bool b_hand_up_allowed = n_panel_handup != 0;;) {
	panel_pkg.clear();
	panel_off.clear();
	panel_out_off.clear();
	panel_units.clear();
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
				bool b_fits = n_cols >= 1 && n_cols <= int(PANEL_COLS);
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
				b_fits = b_fits && n_slots <= n_slot_cap;
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
				b_fits = b_fits && n_units <= size_t(PANEL_UNITS);
				n_stage_max_slots = std::max(n_stage_max_slots, n_slots);
				n_stage_max_units = std::max(n_stage_max_units, int64_t(n_units));
				n_stage_rest += !b_fits;
				if(!b_fits) {
					if(n_slots <= n_slot_cap && n_cols >= 1 && n_cols <= int(PANEL_COLS))
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
				panel_units.push_back(int32_t(n_units));
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
	for(size_t i = 0; i < panel_cfg.size() && !panel_off.empty(); ++ i) {
		b_lds_fits = b_lds_fits && size_t(panel_lds(P.max_dim, true, panel_cfg[i]).TOTAL) * sizeof(double) <= PANEL_LDS_BUDGET;
		if(b_timing && i + 1 < panel_ptr.size() && panel_ptr[i + 1] > panel_ptr[i]) {
			const TPanelLds l = panel_lds(P.max_dim, true, panel_cfg[i]);
			fprintf(stderr, "[setup] stage %d panel launch: %d waves a task, LDS %zu bytes (package %d, blocks %d, inverses + tiles %d, operands %d, fresh %d, hand-up list %d doubles)\n",
				int(i), panel_cfg[i].n_waves, size_t(l.TOTAL) * sizeof(double), l.IMAGE, l.VEC - l.IMAGE, l.OPS - l.VEC, l.YV - l.OPS, l.OUT - l.YV, l.TOTAL - l.OUT);
		}
	}
	if(b_lds_fits || !b_hand_up_allowed)
		break;
	b_hand_up_allowed = false;
	}
	// The launch order inside a stage (round 6): workgroups start in the order of their index, and a launch that holds its
	// tasks more than once over (C3's 2 066-task stage: five workgroups a CU by their LDS, two rounds) ends when the last round's
	// longest task does.  In the plan's order long and short tasks are mixed, so both rounds last as long as a long task; with the
	// big packages first the last round is made of short ones.  Nothing on the host refers to a package by its position
	// any more at this point; the device reads pkg_off[blockIdx.x] and out_off[blockIdx.x] only.
	if(dev_knob("SLAMPP_HIP_DEV_PANEL_ORDER", 1) != 0) { // (development aid, plan.h: 0 = the plan's order)
		std::vector<int32_t> order;
		std::vector<int64_t> off_sorted, out_off_sorted;
		for(size_t st = 0; st + 1 < panel_ptr.size(); ++ st) {
			const int32_t n_first = panel_ptr[st], n_num = panel_ptr[st + 1] - n_first;
			if(n_num < 2)
				continue;
			order.resize(size_t(n_num));
			for(int32_t i = 0; i < n_num; ++ i)
				order[size_t(i)] = n_first + i;
			std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return panel_units[size_t(a)] > panel_units[size_t(b)]; });
			off_sorted.resize(size_t(n_num));
			out_off_sorted.resize(size_t(n_num));
			for(int32_t i = 0; i < n_num; ++ i) {
				off_sorted[size_t(i)] = panel_off[size_t(order[size_t(i)])];
				out_off_sorted[size_t(i)] = panel_out_off[size_t(order[size_t(i)])];
			}
			std::copy(off_sorted.begin(), off_sorted.end(), panel_off.begin() + n_first);
			std::copy(out_off_sorted.begin(), out_off_sorted.end(), panel_out_off.begin() + n_first);
		}
	}
	};
	std::exception_ptr p_panel_error;
	struct TJoinPanel { std::thread t; ~TJoinPanel() { if(t.joinable()) t.join(); } } t_panel_thread;
	if(b_small)
		Build_Panel_Packages();
	else {
		t_panel_thread.t = std::thread([&]() {
			try {
				const double t_panel = wall_ms();
				Build_Panel_Packages();
				if(b_timing)
					fprintf(stderr, "[setup] %-12s %8.2f ms on their own thread\n", "panel pkgs", wall_ms() - t_panel);
			} catch(...) {
				p_panel_error = std::current_exception();
			}
		});
	}
	Fill_Rest();
	// column packages for the upper stages (see sparse_kernels.h); the limits are those of factor_stage_kernel's staged path
	raw_vector<longlong2> pkg;
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
	// the records go to the device beside the host work that follows (the packages of the separator tasks: ~10 ms each at C3,
	// and an upload from a std::vector is a staged copy that keeps its thread): a thread of its own, joined before the rest
	// of the uploads.  The vectors it reads are not written from here on.
	std::exception_ptr p_upload_error;
	struct TJoinUpload { std::thread t; ~TJoinUpload() { if(t.joinable()) t.join(); } } t_upload_thread;
	auto Upload_Records = [&]() {
		try {
			Join_Bringup(); // (a handle fresh from slampp_hip_create: its streams came up beside the plan and the records -- solver.h)
			SLAMPP_HIP_CHECK(hipSetDevice(n_device));
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
		} catch(...) {
			p_upload_error = std::current_exception();
		}
	};
	if(b_small) {
		Upload_Records();
		SETUP_PHASE("record uploads");
	} else
		t_upload_thread.t = std::thread(Upload_Records);
	Join_Bringup(); // (this thread's own uploads and allocations begin below)
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
		SETUP_PHASE("dense records");
		d_dense_blks.Upload(dense_blks, stream);
		d_dense_blk_loff.Upload(dense_blk_loff, stream);
		SETUP_PHASE("dense upload 1");
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
			SETUP_PHASE("dense upload 2");
			SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // the vectors live in this scope
		}
		SETUP_PHASE("dense lists");
		d_dense.Alloc(size_t(n_dense_pad) * n_dense_pad);
		b_dense_clean = false;
		d_dense_invdiag.Alloc(size_t(n_dense_pad / dense_NB) * dense_NB * dense_NB);
		d_dense_z.Alloc(n_dense_pad);
		d_dense_x.Alloc(n_dense_pad);
		SETUP_PHASE("dense allocs");
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
	SETUP_PHASE("tile schedule");
	if(t_panel_thread.t.joinable())
		t_panel_thread.t.join();
	if(p_panel_error)
		std::rethrow_exception(p_panel_error);
	SETUP_PHASE("packages");
	d_panel_upd_slots.Upload(upd_slots, stream);
	d_panel_upd_ents.Upload(upd_ents, stream);
	d_panel_pkg.Upload(panel_pkg, stream);
	d_panel_off.Upload(panel_off, stream);
	d_panel_out_off.Upload(panel_out_off, stream);
	d_handup.Alloc(size_t(std::max<int64_t>(n_handup_doubles, int64_t(P.max_dim) * P.max_dim + 8))); // (every wave of a fused panel launch prefetches one block + 8 from offset 0, hand-ups or not)
	d_panel_rest.Upload(panel_rest, stream);
	if(t_upload_thread.t.joinable())
		t_upload_thread.t.join();
	if(p_upload_error)
		std::rethrow_exception(p_upload_error);
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
		if(t_simt_thread.t.joinable())
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
	if(!b_small) {
		// the record vectors are on the device: giving their memory back to the system (70 MB at C3: 3 - 4 ms of page-table
		// work) is nobody's critical path -- a thread does it behind the analysis' return (solver.h: TTrash, t_discard)
		Join_Discard();
		Discard_Later(analysis_trash, cols); Discard_Later(analysis_trash, blks); Discard_Later(analysis_trash, pairs);
		Discard_Later(analysis_trash, rents); Discard_Later(analysis_trash, pkg); Discard_Later(analysis_trash, task_pkg);
		Discard_Later(analysis_trash, panel_pkg); Discard_Later(analysis_trash, upd_slots); Discard_Later(analysis_trash, upd_ents);
		try {
			t_discard = std::thread([this]() { analysis_trash.clear(); host_pool_release(); });
		} catch(std::system_error&) {
			analysis_trash.clear();
			host_pool_release();
		}
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
	raw_vector<int64_t> &tab = simt_host_tab;
	size_t n_tab_size = 0, n_bwd_tab_size = 0; // (the tables are laid out first and made in one piece after the layout of a stage: round 6 --
	// grown chunk by chunk, zero-filled and moved as they grew, they were most of the 5 ms the layout took at C3)
	// Round 6: the tasks' programs on several threads (a task's program depends on nothing but the plan), the shapes told
	// apart by a hash of the program with one full comparison per task against its shape's first member instead of a
	// std::map keyed by the programs (16 000 insertions of 200-word keys at C3), the tables of a shape's chunks on several
	// threads again.  Shapes, chunks and tables come out in the order the map gave them (programs in lexicographic order,
	// the tasks of a shape ascending).
	struct TTask { int32_t n_task; bool b_fits; uint64_t n_hash; std::vector<int32_t> prog, ops, ys; };
	simt_chunk_ptr.push_back(0);
	simt_rest_ptr.push_back(0);
	const size_t W = size_t(n_simt_width);
	for(int s = 0; s < n_bottom_stages && s < n_stages && s < n_simt_stages; ++ s) {
		const int32_t t0 = P.stage_ptr[s], n_stage_tasks = P.stage_ptr[s + 1] - P.stage_ptr[s];
		const bool b_simt_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
		double t_simt_phase = wall_ms();
		auto Simt_Phase = [&](const char *p_s_name) { if(b_simt_timing) { const double t_ = wall_ms();
			fprintf(stderr, "[shapes] stage %d %-12s %8.2f ms\n", s, p_s_name, t_ - t_simt_phase); t_simt_phase = t_; } };
		std::vector<TTask> tasks_all(size_t(std::max(n_stage_tasks, 0)));
		// (the threads' index arrays: made here, before the threads, and given back after them -- an array of megabytes made
		// and freed by a thread is an mmap and a munmap while a dozen other threads of the analysis run: see the panel packages)
		enum { SIMT_THREADS = 8 };
		raw_vector<int32_t> index_pool((P.lrow.size() + size_t(P.n)) * SIMT_THREADS); // (every thread fills its own slice)
		std::atomic<int> n_next_slice(0);
		Parallel_Ranges(n_stage_tasks, 512, [&](int64_t n_b, int64_t n_e) {
			const size_t n_slice = size_t(n_next_slice.fetch_add(1)) % SIMT_THREADS;
			int32_t *op_index = &index_pool[(P.lrow.size() + size_t(P.n)) * n_slice], *y_index = op_index + P.lrow.size();
			std::fill(op_index, op_index + P.lrow.size() + size_t(P.n), -1);
			std::vector<int32_t> touch, body;
			for(int64_t n_i = n_b; n_i < n_e; ++ n_i) {
				const int32_t t = t0 + int32_t(n_i);
				TTask &tt = tasks_all[size_t(n_i)];
				std::vector<int32_t> &prog = tt.prog;
				prog.assign(4, 0);
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
					touch.clear();
					body.clear();
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
				// (round 6) behind the program proper: for every block below a diagonal, which of the task's columns its row is, or
				// -1 for a row outside the task -- what the backward kernel keeps x of in LDS.  Implied by the program (a block whose
				// row is column m of the task is a row entry of m), and part of the shape's key all the same
				for(int64_t i = P.task_ptr[t]; i < P.task_ptr[t + 1] && b_fits; ++ i) {
					const int32_t j = P.task_cols[i];
					for(int64_t k = P.lptr[j] + 1; k < P.lptr[j + 1]; ++ k) {
						int32_t n_local = -1;
						for(int64_t i2 = i + 1; i2 < P.task_ptr[t + 1] && n_local < 0; ++ i2) {
							if(P.task_cols[i2] == P.lrow[k])
								n_local = int32_t(i2 - P.task_ptr[t]);
						}
						prog.push_back(n_local);
					}
				}
				prog[0] = n_cols;
				prog[1] = n_blocks;
				prog[2] = int32_t(tt.ops.size());
				prog[3] = int32_t(tt.ys.size());
				tt.b_fits = b_fits && size_t(4 * n_cols + n_blocks) + tt.ops.size() + tt.ys.size() <= MAX_TABLE_BYTES / (8 * W); // (the table is staged in LDS)
				uint64_t h = 0x9e3779b97f4a7c15ull ^ prog.size();
				for(size_t k = 0; k < prog.size(); ++ k) {
					h ^= uint64_t(uint32_t(prog[k])) + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
					h *= 0xff51afd7ed558ccdull;
				}
				tt.n_hash = h;
			}
		}, 8);
		Simt_Phase("programs");
		// shapes: tasks of one hash whose programs are the same (compared in full against the shape's first task)
		std::vector<std::vector<int32_t> > groups; // indices into tasks_all, ascending
		{
			std::unordered_map<uint64_t, std::vector<int32_t> > by_hash; // hash -> the shapes that have it
			for(int32_t n_i = 0; n_i < n_stage_tasks; ++ n_i) {
				const TTask &tt = tasks_all[size_t(n_i)];
				if(!tt.b_fits) {
					rest.push_back(tt.n_task);
					continue;
				}
				std::vector<int32_t> &r_shapes = by_hash[tt.n_hash];
				int32_t n_group = -1;
				for(size_t k = 0; k < r_shapes.size() && n_group < 0; ++ k) {
					if(tasks_all[size_t(groups[size_t(r_shapes[k])][0])].prog == tt.prog)
						n_group = r_shapes[k];
				}
				if(n_group < 0) {
					n_group = int32_t(groups.size());
					groups.push_back(std::vector<int32_t>());
					r_shapes.push_back(n_group);
				}
				groups[size_t(n_group)].push_back(n_i);
			}
			std::sort(groups.begin(), groups.end(), [&](const std::vector<int32_t> &r_a, const std::vector<int32_t> &r_b) {
				return tasks_all[size_t(r_a[0])].prog < tasks_all[size_t(r_b[0])].prog; });
		}
		Simt_Phase("grouping");
		// where every shape's program, chunks and tables go
		struct TChunkJob { int32_t n_group; size_t n_first; int n_fields, n_bwd_fields; int64_t n_tab_off, n_bwd_tab_off; };
		std::vector<TChunkJob> jobs;
		int32_t n_stage_lds = 0, n_stage_bwd_lds = 0;
		for(size_t g = 0; g < groups.size(); ++ g) {
			const std::vector<int32_t> &r_members = groups[g];
			const std::vector<int32_t> &prog = tasks_all[size_t(r_members[0])].prog;
			if(r_members.size() < MIN_GROUP) {
				for(int32_t n_i : r_members)
					rest.push_back(tasks_all[size_t(n_i)].n_task);
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
			n_stage_bwd_lds = std::max(n_stage_bwd_lds, int32_t((n_bwd_fields + n_cols * P.max_dim) * W * 8)); // (the table, and x of the task's own columns)
			simt_host_bwd_prog.push_back(n_cols);
			simt_host_bwd_prog.push_back(n_blocks - n_cols);
			{
				const TTask &tt = tasks_all[size_t(r_members[0])];
				for(int64_t i = P.task_ptr[tt.n_task]; i < P.task_ptr[tt.n_task + 1]; ++ i)
					simt_host_bwd_prog.push_back(int32_t(P.lptr[P.task_cols[i] + 1] - P.lptr[P.task_cols[i]]));
				// which of the task's columns every below-diagonal block's row is (the tail of the forward program: see there)
				simt_host_bwd_prog.insert(simt_host_bwd_prog.end(), prog.end() - (n_blocks - n_cols), prog.end());
			}
			for(size_t n_first = 0; n_first < r_members.size(); n_first += W) {
				const size_t n_in_chunk = std::min<size_t>(W, r_members.size() - n_first);
				TSimtChunk ch;
				ch.prog_off = n_prog_off;
				ch.n_tasks = int32_t(n_in_chunk);
				ch.tab_off = int64_t(n_tab_size);
				chunks.push_back(ch);
				TSimtChunk ch_bwd;
				ch_bwd.prog_off = n_bwd_prog_off;
				ch_bwd.n_tasks = int32_t(n_in_chunk);
				ch_bwd.tab_off = int64_t(n_bwd_tab_size);
				simt_host_bwd_chunks.push_back(ch_bwd);
				TChunkJob t_job = {int32_t(g), n_first, n_fields, n_bwd_fields, ch.tab_off, ch_bwd.tab_off};
				jobs.push_back(t_job);
				n_tab_size += size_t(n_fields) * W;
				n_bwd_tab_size += size_t(n_bwd_fields) * W;
			}
		}
		tab.resize(n_tab_size); // (raw_vector: what was there stays, the new part is written in full below)
		simt_host_bwd_tab.resize(n_bwd_tab_size);
		Simt_Phase("layout");
		Parallel_Ranges(int64_t(jobs.size()), 32, [&](int64_t n_b, int64_t n_e) {
			for(int64_t n_job = n_b; n_job < n_e; ++ n_job) {
				const TChunkJob &r_job = jobs[size_t(n_job)];
				const std::vector<int32_t> &r_members = groups[size_t(r_job.n_group)];
				const size_t n_first = r_job.n_first, n_in_chunk = std::min<size_t>(W, r_members.size() - n_first);
				const int n_fields = r_job.n_fields, n_bwd_fields = r_job.n_bwd_fields;
				int64_t *p_tab = &tab[size_t(r_job.n_tab_off)];
				for(int n_lane = 0; n_lane < int(W); ++ n_lane) {
					const TTask &tt = tasks_all[size_t(r_members[n_first + std::min<size_t>(n_lane, n_in_chunk - 1)])]; // spare lanes repeat the last task
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
				int64_t *p_bwd = &simt_host_bwd_tab[size_t(r_job.n_bwd_tab_off)];
				for(int n_lane = 0; n_lane < int(W); ++ n_lane) {
					const TTask &tt = tasks_all[size_t(r_members[n_first + std::min<size_t>(n_lane, n_in_chunk - 1)])];
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
		}, 8);
		Simt_Phase("tables");
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
	{ raw_vector<int64_t> e; simt_host_bwd_tab.swap(e); }
	{ std::vector<TSimtChunk> e; simt_host_chunks.swap(e); }
	{ std::vector<int32_t> e0, e1; simt_host_prog.swap(e0); simt_host_rest.swap(e1); }
	{ raw_vector<int64_t> e; simt_host_tab.swap(e); }
	if(getenv("SLAMPP_HIP_PLAN_TIMING")) {
		for(size_t s = 0; s + 1 < simt_chunk_ptr.size(); ++ s) {
			fprintf(stderr, "[setup] stage %zu: %d tasks -> %d chunks of 64 lanes, %d tasks left to the wave-per-task kernel\n", s,
				P.stage_ptr[s + 1] - P.stage_ptr[s], simt_chunk_ptr[s + 1] - simt_chunk_ptr[s], simt_rest_ptr[s + 1] - simt_rest_ptr[s]);
		}
	}
}

