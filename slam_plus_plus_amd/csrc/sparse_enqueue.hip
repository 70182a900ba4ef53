// sparse_enqueue.hip -- the launches of one numeric factorization + substitutions of the sparse block path (warm path)
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
		(P.max_dim % 2 != 0 || ((reinterpret_cast<uintptr_t>(p_rhs_dev) & 15) == 0 && t_batch.b % 2 == 0));
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
			(((reinterpret_cast<uintptr_t>(p_values_dev) | reinterpret_cast<uintptr_t>(p_rhs_dev)) & 15) == 0 && (t_batch.a | t_batch.b) % 2 == 0));
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
					d_panel_upd_ents.p(), p_values_dev, d_L.p(), p_rhs_dev, d_w.p(), stream, t_batch);
			const int n_next = (s + 1 < n_stages && panel_ride[s + 1] == 1)? panel_upd_ptr[s + 2] - panel_upd_ptr[s + 1] : 0;
			if(!launch_factor_panel(P.max_dim, b_panel_fused, (n_panel_rows < 0)? P.max_dim >= 6 : n_panel_rows != 0, panel_cfg[s], d_panel_pkg.p(), d_panel_off.p() + panel_ptr[s],
				d_panel_out_off.p() + panel_ptr[s], n_panels,
				d_panel_upd_slots.p() + ((n_next > 0)? panel_upd_ptr[s + 1] : 0), n_next, d_panel_upd_ents.p(), p_values_dev, p_rhs_dev,
				d_L.p(), d_Linv.p(), d_w.p(), d_handup.p(), p_flag, stream, dplan.p_timing, t_batch))
				throw CDeviceError("panel launch refused: block size or LDS request outside what the analysis planned for");
			if(panel_rest_ptr[s + 1] > panel_rest_ptr[s]) {
				TDevPlan t_rest = dplan;
				t_rest.task_map = d_panel_rest.p();
				launch_factor_stage(t_rest, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), panel_rest_ptr[s],
					panel_rest_ptr[s + 1] - panel_rest_ptr[s], b_bottom, p_flag, stream, t_batch);
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
					p_values_dev, d_L.p(), b_store_linv? d_Linv.p() : 0, p_rhs_dev, d_w.p(), p_flag, stream, dplan.p_timing, t_batch);
				b_leaf_linv_valid = b_leaf_linv_valid && b_store_linv;
				if(n_rest > 0) {
					TDevPlan t_rest = dplan;
					t_rest.task_map = d_simt_rest.p();
					launch_factor_stage(t_rest, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), simt_rest_ptr[s], n_rest,
						true, p_flag, stream, t_batch);
				}
			} else if(s == 0 && !panel_ptr.empty() && panel_ptr[1] > panel_ptr[0]) {
				Launch_Panels(s, true); // few leaf subtrees: as panels (they receive no updates: the update just copies Lambda's blocks over)
			} else if(s > 0 && s < n_bottom_stages && dplan.task_pkg)
				launch_factor_wide(dplan, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), P.stage_ptr[s],
					P.stage_ptr[s + 1] - P.stage_ptr[s], p_flag, stream, t_batch);
			else if(s >= n_bottom_stages && !panel_ptr.empty()) {
				Launch_Panels(s, false); // separators: as panels in LDS where they fit, column by column otherwise
			} else
			launch_factor_stage(dplan, p_values_dev, d_L.p(), d_Linv.p(), p_rhs_dev, d_w.p(), P.stage_ptr[s],
				P.stage_ptr[s + 1] - P.stage_ptr[s], s < n_bottom_stages, p_flag, stream, t_batch);
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
				d_simt_bwd_prog.p(), d_simt_bwd_tab.p(), P.max_dim, d_L.p(), d_w.p(), p_rhs_dev, stream, t_batch);
			if(n_rest > 0) {
				TDevPlan t_rest = dplan;
				t_rest.task_map = d_simt_rest.p();
				launch_backward_stage(t_rest, d_L.p(), d_Linv.p(), d_w.p(), p_rhs_dev, simt_rest_ptr[s - 1], n_rest, stream, t_batch);
			}
			continue;
		}
		if(s < int(simt_chunk_ptr.size()))
			Ensure_Leaf_Inverses(); // (the wave-per-task kernel on a lane-per-task stage: unaligned caller vector)
		launch_backward_stage(dplan, d_L.p(), d_Linv.p(), d_w.p(), p_rhs_dev, P.stage_ptr[s - 1],
			P.stage_ptr[s] - P.stage_ptr[s - 1], stream, t_batch);
	}
	Phase_End();
	SLAMPP_HIP_CHECK(hipGetLastError());
}

