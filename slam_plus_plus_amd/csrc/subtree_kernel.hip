// subtree_kernel.hip -- bottom stages of the sparse block factorization: one wave per elimination subtree,
// working out of LDS.
//
// The first version of this stage (factor_subtree_kernel in sparse_kernels.hip, still used for mixed block
// sizes) walks a subtree column by column through global memory: column record -> block record -> update pair
// -> operand blocks, every arrow a dependent load of 1-2 us under load.  Measured at 100k, 1M and 4M poses it
// moves the same 0.48 TB/s of algorithmic bytes: ten dependent round trips per column with at most 8 waves per
// SIMD to hide them, not bandwidth and not instruction issue.  A subtree is self-contained, though -- every
// operand of its columns is a block the same wave produced earlier -- and its records, factor blocks, update
// pairs and row entries are contiguous ranges.  So the wave
//   1. fetches the records of the whole task with a handful of coalesced loads (column records, block records,
//      row entries, update pairs: 16 B per lane) and all the Lambda blocks of the task back to back into an LDS
//      image of its part of the factor: three dependent round trips per task instead of ten per column;
//   2. factorizes in that image: operands come from LDS, finished blocks go to LDS and to global memory.
// Ranges that exceed the LDS capacities, tasks whose columns are not consecutive, and the few-column tasks of the
// stages right above the bottom one (whose operands were produced by other waves) take the global path item by
// item; the arithmetic and its order are those of the first version (bit-identical results).
// Own translation unit (see dense_tiles.hip for why).
#include <hip/hip_runtime.h>
#include "sparse_kernels.h"

namespace slampp {

#include "sparse_device.inl"

template <int D, int CAP_BLK>
__global__ void __launch_bounds__(64)
factor_subtree_image_kernel(TDevPlan p, const double *__restrict__ A, double *L, double *Linv,
	const double *__restrict__ b, double *w, int task_begin, int *p_flag, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; A += n_member * t_batch.a; L += n_member * t_batch.l; Linv += n_member * t_batch.linv; b += n_member * t_batch.b; w += n_member * t_batch.w; p_flag += n_member; } // (TBatch: sparse_kernels.h)

	enum { DD = D * D, CAP_COL = 8, CAP_PAIR = 2 * CAP_BLK, CAP_RENT = CAP_BLK };
	__shared__ double s_linv[64];
	__shared__ double s_tile[64];
	__shared__ int64_t s_cols_q[CAP_COL * 8];
	__shared__ longlong2 s_blks_q[CAP_BLK * 2];
	__shared__ longlong2 s_rents_q[CAP_RENT];
	__shared__ longlong2 s_pairs[CAP_PAIR];
	__shared__ double s_L[CAP_BLK * DD];
	__shared__ double s_w[CAP_COL * 8];
	const TColDesc *s_cols = reinterpret_cast<const TColDesc*>(s_cols_q);
	const TBlkDesc *s_blks = reinterpret_cast<const TBlkDesc*>(s_blks_q);
	const TRowEnt *s_rents = reinterpret_cast<const TRowEnt*>(s_rents_q);
	static_assert(sizeof(TColDesc) == 64 && sizeof(TBlkDesc) == 32 && sizeof(TRowEnt) == 16, "record sizes");

	const int lane = threadIdx.x;
	const int task = p.task_map? p.task_map[task_begin + blockIdx.x] : task_begin + blockIdx.x;
	long long *p_tm = 0; // development aid (SLAMPP_HIP_STAGE_TIMING): clock samples of a workgroup in the middle of the grid
	int n_tm = 0;
	if(p.p_timing && blockIdx.x == gridDim.x / 2 && lane == 0) {
		const unsigned long long n_tm_record = atomicAdd((unsigned long long*)p.p_timing, 1ull);
		p_tm = (n_tm_record < 4096)? p.p_timing + 1 + 32 * n_tm_record : 0; // the buffer holds 4096 launch records: later launches go unrecorded
		if(p_tm)
			p_tm[n_tm ++] = wall_clock64();
	}
#define SUBTREE_TICK() do { if(p_tm && n_tm < 32) p_tm[n_tm ++] = wall_clock64(); } while(0)
	const int64_t c_begin = p.task_ptr[task], c_end = p.task_ptr[task + 1];
	const int n_cols = int(c_end - c_begin), n_cached_cols = (n_cols < CAP_COL)? n_cols : int(CAP_COL);
	if(lane < n_cached_cols * 8)
		s_cols_q[lane] = reinterpret_cast<const int64_t*>(p.cols + c_begin)[lane];
	wave_sync();
	SUBTREE_TICK(); // column records
	const int64_t k_begin = s_cols[0].k0, r_begin = s_cols[0].r0, w_base = s_cols[0].cs_new;
	const TColDesc cd_last = s_cols[n_cached_cols - 1];
	// consecutive columns: their factor blocks (and pairs, row entries, entries of y) form one range each, and
	// everything in the range up to the block being computed was produced by this wave
	bool b_follows = true;
	if(lane > 0 && lane < n_cached_cols)
		b_follows = s_cols[lane].k0 == s_cols[lane - 1].k0 + s_cols[lane - 1].nb;
	const bool b_consec = n_cols <= CAP_COL && __all(b_follows);
	int64_t n_tmp = cd_last.k0 + cd_last.nb;
	n_tmp = ((n_tmp < p.n_blks)? n_tmp : p.n_blks) - k_begin;
	const int n_slots = (n_tmp < CAP_BLK)? int(n_tmp) : int(CAP_BLK);
	for(int i = lane; i < n_slots * 2; i += 64)
		s_blks_q[i] = reinterpret_cast<const longlong2*>(p.blks + k_begin)[i];
	n_tmp = cd_last.r0 + cd_last.nr;
	n_tmp = ((n_tmp < p.n_rents)? n_tmp : p.n_rents) - r_begin;
	const int n_rents = (n_tmp < CAP_RENT)? int(n_tmp) : int(CAP_RENT);
	for(int i = lane; i < n_rents; i += 64)
		s_rents_q[i] = reinterpret_cast<const longlong2*>(p.rents + r_begin)[i];
	if(b_consec) {
		const int ci = lane / D, t = lane - ci * D;
		if(ci < n_cols)
			s_w[ci * D + t] = b[s_cols[ci].cs_src + t];
	}
	wave_sync();
	SUBTREE_TICK(); // block records, row entries, b
	const int64_t p_begin = s_blks[0].p0, l_base = s_blks[0].loff;
	const int64_t n_image = int64_t(n_slots) * DD; // doubles of the factor held in LDS, from l_base
	{
		const TBlkDesc bd_last = s_blks[n_slots - 1];
		n_tmp = bd_last.p0 + int64_t(bd_last.np_di & 0xffffff);
		n_tmp = ((n_tmp < p.n_pairs)? n_tmp : p.n_pairs) - p_begin;
	}
	const int n_pairs = (n_tmp < CAP_PAIR)? int(n_tmp) : int(CAP_PAIR);
	for(int i = lane; i < n_pairs; i += 64)
		s_pairs[i] = p.pairs[p_begin + i];
	const TLaneMap mm = lane_map(lane, D, D);
	for(int s0 = 0; s0 < n_slots; s0 += 8) { // all requests of a batch are issued before the first LDS store
		double v[8];
		#pragma unroll
		for(int u = 0; u < 8; ++ u)
			v[u] = (s0 + u < n_slots && mm.b_act)? lambda_element(A, s_blks[s0 + u].asrc, mm.r, mm.q, D, D, false) : 0.0;
		#pragma unroll
		for(int u = 0; u < 8; ++ u) {
			if(s0 + u < n_slots && mm.b_act)
				s_L[(s0 + u) * DD + lane] = v[u];
		}
	}
	wave_sync();
	SUBTREE_TICK(); // pairs, Lambda blocks

	const bool b_y = lane >= Y_LANE0 && lane < Y_LANE0 + D;
	const int yq = b_y? lane - Y_LANE0 : mm.q;
	for(int ci = 0; ci < n_cols; ++ ci) {
		const TColDesc cd = (ci < CAP_COL)? s_cols[ci] : p.cols[c_begin + ci];
		const int64_t slot0 = cd.k0 - k_begin;
		{
			const bool b_slot = slot0 < n_slots;
			const TBlkDesc bd = b_slot? s_blks[slot0] : p.blks[cd.k0];
			double init;
			if(b_y)
				init = b_consec? s_w[ci * D + yq] : b[cd.cs_src + yq];
			else if(!mm.b_act)
				init = 0;
			else
				init = b_slot? s_L[slot0 * DD + lane] : lambda_element(A, bd.asrc, mm.r, mm.q, D, D, false);
			double sum = 0;
			for(int e = 0; e < cd.nr; ++ e) {
				const int64_t re = cd.r0 - r_begin + e;
				const TRowEnt en = (re < n_rents)? s_rents[re] : p.rents[cd.r0 + e];
				const int64_t lo = en.off - l_base;
				if(b_consec && lo >= 0 && lo < n_image)
					sum += row_product_image<D>(s_L + lo, s_w + (en.ycs - w_base), mm.r, yq, b_y);
				else
					sum += row_product<D>(en, L, w, mm.r, yq, D, b_y);
			}
			const double acc = init - sum;
			finish_diagonal_fixed<D>(cd, acc, acc, lane, b_y? 0 : mm.r, b_y? 0 : mm.q, mm.b_act, L, Linv, w, bd.loff, p_flag, s_linv,
				(b_slot && b_consec)? s_L + slot0 * DD : (double*)0, b_consec? s_w + ci * D : (double*)0);
		}
		SUBTREE_TICK(); // diagonal block
		for(int kb = 1; kb < cd.nb; ++ kb) {
			const int64_t slot = slot0 + kb;
			const bool b_slot = slot < n_slots;
			const TBlkDesc bd = b_slot? s_blks[slot] : p.blks[cd.k0 + kb];
			const int np = int(bd.np_di & 0xffffff);
			const double init = !mm.b_act? 0.0 : (b_slot? s_L[slot * DD + lane] : lambda_element(A, bd.asrc, mm.r, mm.q, D, D, false));
			double sum = 0;
			for(int e = 0; e < np; ++ e) {
				const int64_t pe = bd.p0 - p_begin + e;
				const longlong2 pr = (pe < n_pairs)? s_pairs[pe] : p.pairs[bd.p0 + e];
				const int64_t oa = (pr.x & PAIR_OFF_MASK) - l_base, ob = pr.y - l_base;
				if(b_consec && oa >= 0 && oa < n_image && ob >= 0 && ob < n_image)
					sum += pair_product_image<D>(s_L + oa, s_L + ob, mm.r, mm.q);
				else
					sum += pair_product<D>(pr, L, mm.r, mm.q, D, D);
			}
			finish_offdiagonal<D>(init - sum, lane, mm.r, mm.q, mm.b_act, D, L, bd.loff, s_tile, s_linv,
				(b_slot && b_consec)? s_L + slot * DD : (double*)0);
		}
		__syncthreads(); // column j (and y_j) complete and visible to this wave before the next column reads them
		SUBTREE_TICK(); // sub-diagonal blocks
	}
#undef SUBTREE_TICK
}

bool launch_factor_subtree_image(const TDevPlan &p, const double *A, double *L, double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, int *p_flag, hipStream_t stream, const TBatch &t_batch)
{
	switch(p.uniform_dim) {
	case 3:
		hipLaunchKernelGGL((factor_subtree_image_kernel<3, 64>), dim3(n_tasks, t_batch.n), dim3(64), 0, stream, p, A, L, Linv, b, w,
			task_begin, p_flag, t_batch);
		return true;
	case 6:
		hipLaunchKernelGGL((factor_subtree_image_kernel<6, 24>), dim3(n_tasks, t_batch.n), dim3(64), 0, stream, p, A, L, Linv, b, w,
			task_begin, p_flag, t_batch);
		return true;
	case 7:
		hipLaunchKernelGGL((factor_subtree_image_kernel<7, 20>), dim3(n_tasks, t_batch.n), dim3(64), 0, stream, p, A, L, Linv, b, w,
			task_begin, p_flag, t_batch);
		return true;
	default:
		return false;
	}
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(subtree_kernel) // (the handle's bring-up thread loads this unit's code object: capi.hip)
