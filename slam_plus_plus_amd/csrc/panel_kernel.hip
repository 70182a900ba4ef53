// panel_kernel.hip -- separator tasks of the sparse block factorization as panels in LDS.
//
// factor_stage_kernel (sparse_kernels.hip) takes a separator column by column: package of the column, its update pairs'
// operands from global memory, the 6 x 6 Cholesky, the blocks below it, out to global memory -- and the next column of
// the same separator starts by reading some of those blocks back.  Three or four dependent round trips per column, 11 us
// per column on the reduced camera system of C4 (separators of three columns, eight blocks each: 35 us per stage).
// Most of that does not depend on the order of the columns: of a task's updates, only those whose operands the task
// itself produces have to wait.  So here a workgroup
// The host splits every update of the task into external ones (operands from earlier stages) and internal ones (operands
// among the task's own blocks).  Then, per stage,
//   1. panel_update_kernel applies the external updates of all the stage's tasks, one workgroup per factor block: near
//      the root a separator receives hundreds of them (left-looking: everything its descendants owe it), and streamed
//      by the task's own workgroup they were 4 - 42 us of its 10 - 50; spread over the chip they are a few.  Its waves
//      take the block's entries in turn, operand blocks fetched whole, eight entries in flight per wave; partial sums
//      are combined in a fixed order (no atomics); L(block) = Lambda(block) - sum waits in the factor's own storage;
//   2. factor_panel_kernel fetches the task's *panel package* (columns, blocks, internal updates as slot numbers) and
//      its blocks into an LDS image with one round of coalesced loads, and walks the columns inside the image: internal
//      updates from LDS, diagonal block (wave 0), blocks below (all waves), two barriers and no memory round trip per
//      column (2 us); finished blocks go to global memory as they are made.
// Tasks whose columns are not consecutive, or that exceed the capacities, stay with factor_stage_kernel.
// Own translation unit (see dense_tiles.hip for why).
#include <hip/hip_runtime.h>
#include "sparse_kernels.h"

namespace slampp {

#include "sparse_device.inl"

// the updates of one factor block from stages further down: wave v of the n_waves waves that share the block takes the
// entries v * BATCH .., (v + n_waves) * BATCH .., both operand blocks fetched whole, all requests of a batch before the
// first product; the partial sums are added up in wave order (s_part), and the block goes out as Lambda - sum (a
// diagonal block with its right-hand side b - sum).  Every wave of the workgroup must call it (one barrier inside).
template <int D, int BATCH>
__device__ __forceinline__ void panel_update_block(const TUpdSlot &sl, bool b_valid, const TUpdEnt *__restrict__ ents,
	const double *__restrict__ A, double *L, const double *__restrict__ b, double *w, int n_sub_wave, int n_waves, int lane,
	double *s_ops /* this wave's 2 BATCH blocks */, double *s_yv /* this wave's 8 BATCH doubles */, double *s_part /* the block's n_waves x 64 */)
{
	enum { DD = D * D };
	const TLaneMap mm = lane_map(lane, D, D);
	const bool b_diag = sl.kind != 0;
	const bool b_y = b_diag && lane >= Y_LANE0 && lane < Y_LANE0 + D;
	const int yq = b_y? lane - Y_LANE0 : mm.q;
	double init = 0; // Lambda's element (requested now, used last)
	if(b_valid && n_sub_wave == 0)
		init = b_y? b[sl.cs_src + yq] : (mm.b_act? lambda_element(A, sl.asrc, mm.r, mm.q, D, D, false) : 0.0);
	double sum = 0;
	const TUpdEnt *p_ent = ents + sl.e0;
	const int ne = b_valid? sl.ne : 0;
	for(int e0 = n_sub_wave * BATCH; e0 < ne; e0 += n_waves * BATCH) {
		const int n_here = min(int(BATCH), ne - e0);
		double va[BATCH], vb[BATCH];
		#pragma unroll
		for(int u = 0; u < BATCH; ++ u) {
			const TUpdEnt en = p_ent[e0 + min(u, n_here - 1)]; // the tail repeats the last entry: its product is skipped below
			va[u] = L[en.a_off + (mm.b_act? lane : 0)];
			// the other operand: the block L(j,c) of a pair; for a row entry y_c in the right-hand side lanes (one load, the
			// lane picks its address: a load behind a branch would wait for the others)
			const double *p_other = b_diag? w + en.b_off + (b_y? yq : 0) : L + en.b_off + (mm.b_act? lane : 0);
			vb[u] = *p_other;
		}
		#pragma unroll
		for(int u = 0; u < BATCH; ++ u) {
			if(mm.b_act) {
				s_ops[(2 * u) * DD + lane] = va[u];
				if(!b_diag)
					s_ops[(2 * u + 1) * DD + lane] = vb[u];
			}
			if(b_y)
				s_yv[u * 8 + yq] = vb[u];
		}
		wave_sync();
		#pragma unroll
		for(int u = 0; u < BATCH; ++ u) {
			if(u < n_here) { // wave-uniform
				const double *pa = b_y? s_yv + u * 8 : s_ops + (2 * u) * DD + mm.r;
				const double *pb = s_ops + (2 * u + (b_diag? 0 : 1)) * DD + yq; // (yq = q in the matrix lanes)
				const int as = b_y? 1 : D;
				#pragma unroll
				for(int t = 0; t < D; ++ t)
					sum += pa[t * as] * pb[t * D];
			}
		}
		wave_sync();
	}
	s_part[n_sub_wave * 64 + lane] = sum;
	__syncthreads();
	if(n_sub_wave != 0 || !b_valid)
		return;
	double total = 0;
	for(int v = 0; v < n_waves; ++ v)
		total += s_part[v * 64 + lane];
	if(b_y)
		w[sl.cs_new + yq] = init - total;
	else if(mm.b_act)
		L[sl.loff + lane] = init - total;
}

// (b_fused: some stage of the plan has its updates from further down applied inside the launch of the stage below -- the
// launch then holds update workgroups next to the panel ones, and the panel tasks bring in fresh updates themselves)
// one block column of a level as rows (column_rows_steps, sparse_device.inl): chunk n_chunk of its rows below the diagonal
// block -- CAP of them per wave: the lanes 0 .. D - 1 of every row of 16 lanes hold the diagonal block, every chunk factors
// it again (the rows below and the right-hand side come out of the same steps); chunk 0 writes it -- to p_diag_out, not
// into the image, where the other chunks of the column (other waves, any order) read the block as it was.  Results go to
// LDS only; panel_copy_out sends them to memory, off the chain.
template <int D>
__device__ __forceinline__ void panel_column_rows(const TPanelCol col, int ci, int n_chunk, int lane, double *s_L, double *s_w, double *p_diag_out,
	int *p_flag)
{
	enum { DD = D * D, OTHER = 16 - D, CAP = 4 * OTHER };
	const int g = lane & 15, R = lane >> 4;
	const int n_other = (col.nb - 1) * D + 1; // rows of the blocks below the diagonal + the right-hand side
	const bool b_diag = g < D;
	const int o = n_chunk * CAP + R * OTHER + (g - D);
	const bool b_rhs = !b_diag && o == n_other - 1, b_row = !b_diag && o < n_other - 1;
	const int kb = b_row? 1 + o / D : 0, r = b_diag? g : (b_row? o % D : 0);
	double *p_blk = s_L + (col.slot0 + kb) * DD + r, *p_y = s_w + ci * D;
	double a[D], piv_raw[D];
	#pragma unroll
	for(int t = 0; t < D; ++ t)
		a[t] = b_rhs? p_y[t] : p_blk[D * t];
	column_rows_steps<D, 0>(a, piv_raw);
	if(b_diag) {
		if(R == 0 && n_chunk == 0) {
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				p_diag_out[r + D * t] = (t <= r)? a[t] : 0.0;
		}
	} else if(b_row) {
		#pragma unroll
		for(int t = 0; t < D; ++ t)
			p_blk[D * t] = a[t];
	} else if(b_rhs) {
		#pragma unroll
		for(int t = 0; t < D; ++ t)
			p_y[t] = a[t];
	}
	if(lane == 0 && n_chunk == 0) {
		bool b_bad = false;
		#pragma unroll
		for(int t = 0; t < D; ++ t)
			b_bad = b_bad || !(piv_raw[t] > 0); // also catches NaN
		if(b_bad)
			atomicOr(p_flag, 1);
	}
}

// the finished columns ci0 .. ci1 - 1 of the image out to memory: their blocks as they lie (a lane per element, one
// coalesced store each), inv(L_jj) computed here -- nothing in the factorization waits for it any more -- and y_j.
template <int D, int W>
__device__ __forceinline__ void panel_copy_out(const TPanelCol *s_col, const TPanelSlot *s_slot, int ci0, int ci1, int wave, int lane,
	const TLaneMap mm, const double *s_L, const double *s_w, const double *s_diag, double *s_tile, double *L, double *Linv, double *w)
{
	enum { DD = D * D };
	int n_item = 0;
	for(int ci = ci0; ci < ci1; ++ ci) {
		const int n_slot0 = s_col[ci].slot0, n_nb = s_col[ci].nb;
		for(int kb = 0; kb < n_nb; ++ kb, ++ n_item) {
			if((W - 1) - n_item % W != wave)
				continue;
			const double v = mm.b_act? (kb? s_L[(n_slot0 + kb) * DD + lane] : s_diag[ci * 64 + lane]) : 0.0;
			if(mm.b_act)
				L[s_slot[n_slot0 + kb].loff + lane] = v;
			if(!kb) {
				invert_diagonal_fixed<D>(v, lane, mm.r, mm.q, mm.b_act, Linv, s_col[ci].linv_off, s_tile);
				if(lane < D)
					w[s_col[ci].cs_new + lane] = s_w[ci * D + lane];
			}
		}
	}
}

// the blocks the task hands up (TPanelOut): sums of products of its own finished blocks, out of the image; a wave per block:
// records [n_first, n_last) of the list, record n_first + i by wave n_wave0 + i mod n_waves_here (the waves a level's column
// work leaves idle take the records whose operands are final already; what is left is done by all waves at the end)
template <int D>
__device__ __forceinline__ void panel_hand_up(const longlong2 *s_out, int n_out, int n_first, int n_last, int n_rank, int n_ranks, int lane,
	const TLaneMap mm, bool b_y, int yq, const double *s_L, const double *s_w, double *H)
{
	enum { DD = D * D };
	const TPanelOut *s_rec = reinterpret_cast<const TPanelOut*>(s_out + 3);
	const uint32_t *s_opair = reinterpret_cast<const uint32_t*>(s_out + 3 + n_out);
	for(int o = n_first + n_rank; o < n_last; o += n_ranks) {
		const TPanelOut rec = s_rec[o];
		const bool b_diag = (rec.dst >> 62) != 0;
		const int64_t n_dst = rec.dst & ((int64_t(1) << 62) - 1);
		double sum = 0;
		if(b_diag) {
			for(int e = 0; e < rec.onp; ++ e) {
				const uint32_t en = s_opair[rec.op0 + e];
				sum += row_product_image<D>(s_L + int(en & 0xffff) * DD, s_w + int(en >> 16) * D, mm.r, yq, b_y);
			}
			if(b_y)
				H[n_dst + DD + yq] = sum;
		} else {
			for(int e = 0; e < rec.onp; ++ e) {
				const uint32_t en = s_opair[rec.op0 + e];
				sum += pair_product_image<D>(s_L + int(en & 0xffff) * DD, s_L + int(en >> 16) * DD, mm.r, mm.q);
			}
		}
		if(mm.b_act)
			H[n_dst + lane] = sum;
	}
}

template <int D, int W, bool b_fused, bool b_rows>
__global__ void __launch_bounds__(64 * W)
factor_panel_kernel(const longlong2 *__restrict__ pkg, const int64_t *__restrict__ pkg_off, const int64_t *__restrict__ out_off, int n_panels,
	TPanelLaunch t_cfg, const TUpdSlot *__restrict__ upd_slots, int n_upd_slots, const TUpdEnt *__restrict__ upd_ents, const double *__restrict__ A,
	const double *__restrict__ b, double *L, double *Linv, double *w, double *H, int *p_flag, long long *p_timing, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; A += n_member * t_batch.a; b += n_member * t_batch.b; L += n_member * t_batch.l; Linv += n_member * t_batch.linv; w += n_member * t_batch.w; H += n_member * t_batch.h; p_flag += n_member; } // (TBatch: sparse_kernels.h)

	enum { DD = D * D, BATCH = panel_fresh_batch(W), UPD_BATCH = PANEL_UPD_BATCH, UPD_W = (W < int(PANEL_UPD_W))? W : int(PANEL_UPD_W), // (two waves per task: two per update block)
		N_UPD_GROUPS = W / UPD_W };
	extern __shared__ __attribute__((aligned(16))) double s_raw[];
	const TPanelLds t_lds = panel_lds(D, b_fused, t_cfg);
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	if(b_fused && int(blockIdx.x) >= n_panels) {
		// update role: the blocks of the next stage's panel tasks, one per group of PANEL_UPD_W waves
		const int n_group = wave / UPD_W, n_sub = wave % UPD_W;
		const int n_slot = N_UPD_GROUPS * (int(blockIdx.x) - n_panels) + n_group;
		const bool b_valid = n_slot < n_upd_slots;
		const TUpdSlot sl = upd_slots[b_valid? n_slot : n_upd_slots - 1];
		panel_update_block<D, UPD_BATCH>(sl, b_valid, upd_ents, A, L, b, w, n_sub, UPD_W, lane,
			s_raw + wave * 2 * UPD_BATCH * DD, s_raw + W * 2 * UPD_BATCH * DD + wave * UPD_BATCH * 8,
			s_raw + W * 2 * UPD_BATCH * DD + W * UPD_BATCH * 8 + n_group * UPD_W * 64);
		return;
	}
	longlong2 *s_pkg = reinterpret_cast<longlong2*>(s_raw);
	double *s_L = s_raw + t_lds.IMAGE, *s_w = s_raw + t_lds.VEC, *s_linv = s_raw + t_lds.LINV;
	double *s_tile = s_raw + t_lds.TILE + wave * 64;
	double *s_ops = s_raw + t_lds.OPS + wave * 2 * BATCH * DD, *s_yv = s_raw + t_lds.YV + wave * BATCH * 8;
	long long *p_tm = 0; // development aid (SLAMPP_HIP_STAGE_TIMING): clock samples of workgroup 0
	int n_tm = 0;
	if(p_timing && blockIdx.x == 0 && tid == 0) {
		const unsigned long long n_tm_record = atomicAdd((unsigned long long*)p_timing, 1ull);
		p_tm = (n_tm_record < 4096)? p_timing + 1 + 32 * n_tm_record : 0; // the buffer holds 4096 launch records: later launches go unrecorded
		if(p_tm)
			p_tm[n_tm ++] = wall_clock64();
	}
#define PANEL_TICK() do { if(p_tm && n_tm < 32) p_tm[n_tm ++] = wall_clock64(); } while(0)
	const int64_t n_off = pkg_off[blockIdx.x];
	s_pkg[tid] = pkg[n_off + tid]; // before the size is known: the head and, for most tasks, everything
	__syncthreads();
	// (the head field by field, its table read where it lies: a copy of the record indexed by the wave number lives in
	// scratch memory, and a kernel that uses scratch takes longer to launch)
	const TPanelHead *p_hd = reinterpret_cast<const TPanelHead*>(s_pkg);
	struct { int n_cols, n_slots, n_units, n_int_rows; const int32_t *ext_ptr; } hd = {p_hd->n_cols, p_hd->n_slots, p_hd->n_units,
		p_hd->n_int_rows, p_hd->ext_ptr};
	for(int e = 64 * W + tid; e < hd.n_units; e += 64 * W)
		s_pkg[e] = pkg[n_off + e];
	// what the task hands up to the next stage's tasks: the list is needed at the very end, so it is requested now
	const int n_out = p_hd->ext_ptr[10], n_out_units = p_hd->ext_ptr[11];
	longlong2 *s_out = reinterpret_cast<longlong2*>(s_raw + t_lds.OUT);
	if(n_out_units > 0) {
		const int64_t n_out_off = out_off[blockIdx.x];
		for(int e = tid; e < n_out_units; e += 64 * W)
			s_out[e] = pkg[n_out_off + e];
	}
	const int n_cols = hd.n_cols, n_slots = hd.n_slots;
	const TPanelCol *s_col = reinterpret_cast<const TPanelCol*>(s_pkg + 4);
	const TPanelSlot *s_slot = reinterpret_cast<const TPanelSlot*>(s_pkg + 4 + 3 * n_cols);
	const uint32_t *s_irow = reinterpret_cast<const uint32_t*>(s_pkg + 4 + 3 * n_cols + 2 * n_slots);
	const uint32_t *s_ipair = s_irow + 4 * ((hd.n_int_rows + 3) / 4);
	const TPanelExt *s_ext = reinterpret_cast<const TPanelExt*>(s_pkg + hd.n_units) - hd.ext_ptr[W];
	__syncthreads();
	PANEL_TICK(); // package

	const TLaneMap mm = lane_map(lane, D, D);
	const bool b_y = lane >= Y_LANE0 && lane < Y_LANE0 + D;
	const int yq = b_y? lane - Y_LANE0 : mm.q;
	// The first blocks handed up by the stage below (this wave's: they target its slots) are requested here, together with the
	// image: a read of something the launch before has written is two or three microseconds whatever its size, and the
	// image's blocks are such reads too -- one trip for both.
	enum { UPB = 8 };
	int n_ext0 = hd.ext_ptr[wave];
	const int n_ext1 = b_fused? hd.ext_ptr[wave + 1] : n_ext0;
	int n_up_first = 0;
	double va_up[UPB], vy_up[UPB];
	if(b_fused) {
		while(n_up_first < UPB && n_ext0 + n_up_first < n_ext1 && s_ext[n_ext0 + n_up_first].kind >= 2)
			++ n_up_first;
		#pragma unroll
		for(int u = 0; u < UPB; ++ u) {
			const TPanelExt en = s_ext[n_ext0 + min(u, max(n_up_first - 1, 0))];
			const int64_t n_at = n_up_first? en.a_off : 0;
			va_up[u] = H[n_at + (mm.b_act? lane : 0)];
			vy_up[u] = H[n_at + DD + (b_y? yq : 0)];
		}
	}
	// the image: the task's blocks as the update role left them (Lambda minus the updates from further down), y likewise;
	// every wave its own slots (v, v + W, ..: it brings in their fresh updates below)
	for(int s0 = wave; s0 < n_slots; s0 += 4 * W) {
		double v[4];
		if(t_cfg.b_from_lambda) { // (wave-uniform)
			#pragma unroll
			for(int u = 0; u < 4; ++ u)
				v[u] = mm.b_act? lambda_element(A, s_slot[min(s0 + u * W, n_slots - 1)].asrc, mm.r, mm.q, D, D, false) : 0.0;
		} else {
			#pragma unroll
			for(int u = 0; u < 4; ++ u)
				v[u] = L[s_slot[min(s0 + u * W, n_slots - 1)].loff + (mm.b_act? lane : 0)];
		}
		#pragma unroll
		for(int u = 0; u < 4; ++ u) {
			if(s0 + u * W < n_slots && mm.b_act)
				s_L[(s0 + u * W) * DD + lane] = v[u];
		}
	}
	if(tid < n_cols * D)
		s_w[tid] = t_cfg.b_from_lambda? b[s_col[tid / D].cs_src + tid % D] : w[s_col[tid / D].cs_new + tid % D];
	__syncthreads();
	PANEL_TICK(); // image

	// 1. fresh updates (operands from the stage right below: a handful per task): wave v owns the slots v, v + W, ..,
	// streams the entries that target them BATCH at a time and subtracts the products from its slots -- fixed order
	// (a) the blocks handed up by the tasks of the stage below (kinds 2 / 3: the host lists them first): they come as they are,
	// one load each -- eight in flight, subtracted straight from the image; the first eight are here already
	if(b_fused) {
		#pragma unroll
		for(int u = 0; u < UPB; ++ u) {
			if(u < n_up_first) { // wave-uniform
				const TPanelExt en = s_ext[n_ext0 + u];
				if(b_y && en.kind == 3)
					s_w[en.col * D + yq] -= vy_up[u];
				else if(mm.b_act)
					s_L[int(en.slot) * DD + lane] -= va_up[u];
			}
		}
		n_ext0 += n_up_first;
	}
	for(;;) {
		int n_up = 0;
		while(n_up < UPB && n_ext0 + n_up < n_ext1 && s_ext[n_ext0 + n_up].kind >= 2)
			++ n_up;
		if(!n_up)
			break;
		double va[UPB], vy[UPB];
		#pragma unroll
		for(int u = 0; u < UPB; ++ u) {
			const TPanelExt en = s_ext[n_ext0 + min(u, n_up - 1)];
			va[u] = H[en.a_off + (mm.b_act? lane : 0)];
			vy[u] = H[en.a_off + DD + (b_y? yq : 0)];
		}
		#pragma unroll
		for(int u = 0; u < UPB; ++ u) {
			if(u < n_up) { // wave-uniform
				const TPanelExt en = s_ext[n_ext0 + u];
				if(b_y && en.kind == 3)
					s_w[en.col * D + yq] -= vy[u];
				else if(mm.b_act)
					s_L[int(en.slot) * DD + lane] -= va[u];
			}
		}
		n_ext0 += n_up;
	}
	// (b) products whose operands this task fetches itself
	for(int e0 = n_ext0, e1 = n_ext1; e0 < e1; e0 += BATCH) {
		double va[BATCH], vb[BATCH], vy[BATCH];
		#pragma unroll
		for(int u = 0; u < BATCH; ++ u) {
			const TPanelExt en = s_ext[min(e0 + u, e1 - 1)]; // the tail repeats the last entry: its product is skipped below
			va[u] = L[en.a_off + (mm.b_act? lane : 0)];
			vb[u] = L[en.b_off + (mm.b_act? lane : 0)];
			vy[u] = w[(b_y && en.kind)? en.ycs + yq : 0]; // (unconditional: a load behind a branch would wait for the others)
		}
		#pragma unroll
		for(int u = 0; u < BATCH; ++ u) {
			if(mm.b_act) {
				s_ops[(2 * u) * DD + lane] = va[u];
				s_ops[(2 * u + 1) * DD + lane] = vb[u];
			}
			if(b_y)
				s_yv[u * 8 + yq] = vy[u];
		}
		wave_sync();
		#pragma unroll
		for(int u = 0; u < BATCH; ++ u) {
			if(e0 + u < e1) { // wave-uniform
				const TPanelExt en = s_ext[e0 + u];
				const bool b_vec = en.kind && b_y; // the right-hand side rides along the diagonal block's row entries
				const double *pa = b_vec? s_yv + u * 8 : s_ops + (2 * u) * DD + mm.r;
				const double *pb = s_ops + (2 * u + 1) * DD + (en.kind? yq : mm.q);
				const int as = b_vec? 1 : D;
				double sum = 0;
				#pragma unroll
				for(int t = 0; t < D; ++ t)
					sum += pa[t * as] * pb[t * D];
				if(b_vec)
					s_w[en.col * D + yq] -= sum;
				else if(mm.b_act)
					s_L[int(en.slot) * DD + lane] -= sum;
			}
		}
		wave_sync();
	}
	if(b_fused)
		__syncthreads();
	PANEL_TICK(); // fresh updates

	// 2. the columns, inside the image, level by level: the columns of one level of a tall task do not depend on each
	// other (a chain is one column per level)
	if constexpr(b_rows) {
		// round 4: (A) every block of the level's columns, diagonal ones included, gets its internal updates -- products of
		// blocks of earlier levels, out of the image, a lane per element, the blocks dealt over all waves --; (B) one wave per
		// column (and chunk of 40 rows) factors the whole block column as rows, diagonal block, blocks below and right-hand side
		// in the same six steps; meanwhile the other waves send the level before to memory and invert its diagonal blocks.
		enum { CAP = 4 * (16 - D) };
		int n_level = 0, n_out_done = 0;
		// (the finished diagonal blocks wait in s_linv: the image keeps the blocks as they were for the other chunks of their columns)
		for(int ci0 = 0; ci0 < n_cols;) {
			const int n_sub = s_col[ci0].sub;
			int ci1 = ci0 + 1;
			while(ci1 < n_cols && s_col[ci1].sub == n_sub)
				++ ci1;
			int n_before = 0;
			for(int ci = ci0; ci < ci1; ++ ci) {
				const TPanelCol col = s_col[ci];
				for(int kb = ((wave - n_before) % W + W) % W; kb < col.nb; kb += W) {
					if(!kb) {
						if(!col.inr)
							continue;
						const double init = b_y? s_w[ci * D + yq] : (mm.b_act? s_L[col.slot0 * DD + lane] : 0.0);
						double sum = 0;
						for(int e = 0; e < col.inr; ++ e) {
							const uint32_t en = s_irow[col.ir0 + e];
							sum += row_product_image<D>(s_L + int(en & 0xffff) * DD, s_w + int(en >> 16) * D, mm.r, yq, b_y);
						}
						if(b_y)
							s_w[ci * D + yq] = init - sum;
						else if(mm.b_act)
							s_L[col.slot0 * DD + lane] = init - sum;
					} else {
						const int n_slot = col.slot0 + kb;
						const int ip0 = s_slot[n_slot].ip0, inp = s_slot[n_slot].inp;
						if(!inp)
							continue;
						const double init = mm.b_act? s_L[n_slot * DD + lane] : 0.0;
						double sum = 0;
						for(int e = 0; e < inp; ++ e) {
							const uint32_t en = s_ipair[ip0 + e];
							sum += pair_product_image<D>(s_L + int(en & 0xffff) * DD, s_L + int(en >> 16) * DD, mm.r, mm.q);
						}
						if(mm.b_act)
							s_L[n_slot * DD + lane] = init - sum;
					}
				}
				n_before += col.nb;
			}
			__syncthreads();
			PANEL_TICK(); // internal updates of the level
			int n_pair = 0;
			for(int ci = ci0; ci < ci1; ++ ci) {
				const int n_chunks = ((s_col[ci].nb - 1) * D + 1 + CAP - 1) / CAP;
				for(int n_chunk = 0; n_chunk < n_chunks; ++ n_chunk, ++ n_pair) {
					if(n_pair % W == wave)
						panel_column_rows<D>(s_col[ci], ci, n_chunk, lane, s_L, s_w, s_linv + ci * 64, p_flag);
				}
			}
			// the waves without a column here hand up what the levels before this one have finished (at most three blocks a wave:
			// the level's column work takes about as long)
			if(n_out > 0 && n_level > 0 && n_pair < W) {
				const int n_ready = reinterpret_cast<const int32_t*>(s_out)[min(n_level - 1, 11)], n_idle = W - n_pair;
				const int n_take = min(n_ready - n_out_done, 3 * n_idle);
				if(wave >= n_pair && n_take > 0)
					panel_hand_up<D>(s_out, n_out, n_out_done, n_out_done + n_take, wave - n_pair, n_idle, lane, mm, b_y, yq, s_L, s_w, H);
				n_out_done += max(n_take, 0);
			}
			++ n_level;
			__syncthreads(); // the level's columns and their y complete in the image
			PANEL_TICK();
			ci0 = ci1;
		}
		// everything out to memory, and the inverses of the diagonal blocks: by all waves, after the chain
		panel_copy_out<D, W>(s_col, s_slot, 0, n_cols, wave, lane, mm, s_L, s_w, s_linv, s_tile, L, Linv, w);
		PANEL_TICK();
		// (the finished diagonal blocks live in s_linv in this walk, not in the image: nothing handed up reads a diagonal block)
		panel_hand_up<D>(s_out, n_out, n_out_done, n_out, wave, W, lane, mm, b_y, yq, s_L, s_w, H);
		return;
	}
	// (the block-wise form of rounds 2 and 3, kept for comparison: option "panel_rows" = 0) their diagonal blocks go to one
	// wave each, their blocks below in turn over all waves; two barriers and no memory round trip per level
	for(int ci0 = 0; ci0 < n_cols;) {
		const int n_sub = s_col[ci0].sub;
		int ci1 = ci0 + 1;
		while(ci1 < n_cols && s_col[ci1].sub == n_sub)
			++ ci1;
		for(int ci = ci0 + wave; ci < ci1; ci += W) {
			const TPanelCol col = s_col[ci];
			const double init = b_y? s_w[ci * D + yq] : (mm.b_act? s_L[col.slot0 * DD + lane] : 0.0);
			double sum = 0;
			for(int e = 0; e < col.inr; ++ e) {
				const uint32_t en = s_irow[col.ir0 + e];
				sum += row_product_image<D>(s_L + int(en & 0xffff) * DD, s_w + int(en >> 16) * D, mm.r, yq, b_y);
			}
			TColDesc cd; // (finish_diagonal_fixed reads where the inverse and y go)
			cd.linv_off = col.linv_off;
			cd.cs_new = col.cs_new;
			const double acc = init - sum;
			finish_diagonal_fixed<D>(cd, acc, acc, lane, b_y? 0 : mm.r, b_y? 0 : mm.q, mm.b_act, L, Linv, w, s_slot[col.slot0].loff,
				p_flag, s_linv + (ci - ci0) * 64, s_L + col.slot0 * DD, s_w + ci * D);
		}
		__syncthreads();
		PANEL_TICK(); // diagonal blocks of the level
		int n_before = 0; // blocks below the diagonal in the level's columns before this one: block n of the level goes to wave n mod W
		for(int ci = ci0; ci < ci1; ++ ci) {
			const int n_slot0 = s_col[ci].slot0, n_nb = s_col[ci].nb;
			for(int kb = 1 + ((wave - n_before) % W + W) % W; kb < n_nb; kb += W) {
				const int n_slot = n_slot0 + kb;
				const TPanelSlot sd = s_slot[n_slot];
				const double init = mm.b_act? s_L[n_slot * DD + lane] : 0.0;
				double sum = 0;
				for(int e = 0; e < sd.inp; ++ e) {
					const uint32_t en = s_ipair[sd.ip0 + e];
					sum += pair_product_image<D>(s_L + int(en & 0xffff) * DD, s_L + int(en >> 16) * DD, mm.r, mm.q);
				}
				finish_offdiagonal<D>(init - sum, lane, mm.r, mm.q, mm.b_act, D, L, sd.loff, s_tile, s_linv + (ci - ci0) * 64, s_L + n_slot * DD);
			}
			n_before += n_nb - 1;
		}
		__syncthreads(); // the level's columns and their y complete in the image
		PANEL_TICK();
		ci0 = ci1;
	}
	panel_hand_up<D>(s_out, n_out, 0, n_out, wave, W, lane, mm, b_y, yq, s_L, s_w, H);
	PANEL_TICK();
}

bool launch_factor_panel(int n_dim, bool b_fused, bool b_rows, const TPanelLaunch &r_cfg, const longlong2 *pkg, const int64_t *pkg_off, const int64_t *out_off,
	int n_tasks, const TUpdSlot *upd_slots, int n_upd_slots, const TUpdEnt *upd_ents, const double *A, const double *b, double *L, double *Linv, double *w,
	double *H, int *p_flag, hipStream_t stream, long long *p_timing, const TBatch &t_batch)
{
	if(!b_fused)
		n_upd_slots = 0;
	const int W = r_cfg.n_waves, n_groups = W / std::min(W, int(PANEL_UPD_W)); // (two waves per task: two waves per update block as well)
	const int n_grid = n_tasks + (n_upd_slots + n_groups - 1) / n_groups;
	if(n_grid <= 0)
		return true;
	const size_t n_lds_bytes = size_t(panel_lds(n_dim, b_fused, r_cfg).TOTAL) * sizeof(double);
	if(n_lds_bytes > PANEL_LDS_BUDGET)
		return false; // (the analysis keeps every stage inside the budget: solver.hip, the hand-up lists; a launch past it would fail with a device error on a valid system)
	// (beyond 64 KB of dynamic LDS a kernel has to be told once)
#define LAUNCH_PANEL_INSTANCE(D, WW, F, RW) do { \
		static bool b_attribute_set = false; \
		if(!b_attribute_set) { \
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_panel_kernel<D, WW, F, RW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
			b_attribute_set = true; \
		} \
		hipLaunchKernelGGL((factor_panel_kernel<D, WW, F, RW>), dim3(n_grid, t_batch.n), dim3(64 * WW), n_lds_bytes, stream, pkg, pkg_off, out_off, n_tasks, r_cfg, \
			upd_slots, n_upd_slots, upd_ents, A, b, L, Linv, w, H, p_flag, p_timing, t_batch); } while(0)
#define LAUNCH_PANEL_F(D, WW, F) do { if(b_rows) LAUNCH_PANEL_INSTANCE(D, WW, F, true); else LAUNCH_PANEL_INSTANCE(D, WW, F, false); } while(0)
#define LAUNCH_PANEL(D) do { \
		if(W == 2) { if(b_fused) LAUNCH_PANEL_F(D, 2, true); else LAUNCH_PANEL_F(D, 2, false); } \
		else if(W == 4) { if(b_fused) LAUNCH_PANEL_F(D, 4, true); else LAUNCH_PANEL_F(D, 4, false); } \
		else { if(b_fused) LAUNCH_PANEL_F(D, 8, true); else LAUNCH_PANEL_F(D, 8, false); } } while(0)
	switch(n_dim) {
	case 3:
		LAUNCH_PANEL(3);
		return true;
	case 6:
		LAUNCH_PANEL(6);
		return true;
	case 7:
		LAUNCH_PANEL(7);
		return true;
	default:
		return false;
	}
#undef LAUNCH_PANEL
#undef LAUNCH_PANEL_F
#undef LAUNCH_PANEL_INSTANCE
}

// the same updates as a launch of their own (the first panel stage: nothing below it to ride in): one workgroup of
// PANEL_UPD_W waves per factor block
template <int D>
__global__ void __launch_bounds__(64 * PANEL_UPD_W)
panel_update_kernel(const TUpdSlot *__restrict__ slots, const TUpdEnt *__restrict__ ents, const double *__restrict__ A, double *L,
	const double *__restrict__ b, double *w, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; A += n_member * t_batch.a; L += n_member * t_batch.l; b += n_member * t_batch.b; w += n_member * t_batch.w; } // (TBatch: sparse_kernels.h)

	enum { W = PANEL_UPD_W, DD = D * D, BATCH = 8 };
	__shared__ double s_ops[W][2 * BATCH * DD];
	__shared__ double s_yv[W][BATCH * 8];
	__shared__ double s_part[W][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const TUpdSlot sl = slots[blockIdx.x];
	panel_update_block<D, BATCH>(sl, true, ents, A, L, b, w, wave, W, lane, s_ops[wave], s_yv[wave], s_part[0]);
}

void launch_panel_update(int n_dim, const TUpdSlot *slots, int n_slots, const TUpdEnt *ents, const double *A, double *L,
	const double *b, double *w, hipStream_t stream, const TBatch &t_batch)
{
	if(n_slots <= 0)
		return;
	if(n_dim == 3)
		hipLaunchKernelGGL((panel_update_kernel<3>), dim3(n_slots, t_batch.n), dim3(64 * PANEL_UPD_W), 0, stream, slots, ents, A, L, b, w, t_batch);
	else if(n_dim == 6)
		hipLaunchKernelGGL((panel_update_kernel<6>), dim3(n_slots, t_batch.n), dim3(64 * PANEL_UPD_W), 0, stream, slots, ents, A, L, b, w, t_batch);
	else
		hipLaunchKernelGGL((panel_update_kernel<7>), dim3(n_slots, t_batch.n), dim3(64 * PANEL_UPD_W), 0, stream, slots, ents, A, L, b, w, t_batch);
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(panel_kernel) // (the handle's bring-up thread loads this unit's code object: capi.hip)
