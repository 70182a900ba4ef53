// sparse_kernels.hip -- gfx950 kernels of the sparse block Cholesky path (pose graphs).
//
// What the reference does on one CPU core (native block up-looking Cholesky,
// /root/reference/src/slam/BlockMatrix.cpp:9547-9785, and the two block triangular solves,
// BlockMatrix.cpp:8637-8719,8898-) is restated here as a *left-looking, stage-scheduled* block
// factorization: the host plan (plan.cpp) lists, for every block L(i,j), the pairs of already
// finished blocks whose product updates it, and groups the block columns into stages of mutually
// independent tasks.  A task is a chain of columns eliminated in order by one workgroup; the
// bottom stage holds whole elimination subtrees (thousands of them for a nested-dissection
// ordering of a 100k-pose graph), upper stages hold the separators.
//
// Per block column j (d_j x d_j diagonal, blocks stored column-major, dimension <= 8):
//   acc(i,j)  = Lambda(i,j) - sum_pairs L(i,c) L(j,c)^T          (lanes = elements of the block)
//   L(j,j)    = chol(acc(j,j)),  Linv(j) = inv(L(j,j))           (in-wave, shuffles + LDS)
//   L(i,j)    = acc(i,j) Linv(j)^T
//   y(j)      = Linv(j) (b(j) - sum_c L(j,c) y(c))               (forward substitution, fused: the
//               blocks L(j,c) it needs are exactly the operands of the diagonal block's update)
// Traffic is one read of Lambda, one write of L, plus re-reads of L blocks by the updates that
// mostly hit L2 (the producer ran in the same wave or in the previous stage): HBM-bound on paper,
// latency-bound by the dependency chain in practice -- see DESIGN.md.
//
// Kernels are instantiated for uniform block dimension D = 3 / 6 / 7 (SE(2), SE(3), Sim(3) pose
// graphs: inner loops fully unrolled, all operand loads of an update issued before the FMAs) and
// for D = 0 (any mix of dimensions <= 8, e.g. poses + landmarks).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "sparse_kernels.h"

namespace slampp {

#include "sparse_device.inl"

// ---- bottom stage: whole elimination subtrees, one wave each (touches most of Lambda and L) ----
// First version, through global memory; used for mixed block sizes (D = 0) and, with SLAMPP_HIP_DEV_SUBTREE_V1 set (and SLAMPP_HIP_DEV=1), for
// A/B timing against subtree_kernel.hip, which took over the fixed block sizes.  With 8 waves per SIMD resident it is
// bound by instruction issue, not by memory latency (the LDS version runs at the same speed: DESIGN.md section 4.1),
// so it keeps the instruction count per column low: one block, one update at a time.
template <int D>
__global__ void __launch_bounds__(64)
factor_subtree_kernel(TDevPlan p, const double *__restrict__ A, double *L, double *Linv,
	const double *__restrict__ b, double *w, int task_begin, int *p_flag, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; A += n_member * t_batch.a; L += n_member * t_batch.l; Linv += n_member * t_batch.linv; b += n_member * t_batch.b; w += n_member * t_batch.w; p_flag += n_member; } // (TBatch: sparse_kernels.h)

	__shared__ double s_linv[64];  // inv(L_jj), element (r,c) at r + 8 c
	__shared__ double s_rdiag[8];  // 1 / L_jj(k,k)
	__shared__ double s_tile[64];
	const int lane = threadIdx.x;
	const int task = p.task_map? p.task_map[task_begin + blockIdx.x] : task_begin + blockIdx.x;
	const int64_t c_begin = p.task_ptr[task], c_end = p.task_ptr[task + 1];
	TColDesc cd_next = p.cols[c_begin];
	for(int64_t c = c_begin; c < c_end; ++ c) {
		const TColDesc cd = cd_next;
		if(c + 1 < c_end)
			cd_next = p.cols[c + 1]; // index data does not depend on the numbers: fetch it a column ahead
		const int dj = D? D : cd.dj;
		const bool b_y_inline = dj <= 7;
		const bool b_y = b_y_inline && lane >= Y_LANE0 && lane < Y_LANE0 + dj;
		{
			const TBlkDesc bd = p.blks[cd.k0];
			const TLaneMap md = lane_map(lane, dj, dj);
			const int yq = b_y? lane - Y_LANE0 : md.q;
			const double init = b_y? b[cd.cs_src + yq] : (md.b_act? lambda_element(A, bd.asrc, md.r, md.q, dj, dj, true) : 0);
			const double acc = init - accumulate_row<D>(p.rents, cd.r0, cd.nr, 0, 1, L, w, md.r, yq, dj, b_y);
			if(D)
				finish_diagonal_fixed<(D? D : 1)>(cd, acc, acc, lane, b_y? 0 : md.r, b_y? 0 : md.q, md.b_act, L, Linv, w, bd.loff,
					p_flag, s_linv);
			else
				finish_diagonal<D>(cd, acc, acc, lane, b_y? 0 : md.r, b_y? 0 : md.q, md.b_act, b_y_inline, p, L, Linv, b, w,
					bd.loff, p_flag, s_linv, s_rdiag, s_tile);
			if(!D && !b_y_inline)
				finish_rhs_wide(cd, lane, p, L, b, w, s_linv);
		}
		for(int kb = 1; kb < cd.nb; ++ kb) {
			const TBlkDesc bd = p.blks[cd.k0 + kb];
			const int di = D? D : int(bd.np_di >> 24), np = int(bd.np_di & 0xffffff);
			const TLaneMap m = lane_map(lane, di, dj);
			const double acc = lambda_element(A, bd.asrc, m.r, m.q, di, dj, false) -
				accumulate_pairs<D>(p.pairs, bd.p0, np, 0, 1, L, m.r, m.q, di, dj);
			finish_offdiagonal<D>(acc, lane, m.r, m.q, m.b_act, dj, L, bd.loff, s_tile, s_linv);
		}
		__syncthreads(); // column j (and y_j) complete and visible to this wave before the next column reads them
	}
}

// ---- upper stages: separator columns; W waves split the updates of the column ----
// Fast path (<= CH blocks, <= NR row entries, <= NP pairs): index records staged through
// LDS by the whole workgroup, update e handled by wave e mod W into that wave's private partial sums.

// CH / NR / NP: capacities of the staged path (blocks, row entries, update pairs of a column); the separators near the
// root want (16, 128, 512), the thousands of small columns right above the leaves (8, 32, 48) and one wave each -- a
// twentieth of the LDS, so that many of them are resident and hide each other's round trips
template <int D, int W, int CH, int NR, int NP>
__global__ void __launch_bounds__(64 * W)
factor_stage_kernel(TDevPlan p, const double *__restrict__ A, double *L, double *Linv,
	const double *__restrict__ b, double *w, int task_begin, int *p_flag, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; A += n_member * t_batch.a; L += n_member * t_batch.l; Linv += n_member * t_batch.linv; b += n_member * t_batch.b; w += n_member * t_batch.w; p_flag += n_member; } // (TBatch: sparse_kernels.h)

	__shared__ double s_linv[64];
	__shared__ double s_rdiag[8];
	__shared__ double s_tile[W][64];
	__shared__ double s_part[W][CH][64]; // partial sums: [wave][block][lane]
	__shared__ TBlkDesc s_blk[CH];
	__shared__ TRowEnt s_rent[NR];
	__shared__ longlong2 s_prec[NP];
	__shared__ unsigned char s_ptag[NP];
	// fixed block size: the column's package (descriptor, block records, then row entries and pairs alike as (a, b)
	// operand offsets with their right-hand side offsets and target tags), copied from the plan as it is
	enum { PKG_UNITS = 4 + 2 * CH + (NR + NP) + (NR + NP + 3) / 4 + (NR + NP + 15) / 16 };
	__shared__ longlong2 s_pkg[D? PKG_UNITS : 1];
	__shared__ double s_ops[(W == 1 && D)? 16 * D * D : 1]; // one wave per column: operand blocks of eight entries
	__shared__ double s_yv[(W == 1 && D)? 64 : 1];          // and the y vectors of their row entries
	static_assert(64 * W <= PKG_SPECULATIVE && 64 * W <= PKG_UNITS, "one speculative unit per thread");
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
	const int task = p.task_map? p.task_map[task_begin + blockIdx.x] : task_begin + blockIdx.x;
	long long *p_tm = 0;
	int n_tm = 0;
	if(p.p_timing && blockIdx.x == 0 && tid == 0) {
		const unsigned long long n_tm_record = atomicAdd((unsigned long long*)p.p_timing, 1ull);
		p_tm = (n_tm_record < 4096)? p.p_timing + 1 + 32 * n_tm_record : 0; // the buffer holds 4096 launch records: later launches go unrecorded
		if(p_tm)
			p_tm[n_tm ++] = wall_clock64();
	}
#define STAGE_TICK() do { if(p_tm && n_tm < 32) p_tm[n_tm ++] = wall_clock64(); } while(0)
	const int64_t c_begin = p.task_ptr[task], c_end = p.task_ptr[task + 1];
	int64_t n_pkg_at = (D && p.task_pkg)? p.task_pkg[task] : -1;
	for(int64_t c = c_begin; c < c_end; ++ c) {
		TColDesc cd;
		bool b_packaged = false;
		if(D && n_pkg_at >= 0) {
			// one unit per thread before the size is known: the descriptor and, for all but the largest columns, every record
			s_pkg[tid] = p.pkg[n_pkg_at + tid];
			__syncthreads();
			cd = *reinterpret_cast<const TColDesc*>(s_pkg);
			b_packaged = cd.nb <= CH && cd.nr <= NR && cd.np <= NP;
			const int n_units = b_packaged? package_units(cd.nb, cd.nr + cd.np) : 4;
			for(int e = 64 * W + tid; e < n_units; e += 64 * W)
				s_pkg[e] = p.pkg[n_pkg_at + e];
			n_pkg_at += n_units;
			__syncthreads();
		} else
			cd = p.cols[c];
		const int dj = D? D : cd.dj;
		STAGE_TICK(); // column descriptor (and records, if packaged) here
		const bool b_y_inline = dj <= 7;
		const bool b_y = b_y_inline && lane >= Y_LANE0 && lane < Y_LANE0 + dj;
		const TLaneMap md = lane_map(lane, dj, dj);
		const int yq = b_y? lane - Y_LANE0 : md.q;
		const bool b_staged = D != 0;
		if(cd.nb <= CH && cd.nr <= NR && cd.np <= NP) {
			// where the staged records sit in the package image
			const int ne_all = cd.nr + cd.np;
			TBlkDesc *s_pblk = reinterpret_cast<TBlkDesc*>(s_pkg + 4);
			longlong2 *s_ent = s_pkg + 4 + 2 * cd.nb;
			int *s_ycs = reinterpret_cast<int*>(s_ent + ne_all);
			unsigned char *s_tag = reinterpret_cast<unsigned char*>(s_ent + ne_all + (ne_all + 3) / 4);
			if(b_staged && b_packaged) {
				if(tid < cd.nb)
					s_blk[tid] = s_pblk[tid]; // (the code below indexes s_blk)
			} else if(b_staged) { // rows and pairs as one list of (a, b) operand offsets for the unified loop below
				for(int e = tid; e < cd.nb; e += 64 * W)
					s_blk[e] = p.blks[cd.k0 + e];
				for(int e = tid; e < cd.nr; e += 64 * W) {
					const TRowEnt en = p.rents[cd.r0 + e];
					s_ent[e] = longlong2{en.off, en.off};
					s_ycs[e] = en.ycs;
					s_tag[e] = 0;
				}
				for(int e = tid; e < cd.np; e += 64 * W) {
					const longlong2 pr = p.pairs[cd.p0 + e];
					s_ent[cd.nr + e] = longlong2{pr.x & PAIR_OFF_MASK, pr.y};
					s_tag[cd.nr + e] = (unsigned char)((pr.x >> 48) & 0xff);
				}
			} else {
				for(int e = tid; e < cd.nb; e += 64 * W)
					s_blk[e] = p.blks[cd.k0 + e];
				for(int e = tid; e < cd.nr; e += 64 * W)
					s_rent[e] = p.rents[cd.r0 + e];
				for(int e = tid; e < cd.np; e += 64 * W)
					s_prec[e] = p.pairs[cd.p0 + e];
			}
			__syncthreads();
			STAGE_TICK(); // records staged
			if(b_staged) {
				// phase A: the Lambda elements are requested first and join the wave's partial sums last, so that
				// their latency hides behind the products
				double init[(CH + W - 1) / W]; // Lambda elements of the blocks wave, wave + W, ...
				#pragma unroll
				for(int i = 0; i < (CH + W - 1) / W; ++ i) {
					const int kb = wave + i * W;
					init[i] = 0;
					if(kb == 0)
						init[i] = b_y? -b[cd.cs_src + yq] : (md.b_act? -lambda_element(A, s_blk[0].asrc, md.r, md.q, D, D, true) : 0);
					else if(kb < cd.nb)
						init[i] = md.b_act? -lambda_element(A, s_blk[kb].asrc, md.r, md.q, D, D, false) : 0;
				}
				const int ne = cd.nr + cd.np;
				// every update (row entries of the diagonal block and pairs of the others alike) goes through one loop
				if(W == 1) {
					// one wave per column (the wide stages): the operand blocks are fetched whole, one coalesced 288-byte load
					// each, four entries = eight blocks in flight together, and multiplied out of LDS -- a lane fetching its
					// own twelve operands per entry kept the column 7-10 us in this loop (a sixth of the loads this way, and a
					// twelfth of the cache lines they touch).  The next four entries are requested before these are
					// multiplied; the sums start from minus the Lambda elements (they are here by the time the first operands
					// are).  Four, not eight, at a time: 149 registers instead of 171, three waves per SIMD instead of two
					// (C3's three wide stages 96 -> 86 us, at a million poses 1.23 -> 1.03 ms)
					enum { BATCH = 4, DD = (D? D * D : 1) };
					double va[BATCH], vb[BATCH], vy[BATCH];
					#pragma unroll
					for(int u = 0; u < BATCH; ++ u) {
						const int e = max(min(u, ne - 1), 0); // the tail repeats the last entry: its product is skipped below
						const longlong2 en = s_ent[e];
						va[u] = (ne > 0)? L[en.x + (md.b_act? lane : 0)] : 0.0;
						vb[u] = (ne > 0)? L[en.y + (md.b_act? lane : 0)] : 0.0;
						vy[u] = (ne > 0 && b_y && s_tag[e] == 0)? w[s_ycs[e] + yq] : 0.0;
					}
					#pragma unroll
					for(int i = 0; i < CH; ++ i)
						if(i < cd.nb) s_part[0][i][lane] = init[i];
					for(int e0 = 0; e0 < ne; e0 += BATCH) {
						#pragma unroll
						for(int u = 0; u < BATCH; ++ u) {
							if(md.b_act) {
								s_ops[(2 * u) * DD + lane] = va[u];
								s_ops[(2 * u + 1) * DD + lane] = vb[u];
							}
							if(b_y)
								s_yv[u * 8 + yq] = vy[u];
						}
						wave_sync();
						#pragma unroll
						for(int u = 0; u < BATCH; ++ u) { // the next batch (clamped addresses, no branch around the requests)
							const int e = min(e0 + BATCH + u, ne - 1);
							const longlong2 en = s_ent[e];
							va[u] = L[en.x + (md.b_act? lane : 0)];
							vb[u] = L[en.y + (md.b_act? lane : 0)];
							vy[u] = (b_y && s_tag[e] == 0)? w[s_ycs[e] + yq] : 0.0;
						}
						#pragma unroll
						for(int u = 0; u < BATCH; ++ u) {
							if(e0 + u < ne) { // wave-uniform
								const int tag = s_tag[e0 + u];
								const bool b_vec = tag == 0 && b_y;
								const double *pa = b_vec? s_yv + u * 8 : s_ops + (2 * u) * DD + md.r;
								const double *pb = s_ops + (2 * u + 1) * DD + ((tag == 0)? yq : md.q);
								const int as = b_vec? 1 : D;
								double sum = 0;
								#pragma unroll
								for(int t = 0; t < D; ++ t)
									sum += pa[t * as] * pb[t * D];
								s_part[0][tag][lane] += sum;
							}
						}
						wave_sync();
					}
				} else {
				for(int kb = 0; kb < cd.nb; ++ kb)
					s_part[wave][kb][lane] = 0;
				#pragma unroll 4
				for(int e = wave; e < ne; e += W) {
					const longlong2 en = s_ent[e];
					const int tag = s_tag[e];
					const bool b_vec = tag == 0 && b_y; // the right-hand side rides along the diagonal block
					const double *pa = b_vec? w + s_ycs[e] : L + en.x + md.r;
					const double *pb = L + en.y + ((tag == 0)? yq : md.q);
					const int as = b_vec? 1 : D;
					double av[D? D : 1], bv[D? D : 1];
					#pragma unroll
					for(int t = 0; t < D; ++ t) {
						av[t] = pa[t * as];
						bv[t] = pb[t * D];
					}
					double sum = 0;
					#pragma unroll
					for(int t = 0; t < D; ++ t)
						sum += av[t] * bv[t];
					s_part[wave][tag][lane] += sum;
				}
				}
				if(W > 1) {
					#pragma unroll
					for(int i = 0; i < (CH + W - 1) / W; ++ i) {
						if(wave + i * W < cd.nb)
							s_part[wave][wave + i * W][lane] += init[i];
					}
				}
			} else {
			for(int e = tid; e < cd.np; e += 64 * W)
				s_ptag[e] = (unsigned char)pair_block(s_blk, cd.nb, cd.p0 + e);
			// phase A: partial sums start from minus the Lambda element in the wave that owns the block
			double acc0 = 0;
			if(wave == 0)
				acc0 = b_y? -b[cd.cs_src + yq] : (md.b_act? -lambda_element(A, s_blk[0].asrc, md.r, md.q, dj, dj, true) : 0);
			for(int kb = 1; kb < cd.nb; ++ kb) {
				double init = 0;
				if(kb % W == wave) {
					const TBlkDesc bd = s_blk[kb];
					const int di = D? D : int(bd.np_di >> 24);
					const TLaneMap m = lane_map(lane, di, dj);
					init = -lambda_element(A, bd.asrc, m.r, m.q, di, dj, false);
				}
				s_part[wave][kb][lane] = init;
			}
			__syncthreads(); // tags visible
			#pragma unroll 4
			for(int e = wave; e < cd.nr; e += W)
				acc0 += row_product<D>(s_rent[e], L, w, md.r, yq, dj, b_y);
			s_part[wave][0][lane] = acc0;
			#pragma unroll 4
			for(int e = wave; e < cd.np; e += W) {
				const int kb = s_ptag[e];
				const int di = D? D : int(s_blk[kb].np_di >> 24);
				const TLaneMap m = lane_map(lane, di, dj);
				s_part[wave][kb][lane] += pair_product<D>(s_prec[e], L, m.r, m.q, di, dj);
			}
			}
			__syncthreads();
			STAGE_TICK(); // products done
			// phase B: the diagonal block first (wave 0), then the sub-diagonal blocks round-robin
			if(wave == 0) {
				double acc = 0;
				#pragma unroll
				for(int ww = 0; ww < W; ++ ww)
					acc -= s_part[ww][0][lane];
				if(D)
					finish_diagonal_fixed<(D? D : 1)>(cd, acc, acc, lane, b_y? 0 : md.r, b_y? 0 : md.q, md.b_act, L, Linv, w,
						s_blk[0].loff, p_flag, s_linv);
				else
					finish_diagonal<D>(cd, acc, acc, lane, b_y? 0 : md.r, b_y? 0 : md.q, md.b_act, b_y_inline, p, L, Linv, b, w,
						s_blk[0].loff, p_flag, s_linv, s_rdiag, s_tile[0]);
				if(!D && !b_y_inline)
					finish_rhs_wide(cd, lane, p, L, b, w, s_linv);
			}
			__syncthreads();
			STAGE_TICK(); // diagonal block done
			for(int kb = 1 + wave; kb < cd.nb; kb += W) {
				const TBlkDesc bd = s_blk[kb];
				const int di = D? D : int(bd.np_di >> 24);
				const TLaneMap m = lane_map(lane, di, dj);
				double acc = 0;
				#pragma unroll
				for(int ww = 0; ww < W; ++ ww)
					acc -= s_part[ww][kb][lane];
				finish_offdiagonal<D>(acc, lane, m.r, m.q, m.b_act, dj, L, bd.loff, s_tile[wave], s_linv);
			}
			__syncthreads(); // LDS records and partial sums consumed; column complete
			STAGE_TICK(); // column done
			continue;
		}
		// general path: any number of blocks, CH at a time, records read from global memory
		for(int kb0 = 0; kb0 < cd.nb; kb0 += CH) {
			const int kbn = min(int(CH), cd.nb - kb0);
			for(int s = 0; s < kbn; ++ s) {
				const int kb = kb0 + s;
				double part;
				if(kb == 0)
					part = accumulate_row<D>(p.rents, cd.r0, cd.nr, wave, W, L, w, md.r, yq, dj, b_y);
				else {
					const TBlkDesc bd = p.blks[cd.k0 + kb];
					const int di = D? D : int(bd.np_di >> 24), np = int(bd.np_di & 0xffffff);
					const TLaneMap m = lane_map(lane, di, dj);
					part = accumulate_pairs<D>(p.pairs, bd.p0, np, wave, W, L, m.r, m.q, di, dj);
				}
				s_part[wave][s][lane] = part;
			}
			__syncthreads();
			if(kb0 == 0) {
				if(wave == 0) {
					const TBlkDesc bd = p.blks[cd.k0];
					double acc = b_y? b[cd.cs_src + yq] : (md.b_act? lambda_element(A, bd.asrc, md.r, md.q, dj, dj, true) : 0);
					#pragma unroll
					for(int ww = 0; ww < W; ++ ww)
						acc -= s_part[ww][0][lane];
					if(D)
						finish_diagonal_fixed<(D? D : 1)>(cd, acc, acc, lane, b_y? 0 : md.r, b_y? 0 : md.q, md.b_act, L, Linv, w,
							bd.loff, p_flag, s_linv);
					else
						finish_diagonal<D>(cd, acc, acc, lane, b_y? 0 : md.r, b_y? 0 : md.q, md.b_act, b_y_inline, p, L, Linv, b, w,
							bd.loff, p_flag, s_linv, s_rdiag, s_tile[0]);
					if(!D && !b_y_inline)
						finish_rhs_wide(cd, lane, p, L, b, w, s_linv);
				}
				__syncthreads();
			}
			for(int s = wave; s < kbn; s += W) {
				const int kb = kb0 + s;
				if(kb == 0)
					continue;
				const TBlkDesc bd = p.blks[cd.k0 + kb];
				const int di = D? D : int(bd.np_di >> 24);
				const TLaneMap m = lane_map(lane, di, dj);
				double acc = lambda_element(A, bd.asrc, m.r, m.q, di, dj, false);
				#pragma unroll
				for(int ww = 0; ww < W; ++ ww)
					acc -= s_part[ww][s][lane];
				finish_offdiagonal<D>(acc, lane, m.r, m.q, m.b_act, dj, L, bd.loff, s_tile[wave], s_linv);
			}
			__syncthreads(); // partial sums consumed; column complete after the last chunk
		}
	}
}

// ---- substitutions: one wave per task, one lane per block of the row / column ----
template <int D>
__device__ __forceinline__ void reduce_over_wave(double (&v)[D? D : 8])
{
	#pragma unroll
	for(int i = 0; i < (D? D : 8); ++ i) {
		#pragma unroll
		for(int m = 32; m >= 1; m >>= 1)
			v[i] += __shfl_xor(v[i], m);
	}
}

// stand-alone forward substitution y_j = inv(L_jj) (b_j - sum_c L(j,c) y_c) (solve_again)
template <int D>
__global__ void __launch_bounds__(64)
forward_stage_kernel(TDevPlan p, const double *L, const double *Linv, const double *__restrict__ b,
	double *w, int task_begin)
{
	enum { DM = D? D : 8 };
	const int lane = threadIdx.x;
	const int task = task_begin + blockIdx.x;
	const int64_t c_end = p.task_ptr[task + 1];
	for(int64_t c = p.task_ptr[task]; c < c_end; ++ c) {
		const TColDesc cd = p.cols[c];
		const int dj = D? D : cd.dj;
		double v[DM];
		#pragma unroll
		for(int i = 0; i < DM; ++ i)
			v[i] = 0;
		for(int e = lane; e < cd.nr; e += 64) {
			const TRowEnt en = p.rents[cd.r0 + e];
			const double *Lb = L + en.off, *y = w + en.ycs;
			const int dc = D? D : en.dc;
			#pragma unroll
			for(int t = 0; t < DM; ++ t) {
				if(t < dc) {
					const double yt = y[t];
					#pragma unroll
					for(int i = 0; i < DM; ++ i)
						if(i < dj) v[i] += Lb[i + t * dj] * yt;
				}
			}
		}
		reduce_over_wave<D>(v);
		double val = 0;
		#pragma unroll
		for(int i = 0; i < DM; ++ i)
			if(lane == i) val = v[i];
		val = (lane < dj)? b[cd.cs_src + lane] - val : 0;
		const double *Li = Linv + cd.linv_off;
		double y = 0;
		for(int t = 0; t < dj; ++ t) {
			const double vt = __shfl(val, t);
			if(lane < dj && t <= lane)
				y += Li[lane + t * dj] * vt;
		}
		if(lane < dj)
			w[cd.cs_new + lane] = y;
		__syncthreads(); // single-wave workgroup: y_j visible to the following columns
	}
}

// backward substitution x_j = inv(L_jj)^T (y_j - sum_i L(i,j)^T x_i); tasks and columns in reverse.
// Lanes = 8 block slots x 8 columns of the block: lane (g, q) sums L(i,j)[:,q]^T x_i over the blocks
// kb = 1 + g, 9 + g, ...; three shuffle steps combine the slots (a column of an elimination subtree
// has 2-3 sub-diagonal blocks, a separator column a few dozen).
// Overwrites the workspace in place and scatters x to its original position.
template <int D>
__global__ void __launch_bounds__(64)
backward_stage_kernel(TDevPlan p, const double *L, const double *Linv, double *w,
	double *__restrict__ x_out, int task_begin, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; L += n_member * t_batch.l; Linv += n_member * t_batch.linv; w += n_member * t_batch.w; x_out += n_member * t_batch.b; } // (TBatch: sparse_kernels.h)

	const int lane = threadIdx.x, g = lane >> 3, q = lane & 7;
	const int task = p.task_map? p.task_map[task_begin + blockIdx.x] : task_begin + blockIdx.x;
	const int64_t c_begin = p.task_ptr[task], c_end = p.task_ptr[task + 1];
	if(c_end <= c_begin)
		return;
	if(D) {
		// Fixed block size.  A column step should cost one trip to memory, not three (column record -> block record ->
		// values): the records do not depend on any x, so they are requested ahead (the column's: two columns, the lane
		// group's first block's: one), as scalars -- a record selected by a condition or copied as a whole goes through
		// scratch memory, and a kernel with scratch takes 14 us longer to launch -- and everything a step multiplies (x of
		// the block's row, which the step before may just have written, the block, inv(L_jj), y_j) is requested together
		// at its start.  (Values requested a column ahead as well cost 30 more registers and a third of the waves per CU for
		// nothing: C3's 22 backward launches 126 us that way, 122 us this way, 184 us with the three trips.)
		enum { DN = D? D : 1 };
		const int qq = (q < D)? q : 0;
		// column records: this column, the next one (c - 2), the one after (c - 3); block records: this column, the next one
		int64_t n_linv, n_cs, n_src, n_k0_1, n_linv_1, n_cs_1, n_src_1, n_k0_2, n_linv_2, n_cs_2, n_src_2, n_loff, n_loff_1;
		int n_nb, n_nb_1, n_nb_2, n_xcs, n_xcs_1;
		{
			const TColDesc &r_cd = p.cols[c_end - 1], &r_cd_1 = p.cols[max(c_end - 2, c_begin)], &r_cd_2 = p.cols[max(c_end - 3, c_begin)];
			n_nb = r_cd.nb; n_linv = r_cd.linv_off; n_cs = r_cd.cs_new; n_src = r_cd.cs_src;
			n_k0_1 = r_cd_1.k0; n_nb_1 = r_cd_1.nb; n_linv_1 = r_cd_1.linv_off; n_cs_1 = r_cd_1.cs_new; n_src_1 = r_cd_1.cs_src;
			n_k0_2 = r_cd_2.k0; n_nb_2 = r_cd_2.nb; n_linv_2 = r_cd_2.linv_off; n_cs_2 = r_cd_2.cs_new; n_src_2 = r_cd_2.cs_src;
			const TBlkDesc &r_bd = p.blks[r_cd.k0 + min(1 + g, n_nb - 1)], &r_bd_1 = p.blks[n_k0_1 + min(1 + g, n_nb_1 - 1)];
			n_loff = r_bd.loff; n_xcs = r_bd.xcs;
			n_loff_1 = r_bd_1.loff; n_xcs_1 = r_bd_1.xcs;
		}
		for(int64_t c = c_end; c > c_begin; -- c) {
			// this step's numbers (no branch around the requests: the addresses are valid either way)
			double xv[DN], lv[DN], li[DN];
			#pragma unroll
			for(int t = 0; t < DN; ++ t)
				xv[t] = w[n_xcs + t];
			#pragma unroll
			for(int t = 0; t < DN; ++ t) {
				lv[t] = L[n_loff + qq * D + t];
				li[t] = Linv[n_linv + t + qq * D];
			}
			const double y = w[n_cs + qq];
			// records for later steps: the column three steps on, the block two steps on (its column's record is here already)
			const TColDesc &r_cd_3 = p.cols[max(c - 4, c_begin)];
			const int64_t n_k0_3 = r_cd_3.k0, n_linv_3 = r_cd_3.linv_off, n_cs_3 = r_cd_3.cs_new, n_src_3 = r_cd_3.cs_src;
			const int n_nb_3 = r_cd_3.nb;
			const TBlkDesc &r_bd_2 = p.blks[n_k0_2 + min(1 + g, n_nb_2 - 1)];
			const int64_t n_loff_2 = r_bd_2.loff;
			const int n_xcs_2 = r_bd_2.xcs;
			double acc = 0;
			#pragma unroll
			for(int t = 0; t < DN; ++ t)
				acc += lv[t] * xv[t];
			acc = (1 + g < n_nb)? acc : 0.0;
			if(n_nb > 9) { // (more than eight blocks below the diagonal: the rest as they come)
				const int64_t n_k0 = p.cols[c - 1].k0;
				for(int kb = 9 + g; kb < n_nb; kb += 8) {
					const int64_t n_loff_k = p.blks[n_k0 + kb].loff;
					const int n_xcs_k = p.blks[n_k0 + kb].xcs;
					#pragma unroll
					for(int t = 0; t < DN; ++ t)
						acc += L[n_loff_k + qq * D + t] * w[n_xcs_k + t];
				}
			}
			acc += __shfl_xor(acc, 8);
			acc += __shfl_xor(acc, 16);
			acc += __shfl_xor(acc, 32);
			const double val = (q < D)? y - acc : 0; // every slot g holds the same totals
			double x = 0;
			#pragma unroll
			for(int t = 0; t < DN; ++ t) {
				const double vt = __shfl(val, t);
				if(t >= qq)
					x += li[t] * vt;
			}
			if(lane < D) {
				w[n_cs + lane] = x;
				x_out[n_src + lane] = x;
			}
			__syncthreads(); // x_j is where the next column's blocks look for it
			n_nb = n_nb_1; n_linv = n_linv_1; n_cs = n_cs_1; n_src = n_src_1;
			n_k0_1 = n_k0_2; n_nb_1 = n_nb_2; n_linv_1 = n_linv_2; n_cs_1 = n_cs_2; n_src_1 = n_src_2;
			n_k0_2 = n_k0_3; n_nb_2 = n_nb_3; n_linv_2 = n_linv_3; n_cs_2 = n_cs_3; n_src_2 = n_src_3;
			n_loff = n_loff_1; n_xcs = n_xcs_1;
			n_loff_1 = n_loff_2; n_xcs_1 = n_xcs_2;
		}
		return;
	}
	TColDesc cd_next = p.cols[c_end - 1];
	for(int64_t c = c_end; c > c_begin; -- c) {
		const TColDesc cd = cd_next;
		if(c - 1 > c_begin)
			cd_next = p.cols[c - 2]; // index data does not depend on the numbers: fetch it a column ahead
		const int dj = cd.dj;
		const int qq = (q < dj)? q : 0;
		double acc = 0;
		for(int kb = 1 + g; kb < cd.nb; kb += 8) {
			const TBlkDesc bd = p.blks[cd.k0 + kb];
			const int di = int(bd.np_di >> 24);
			const double *Lb = L + bd.loff + qq * di, *x = w + bd.xcs;
			for(int t = 0; t < di; ++ t)
				acc += Lb[t] * x[t];
		}
		acc += __shfl_xor(acc, 8);
		acc += __shfl_xor(acc, 16);
		acc += __shfl_xor(acc, 32);
		const double val = (q < dj)? w[cd.cs_new + q] - acc : 0; // every slot g holds the same totals
		const double *Li = Linv + cd.linv_off;
		double x = 0;
		for(int t = 0; t < dj; ++ t) {
			const double vt = __shfl(val, t);
			if(t >= qq)
				x += Li[t + qq * dj] * vt;
		}
		__syncthreads(); // every lane has read y_j before it is overwritten
		if(lane < dj) {
			w[cd.cs_new + lane] = x;
			x_out[cd.cs_src + lane] = x;
		}
		__syncthreads();
	}
}

// ---- dense top: assemble the Schur complement onto the dense-top columns ----
// one workgroup of 4 waves per block (i,j), j in the dense top: D(i,j) = Lambda(i,j) - sum of the updates
// from block-eliminated columns (the updates among dense-top columns happen in dense_cholesky);
// diagonal blocks also produce the right-hand side  b_j - sum_c L(j,c) y_c  into row ld-1
template <int D>
__global__ void __launch_bounds__(256)
dense_assemble_kernel(TDevPlan p, const TDenseBlk *__restrict__ blks, const double *__restrict__ A, const double *L,
	const double *__restrict__ b, const double *w, double *Dm, int ld, int b_rhs_only)
{
	__shared__ double s_part[4][64];
	const TDenseBlk bd = blks[blockIdx.x];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int di = D? D : bd.di, dj = D? D : bd.dj;
	const bool b_diag = bd.nr >= 0;
	if(b_rhs_only && !b_diag)
		return;
	const bool b_y_inline = dj <= 7;
	const bool b_y = b_diag && b_y_inline && lane >= Y_LANE0 && lane < Y_LANE0 + dj;
	const TLaneMap m = lane_map(lane, di, dj);
	const int yq = b_y? lane - Y_LANE0 : m.q;
	double part;
	if(b_y)
		part = accumulate_row<D>(p.rents, bd.r0, bd.nr, wave, 4, L, w, 0, yq, dj, true);
	else
		part = b_rhs_only? 0.0 : accumulate_pairs<D>(p.pairs, bd.p0, bd.np, wave, 4, L, m.r, m.q, di, dj);
	s_part[wave][lane] = part;
	__syncthreads();
	if(wave == 0) {
		const double sum = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
		if(b_y)
			Dm[size_t(ld - 1) + size_t(bd.pos + yq) * ld] = b[bd.cs_src + yq] - sum;
		else if(m.b_act && !b_rhs_only)
			Dm[bd.dst + m.r + size_t(m.q) * ld] = lambda_element(A, bd.asrc, m.r, m.q, di, dj, b_diag) - sum;
		if(!D && b_diag && !b_y_inline) { // 8-wide column: right-hand side by lanes 0..7 in a second pass
			const int q = lane & 7;
			double ay = accumulate_row<0>(p.rents, bd.r0, bd.nr, lane >> 3, 8, L, w, 0, q, dj, true);
			ay += __shfl_xor(ay, 8);
			ay += __shfl_xor(ay, 16);
			ay += __shfl_xor(ay, 32);
			if(lane < dj)
				Dm[size_t(ld - 1) + size_t(bd.pos + lane) * ld] = b[bd.cs_src + lane] - ay;
		}
	}
}

// ---- launchers ----
#define DISPATCH_DIM(D_runtime, CALL) do { switch(D_runtime) { \
	case 3: { enum { D = 3 }; CALL; } break; \
	case 6: { enum { D = 6 }; CALL; } break; \
	case 7: { enum { D = 7 }; CALL; } break; \
	default: { enum { D = 0 }; CALL; } break; } } while(0)

void launch_factor_stage(const TDevPlan &p, const double *A, double *L, double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, bool b_bottom_stage, int *p_flag, hipStream_t stream, const TBatch &t_batch)
{
	if(n_tasks <= 0)
		return;
	if(b_bottom_stage) { // one wave per task (the host decides which stages: solver.hip, n_bottom_stages)
		const bool b_first_version = dev_knob_set("SLAMPP_HIP_DEV_SUBTREE_V1"); // development aid: A/B timing (plan.h)
		if(!b_first_version && launch_factor_subtree_image(p, A, L, Linv, b, w, task_begin, n_tasks, p_flag, stream, t_batch))
			return;
		DISPATCH_DIM(p.uniform_dim, hipLaunchKernelGGL((factor_subtree_kernel<D>), dim3(n_tasks, t_batch.n), dim3(64), 0, stream,
			p, A, L, Linv, b, w, task_begin, p_flag, t_batch));
	} else {
		DISPATCH_DIM(p.uniform_dim, hipLaunchKernelGGL((factor_stage_kernel<D, 8, CHUNK, UP_NR, UP_NP>), dim3(n_tasks, t_batch.n), dim3(512), 0, stream,
			p, A, L, Linv, b, w, task_begin, p_flag, t_batch));
	}
}

void launch_factor_wide(const TDevPlan &p, const double *A, double *L, double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, int *p_flag, hipStream_t stream, const TBatch &t_batch)
{
	if(n_tasks > 0)
		DISPATCH_DIM(p.uniform_dim, hipLaunchKernelGGL((factor_stage_kernel<D, 1, WIDE_CHUNK, WIDE_NR, WIDE_NP>), dim3(n_tasks, t_batch.n), dim3(64), 0, stream,
			p, A, L, Linv, b, w, task_begin, p_flag, t_batch));
}

void launch_forward_stage(const TDevPlan &p, const double *L, const double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, hipStream_t stream)
{
	if(n_tasks > 0)
		DISPATCH_DIM(p.uniform_dim, hipLaunchKernelGGL((forward_stage_kernel<D>), dim3(n_tasks), dim3(64), 0, stream,
			p, L, Linv, b, w, task_begin));
}

void launch_backward_stage(const TDevPlan &p, const double *L, const double *Linv, double *w,
	double *x_out, int task_begin, int n_tasks, hipStream_t stream, const TBatch &t_batch)
{
	if(n_tasks > 0)
		DISPATCH_DIM(p.uniform_dim, hipLaunchKernelGGL((backward_stage_kernel<D>), dim3(n_tasks, t_batch.n), dim3(64), 0, stream,
			p, L, Linv, w, x_out, task_begin, t_batch));
}

void launch_dense_assemble(const TDevPlan &p, const TDenseBlk *blks, int n_blks, const double *A, const double *L,
	const double *b, const double *w, double *Dm, int ld, bool b_rhs_only, hipStream_t stream)
{
	if(n_blks > 0)
		DISPATCH_DIM(p.uniform_dim, hipLaunchKernelGGL((dense_assemble_kernel<D>), dim3(n_blks), dim3(256), 0, stream,
			p, blks, A, L, b, w, Dm, ld, int(b_rhs_only)));
}

// the dense top's part of the factor back into the factor's block layout (slampp_hip_factorize: the caller wants every
// column of L): block e of the dense-top columns from the dense matrix, zeros above the diagonal of a diagonal block
__global__ void __launch_bounds__(64)
dense_gather_factor_kernel(const TDenseBlk *__restrict__ blks, const int64_t *__restrict__ loffs, const double *__restrict__ Dm, int ld,
	double *__restrict__ L)
{
	const TDenseBlk bd = blks[blockIdx.x];
	const int lane = threadIdx.x;
	if(lane >= bd.di * bd.dj)
		return;
	const int r = lane % bd.di, q = lane / bd.di;
	const bool b_diag = bd.nr >= 0;
	L[loffs[blockIdx.x] + lane] = (b_diag && r < q)? 0.0 : Dm[bd.dst + r + size_t(q) * ld];
}

void launch_dense_gather_factor(const TDenseBlk *blks, const int64_t *loffs, int n_blks, const double *Dm, int ld, double *L, hipStream_t stream)
{
	if(n_blks > 0)
		hipLaunchKernelGGL(dense_gather_factor_kernel, dim3(n_blks), dim3(64), 0, stream, blks, loffs, Dm, ld, L);
}

// values of a structure whose wide block columns were cut into pieces (solver.hip: Refine_Structure): dst[i] = src[map[i]]
__global__ void gather_values_kernel(const int64_t *__restrict__ p_map, int64_t n, const double *__restrict__ p_src,
	double *__restrict__ p_dst)
{
	for(int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
		p_dst[i] = p_src[p_map[i]];
}

void launch_gather_values(const int64_t *p_map, int64_t n, const double *p_src, double *p_dst, hipStream_t stream)
{
	if(n > 0)
		hipLaunchKernelGGL(gather_values_kernel, dim3(unsigned(std::min<int64_t>((n + 255) / 256, 8192))), dim3(256), 0, stream,
			p_map, n, p_src, p_dst);
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(sparse_kernels) // (the handle's bring-up thread loads this unit's code object: capi.hip)
