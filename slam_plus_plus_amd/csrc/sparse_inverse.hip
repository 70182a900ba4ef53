// sparse_inverse.hip -- the entries of Z = (L L^T)^-1 on the block pattern of the factor L (a "sparse inverse subset",
// Takahashi's equations), for the marginal covariances of BA systems whose reduced camera system is factored by the
// sparse block path: every camera pair that shares a landmark is a block of S, hence of L's pattern, so the landmark
// covariances C_p^-1 + W_p^T Z W_p need nothing outside that pattern -- and neither does the recursion.  The
// reference's counterpart is CMarginals::Calculate_DenseMarginals_Recurrent_FBS (/root/reference/include/slam/
// Marginals.h, called from BAMarginals.h:765-772 for the camera blocks) on its own block Cholesky factor.
//
// Columns are processed from the root of the elimination tree downwards (the factorization's stages in reverse, one
// launch each, one wave per task, a task's columns last to first).  With struct(j) the block rows of column j below
// the diagonal, all of them ancestors of j and therefore done:
//     Z(i,j) = -( sum over k in struct(j) of Z(i,k) L(k,j) ) inv(L_jj)          for i in struct(j)
//     Z(j,j) = inv(L_jj)^T ( inv(L_jj) - sum over k in struct(j) of L(k,j)^T Z(k,j) )
// Z(i,k) is the stored block (max, min) of the pair, transposed when i < k; it exists because struct(j) is a clique
// of the filled graph.  Z lives in an array laid out like L.  |struct|^2 small products per column: the cost of the
// factorization again, latency-bound like it.
// With a dense top (plan.h) the blocks among dense-top columns are not in that array: they are read from the dense
// inverse of the top's Schur complement (dense_inverse.hip), whose lower triangle and diagonal tiles are valid; the
// recursion then covers the block-eliminated columns below it.
#include <hip/hip_runtime.h>
#include "sparse_inverse.h"
#include "solver.h"

#include <algorithm>

namespace slampp {

struct TInvCol { // 40 B
	int64_t zdiag;   // offset of block (j,j) in L / Z
	int64_t linv;    // offset of inv(L_jj)
	int64_t b0;      // first entry of the column's block-offset list (sub-diagonal blocks, rows ascending)
	int64_t t0;      // first of its (nb-1)^2 term records
	int32_t nbm;     // number of sub-diagonal blocks
	int32_t dj;      // dimension of the column (mixed block sizes: D = 0)
};

struct CSparseInverse {
	int D;
	int64_t n_cols;
	CDevArray<TInvCol> d_cols;      // in schedule order (the plan's task_cols)
	CDevArray<int64_t> d_blk_off;   // offsets of the sub-diagonal blocks of every column
	CDevArray<int32_t> d_blk_dim;   // their row dimensions (mixed block sizes only)
	CDevArray<int64_t> d_terms;     // (offset of the stored block of Z(i,k)) << 2 | transposed, or (pos_i << 24 | pos_k) << 2 | 2 for the dense top
	CDevArray<int64_t> d_task_ptr;
};

void sparse_inverse_destroy(CSparseInverse *p) { delete p; }

size_t sparse_inverse_bytes(const CSparseInverse *p)
{
	return p? p->d_cols.n_Bytes() + p->d_blk_off.n_Bytes() + p->d_blk_dim.n_Bytes() + p->d_terms.n_Bytes() + p->d_task_ptr.n_Bytes() : 0;
}

// offset of the factor block (i, k), i >= k, or -1
int64_t plan_block_offset(const Plan &P, int32_t i, int32_t k)
{
	const int32_t *b = P.lrow.data() + P.lptr[k], *e = P.lrow.data() + P.lptr[k + 1];
	const int32_t *f = std::lower_bound(b, e, i);
	return (f != e && *f == i)? P.loff[f - P.lrow.data()] : -1;
}

CSparseInverse *sparse_inverse_setup(const Plan &P, hipStream_t stream, bool b_allow_dense_top)
{
	// one block size of 3, 6 or 7 (the unrolled kernels), or -- round 4 -- any mix of block sizes up to 8 (the reference's
	// CMarginals takes any, Marginals.h:1694: poses and landmarks in one graph), that one without a dense top
	const bool b_fixed = P.uniform_dim && (P.max_dim == 3 || P.max_dim == 6 || P.max_dim == 7);
	if((!b_fixed && (P.max_dim > 8 || P.dense_dim != 0)) || (P.dense_dim != 0 && !b_allow_dense_top) || P.dense_dim >= (1 << 24))
		return 0; // the dense inverse serves these
	const int64_t n_sched = int64_t(P.task_cols.size());
	std::vector<TInvCol> cols(n_sched);
	std::vector<int64_t> blk_off, terms;
	std::vector<int32_t> blk_dim;
	for(int64_t s = 0; s < n_sched; ++ s) {
		const int32_t j = P.task_cols[s];
		TInvCol &c = cols[s];
		c.zdiag = P.loff[P.lptr[j]];
		c.linv = P.linv_off[j];
		c.nbm = int32_t(P.lptr[j + 1] - P.lptr[j] - 1);
		c.dj = P.dim[j];
		c.b0 = int64_t(blk_off.size());
		c.t0 = int64_t(terms.size());
		for(int64_t k = P.lptr[j] + 1; k < P.lptr[j + 1]; ++ k) {
			blk_off.push_back(P.loff[k]);
			if(!b_fixed)
				blk_dim.push_back(P.dim[P.lrow[k]]);
		}
		for(int64_t a = P.lptr[j] + 1; a < P.lptr[j + 1]; ++ a) {
			for(int64_t b = P.lptr[j] + 1; b < P.lptr[j + 1]; ++ b) {
				const int32_t i = P.lrow[a], k = P.lrow[b];
				if(P.dense_dim && P.dense_pos[i] >= 0 && P.dense_pos[k] >= 0) { // both in the dense top: element-wise from its inverse
					terms.push_back((((int64_t(P.dense_pos[i]) << 24) | int64_t(P.dense_pos[k])) << 2) | 2);
					continue;
				}
				const int64_t off = plan_block_offset(P, std::max(i, k), std::min(i, k));
				if(off < 0)
					throw std::logic_error("sparse inverse: the structure of a factor column is not a clique");
				terms.push_back((off << 2) | (i < k));
			}
		}
	}
	CSparseInverse *p = new CSparseInverse();
	try {
		p->D = b_fixed? P.max_dim : 0;
		p->n_cols = n_sched;
		if(!b_fixed)
			p->d_blk_dim.Upload(blk_dim, stream);
		p->d_cols.Upload(cols, stream);
		p->d_blk_off.Upload(blk_off, stream);
		p->d_terms.Upload(terms, stream);
		p->d_task_ptr.Upload(P.task_ptr, stream);
		SLAMPP_HIP_CHECK(hipStreamSynchronize(stream));
	} catch(...) {
		delete p;
		throw;
	}
	return p;
}

__device__ __forceinline__ void inv_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int D>
__global__ void __launch_bounds__(64)
sparse_inverse_stage_kernel(const TInvCol *__restrict__ cols, const int64_t *__restrict__ blk_off,
	const int64_t *__restrict__ terms, const int64_t *__restrict__ task_ptr, int task_begin,
	const double *__restrict__ L, const double *__restrict__ Linv, double *Z, const double *__restrict__ Zd, int ld)
{
	__shared__ double s_linv[64], s_tile[64], s_z[64];
	const int lane = threadIdx.x;
	const bool b_act = lane < D * D;
	const int r = b_act? lane % D : 0, q = b_act? lane / D : 0;
	const int task = task_begin + blockIdx.x;
	for(int64_t c = task_ptr[task + 1]; c > task_ptr[task]; -- c) {
		const TInvCol cd = cols[c - 1];
		if(b_act)
			s_linv[r + 8 * q] = Linv[cd.linv + lane];
		double m_acc = 0; // sum over k of L(k,j)^T Z(k,j), element (r, q)
		for(int kb = 0; kb < cd.nbm; ++ kb) {
			const int64_t zoff_i = blk_off[cd.b0 + kb];
			double acc = 0; // ( sum over k of Z(i,k) L(k,j) )(r, q)
			#pragma unroll 2
			for(int kk = 0; kk < cd.nbm; ++ kk) {
				const int64_t term = terms[cd.t0 + int64_t(kb) * cd.nbm + kk];
				const double *Ls = L + blk_off[cd.b0 + kk] + q * D;
				double zv[D], lv[D];
				if(term & 2) { // both rows in the dense top: element (pos_i + r, pos_k + t) of its inverse, read as (max, min)
					const int64_t pi = (term >> 26) + r, pk = (term >> 2) & 0xffffff;
					#pragma unroll
					for(int t = 0; t < D; ++ t) {
						const int64_t a = pi, b = pk + t;
						zv[t] = Zd[((a > b)? a : b) + ((a > b)? b : a) * int64_t(ld)];
						lv[t] = Ls[t];
					}
				} else {
					const double *Zs = Z + (term >> 2);
					const int zs = (term & 1)? 1 : D, z0 = (term & 1)? r * D : r; // Z(i,k)(r, t): stored block or its transpose
					#pragma unroll
					for(int t = 0; t < D; ++ t) {
						zv[t] = Zs[z0 + t * zs];
						lv[t] = Ls[t];
					}
				}
				#pragma unroll
				for(int t = 0; t < D; ++ t)
					acc += zv[t] * lv[t];
			}
			inv_wave_sync(); // the previous uses of the tiles are over
			if(b_act)
				s_tile[r + 8 * q] = acc;
			inv_wave_sync();
			double v = 0;
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				v += s_tile[r + 8 * t] * s_linv[t + 8 * q];
			v = -v; // Z(i,j)
			if(b_act) {
				Z[zoff_i + lane] = v;
				s_z[r + 8 * q] = v;
			}
			inv_wave_sync();
			const double *Lij = L + zoff_i + r * D; // column r of L(i,j): L(i,j)^T's row r
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				m_acc += Lij[t] * s_z[t + 8 * q];
		}
		inv_wave_sync();
		if(b_act)
			s_tile[r + 8 * q] = s_linv[r + 8 * q] - m_acc; // inv(L_jj) - M
		inv_wave_sync();
		double zjj = 0;
		#pragma unroll
		for(int t = 0; t < D; ++ t)
			zjj += s_linv[t + 8 * r] * s_tile[t + 8 * q];
		if(b_act)
			Z[cd.zdiag + lane] = zjj;
		__syncthreads(); // column j is complete and visible before a descendant in this task reads it
	}
}

// The same for any mix of block sizes up to 8 (poses and landmarks in one pose graph, SE(2) and SE(3) vertices together):
// dimensions read from the records, a lane per element of the target block, one wave per task at every stage -- what the
// reference's CMarginals does for such systems on the host (Marginals.h:1694); a completeness path, not a tuned one.
__global__ void __launch_bounds__(64)
sparse_inverse_stage_any_kernel(const TInvCol *__restrict__ cols, const int64_t *__restrict__ blk_off, const int32_t *__restrict__ blk_dim,
	const int64_t *__restrict__ terms, const int64_t *__restrict__ task_ptr, int task_begin,
	const double *__restrict__ L, const double *__restrict__ Linv, double *Z)
{
	__shared__ double s_linv[64], s_tile[64], s_z[64];
	const int lane = threadIdx.x;
	const int task = task_begin + blockIdx.x;
	for(int64_t c = task_ptr[task + 1]; c > task_ptr[task]; -- c) {
		const TInvCol cd = cols[c - 1];
		const int dj = cd.dj;
		inv_wave_sync();
		if(lane < dj * dj)
			s_linv[lane % dj + 8 * (lane / dj)] = Linv[cd.linv + lane];
		inv_wave_sync();
		const int rj = (lane < dj * dj)? lane % dj : 0, qj = (lane < dj * dj)? lane / dj : 0; // element of a dj x dj block
		double m_acc = 0; // sum over i of L(i,j)^T Z(i,j), element (rj, qj)
		for(int kb = 0; kb < cd.nbm; ++ kb) {
			const int64_t zoff_i = blk_off[cd.b0 + kb];
			const int di = blk_dim[cd.b0 + kb];
			const bool b_act = lane < di * dj;
			const int r = b_act? lane % di : 0, q = b_act? lane / di : 0; // element of the di x dj target
			double acc = 0; // ( sum over k of Z(i,k) L(k,j) )(r, q)
			for(int kk = 0; kk < cd.nbm; ++ kk) {
				const int64_t term = terms[cd.t0 + int64_t(kb) * cd.nbm + kk];
				const int dk = blk_dim[cd.b0 + kk];
				const double *Ls = L + blk_off[cd.b0 + kk] + q * dk; // column q of L(k,j)
				const double *Zs = Z + (term >> 2);
				// Z(i,k)(r, t): the stored block (i,k), di x dk, or the transpose of the stored block (k,i), dk x di
				const int zs = (term & 1)? 1 : di, z0 = (term & 1)? r * dk : r;
				for(int t = 0; t < dk; ++ t)
					acc += Zs[z0 + t * zs] * Ls[t];
			}
			inv_wave_sync(); // the previous uses of the tiles are over
			if(b_act)
				s_tile[r + 8 * q] = acc;
			inv_wave_sync();
			double v = 0;
			for(int t = 0; t < dj; ++ t)
				v += s_tile[r + 8 * t] * s_linv[t + 8 * q];
			v = -v; // Z(i,j)
			if(b_act) {
				Z[zoff_i + lane] = v;
				s_z[r + 8 * q] = v;
			}
			inv_wave_sync();
			if(lane < dj * dj) {
				const double *Lij = L + zoff_i + rj * di; // column rj of L(i,j): row rj of its transpose
				for(int t = 0; t < di; ++ t)
					m_acc += Lij[t] * s_z[t + 8 * qj];
			}
		}
		inv_wave_sync();
		if(lane < dj * dj)
			s_tile[rj + 8 * qj] = s_linv[rj + 8 * qj] - m_acc; // inv(L_jj) - M
		inv_wave_sync();
		double zjj = 0;
		for(int t = 0; t < dj; ++ t)
			zjj += s_linv[t + 8 * rj] * s_tile[t + 8 * qj];
		if(lane < dj * dj)
			Z[cd.zdiag + lane] = zjj;
		__syncthreads(); // column j is complete and visible before a descendant in this task reads it
	}
}

// The same with W waves per task for the stages near the root, where a stage is a handful of columns with many blocks
// each and one wave walked (|struct|^2 terms) through them alone: wave w takes the target blocks w, w + W, ... of the
// column, finishes them, and the partial sums of L(k,j)^T Z(k,j) meet in LDS in a fixed order for Z(j,j).
template <int D, int W>
__global__ void __launch_bounds__(64 * W)
sparse_inverse_wide_kernel(const TInvCol *__restrict__ cols, const int64_t *__restrict__ blk_off,
	const int64_t *__restrict__ terms, const int64_t *__restrict__ task_ptr, int task_begin,
	const double *__restrict__ L, const double *__restrict__ Linv, double *Z, const double *__restrict__ Zd, int ld)
{
	__shared__ double s_linv[64], s_n[64], s_tile[W][64], s_z[W][64], s_m[W][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const bool b_act = lane < D * D;
	const int r = b_act? lane % D : 0, q = b_act? lane / D : 0;
	const int task = task_begin + blockIdx.x;
	for(int64_t c = task_ptr[task + 1]; c > task_ptr[task]; -- c) {
		const TInvCol cd = cols[c - 1];
		if(wave == 0 && b_act)
			s_linv[r + 8 * q] = Linv[cd.linv + lane];
		__syncthreads();
		double m_acc = 0;
		for(int kb = wave; kb < cd.nbm; kb += W) {
			const int64_t zoff_i = blk_off[cd.b0 + kb];
			double acc = 0;
			#pragma unroll 2
			for(int kk = 0; kk < cd.nbm; ++ kk) {
				const int64_t term = terms[cd.t0 + int64_t(kb) * cd.nbm + kk];
				const double *Ls = L + blk_off[cd.b0 + kk] + q * D;
				double zv[D], lv[D];
				if(term & 2) {
					const int64_t pi = (term >> 26) + r, pk = (term >> 2) & 0xffffff;
					#pragma unroll
					for(int t = 0; t < D; ++ t) {
						const int64_t a = pi, b = pk + t;
						zv[t] = Zd[((a > b)? a : b) + ((a > b)? b : a) * int64_t(ld)];
						lv[t] = Ls[t];
					}
				} else {
					const double *Zs = Z + (term >> 2);
					const int zs = (term & 1)? 1 : D, z0 = (term & 1)? r * D : r;
					#pragma unroll
					for(int t = 0; t < D; ++ t) {
						zv[t] = Zs[z0 + t * zs];
						lv[t] = Ls[t];
					}
				}
				#pragma unroll
				for(int t = 0; t < D; ++ t)
					acc += zv[t] * lv[t];
			}
			inv_wave_sync();
			if(b_act)
				s_tile[wave][r + 8 * q] = acc;
			inv_wave_sync();
			double v = 0;
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				v += s_tile[wave][r + 8 * t] * s_linv[t + 8 * q];
			v = -v;
			if(b_act) {
				Z[zoff_i + lane] = v;
				s_z[wave][r + 8 * q] = v;
			}
			inv_wave_sync();
			const double *Lij = L + zoff_i + r * D;
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				m_acc += Lij[t] * s_z[wave][t + 8 * q];
		}
		if(b_act)
			s_m[wave][r + 8 * q] = m_acc;
		__syncthreads();
		if(wave == 0) {
			double m = 0;
			#pragma unroll
			for(int ww = 0; ww < W; ++ ww)
				m += b_act? s_m[ww][r + 8 * q] : 0.0;
			if(b_act)
				s_n[r + 8 * q] = s_linv[r + 8 * q] - m;
			inv_wave_sync();
			double zjj = 0;
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				zjj += s_linv[t + 8 * r] * s_n[t + 8 * q];
			if(b_act)
				Z[cd.zdiag + lane] = zjj;
		}
		__syncthreads(); // column j is complete and visible before a descendant in this task reads it
	}
}

void sparse_inverse_enqueue(const CSparseInverse &r_inv, const Plan &P, const double *L, const double *Linv, double *Z,
	hipStream_t stream, const double *p_dense_top_inverse, int n_dense_ld)
{
	const int n_stages = int(P.stage_ptr.size()) - 1;
	for(int s = n_stages - 1; s >= 0; -- s) {
		const int n_tasks = P.stage_ptr[s + 1] - P.stage_ptr[s];
		if(n_tasks <= 0)
			continue;
		if(!r_inv.D) { // mixed block sizes
			hipLaunchKernelGGL(sparse_inverse_stage_any_kernel, dim3(n_tasks), dim3(64), 0, stream, r_inv.d_cols.p(), r_inv.d_blk_off.p(),
				r_inv.d_blk_dim.p(), r_inv.d_terms.p(), r_inv.d_task_ptr.p(), P.stage_ptr[s], L, Linv, Z);
			continue;
		}
		const bool b_wide = n_tasks <= 1024; // as the factorization splits its stages between one wave and eight per task
#define LAUNCH_INV(DD) do { if(b_wide) \
			hipLaunchKernelGGL((sparse_inverse_wide_kernel<DD, 8>), dim3(n_tasks), dim3(512), 0, stream, r_inv.d_cols.p(), \
				r_inv.d_blk_off.p(), r_inv.d_terms.p(), r_inv.d_task_ptr.p(), P.stage_ptr[s], L, Linv, Z, p_dense_top_inverse, n_dense_ld); \
		else \
			hipLaunchKernelGGL((sparse_inverse_stage_kernel<DD>), dim3(n_tasks), dim3(64), 0, stream, r_inv.d_cols.p(), \
				r_inv.d_blk_off.p(), r_inv.d_terms.p(), r_inv.d_task_ptr.p(), P.stage_ptr[s], L, Linv, Z, p_dense_top_inverse, n_dense_ld); \
		} while(0)
		switch(r_inv.D) {
		case 3: LAUNCH_INV(3); break;
		case 6: LAUNCH_INV(6); break;
		default: LAUNCH_INV(7); break;
		}
#undef LAUNCH_INV
	}
}

// diagonal blocks of the inverse, one d x d column-major block per block column in the caller's order: from Z, or
// (p_where[c] = -(position + 1)) from the dense top's inverse
__global__ void inverse_diag_blocks_kernel(int64_t n, int d, const int64_t *__restrict__ p_where, const double *__restrict__ Z,
	const double *__restrict__ Zd, int ld, double *out)
{
	const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(gid >= n * d * d)
		return;
	const int64_t c = gid / (d * d);
	const int e = int(gid - c * (d * d)), r = e % d, q = e / d;
	const int64_t at = p_where[c];
	if(at >= 0)
		out[gid] = Z[at + e];
	else {
		const int64_t pos = -at - 1, a = pos + ((r > q)? r : q), b = pos + ((r > q)? q : r);
		out[gid] = Zd[a + b * int64_t(ld)];
	}
}

// ... for mixed block sizes: block c is p_dim[c] x p_dim[c], read at p_where[c], written at p_out_off[c]
__global__ void inverse_diag_blocks_any_kernel(int64_t n, const int32_t *__restrict__ p_dim, const int64_t *__restrict__ p_where,
	const int64_t *__restrict__ p_out_off, const double *__restrict__ Z, double *out)
{
	const int64_t c = int64_t(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64;
	const int e = threadIdx.x & 63;
	if(c < n && e < p_dim[c] * p_dim[c])
		out[p_out_off[c] + e] = Z[p_where[c] + e];
}

void inverse_diag_blocks_any_launch(int64_t n, const int32_t *p_dim, const int64_t *p_where, const int64_t *p_out_off, const double *Z,
	double *out, hipStream_t stream)
{
	hipLaunchKernelGGL(inverse_diag_blocks_any_kernel, dim3(unsigned((n + 3) / 4)), dim3(256), 0, stream, n, p_dim, p_where, p_out_off, Z, out);
}

void inverse_diag_blocks_launch(int64_t n, int d, const int64_t *p_where, const double *Z, const double *Zd, int ld, double *out,
	hipStream_t stream)
{
	hipLaunchKernelGGL(inverse_diag_blocks_kernel, dim3(unsigned((n * d * d + 255) / 256)), dim3(256), 0, stream, n, d, p_where, Z,
		Zd, ld, out);
}

// the factor of the dense top carries the right-hand side as its last row: an identity row in the copy that gets inverted
__global__ void dense_top_clear_rhs_row_kernel(double *M, int ld)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i < ld)
		M[size_t(ld - 1) + size_t(i) * ld] = (i == ld - 1)? 1.0 : 0.0;
}

void dense_top_clear_rhs_row(double *M, int ld, hipStream_t stream)
{
	hipLaunchKernelGGL(dense_top_clear_rhs_row_kernel, dim3((ld + 255) / 256), dim3(256), 0, stream, M, ld);
}

} // namespace slampp
