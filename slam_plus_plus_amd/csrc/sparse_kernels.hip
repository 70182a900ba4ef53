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
// Traffic is one read of Lambda, one write of L, plus re-reads of L blocks by the updates that
// mostly hit L2 (the producer ran in the same wave or in the previous stage): HBM-bound, and in
// practice latency-bound by the dependency chain -- see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sparse_kernels.h"

namespace slampp {

__device__ __forceinline__ void wave_sync()
{
	// LDS operations of one wave execute in order; this only stops the compiler from moving
	// LDS accesses across the point where lanes exchange data
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc(r,q) of factor block k: source block of Lambda minus the update pairs
__device__ __forceinline__ double accumulate_block(const TDevPlan &p, const double *__restrict__ A,
	const double *L, int64_t k, int r, int q, int di, int dj, bool b_diag)
{
	double acc = 0;
	const int64_t enc = p.asrc[k];
	if(enc >= 0) {
		const int64_t off = enc >> 1;
		// diagonal blocks: read the upper triangle, which is what the reference's solvers consume
		// (src/slam/LinearSolver_CholMod.cpp:57); off-diagonal: as stored or transposed
		const bool b_trans = (enc & 1) || b_diag;
		acc = b_trans? A[off + q + int64_t(r) * dj] : A[off + r + int64_t(q) * di];
	}
	const int64_t p1 = p.pptr[k + 1];
	for(int64_t e = p.pptr[k]; e < p1; ++ e) {
		const longlong2 pr = p.pairs[e];
		const int dc = int(pr.x >> 56);
		const double *a = L + (pr.x & ((int64_t(1) << 56) - 1)) + r;
		const double *b = L + pr.y + q;
		#pragma unroll 2
		for(int t = 0; t < dc; ++ t)
			acc -= a[t * di] * b[t * dj];
	}
	return acc;
}

// one task = a chain of block columns eliminated in order by one workgroup of W waves
template <int W>
__device__ __forceinline__ void factor_task(const TDevPlan &p, const double *__restrict__ A, double *L,
	double *Linv, int task, int *p_flag)
{
	__shared__ double s_linv[64];      // inv(L_jj), element (r,c) at r + 8 c
	__shared__ double s_rdiag[8];      // 1 / L_jj(k,k)
	__shared__ double s_tile[W][64];

	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int64_t c_end = p.task_ptr[task + 1];
	for(int64_t c = p.task_ptr[task]; c < c_end; ++ c) {
		const int j = p.task_cols[c];
		const int dj = p.dim[j];
		const int64_t k0 = p.lptr[j];
		const int nb = int(p.lptr[j + 1] - k0);

		if(wave == 0) {
			// ---- diagonal block: lanes (r, q), lane = r + q dj ----
			const bool b_act = lane < dj * dj;
			const int r = b_act? lane % dj : 0, q = b_act? lane / dj : 0;
			double a = accumulate_block(p, A, L, k0, r, q, dj, dj, true);
			if(r < q)
				a = 0;
			bool b_bad = false;
			for(int kk = 0; kk < dj; ++ kk) {
				double piv = __shfl(a, kk + kk * dj);
				if(!(piv > 0)) { // also catches NaN
					b_bad = true;
					piv = 1;
				}
				const double s = 1.0 / sqrt(piv);
				const double lcol = a * s; // meaningful in lanes (., kk)
				const double lr = __shfl(lcol, r + kk * dj);
				const double lq = __shfl(lcol, q + kk * dj);
				if(q == kk)
					a = (r >= kk)? lcol : 0;
				else if(q > kk && r >= q)
					a -= lr * lq;
				if(lane == 0)
					s_rdiag[kk] = s;
			}
			if(b_bad && lane == 0)
				atomicOr(p_flag, 1);
			if(b_act) {
				L[p.loff[k0] + lane] = a;
				s_tile[0][r + 8 * q] = a;
			}
			wave_sync();
			// inverse of the lower-triangular L_jj: lane q computes column q by forward substitution
			if(lane < dj) {
				const int cq = lane;
				for(int rr = 0; rr < dj; ++ rr) {
					double x;
					if(rr < cq)
						x = 0;
					else if(rr == cq)
						x = s_rdiag[rr];
					else {
						double sum = 0;
						for(int t = cq; t < rr; ++ t)
							sum += s_tile[0][rr + 8 * t] * s_linv[t + 8 * cq];
						x = -sum * s_rdiag[rr];
					}
					s_linv[rr + 8 * cq] = x;
				}
			}
			wave_sync();
			if(b_act)
				Linv[p.linv_off[j] + lane] = s_linv[r + 8 * q];
		}
		__syncthreads();

		// ---- sub-diagonal blocks: L(i,j) = acc(i,j) * inv(L_jj)^T ----
		for(int kb = 1 + wave; kb < nb; kb += W) {
			const int64_t k = k0 + kb;
			const int di = p.dim[p.lrow[k]];
			const bool b_act = lane < di * dj;
			const int r = b_act? lane % di : 0, q = b_act? lane / di : 0;
			const double acc = accumulate_block(p, A, L, k, r, q, di, dj, false);
			wave_sync(); // the previous iteration's reads of the tile are done
			if(b_act)
				s_tile[wave][r + 8 * q] = acc;
			wave_sync();
			double v = 0;
			for(int t = 0; t <= q; ++ t)
				v += s_tile[wave][r + 8 * t] * s_linv[q + 8 * t];
			if(b_act)
				L[p.loff[k] + lane] = v;
		}
		__syncthreads(); // column j is complete (and visible to this workgroup) before the next one starts
	}
}

// bottom stage: whole elimination subtrees, one wave each (the launch that touches most of Lambda and L)
__global__ void __launch_bounds__(64)
factor_subtree_kernel(TDevPlan p, const double *__restrict__ A, double *L, double *Linv,
	int task_begin, int *p_flag)
{
	factor_task<1>(p, A, L, Linv, task_begin + blockIdx.x, p_flag);
}

// upper stages: separator columns / chains, W waves per column
template <int W>
__global__ void __launch_bounds__(64 * W)
factor_stage_kernel(TDevPlan p, const double *__restrict__ A, double *L, double *Linv,
	int task_begin, int *p_flag)
{
	factor_task<W>(p, A, L, Linv, task_begin + blockIdx.x, p_flag);
}

// forward substitution  y_j = inv(L_jj) (b_j - sum_c L(j,c) y_c), one wave per task, lanes = 8 entry
// groups x 8 rows.  Reads b at its original (unpermuted) position, writes y to the permuted workspace.
__global__ void __launch_bounds__(64)
forward_stage_kernel(TDevPlan p, const double *L, const double *Linv, const double *__restrict__ b,
	double *w, int task_begin)
{
	const int lane = threadIdx.x & 63, g = lane >> 3, r = lane & 7;
	const int task = task_begin + blockIdx.x;
	const int64_t c_end = p.task_ptr[task + 1];
	for(int64_t c = p.task_ptr[task]; c < c_end; ++ c) {
		const int j = p.task_cols[c];
		const int dj = p.dim[j];
		const int rr = (r < dj)? r : 0;
		double acc = 0;
		const int64_t e1 = p.rptr[j + 1];
		for(int64_t e = p.rptr[j] + g; e < e1; e += 8) {
			const int cc = p.rcol[e];
			const int dc = p.dim[cc];
			const double *Lb = L + p.roff[e] + rr;
			const double *y = w + p.cs_new[cc];
			for(int t = 0; t < dc; ++ t)
				acc += Lb[t * dj] * y[t];
		}
		acc += __shfl_xor(acc, 8);
		acc += __shfl_xor(acc, 16);
		acc += __shfl_xor(acc, 32);
		const double v = (r < dj)? b[p.cs_src[j] + r] - acc : 0;
		const double *Li = Linv + p.linv_off[j];
		double y = 0;
		for(int t = 0; t < dj; ++ t) {
			const double vt = __shfl(v, t);
			if(t <= rr)
				y += Li[rr + t * dj] * vt;
		}
		if(lane < dj)
			w[p.cs_new[j] + lane] = y;
		__syncthreads(); // single-wave workgroup: makes y_j visible to the following columns
	}
}

// backward substitution  x_j = inv(L_jj)^T (y_j - sum_i L(i,j)^T x_i); tasks and columns in reverse.
// Overwrites the workspace in place and scatters x to its original position.
__global__ void __launch_bounds__(64)
backward_stage_kernel(TDevPlan p, const double *L, const double *Linv, double *w,
	double *__restrict__ x_out, int task_begin)
{
	const int lane = threadIdx.x & 63, g = lane >> 3, q = lane & 7;
	const int task = task_begin + blockIdx.x;
	const int64_t c_begin = p.task_ptr[task];
	for(int64_t c = p.task_ptr[task + 1]; c > c_begin; -- c) {
		const int j = p.task_cols[c - 1];
		const int dj = p.dim[j];
		const int qq = (q < dj)? q : 0;
		const int64_t k0 = p.lptr[j];
		const int nb = int(p.lptr[j + 1] - k0);
		double acc = 0;
		for(int kb = 1 + g; kb < nb; kb += 8) {
			const int64_t k = k0 + kb;
			const int i = p.lrow[k];
			const int di = p.dim[i];
			const double *Lb = L + p.loff[k] + int64_t(qq) * di;
			const double *x = w + p.cs_new[i];
			for(int t = 0; t < di; ++ t)
				acc += Lb[t] * x[t];
		}
		acc += __shfl_xor(acc, 8);
		acc += __shfl_xor(acc, 16);
		acc += __shfl_xor(acc, 32);
		const double v = (q < dj)? w[p.cs_new[j] + q] - acc : 0;
		const double *Li = Linv + p.linv_off[j];
		double x = 0;
		for(int t = 0; t < dj; ++ t) {
			const double vt = __shfl(v, t);
			if(t >= qq)
				x += Li[t + qq * dj] * vt;
		}
		__syncthreads(); // every lane has read y_j before it is overwritten
		if(lane < dj) {
			w[p.cs_new[j] + lane] = x;
			x_out[p.cs_src[j] + lane] = x;
		}
		__syncthreads();
	}
}

void launch_factor_stage(const TDevPlan &p, const double *A, double *L, double *Linv,
	int task_begin, int n_tasks, int n_waves, int *p_flag, hipStream_t stream)
{
	if(n_tasks <= 0)
		return;
	if(n_waves <= 0)
		hipLaunchKernelGGL(factor_subtree_kernel, dim3(n_tasks), dim3(64), 0, stream, p, A, L, Linv, task_begin, p_flag);
	else if(n_waves <= 1)
		hipLaunchKernelGGL(factor_stage_kernel<1>, dim3(n_tasks), dim3(64), 0, stream, p, A, L, Linv, task_begin, p_flag);
	else if(n_waves <= 4)
		hipLaunchKernelGGL(factor_stage_kernel<4>, dim3(n_tasks), dim3(256), 0, stream, p, A, L, Linv, task_begin, p_flag);
	else
		hipLaunchKernelGGL(factor_stage_kernel<16>, dim3(n_tasks), dim3(1024), 0, stream, p, A, L, Linv, task_begin, p_flag);
}

void launch_forward_stage(const TDevPlan &p, const double *L, const double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, hipStream_t stream)
{
	if(n_tasks > 0)
		hipLaunchKernelGGL(forward_stage_kernel, dim3(n_tasks), dim3(64), 0, stream, p, L, Linv, b, w, task_begin);
}

void launch_backward_stage(const TDevPlan &p, const double *L, const double *Linv, double *w,
	double *x_out, int task_begin, int n_tasks, hipStream_t stream)
{
	if(n_tasks > 0)
		hipLaunchKernelGGL(backward_stage_kernel, dim3(n_tasks), dim3(64), 0, stream, p, L, Linv, w, x_out, task_begin);
}

} // namespace slampp
