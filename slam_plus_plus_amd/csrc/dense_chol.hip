// dense_chol.hip -- dense fp64 Cholesky + substitutions for the reduced camera system S (BA path).
//
// Stands where the reference calls Eigen::LLT on the densified Schur complement
// (/root/reference/src/slam/LinearSolver_Schur.cpp:2314-2331; 65-70 % of its Schur solve time) or
// CULA culaDevicePosv on its CUDA build (src/slam/LinearSolver_Schur_GPU.cpp:736-796).
//
// Right-looking blocked factorization with 64-wide panels, lower triangle, column-major:
//   potrf_diag : one workgroup factors the 64x64 diagonal tile in registers (4x4 per thread) and
//                inverts it (so that the panel solve becomes a GEMM)
//   trsm       : L21 = A21 inv(L11)^T, one workgroup per 64-row tile          (MFMA f64 16x16x4)
//   syrk       : A22 -= L21 L21^T, one workgroup per lower 64x64 tile        (MFMA f64 16x16x4),
//                two-level blocked: panel-local after every step, trailing matrix once per 256 columns
// The right-hand side rides along as the last row of the matrix, so the forward substitution
// costs nothing extra; the backward substitution is right-looking, one launch per panel.
// This is the MFMA-bound kernel of the path: n^3/3 flops (72 GFLOP at 1k cameras).
#include <hip/hip_runtime.h>
#include "dense_chol.h"

namespace slampp {

typedef double v4f64 __attribute__((ext_vector_type(4)));

enum { NB = dense_NB, LDS_LD = 80 }; // 80: k-groups of a fragment read land on disjoint LDS banks

__global__ void dense_pad_kernel(double *M, int ld, int n)
{
	const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
	if(i < ld)
		M[i + size_t(i) * ld] = 1.0;
}

void dense_prepare_padding(double *M, int n_pad, int n, hipStream_t stream)
{
	const int cnt = n_pad - n;
	hipLaunchKernelGGL(dense_pad_kernel, dim3((cnt + 63) / 64), dim3(64), 0, stream, M, n_pad, n);
}

// ---- 64 x 64 x 64 tile product on the matrix cores ----
// acc[c][reg] (+)= sum_k Q[i][k] P[j][k] with i = 16 wave + (lane >> 4) + 4 reg, j = 16 c + (lane & 15);
// both operands live in LDS as [k][row] with leading dimension LDS_LD.
__device__ __forceinline__ void tile_product(const double *Ps, const double *Qs, int wave, int lane, v4f64 acc[4])
{
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int ks = 0; ks < NB / 4; ++ ks) {
		const int k = ks * 4 + hi;
		const double a = Qs[k * LDS_LD + 16 * wave + lo];
		#pragma unroll
		for(int c = 0; c < 4; ++ c) {
			const double b = Ps[k * LDS_LD + 16 * c + lo];
			acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
		}
	}
}

// loads the 64 x 64 tile at (row0, col0) of the column-major matrix into LDS as [col][row]
__device__ __forceinline__ void load_tile(double *Ts, const double *M, int ld, int row0, int col0)
{
	// 16-byte loads: thread t moves rows 2 (t & 31), +1 of columns t >> 5, +8, ...; all loads are
	// issued before the first LDS store (row0, ld and LDS_LD are even, so everything is 16-B aligned)
	const int r = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
	double2 v[NB / 8];
	#pragma unroll
	for(int i = 0; i < NB / 8; ++ i)
		v[i] = *reinterpret_cast<const double2*>(M + size_t(row0 + r) + size_t(col0 + c0 + 8 * i) * ld);
	#pragma unroll
	for(int i = 0; i < NB / 8; ++ i)
		*reinterpret_cast<double2*>(Ts + (c0 + 8 * i) * LDS_LD + r) = v[i];
}

// ---- diagonal tile: Cholesky in registers + inverse ----
// Cholesky: one wave, thread r owns row r of the 64 x 64 tile in 64 registers; the 64 elimination
// steps are fully unrolled, pivots and the scaled pivot column travel by v_readlane with constant
// lane numbers -- no LDS, no barrier on the critical path (a 4-wave version with one barrier per
// step took 19 us, this one takes about half).  Inverse: recursive on 16 / 32 / 64 blocks with all
// 256 threads through LDS,  inv([A 0; B C]) = [inv(A) 0; -inv(C) B inv(A), inv(C)].
__device__ __forceinline__ double dense_read_lane(double v, int n_lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), n_lane);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), n_lane);
	return __hiloint2double(hi, lo);
}

template <bool b_chol, bool b_inverse>
__device__ __forceinline__ void potrf_diag_body(double *M, int ld, int kb, int n, double *invL, int *p_flag)
{
	__shared__ double s_L[NB][NB + 1]; // [row][col]
	__shared__ double s_X[NB][NB + 1]; // inverse, [row][col]
	__shared__ double s_T[NB][NB + 1];

	const int t = threadIdx.x;
	const int o = kb * NB;
	for(int e = t; e < NB * NB; e += 256) { // coalesced along rows of the column-major tile
		const int r = e & 63, c = e >> 6;
		s_L[r][c] = (c <= r)? M[size_t(o + r) + size_t(o + c) * ld] : 0.0;
	}
	__syncthreads();
	bool b_bad = false;
	if(b_chol) {
		// blocked by 16 columns: wave 0 factors the 64 x 16 panel in registers (thread r = row r,
		// pivots and the scaled pivot column by v_readlane), then all four waves apply the rank-16
		// update to the rest of the tile through LDS
		for(int J = 0; J < NB / 16; ++ J) {
			const int c0 = 16 * J;
			if(t < NB) {
				const int r = t;
				double a[16];
				#pragma unroll
				for(int c = 0; c < 16; ++ c)
					a[c] = s_L[r][c0 + c];
				#pragma unroll
				for(int k = 0; k < 16; ++ k) {
					double piv = dense_read_lane(a[k], c0 + k);
					const bool b_neg = !(piv > 0);
					b_bad = b_bad || (b_neg && o + c0 + k < n);
					piv = b_neg? 1.0 : piv;
					double rs = __builtin_amdgcn_rsq(piv);
					const double h = 0.5 * piv;
					rs = rs * (1.5 - h * rs * rs);
					rs = rs * (1.5 - h * rs * rs);
					const double lk = a[k] * rs; // L(r, c0 + k), meaningful for r >= c0 + k
					a[k] = lk;
					#pragma unroll
					for(int c = k + 1; c < 16; ++ c)
						a[c] -= lk * dense_read_lane(lk, c0 + c);
				}
				#pragma unroll
				for(int c = 0; c < 16; ++ c)
					s_L[r][c0 + c] = (r >= c0 + c)? a[c] : 0.0;
			}
			__syncthreads();
			const int m = NB - c0 - 16; // trailing size
			for(int e = t; e < m * m; e += 256) {
				const int i = e / m, j = e % m;
				if(j > i)
					continue;
				const int r = c0 + 16 + i, c = c0 + 16 + j;
				double sum = 0;
				#pragma unroll
				for(int u = 0; u < 16; ++ u)
					sum += s_L[r][c0 + u] * s_L[c][c0 + u];
				s_L[r][c] -= sum;
			}
			__syncthreads();
		}
	}
	if(b_bad && (t & 63) == 0)
		atomicOr(p_flag, 1);
	for(int e = t; e < NB * NB; e += 256) {
		const int r = e & 63, c = e >> 6;
		if(c <= r)
			M[size_t(o + r) + size_t(o + c) * ld] = s_L[r][c];
	}
	if(!b_inverse)
		return;
	// level 0: the four 16 x 16 diagonal blocks, one thread per column
	if(t < NB) {
		const int b0 = (t >> 4) * 16, c = t & 15;
		double x[16];
		#pragma unroll
		for(int r = 0; r < 16; ++ r) {
			double sum = 0;
			#pragma unroll
			for(int u = 0; u < r; ++ u)
				sum += s_L[b0 + r][b0 + u] * x[u];
			x[r] = (((r == c)? 1.0 : 0.0) - sum) / s_L[b0 + r][b0 + r];
		}
		#pragma unroll
		for(int r = 0; r < 16; ++ r)
			s_X[b0 + r][b0 + c] = x[r];
	}
	// zero the strictly upper block part once (the tile is consumed as a full 64 x 64 operand)
	for(int e = t; e < NB * NB; e += 256) {
		const int r = e >> 6, c = e & 63;
		if((c >> 4) > (r >> 4))
			s_X[r][c] = 0.0;
	}
	__syncthreads();
	// levels 1 and 2: X21 = -X22 (L21 X11) for block size h = 16 (two 32 x 32 blocks), then h = 32
	#pragma unroll
	for(int h = 16; h <= 32; h *= 2) {
		const int n_groups = NB / (2 * h);        // independent 2h x 2h diagonal blocks
		const int per_group = h * h;              // elements of one off-diagonal block
		for(int e = t; e < n_groups * per_group; e += 256) {
			const int g = e / per_group, i = (e % per_group) / h, j = e % h;
			const int b0 = g * 2 * h;
			double sum = 0;
			for(int u = 0; u < h; ++ u)
				sum += s_L[b0 + h + i][b0 + u] * s_X[b0 + u][b0 + j];
			s_T[b0 + h + i][b0 + j] = sum;
		}
		__syncthreads();
		for(int e = t; e < n_groups * per_group; e += 256) {
			const int g = e / per_group, i = (e % per_group) / h, j = e % h;
			const int b0 = g * 2 * h;
			double sum = 0;
			for(int u = 0; u < h; ++ u)
				sum += s_X[b0 + h + i][b0 + h + u] * s_T[b0 + h + u][b0 + j];
			s_X[b0 + h + i][b0 + j] = -sum;
		}
		__syncthreads();
	}
	for(int e = t; e < NB * NB; e += 256) {
		const int r = e & 63, c = e >> 6;
		invL[r + c * NB] = s_X[r][c]; // column-major inverse
	}
}

__global__ void __launch_bounds__(256)
potrf_diag_kernel(double *M, int ld, int kb, int n, double *invL, int *p_flag)
{
	potrf_diag_body<true, true>(M, ld, kb, n, invL, p_flag);
}

#ifdef POTRF_VARIANTS // tools/bench_potrf.hip: timing of the two halves
template <bool b_chol, bool b_inverse>
__global__ void __launch_bounds__(256)
potrf_diag_variant(double *M, int ld, int kb, int n, double *invL, int *p_flag)
{
	potrf_diag_body<b_chol, b_inverse>(M, ld, kb, n, invL, p_flag);
}
#endif

// ---- panel solve: L21 = A21 inv(L11)^T ----
__global__ void __launch_bounds__(256)
trsm_kernel(double *M, int ld, int kb, const double *invL)
{
	__shared__ double Ps[NB * LDS_LD];
	__shared__ double Qs[NB * LDS_LD];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int row0 = (kb + 1 + blockIdx.x) * NB, col0 = kb * NB;
	load_tile(Ps, M, ld, row0, col0);
	load_tile(Qs, invL, NB, 0, 0);
	__syncthreads();
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	tile_product(Ps, Qs, wave, lane, acc);
	// every thread has read its operands out of LDS; the tile in global memory can be overwritten
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 16 * c + lo) + size_t(col0 + 16 * wave + hi + 4 * reg) * ld] = acc[c][reg];
}

// ---- symmetric update: A(ti,tj) -= sum over K tiles kt in [k0, k1) of L(ti,kt) L(tj,kt)^T ----
// for target column tiles tj in [c0, c1) and row tiles ti in [tj, n_blocks).  One workgroup per
// 64 x 64 target tile; the read-modify-write of the target happens once, after the whole K range.
// Two-level blocking: inside an outer panel (4 tiles = 256 columns) only the panel's own columns are
// updated after every 64-wide step (k1 - k0 = 1); the big trailing matrix is touched once per outer
// panel with K = 256, which cuts its HBM traffic fourfold compared with 64-wide right-looking.
__global__ void __launch_bounds__(256)
syrk_kernel(double *M, int ld, int n_blocks, int k0, int k1, int c0, int c1)
{
	__shared__ double Ps[NB * LDS_LD];
	__shared__ double Qs[NB * LDS_LD];
	// linear index -> (tj, ti): column tile by column tile, rows tj .. n_blocks-1
	int tj = c0, idx = int(blockIdx.x);
	if(c1 == n_blocks) { // full lower triangle of the trailing matrix: closed form on the reversed index
		const int T = n_blocks - c0;
		const int total = T * (T + 1) / 2, rev = total - 1 - idx; // rev counts from the last (smallest) column
		int m = int((sqrt(8.0 * double(rev) + 1.0) - 1.0) * 0.5);
		while((m + 1) * (m + 2) / 2 <= rev) ++ m;
		while(m * (m + 1) / 2 > rev) -- m;
		// column with m+1 rows is tj = n_blocks - 1 - m; position inside it, counted from the end
		tj = n_blocks - 1 - m;
		idx = m - (rev - m * (m + 1) / 2);
	} else {
		while(idx >= n_blocks - tj) {
			idx -= n_blocks - tj;
			++ tj;
		}
	}
	const int ti = tj + idx;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int row0 = ti * NB, colq = tj * NB;
	const int lo = lane & 15, hi = lane >> 4;
	// the target tile is requested first: its HBM latency hides behind the K loop
	double cv[4][4];
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			cv[c][reg] = M[size_t(row0 + 16 * c + lo) + size_t(colq + 16 * wave + hi + 4 * reg) * ld];
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	for(int kt = k0; kt < k1; ++ kt) {
		if(kt > k0)
			__syncthreads(); // the previous K tile has been consumed
		load_tile(Ps, M, ld, row0, kt * NB);
		load_tile(Qs, M, ld, colq, kt * NB);
		__syncthreads();
		tile_product(Ps, Qs, wave, lane, acc);
	}
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 16 * c + lo) + size_t(colq + 16 * wave + hi + 4 * reg) * ld] = cv[c][reg] - acc[c][reg];
}

enum { OUTER_TILES = 4 }; // outer panel = 4 x 64 columns

void dense_cholesky(double *M, int n_pad, int n, double *p_invdiag, int *p_flag, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	for(int ob = 0; ob < n_blocks; ob += OUTER_TILES) {
		const int oe = (ob + OUTER_TILES < n_blocks)? ob + OUTER_TILES : n_blocks;
		for(int kb = ob; kb < oe; ++ kb) {
			double *invL = p_invdiag + size_t(kb) * NB * NB;
			hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, stream, M, n_pad, kb, n, invL, p_flag);
			const int n_below = n_blocks - kb - 1;
			if(n_below > 0)
				hipLaunchKernelGGL(trsm_kernel, dim3(n_below), dim3(256), 0, stream, M, n_pad, kb, invL);
			if(kb + 1 < oe) { // update the rest of the outer panel with this 64-wide step
				int n_tiles = 0;
				for(int tj = kb + 1; tj < oe; ++ tj)
					n_tiles += n_blocks - tj;
				hipLaunchKernelGGL(syrk_kernel, dim3(n_tiles), dim3(256), 0, stream, M, n_pad, n_blocks, kb, kb + 1, kb + 1, oe);
			}
		}
		if(oe < n_blocks) { // trailing matrix, once per outer panel, K = the whole panel
			const int T = n_blocks - oe;
			hipLaunchKernelGGL(syrk_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, stream, M, n_pad, n_blocks, ob, oe, oe, n_blocks);
		}
	}
}

// ---- backward substitution x = L^-T y, right-looking ----
__global__ void dense_backsolve_init_kernel(const double *M, int ld, int n, double *z)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i < ld)
		z[i] = (i < n)? M[size_t(ld - 1) + size_t(i) * ld] : 0.0;
}

// ---- stand-alone forward substitution y = L^-1 r with a kept factor (another right-hand side) ----
// r is taken from row ld-1 of M (where dense_assemble leaves it), y is written back there, so that
// dense_backsolve finds it where the fused factorization would have put it
__global__ void __launch_bounds__(256)
dense_forward_step_kernel(double *M, int ld, int kb, const double *invL)
{
	__shared__ double s_y[NB];
	__shared__ double s_red[4][NB];
	const int t = threadIdx.x, c = t >> 2, part = t & 3;
	const int jb = kb + 1 + blockIdx.x; // row tile to update; the extra last workgroup only publishes y_kb
	const size_t last = size_t(ld - 1);
	{
		double sum = 0; // y_kb = inv(L_kk) r_kb: row c of the inverse
		for(int u = part; u <= c; u += 4)
			sum += invL[c + u * NB] * M[last + size_t(kb * NB + u) * ld];
		sum += __shfl_xor(sum, 1);
		sum += __shfl_xor(sum, 2);
		if(part == 0)
			s_y[c] = sum;
	}
	__syncthreads();
	const int n_blocks = ld / NB;
	if(jb >= n_blocks) { // the publishing workgroup runs after no one needs r_kb any more? no: it must not
		return;
	}
	// r_jb -= L(jb, kb) y_kb; thread (r, part): rows contiguous, 4 partial sums over the 64 columns
	const int r = t & 63, pc = t >> 6;
	double sum = 0;
	for(int u = pc * 16; u < pc * 16 + 16; ++ u)
		sum += M[size_t(jb * NB + r) + size_t(kb * NB + u) * ld] * s_y[u];
	s_red[pc][r] = sum;
	__syncthreads();
	if(t < NB) {
		const double tot = (s_red[0][t] + s_red[1][t]) + (s_red[2][t] + s_red[3][t]);
		const int row = jb * NB + t;
		if(row != ld - 1) // the last row of the matrix is the right-hand side itself, not an equation
			M[last + size_t(row) * ld] -= tot;
	}
}

__global__ void __launch_bounds__(256)
dense_forward_finish_kernel(double *M, int ld, int kb, const double *invL)
{
	// r_kb <- y_kb in place, after every reader of r_kb (own launch)
	__shared__ double s_r[NB];
	const int t = threadIdx.x, c = t >> 2, part = t & 3;
	const size_t last = size_t(ld - 1);
	if(t < NB)
		s_r[t] = M[last + size_t(kb * NB + t) * ld];
	__syncthreads();
	double sum = 0;
	for(int u = part; u <= c; u += 4)
		sum += invL[c + u * NB] * s_r[u];
	sum += __shfl_xor(sum, 1);
	sum += __shfl_xor(sum, 2);
	if(part == 0 && kb * NB + c != ld - 1)
		M[last + size_t(kb * NB + c) * ld] = sum;
}

void dense_forwardsolve(double *M, int n_pad, const double *p_invdiag, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	for(int kb = 0; kb < n_blocks; ++ kb) {
		const double *invL = p_invdiag + size_t(kb) * NB * NB;
		if(kb + 1 < n_blocks)
			hipLaunchKernelGGL(dense_forward_step_kernel, dim3(n_blocks - kb - 1), dim3(256), 0, stream, M, n_pad, kb, invL);
		hipLaunchKernelGGL(dense_forward_finish_kernel, dim3(1), dim3(256), 0, stream, M, n_pad, kb, invL);
	}
}

// one launch per diagonal tile kb (descending): every workgroup recomputes x_kb = inv(L_kk)^T z_kb
// (4 lanes per entry), workgroup 0 publishes it to x, workgroup jb < kb applies z_jb -= L(kb,jb)^T x_kb
__global__ void __launch_bounds__(256)
dense_backsolve_step_kernel(const double *M, int ld, int kb, const double *invL, double *z, double *x)
{
	__shared__ double s_x[NB];
	const int t = threadIdx.x, c = t >> 2, part = t & 3;
	const int jb = blockIdx.x;
	{
		double sum = 0;
		for(int r = c + part; r < NB; r += 4)
			sum += invL[r + c * NB] * z[kb * NB + r];
		sum += __shfl_xor(sum, 1);
		sum += __shfl_xor(sum, 2);
		if(part == 0) {
			s_x[c] = sum;
			if(jb == 0)
				x[kb * NB + c] = sum;
		}
	}
	__syncthreads();
	if(jb >= kb)
		return;
	double sum = 0;
	const double *col = M + size_t(kb * NB) + size_t(jb * NB + c) * ld;
	for(int r = part; r < NB; r += 4)
		sum += col[r] * s_x[r];
	sum += __shfl_xor(sum, 1);
	sum += __shfl_xor(sum, 2);
	if(part == 0)
		z[jb * NB + c] -= sum;
}

void dense_backsolve(const double *M, int n_pad, int n, const double *p_invdiag, double *p_z, double *p_x, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	hipLaunchKernelGGL(dense_backsolve_init_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, stream, M, n_pad, n, p_z);
	for(int kb = n_blocks - 1; kb >= 0; -- kb) {
		const double *invL = p_invdiag + size_t(kb) * NB * NB;
		hipLaunchKernelGGL(dense_backsolve_step_kernel, dim3(kb > 0? kb : 1), dim3(256), 0, stream, M, n_pad, kb, invL, p_z, p_x);
	}
}

} // namespace slampp
