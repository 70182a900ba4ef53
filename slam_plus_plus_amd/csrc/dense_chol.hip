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
// Cholesky: thread (br, bc) owns a 4 x 4 register tile; step k broadcasts column k through a
// ping-pong LDS buffer (one barrier per step).  Inverse: thread c of wave 0 owns column c of
// inv(L) in 64 registers; the two loops are fully unrolled so that the column never leaves the
// register file, L(r,t) comes from LDS as a broadcast read.
__global__ void __launch_bounds__(256)
potrf_diag_kernel(double *M, int ld, int kb, int n, double *invL, int *p_flag)
{
	__shared__ double s_col[2][NB];    // column k of the trailing matrix at step k (unscaled), double-buffered
	__shared__ double s_piv[NB];
	__shared__ double s_L[NB][NB + 1];

	const int t = threadIdx.x;
	const int br = t >> 4, bc = t & 15; // 4 x 4 register tile at rows 4 br.., columns 4 bc..
	const int o = kb * NB;
	double a[4][4];
	#pragma unroll
	for(int i = 0; i < 4; ++ i)
		#pragma unroll
		for(int j = 0; j < 4; ++ j) {
			const int r = 4 * br + i, c = 4 * bc + j;
			a[i][j] = (r >= c)? M[size_t(o + r) + size_t(o + c) * ld] : 0.0;
		}
	bool b_bad = false;
	for(int k4 = 0; k4 < NB / 4; ++ k4) {
		#pragma unroll
		for(int kj = 0; kj < 4; ++ kj) {
			const int k = 4 * k4 + kj;
			double *col = s_col[k & 1];
			if(bc == k4) { // owners of column k publish it
				#pragma unroll
				for(int i = 0; i < 4; ++ i)
					col[4 * br + i] = a[i][kj];
			}
			__syncthreads();
			double piv = col[k];
			if(!(piv > 0)) {
				if(o + k < n)
					b_bad = true;
				piv = 1;
			}
			if(t == 0)
				s_piv[k] = piv;
			const double s2 = 1.0 / piv;
			if(br >= bc && 4 * bc + 3 > k) {
				double cr[4], cc[4];
				#pragma unroll
				for(int i = 0; i < 4; ++ i) {
					cr[i] = col[4 * br + i];
					cc[i] = col[4 * bc + i] * s2;
				}
				#pragma unroll
				for(int i = 0; i < 4; ++ i)
					#pragma unroll
					for(int j = 0; j < 4; ++ j) {
						const int r = 4 * br + i, c = 4 * bc + j;
						if(c > k && r >= c)
							a[i][j] -= cr[i] * cc[j];
					}
			}
		}
	}
	if(b_bad && t == 0)
		atomicOr(p_flag, 1);
	__syncthreads();
	// scale the columns: L(r,c) = a(r,c) / sqrt(piv_c), L(c,c) = sqrt(piv_c)
	#pragma unroll
	for(int j = 0; j < 4; ++ j) {
		const int c = 4 * bc + j;
		const double p = s_piv[c], rs = 1.0 / sqrt(p);
		#pragma unroll
		for(int i = 0; i < 4; ++ i) {
			const int r = 4 * br + i;
			double v = 0;
			if(r > c)
				v = a[i][j] * rs;
			else if(r == c)
				v = sqrt(p);
			s_L[r][c] = v;
			if(r >= c)
				M[size_t(o + r) + size_t(o + c) * ld] = v;
		}
	}
	__syncthreads();
	if(t < NB) {
		// column c = t of X = inv(L): x_r = ((r == c) - sum_{u < r} L(r,u) x_u) / L(r,r); x_u = 0 for u < c falls out
		const int c = t;
		double x[NB];
		#pragma unroll
		for(int r = 0; r < NB; ++ r) {
			double sum = 0;
			#pragma unroll
			for(int u = 0; u < r; ++ u)
				sum += s_L[r][u] * x[u];
			x[r] = (((r == c)? 1.0 : 0.0) - sum) / s_L[r][r];
		}
		#pragma unroll
		for(int r = 0; r < NB; ++ r)
			invL[r + c * NB] = x[r]; // column-major inverse
	}
}

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
