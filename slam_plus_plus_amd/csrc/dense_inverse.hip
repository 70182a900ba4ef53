// dense_inverse.hip -- inverse of the dense reduced camera system from its Cholesky factor, for the marginal
// covariances of the BA path (the reference gets the blocks it needs from sparse triangular solves with unit
// bases and a recursive formula, /root/reference/include/slam/BAMarginals.h:579-806; a dense system on the
// matrix cores is inverted outright: 2 n^3 / 3 flops next to the n^3 / 3 of the factorization).
//
//   X = inv(L)      in place, by recursive halving on the 64 x 64 tiles (inverse_level_kernel)
//   Z = X^T X       one launch, one workgroup per lower tile: Z(i,j) = sum_{t >= i} X(t,i)^T X(t,j)
// Own translation unit (see dense_tiles.hip for why).
#include <hip/hip_runtime.h>
#include "dense_chol.h"

namespace slampp {

#include "dense_device.inl"

__device__ __forceinline__ void store_product(double *M, size_t ld, size_t row0, size_t col0, int wave, int lane,
	const v4f64 acc[4], double f_sign)
{
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[row0 + 16 * c + lo + (col0 + 16 * wave + hi + 4 * reg) * ld] = f_sign * acc[c][reg];
}

// operand tiles whose contraction index is the one they are contiguous in (the rows of a column-major tile) sit in
// LDS the way they sit in memory, [column][row] with leading dimension TLD = 68 (16-B aligned columns; a fragment
// read of 16 columns x 4 rows falls on 64 distinct 8-byte words, twice around the 32 x 8-byte banks: the minimum
// for a 512-byte read)
enum { TLD = NB + 4 };

__device__ __forceinline__ void stage_tile_t(double *Ts, const TTileRegs &t_regs)
{
	const int r = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
	#pragma unroll
	for(int i = 0; i < NB / 8; ++ i)
		*reinterpret_cast<v2f64*>(Ts + (c0 + 8 * i) * TLD + r) = t_regs.v[i];
}

// acc[c][reg] += sum_k P(j, k) Q(k, i), i = 16 wave + (lane >> 4) + 4 reg, j = 16 c + (lane & 15): a plain product
// of two column-major tiles, P staged by stage_tile ([k][row], swizzled), Q by stage_tile_t
__device__ __forceinline__ void tile_product_nn(const double *Ps, const double *Qs, int wave, int lane, v4f64 acc[4])
{
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int ks = 0; ks < NB / 4; ++ ks) {
		const int k = ks * 4 + hi;
		const double a = Qs[(16 * wave + lo) * TLD + k];
		#pragma unroll
		for(int c = 0; c < 4; ++ c) {
			const double b = Ps[lds_at(k, 16 * c + lo)];
			acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
		}
	}
}

// X(k,k) = inv(L_kk) as a full tile (zeros above the diagonal) in the place of L(k,k)
__global__ void __launch_bounds__(256)
inverse_diag_kernel(double *M, int ld, const double *__restrict__ p_invdiag)
{
	const int kb = blockIdx.x;
	const double *src = p_invdiag + size_t(kb) * NB * NB;
	for(int e = threadIdx.x; e < NB * NB; e += 256) {
		const int r = e & 63, c = e >> 6;
		M[size_t(kb * NB + r) + size_t(kb * NB + c) * ld] = (c <= r)? src[e] : 0.0;
	}
}

// One level of the recursion inv([L11 0; L21 L22]) = [X11 0; -X22 L21 X11, X22] on diagonal blocks of b tiles: every
// aligned pair of blocks (first tile k0 = 2 b q, second block from mid = k0 + b, clipped at the matrix) in the same launch,
// one workgroup per 64 x 64 tile (i, j) of the off-diagonal block, i in [mid, end), j in [k0, mid):
//   first pass   T(i,j) = sum_{t = j .. mid-1} L(i,t) X(t,j)      (X11 is lower triangular)       -> scratch
//   second pass  X(i,j) = -sum_{t = mid .. i} X(i,t) T(t,j)       (X22 is lower triangular)       -> M, over L21
// log2(n / 64) levels of two launches with (n / 128)^2 tiles each at the top instead of a chain of n / 64 panels.
template <bool b_second>
__global__ void __launch_bounds__(256)
inverse_level_kernel(double *M, int ld, double *T, int n_blocks, int b)
{
	__shared__ double s_buf[NB * NB + NB * TLD];
	double *Ps = s_buf, *Qs = s_buf + NB * NB;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int per = b * b, q = int(blockIdx.x) / per, e = int(blockIdx.x) % per;
	const int k0 = 2 * b * q, mid = k0 + b;
	// the tiles with the longest sums first: rows from the bottom (second pass), columns from the left (first pass)
	const int i = b_second? mid + b - 1 - e / b : mid + e % b;
	const int j = b_second? k0 + e % b : k0 + e / b;
	if(i >= n_blocks)
		return;
	const int t0 = b_second? mid : j, t1 = b_second? i + 1 : mid;
	const double *p_left = M, *p_right = b_second? T : M;
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	TTileRegs t_p, t_q;
	fetch_tile(t_p, p_left, ld, i * NB, t0 * NB);
	fetch_tile(t_q, p_right, ld, t0 * NB, j * NB);
	for(int t = t0; t < t1; ++ t) {
		if(t > t0)
			__syncthreads(); // the previous K tile has been consumed
		stage_tile(Ps, t_p);
		stage_tile_t(Qs, t_q);
		__syncthreads();
		if(t + 1 < t1) {
			fetch_tile(t_p, p_left, ld, i * NB, (t + 1) * NB);
			fetch_tile(t_q, p_right, ld, (t + 1) * NB, j * NB);
		}
		tile_product_nn(Ps, Qs, wave, lane, acc);
	}
	store_product(b_second? M : T, size_t(ld), size_t(i) * NB, size_t(j) * NB, wave, lane, acc, b_second? -1.0 : 1.0);
}

// ---- Z = X^T X: the contraction runs over the rows of both tiles, both operands are staged by stage_tile_t ----
// acc[c][reg] += sum_k Q(k, i) P(k, j), i = 16 wave + (lane >> 4) + 4 reg, j = 16 c + (lane & 15)
__device__ __forceinline__ void tile_product_t(const double *Ps, const double *Qs, int wave, int lane, v4f64 acc[4])
{
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int ks = 0; ks < NB / 4; ++ ks) {
		const int k = ks * 4 + hi;
		const double a = Qs[(16 * wave + lo) * TLD + k];
		#pragma unroll
		for(int c = 0; c < 4; ++ c) {
			const double b = Ps[(16 * c + lo) * TLD + k];
			acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
		}
	}
}

__global__ void __launch_bounds__(256)
inverse_lauum_kernel(const double *__restrict__ X, int ld, int n_blocks, double *Z)
{
	__shared__ double s_buf[2 * NB * TLD];
	double *Ps = s_buf, *Qs = s_buf + NB * TLD;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	// linear index -> (i, j), j <= i, row tile by row tile: the first rows have the longest sums
	const int idx = int(blockIdx.x);
	int i = int((sqrt(8.0 * double(idx) + 1.0) - 1.0) * 0.5);
	while((i + 1) * (i + 2) / 2 <= idx) ++ i;
	while(i * (i + 1) / 2 > idx) -- i;
	const int j = idx - i * (i + 1) / 2;
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	TTileRegs t_p, t_q;
	fetch_tile(t_p, X, ld, i * NB, i * NB);
	fetch_tile(t_q, X, ld, i * NB, j * NB);
	for(int t = i; t < n_blocks; ++ t) {
		if(t > i)
			__syncthreads();
		stage_tile_t(Ps, t_p);
		stage_tile_t(Qs, t_q);
		__syncthreads();
		if(t + 1 < n_blocks) {
			fetch_tile(t_p, X, ld, (t + 1) * NB, i * NB);
			fetch_tile(t_q, X, ld, (t + 1) * NB, j * NB);
		}
		tile_product_t(Ps, Qs, wave, lane, acc); // out(a, b) = sum_k X(t,i)(k, a) X(t,j)(k, b)
	}
	store_product(Z, size_t(ld), size_t(i) * NB, size_t(j) * NB, wave, lane, acc, 1.0);
}

void dense_inverse_from_factor(double *M, int n_pad, const double *p_invdiag, double *Z, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	hipLaunchKernelGGL(inverse_diag_kernel, dim3(n_blocks), dim3(256), 0, stream, M, n_pad, p_invdiag);
	for(int b = 1; b < n_blocks; b *= 2) {
		const int n_pairs = (n_blocks - b + 2 * b - 1) / (2 * b); // pairs whose second block is not empty
		const dim3 grid(unsigned(n_pairs) * b * b);
		hipLaunchKernelGGL(inverse_level_kernel<false>, grid, dim3(256), 0, stream, M, n_pad, Z, n_blocks, b); // Z is the scratch
		hipLaunchKernelGGL(inverse_level_kernel<true>, grid, dim3(256), 0, stream, M, n_pad, Z, n_blocks, b);
	}
	hipLaunchKernelGGL(inverse_lauum_kernel, dim3(n_blocks * (n_blocks + 1) / 2), dim3(256), 0, stream, M, n_pad, n_blocks, Z);
}

} // namespace slampp
