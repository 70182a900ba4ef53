// dense_chol.hip -- dense fp64 Cholesky + substitutions for the reduced camera system S (BA path).
//
// Stands where the reference calls Eigen::LLT on the densified Schur complement
// (/root/reference/src/slam/LinearSolver_Schur.cpp:2314-2331; 65-70 % of its Schur solve time) or
// CULA culaDevicePosv on its CUDA build (src/slam/LinearSolver_Schur_GPU.cpp:736-796).
//
// Right-looking blocked factorization with 64-wide panels, lower triangle, column-major:
//   potrf_diag : one workgroup factors the 64x64 diagonal tile (16-column register panels, MFMA updates)
//                and inverts it (so that the panel solve becomes a GEMM)
//   trsm       : L21 = A21 inv(L11)^T, one workgroup per 64-row tile          (MFMA f64 16x16x4)
//   syrk       : A22 -= L21 L21^T, one workgroup per lower 64x64 tile        (MFMA f64 16x16x4),
//                two-level blocked: panel-local after every step, trailing matrix once per 256 columns
// The right-hand side rides along as the last row of the matrix, so the forward substitution
// costs nothing extra; the backward substitution is right-looking, one launch per panel.
// This is the MFMA-bound kernel of the path: n^3/3 flops (72 GFLOP at 1k cameras).
#include <hip/hip_runtime.h>
#include "dense_chol.h"
#include "plan.h"
#include <algorithm>

namespace slampp {

typedef double v4f64 __attribute__((ext_vector_type(4)));

enum { NB = dense_NB };

// LDS operand tiles are stored [k][row] with leading dimension 64 and rows XOR-swizzled by 16 on odd k: the
// two k-groups a half-wave reads in one fragment load land on disjoint halves of the banks, and a 64 x 64
// operand pair takes 64 KB, so two workgroups (or one next to the diagonal-tile kernel) fit on a CU
__device__ __forceinline__ int lds_at(int k, int row) { return k * NB + (row ^ ((k & 1) << 4)); }

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

__global__ void dense_gap_kernel(double *M, int ld, const int32_t *__restrict__ p_positions, int n_positions)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n_positions)
		M[p_positions[i] + size_t(p_positions[i]) * ld] = 1.0;
}

void dense_prepare_gaps(double *M, int n_pad, const int32_t *p_positions_dev, int n_positions, hipStream_t stream)
{
	if(n_positions > 0)
		hipLaunchKernelGGL(dense_gap_kernel, dim3((n_positions + 63) / 64), dim3(64), 0, stream, M, n_pad, p_positions_dev, n_positions);
}

// ---- 64 x 64 x 64 tile product on the matrix cores ----
// acc[c][reg] (+)= sum_k Q[i][k] P[j][k] with i = 16 wave + (lane >> 4) + 4 reg, j = 16 c + (lane & 15);
// both operands live in LDS as [k][row], swizzled (lds_at).
__device__ __forceinline__ void tile_product(const double *Ps, const double *Qs, int wave, int lane, v4f64 acc[4])
{
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int ks = 0; ks < NB / 4; ++ ks) {
		const int k = ks * 4 + hi;
		const double a = Qs[lds_at(k, 16 * wave + lo)];
		#pragma unroll
		for(int c = 0; c < 4; ++ c) {
			const double b = Ps[lds_at(k, 16 * c + lo)];
			acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
		}
	}
}

// the 64 x 64 tile at (row0, col0) of the column-major matrix, on its way into LDS as [col][row]:
// 16-byte loads, thread t moves rows 2 (t & 31), +1 of columns t >> 5, +8, ...; all loads are issued before
// the first LDS store (row0 and ld are even and the swizzle keeps row pairs together, so everything is
// 16-B aligned).  Split in two so that the loads of the next K tile can fly during the current product.
typedef double v2f64 __attribute__((ext_vector_type(2)));

struct TTileRegs {
	v2f64 v[NB / 8];
};

__device__ __forceinline__ void fetch_tile(TTileRegs &t_regs, const double *M, int ld, int row0, int col0)
{
	const int r = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
	#pragma unroll
	for(int i = 0; i < NB / 8; ++ i)
		t_regs.v[i] = *reinterpret_cast<const v2f64*>(M + size_t(row0 + r) + size_t(col0 + c0 + 8 * i) * ld);
}

__device__ __forceinline__ void stage_tile(double *Ts, const TTileRegs &t_regs)
{
	const int r = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
	#pragma unroll
	for(int i = 0; i < NB / 8; ++ i)
		*reinterpret_cast<v2f64*>(Ts + lds_at(c0 + 8 * i, r)) = t_regs.v[i];
}

__device__ __forceinline__ void load_tile(double *Ts, const double *M, int ld, int row0, int col0)
{
	TTileRegs t_regs;
	fetch_tile(t_regs, M, ld, row0, col0);
	stage_tile(Ts, t_regs);
}

// ---- diagonal tile: Cholesky + inverse ----
// Cholesky, blocked by 16 columns: wave 0 factors a 64 x 16 panel in registers (thread r = row r; pivots and
// the scaled pivot column travel by v_readlane with constant lane numbers -- no LDS, no barrier inside the
// 16 steps), then the rank-16 update of the next panel's columns runs on the matrix cores (one 16 x 16 tile
// per wave); the updates of the columns further right and the inverse of the finished 16 x 16 diagonal block
// are done by waves 1-3 while wave 0 is already inside the next panel.
// Inverse (so that the panel solve becomes a GEMM): recursive on 16 / 32 / 64 blocks,
// inv([A 0; B C]) = [inv(A) 0; -inv(C) B inv(A), inv(C)], the products on the matrix cores.
// All LDS tiles use the odd leading dimension PL: row-wise and column-wise fragment reads both stay at
// most 2-way bank conflicted, so no transposed copies are needed.
__device__ __forceinline__ double dense_read_lane(double v, int n_lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), n_lane);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), n_lane);
	return __hiloint2double(hi, lo);
}

enum { PL = NB + 1, TL = NB / 2 + 1 };

// D[m][n] += sum_{k < K} A[m][k] B[k][n] for one 16 x 16 tile; A[m][k] = p_A[m * a_m + k * a_k], B[k][n] =
// p_B[k * b_k + n * b_n]; lane l holds D[(l >> 4) + 4 reg][l & 15] in acc[reg]
__device__ __forceinline__ v4f64 mfma_tile16(const double *p_A, int a_m, int a_k, const double *p_B, int b_k, int b_n,
	int K, int lane, v4f64 acc)
{
	const int lo = lane & 15, hi = lane >> 4;
	for(int ks = 0; ks < K; ks += 4) {
		const double a = p_A[lo * a_m + (ks + hi) * a_k];
		const double b = p_B[(ks + hi) * b_k + lo * b_n];
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
	}
	return acc;
}

// trailing update inside the diagonal tile: S(ti, tj) -= P(ti) P(tj)^T with the 16-column panel at column c0;
// computed transposed (m = column of S, n = row) so that the read-modify-write runs along rows
__device__ __forceinline__ void potrf_update_tile(double *s_L, int c0, int ti, int tj, int lane)
{
	const v4f64 zero = {0, 0, 0, 0};
	const v4f64 acc = mfma_tile16(s_L + c0 * PL + 16 * tj, 1, PL, s_L + c0 * PL + 16 * ti, PL, 1, 16, lane, zero);
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int reg = 0; reg < 4; ++ reg)
		s_L[(16 * tj + hi + 4 * reg) * PL + 16 * ti + lo] -= acc[reg];
}

// inverse of the 16 x 16 lower triangular diagonal block at b0: lane c < 16 of the calling wave solves column c
__device__ __forceinline__ void potrf_invert_block16(const double *s_L, const double *s_rd, double *s_X, int b0, int lane)
{
	if(lane >= 16)
		return;
	const int c = lane;
	double x[16];
	#pragma unroll
	for(int r = 0; r < 16; ++ r) {
		double sum = (r == c)? 1.0 : 0.0;
		#pragma unroll
		for(int u = 0; u < r; ++ u)
			sum -= s_L[(b0 + u) * PL + b0 + r] * x[u];
		x[r] = sum * s_rd[b0 + r];
	}
	#pragma unroll
	for(int r = 0; r < 16; ++ r)
		s_X[(b0 + r) * PL + b0 + c] = x[r];
}

enum { POTRF_LDS_DOUBLES = 2 * NB * PL + (NB / 2) * TL + NB };

template <bool b_chol, bool b_inverse>
__device__ __forceinline__ void potrf_diag_body(double *M, int ld, int kb, int n, double *invL, int *p_flag, double *s_buf)
{
	double *s_L = s_buf;                  // the tile, [col][row]
	double *s_X = s_L + NB * PL;          // its inverse, [row][col]
	double *s_T = s_X + NB * PL;          // products L21 X11, [row][col]
	double *s_rd = s_T + (NB / 2) * TL;   // reciprocals of the diagonal of L

	const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
	const int o = kb * NB;
	{
		double v[NB * NB / 256];
		#pragma unroll
		for(int i = 0; i < NB * NB / 256; ++ i) { // coalesced along rows of the column-major tile
			const int e = t + 256 * i, r = e & 63, c = e >> 6;
			v[i] = (c <= r)? M[size_t(o + r) + size_t(o + c) * ld] : 0.0;
		}
		#pragma unroll
		for(int i = 0; i < NB * NB / 256; ++ i) {
			const int e = t + 256 * i, r = e & 63, c = e >> 6;
			s_L[c * PL + r] = v[i];
			s_X[c * PL + r] = 0.0;
		}
	}
	if(!b_chol && t < NB)
		s_rd[t] = 1.0 / M[size_t(o + t) + size_t(o + t) * ld];
	__syncthreads();
	bool b_bad = false;
	if(b_chol) {
		for(int J = 0; J < NB / 16; ++ J) {
			const int c0 = 16 * J;
			if(wave == 0) {
				const int r = lane;
				double a[16];
				#pragma unroll
				for(int c = 0; c < 16; ++ c)
					a[c] = s_L[(c0 + c) * PL + r];
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
					if(r == c0 + k)
						s_rd[c0 + k] = rs;
					#pragma unroll
					for(int c = k + 1; c < 16; ++ c)
						a[c] -= lk * dense_read_lane(lk, c0 + c);
				}
				#pragma unroll
				for(int c = 0; c < 16; ++ c)
					s_L[(c0 + c) * PL + r] = (r >= c0 + c)? a[c] : 0.0;
			}
			__syncthreads();
			// the next panel's columns first: tiles (ti, J + 1), ti = J + 1 .. 3, one per wave
			if(J + 1 + wave < NB / 16)
				potrf_update_tile(s_L, c0, J + 1 + wave, J + 1, lane);
			__syncthreads();
			// columns further right and the inverse of this diagonal block: waves 1-3, next to wave 0's next panel
			if(wave > 0) {
				int n_idx = 0;
				for(int tj = J + 2; tj < NB / 16; ++ tj) {
					for(int ti = tj; ti < NB / 16; ++ ti, ++ n_idx) {
						if(n_idx % 3 == wave - 1)
							potrf_update_tile(s_L, c0, ti, tj, lane);
					}
				}
				if(b_inverse && wave == 1 + J % 3)
					potrf_invert_block16(s_L, s_rd, s_X, c0, lane);
			}
		}
		__syncthreads();
	} else if(b_inverse) {
		if(wave < NB / 16)
			potrf_invert_block16(s_L, s_rd, s_X, 16 * wave, lane);
		__syncthreads();
	}
	if(b_bad && lane == 0)
		atomicOr(p_flag, 1);
	#pragma unroll
	for(int i = 0; i < NB * NB / 256; ++ i) {
		const int e = t + 256 * i, r = e & 63, c = e >> 6;
		if(c <= r)
			M[size_t(o + r) + size_t(o + c) * ld] = s_L[c * PL + r];
	}
	if(!b_inverse)
		return;
	const int lo = lane & 15, hi = lane >> 4;
	const v4f64 zero = {0, 0, 0, 0};
	// level 1: the off-diagonal 16 x 16 block of the two 32 x 32 diagonal blocks, X21 = -X22 (L21 X11); waves 0 and 1
	if(wave < 2) {
		const int b0 = 32 * wave;
		const v4f64 acc = mfma_tile16(s_L + b0 * PL + b0 + 16, 1, PL, s_X + b0 * PL + b0, PL, 1, 16, lane, zero);
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			s_T[(16 * wave + hi + 4 * reg) * TL + lo] = acc[reg];
	}
	__syncthreads();
	if(wave < 2) {
		const int b0 = 32 * wave;
		const v4f64 acc = mfma_tile16(s_X + (b0 + 16) * PL + b0 + 16, PL, 1, s_T + 16 * wave * TL, TL, 1, 16, lane, zero);
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			s_X[(b0 + 16 + hi + 4 * reg) * PL + b0 + lo] = -acc[reg];
	}
	__syncthreads();
	// level 2: the 32 x 32 off-diagonal block, one 16 x 16 tile per wave
	{
		const int mt = wave >> 1, nt = wave & 1;
		v4f64 acc = mfma_tile16(s_L + 32 + 16 * mt, 1, PL, s_X + 16 * nt, PL, 1, 32, lane, zero);
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			s_T[(16 * mt + hi + 4 * reg) * TL + 16 * nt + lo] = acc[reg];
		__syncthreads();
		acc = mfma_tile16(s_X + (32 + 16 * mt) * PL + 32, PL, 1, s_T + 16 * nt, TL, 1, 32, lane, zero);
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			s_X[(32 + 16 * mt + hi + 4 * reg) * PL + 16 * nt + lo] = -acc[reg];
	}
	__syncthreads();
	#pragma unroll
	for(int i = 0; i < NB * NB / 256; ++ i) {
		const int e = t + 256 * i, r = e & 63, c = e >> 6;
		invL[r + c * NB] = s_X[r * PL + c]; // column-major inverse
	}
}

#ifdef POTRF_VARIANTS // tools/bench_potrf.hip: timing of the two halves
template <bool b_chol, bool b_inverse>
__global__ void __launch_bounds__(256)
potrf_diag_variant(double *M, int ld, int kb, int n, double *invL, int *p_flag)
{
	__shared__ double s_buf[POTRF_LDS_DOUBLES];
	potrf_diag_body<b_chol, b_inverse>(M, ld, kb, n, invL, p_flag, s_buf);
}
#endif

// ---- panel solve: L21 = A21 inv(L11)^T ----
// With b_update_next the workgroup of the first row tile also applies this step's update to the next diagonal
// tile, A(kb+1,kb+1) -= L(kb+1,kb) L(kb+1,kb)^T, so that the next potrf can follow this launch directly (the rest of
// the 64-wide update rides in that potrf's launch).
__global__ void __launch_bounds__(256)
trsm_kernel(double *M, int ld, int kb, const double *invL, int b_update_next)
{
	__shared__ double Ps[NB * NB];
	__shared__ double Qs[NB * NB];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int row0 = (kb + 1 + blockIdx.x) * NB, col0 = kb * NB;
	load_tile(Ps, M, ld, row0, col0);
	load_tile(Qs, invL, NB, 0, 0);
	const int lo = lane & 15, hi = lane >> 4;
	const bool b_diag = b_update_next && blockIdx.x == 0; // workgroup-uniform
	double cv[4][4];
	if(b_diag) { // the next diagonal tile is requested now, its latency hides behind the two products
		#pragma unroll
		for(int c = 0; c < 4; ++ c)
			#pragma unroll
			for(int reg = 0; reg < 4; ++ reg)
				cv[c][reg] = M[size_t(row0 + 16 * c + lo) + size_t(row0 + 16 * wave + hi + 4 * reg) * ld];
	}
	__syncthreads();
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	tile_product(Ps, Qs, wave, lane, acc);
	// every thread has read its operands out of LDS; the tile in global memory can be overwritten
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 16 * c + lo) + size_t(col0 + 16 * wave + hi + 4 * reg) * ld] = acc[c][reg];
	if(!b_diag)
		return;
	__syncthreads(); // every wave is done with Ps
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			Ps[lds_at(16 * wave + hi + 4 * reg, 16 * c + lo)] = acc[c][reg]; // L(kb+1,kb) as an operand: [k = column][row]
	__syncthreads();
	v4f64 upd[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	tile_product(Ps, Ps, wave, lane, upd);
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 16 * c + lo) + size_t(row0 + 16 * wave + hi + 4 * reg) * ld] = cv[c][reg] - upd[c][reg];
}

// ---- symmetric update: A(ti,tj) -= sum over K tiles kt in [k0, k1) of L(ti,kt) L(tj,kt)^T ----
// for target column tiles tj in [c0, c1) and row tiles ti in [tj, n_blocks).  One workgroup per
// 64 x 64 target tile; the read-modify-write of the target happens once, after the whole K range.
// Two-level blocking: inside an outer panel (4 tiles = 256 columns) only the panel's own columns are
// updated after every 64-wide step (k1 - k0 = 1); the big trailing matrix is touched once per outer
// panel with K = 256, which cuts its HBM traffic fourfold compared with 64-wide right-looking.
struct TSyrkJob { // tiles [tile0, tile0 + n_tiles) of the update (k0, k1, c0, c1)
	int k0, k1, c0, c1, tile0, n_tiles;
};

__device__ __forceinline__ void syrk_tile(double *M, int ld, int n_blocks, int k0, int k1, int c0, int c1, int tile,
	double *Ps, double *Qs)
{
	// linear index -> (tj, ti): column tile by column tile, rows tj .. n_blocks-1
	int tj = c0, idx = tile;
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
	TTileRegs t_p, t_q;
	fetch_tile(t_p, M, ld, row0, k0 * NB);
	fetch_tile(t_q, M, ld, colq, k0 * NB);
	for(int kt = k0; kt < k1; ++ kt) {
		if(kt > k0)
			__syncthreads(); // the previous K tile has been consumed
		stage_tile(Ps, t_p);
		stage_tile(Qs, t_q);
		__syncthreads();
		if(kt + 1 < k1) { // the next K tile travels while the matrix cores work on this one
			fetch_tile(t_p, M, ld, row0, (kt + 1) * NB);
			fetch_tile(t_q, M, ld, colq, (kt + 1) * NB);
		}
		tile_product(Ps, Qs, wave, lane, acc);
	}
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 16 * c + lo) + size_t(colq + 16 * wave + hi + 4 * reg) * ld] = cv[c][reg] - acc[c][reg];
}

__global__ void __launch_bounds__(256)
syrk_kernel(double *M, int ld, int n_blocks, int k0, int k1, int c0, int c1)
{
	__shared__ double s_buf[2 * NB * NB];
	syrk_tile(M, ld, n_blocks, k0, k1, c0, c1, int(blockIdx.x), s_buf, s_buf + NB * NB);
}

// workgroup 0 factors and inverts the diagonal tile kb; the others run tiles of up to three symmetric updates
// that do not depend on it (see the schedule in dense_cholesky): the single-workgroup step that every panel
// has to wait for gives the rest of the chip something to do
__global__ void __launch_bounds__(256)
potrf_diag_kernel(double *M, int ld, int kb, int n, double *invL, int *p_flag, int n_blocks, TSyrkJob t_job_a, TSyrkJob t_job_b,
	TSyrkJob t_job_c)
{
	__shared__ double s_buf[(int(POTRF_LDS_DOUBLES) > 2 * NB * NB)? int(POTRF_LDS_DOUBLES) : 2 * NB * NB];
	if(blockIdx.x == 0) {
		potrf_diag_body<true, true>(M, ld, kb, n, invL, p_flag, s_buf);
		return;
	}
	int idx = int(blockIdx.x) - 1;
	TSyrkJob t_job = t_job_a;
	if(idx >= t_job_a.n_tiles) {
		idx -= t_job_a.n_tiles;
		t_job = t_job_b;
		if(idx >= t_job_b.n_tiles) {
			idx -= t_job_b.n_tiles;
			t_job = t_job_c;
		}
	}
	syrk_tile(M, ld, n_blocks, t_job.k0, t_job.k1, t_job.c0, t_job.c1, t_job.tile0 + idx, s_buf, s_buf + NB * NB);
}

enum { OUTER_TILES = 4 }; // outer panel = 4 x 64 columns

static inline int n_syrk_tiles(int n_blocks, int c0, int c1)
{
	int n_tiles = 0;
	for(int tj = c0; tj < c1; ++ tj)
		n_tiles += n_blocks - tj;
	return n_tiles;
}

static inline void launch_syrk(double *M, int n_pad, int n_blocks, int k0, int k1, int c0, int c1, hipStream_t stream)
{
	const int n_tiles = n_syrk_tiles(n_blocks, c0, c1);
	if(n_tiles > 0)
		hipLaunchKernelGGL(syrk_kernel, dim3(n_tiles), dim3(256), 0, stream, M, n_pad, n_blocks, k0, k1, c0, c1);
}

// Schedule, on one stream.  The matrix is cut into outer panels of OUTER_TILES 64-wide tiles; panel b is
// factored by the chain potrf -> trsm per tile, the trsm also updating the next diagonal tile.  Every other
// update does not run after that chain but inside it, as extra workgroups of the potrf launches (which would
// otherwise keep one CU busy and 255 idle):
//   potrf of tile k >= 1 of panel b  carries  the rest of tile k - 1's 64-wide update of panel b (all but the
//                                             diagonal tile the trsm has done), and
//                                             the 64-wide update of panel b + 1 by tile k - 1,
//   potrf of tile 0 of panel b       carries  the K = 256 update of panel b + 1 by panel b - 1,
//   every potrf of panel b           carries  a slice of the K = 256 update of panels >= b + 2 by panel b - 1.
// Only the 64-wide update of panel b + 1 by the last tile of panel b separates two chains.  Launches on one
// stream serialize the writers of every target tile, and inside one launch no two workgroups share a target.
void dense_cholesky(double *M, int n_pad, int n, double *p_invdiag, int *p_flag, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	const int n_outer = (n_blocks + OUTER_TILES - 1) / OUTER_TILES;
	for(int b = 0; b < n_outer; ++ b) {
		const int t0 = b * OUTER_TILES, t1 = (t0 + OUTER_TILES < n_blocks)? t0 + OUTER_TILES : n_blocks;
		const int u0 = t1, u1 = (u0 + OUTER_TILES < n_blocks)? u0 + OUTER_TILES : n_blocks; // panel b + 1
		const int v0 = u1;                                                                   // panel b + 2 onwards
		const int p0 = t0 - OUTER_TILES, p1 = t0;                                            // panel b - 1
		const int n_next_tiles = n_syrk_tiles(n_blocks, u0, u1);
		const int n_far_tiles = (b > 0)? n_syrk_tiles(n_blocks, v0, n_blocks) : 0;
		for(int kb = t0; kb < t1; ++ kb) {
			const int k = kb - t0, m = t1 - t0;
			TSyrkJob t_inner = {0, 0, 0, 0, 0, 0}, t_near = {0, 0, 0, 0, 0, 0}, t_far = {0, 0, 0, 0, 0, 0};
			if(k > 0) {
				t_inner = TSyrkJob{kb - 1, kb, kb, t1, 1, n_syrk_tiles(n_blocks, kb, t1) - 1}; // tile 0 = (kb, kb): done by the trsm
				t_near = TSyrkJob{kb - 1, kb, u0, u1, 0, n_next_tiles};
			} else if(b > 0)
				t_near = TSyrkJob{p0, p1, u0, u1, 0, n_next_tiles};
			if(n_far_tiles > 0) {
				const int n_begin = int(int64_t(n_far_tiles) * k / m), n_end = int(int64_t(n_far_tiles) * (k + 1) / m);
				t_far = TSyrkJob{p0, p1, v0, n_blocks, n_begin, n_end - n_begin};
			}
			double *invL = p_invdiag + size_t(kb) * NB * NB;
			hipLaunchKernelGGL(potrf_diag_kernel, dim3(1 + t_inner.n_tiles + t_near.n_tiles + t_far.n_tiles), dim3(256), 0, stream,
				M, n_pad, kb, n, invL, p_flag, n_blocks, t_inner, t_near, t_far);
			const int n_below = n_blocks - kb - 1;
			if(n_below > 0)
				hipLaunchKernelGGL(trsm_kernel, dim3(n_below), dim3(256), 0, stream, M, n_pad, kb, invL, int(kb + 1 < t1));
		}
		launch_syrk(M, n_pad, n_blocks, t1 - 1, t1, u0, u1, stream); // the last tile's update of the next panel
	}
}

// ---- tile-sparse, level-scheduled variant (see dense_chol.h) ----
__global__ void __launch_bounds__(256)
tile_potrf_kernel(double *M, int ld, int n, double *p_invdiag, int *p_flag, const int *__restrict__ p_tiles)
{
	__shared__ double s_buf[POTRF_LDS_DOUBLES];
	const int kb = p_tiles[blockIdx.x];
	potrf_diag_body<true, true>(M, ld, kb, n, p_invdiag + size_t(kb) * NB * NB, p_flag, s_buf);
}

__global__ void __launch_bounds__(256)
tile_trsm_kernel(double *M, int ld, const double *p_invdiag, const int2 *__restrict__ p_pairs)
{
	__shared__ double Ps[NB * NB];
	__shared__ double Qs[NB * NB];
	const int2 t_pair = p_pairs[blockIdx.x];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int row0 = t_pair.x * NB, col0 = t_pair.y * NB;
	load_tile(Ps, M, ld, row0, col0);
	load_tile(Qs, p_invdiag + size_t(t_pair.y) * NB * NB, NB, 0, 0);
	__syncthreads();
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	tile_product(Ps, Qs, wave, lane, acc);
	const int lo = lane & 15, hi = lane >> 4;
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 16 * c + lo) + size_t(col0 + 16 * wave + hi + 4 * reg) * ld] = acc[c][reg];
}

// target tile (ti, tj) -= sum over its source tile columns kt of L(ti, kt) L(tj, kt)^T
__global__ void __launch_bounds__(256)
tile_update_kernel(double *M, int ld, const int4 *__restrict__ p_targets, const int *__restrict__ p_sources)
{
	__shared__ double Ps[NB * NB];
	__shared__ double Qs[NB * NB];
	const int4 t_tgt = p_targets[blockIdx.x];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int row0 = t_tgt.x * NB, colq = t_tgt.y * NB;
	const int lo = lane & 15, hi = lane >> 4;
	double cv[4][4];
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			cv[c][reg] = M[size_t(row0 + 16 * c + lo) + size_t(colq + 16 * wave + hi + 4 * reg) * ld];
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	TTileRegs t_p, t_q;
	int kt = p_sources[t_tgt.z];
	fetch_tile(t_p, M, ld, row0, kt * NB);
	fetch_tile(t_q, M, ld, colq, kt * NB);
	for(int e = t_tgt.z; e < t_tgt.w; ++ e) {
		if(e > t_tgt.z)
			__syncthreads(); // the previous source has been consumed
		stage_tile(Ps, t_p);
		stage_tile(Qs, t_q);
		__syncthreads();
		if(e + 1 < t_tgt.w) {
			kt = p_sources[e + 1];
			fetch_tile(t_p, M, ld, row0, kt * NB);
			fetch_tile(t_q, M, ld, colq, kt * NB);
		}
		tile_product(Ps, Qs, wave, lane, acc);
	}
	#pragma unroll
	for(int c = 0; c < 4; ++ c)
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 16 * c + lo) + size_t(colq + 16 * wave + hi + 4 * reg) * ld] = cv[c][reg] - acc[c][reg];
}

void CTileSchedule::Free()
{
	if(d_potrf) (void)hipFree(d_potrf);
	if(d_trsm) (void)hipFree(d_trsm);
	if(d_tgt) (void)hipFree(d_tgt);
	if(d_src) (void)hipFree(d_src);
	d_potrf = 0; d_trsm = 0; d_tgt = 0; d_src = 0;
	n_bytes = 0;
	n_levels = 0;
	n_tiles = 0;
}

bool CTileSchedule::Build(int n_tile_num, const std::vector<char> &r_nonzero, hipStream_t stream)
{
	Free();
	const int T = n_tile_num;
	if(T <= 0 || r_nonzero.size() != size_t(T) * T)
		return false;
	// structure of the tile factor: symbolic elimination at tile granularity (the block-level structure it comes
	// from is already closed; whole tiles are not, e.g. two blocks of different columns sharing a tile column)
	std::vector<char> nz(r_nonzero);
	for(int j = 0; j < T; ++ j) {
		nz[size_t(j) + size_t(j) * T] = 1;
		nz[size_t(T - 1) + size_t(j) * T] = 1; // the right-hand side rides in the last row
	}
	const std::vector<int> height = tile_symbolic(T, nz);
	const int n_max_height = *std::max_element(height.begin(), height.end());
	n_tiles = T;
	n_levels = n_max_height + 1;
	std::vector<int> potrf;
	std::vector<int2> trsm;
	std::vector<int4> tgt;
	std::vector<int> src;
	level_potrf_ptr.assign(1, 0);
	level_trsm_ptr.assign(1, 0);
	level_tgt_ptr.assign(1, 0);
	std::vector<int> tgt_of(size_t(T) * T, -1); // per level: index of the target record of a tile
	for(int l = 0; l < n_levels; ++ l) {
		const size_t n_tgt0 = tgt.size();
		std::vector<std::vector<int> > sources; // per target of this level
		for(int j = 0; j < T; ++ j) {
			if(height[j] != l)
				continue;
			potrf.push_back(j);
			for(int i = j + 1; i < T; ++ i) {
				if(nz[size_t(i) + size_t(j) * T])
					trsm.push_back(int2{i, j});
			}
			for(int i2 = j + 1; i2 < T; ++ i2) {
				if(!nz[size_t(i2) + size_t(j) * T])
					continue;
				for(int i1 = i2; i1 < T; ++ i1) {
					if(!nz[size_t(i1) + size_t(j) * T])
						continue;
					int &r_idx = tgt_of[size_t(i1) + size_t(i2) * T];
					if(r_idx < int(n_tgt0)) { // not seen in this level yet (stale indices of earlier levels are smaller)
						r_idx = int(n_tgt0 + sources.size());
						tgt.push_back(int4{i1, i2, 0, 0});
						sources.push_back(std::vector<int>());
					}
					sources[r_idx - n_tgt0].push_back(j);
				}
			}
		}
		for(size_t k = 0; k < sources.size(); ++ k) {
			tgt[n_tgt0 + k].z = int(src.size());
			src.insert(src.end(), sources[k].begin(), sources[k].end());
			tgt[n_tgt0 + k].w = int(src.size());
		}
		level_potrf_ptr.push_back(int(potrf.size()));
		level_trsm_ptr.push_back(int(trsm.size()));
		level_tgt_ptr.push_back(int(tgt.size()));
	}
	const size_t n_b0 = potrf.size() * sizeof(int), n_b1 = (trsm.size() + 1) * sizeof(int2),
		n_b2 = (tgt.size() + 1) * sizeof(int4), n_b3 = (src.size() + 1) * sizeof(int);
	if(hipMalloc((void**)&d_potrf, n_b0) != hipSuccess || hipMalloc((void**)&d_trsm, n_b1) != hipSuccess ||
	   hipMalloc((void**)&d_tgt, n_b2) != hipSuccess || hipMalloc((void**)&d_src, n_b3) != hipSuccess) {
		(void)hipGetLastError();
		Free();
		return false;
	}
	bool b_ok = hipMemcpyAsync(d_potrf, potrf.data(), n_b0, hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!trsm.empty())
		b_ok = b_ok && hipMemcpyAsync(d_trsm, trsm.data(), trsm.size() * sizeof(int2), hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!tgt.empty())
		b_ok = b_ok && hipMemcpyAsync(d_tgt, tgt.data(), tgt.size() * sizeof(int4), hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!src.empty())
		b_ok = b_ok && hipMemcpyAsync(d_src, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice, stream) == hipSuccess;
	b_ok = b_ok && hipStreamSynchronize(stream) == hipSuccess; // the host vectors live on this stack frame
	if(!b_ok) {
		(void)hipGetLastError();
		Free();
		return false;
	}
	n_bytes = n_b0 + n_b1 + n_b2 + n_b3;
	n_tiles = T;
	n_levels = n_max_height + 1;
	return true;
}

void tile_cholesky(const CTileSchedule &r_s, double *M, int n_pad, int n, double *p_invdiag, int *p_flag, hipStream_t stream)
{
	for(int l = 0; l < r_s.n_levels; ++ l) {
		const int p0 = r_s.level_potrf_ptr[l], p1 = r_s.level_potrf_ptr[l + 1];
		const int t0 = r_s.level_trsm_ptr[l], t1 = r_s.level_trsm_ptr[l + 1];
		const int g0 = r_s.level_tgt_ptr[l], g1 = r_s.level_tgt_ptr[l + 1];
		if(p1 > p0)
			hipLaunchKernelGGL(tile_potrf_kernel, dim3(p1 - p0), dim3(256), 0, stream, M, n_pad, n, p_invdiag, p_flag, r_s.d_potrf + p0);
		if(t1 > t0)
			hipLaunchKernelGGL(tile_trsm_kernel, dim3(t1 - t0), dim3(256), 0, stream, M, n_pad, p_invdiag, r_s.d_trsm + t0);
		if(g1 > g0)
			hipLaunchKernelGGL(tile_update_kernel, dim3(g1 - g0), dim3(256), 0, stream, M, n_pad, r_s.d_tgt + g0, r_s.d_src);
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

// one launch per outer panel [t0, t1) of diagonal tiles (descending): every workgroup solves the small
// triangular system of the panel redundantly -- x_kb = inv(L_kk)^T z_kb, z_j -= L(kb,j)^T x_kb for the tiles j
// of the panel left of kb -- workgroup 0 publishes x, and workgroup jb < t0 then applies
// z_jb -= sum over the panel's tiles kb of L(kb,jb)^T x_kb.  Everything a workgroup will need (up to 14 tiles,
// 8 values of each per thread) is requested before the first dependent step, so the chain of small products
// runs out of registers: one memory latency per launch, a quarter of the launches of a tile-by-tile substitution.
__global__ void __launch_bounds__(512)
dense_backsolve_panel_kernel(const double *M, int ld, int t0, int t1, const double *p_invdiag, double *z, double *x)
{
	enum { PARTS = 8, PER = NB / PARTS, N_PAIRS = OUTER_TILES * (OUTER_TILES - 1) / 2 };
	__shared__ double s_z[OUTER_TILES * NB]; // z of the panel, overwritten by x tile by tile
	const int t = threadIdx.x, c = t / PARTS, part = t % PARTS;
	const int jb = blockIdx.x;
	const int m = t1 - t0;
	const bool b_strip = jb < t0;
	double vi[OUTER_TILES][PER], vl[N_PAIRS][PER], vs[OUTER_TILES][PER];
	#pragma unroll
	for(int a = 0; a < OUTER_TILES; ++ a) {
		if(a < m) {
			const double *invL = p_invdiag + size_t(t0 + a) * NB * NB + c * NB + part;
			#pragma unroll
			for(int i = 0; i < PER; ++ i)
				vi[a][i] = invL[PARTS * i];
			#pragma unroll
			for(int b = 0; b < a; ++ b) {
				const double *col = M + size_t((t0 + a) * NB + part) + size_t((t0 + b) * NB + c) * ld;
				#pragma unroll
				for(int i = 0; i < PER; ++ i)
					vl[a * (a - 1) / 2 + b][i] = col[PARTS * i];
			}
			if(b_strip) {
				const double *col = M + size_t((t0 + a) * NB + part) + size_t(jb * NB + c) * ld;
				#pragma unroll
				for(int i = 0; i < PER; ++ i)
					vs[a][i] = col[PARTS * i];
			}
		}
	}
	for(int i = t; i < m * NB; i += 512)
		s_z[i] = z[t0 * NB + i];
	__syncthreads();
	#pragma unroll
	for(int a = OUTER_TILES - 1; a >= 0; -- a) {
		if(a < m) { // workgroup-uniform
			double *zk = s_z + a * NB;
			double sum = 0; // x_kb[c] = sum_r inv(L_kk)[r][c] z_kb[r]; PARTS lanes per entry
			#pragma unroll
			for(int i = 0; i < PER; ++ i)
				sum += vi[a][i] * zk[part + PARTS * i];
			sum += __shfl_xor(sum, 1);
			sum += __shfl_xor(sum, 2);
			sum += __shfl_xor(sum, 4);
			__syncthreads(); // everyone has read z_kb
			if(part == 0)
				zk[c] = sum;
			__syncthreads();
			#pragma unroll
			for(int b = 0; b < a; ++ b) { // z_j -= L(kb, j)^T x_kb inside the panel
				double upd = 0;
				#pragma unroll
				for(int i = 0; i < PER; ++ i)
					upd += vl[a * (a - 1) / 2 + b][i] * zk[part + PARTS * i];
				upd += __shfl_xor(upd, 1);
				upd += __shfl_xor(upd, 2);
				upd += __shfl_xor(upd, 4);
				if(part == 0)
					s_z[b * NB + c] -= upd; // thread-private entry; the barrier of the next tile orders it
			}
			__syncthreads();
		}
	}
	if(jb == 0) {
		for(int i = t; i < m * NB; i += 512)
			x[t0 * NB + i] = s_z[i];
	}
	if(!b_strip)
		return; // the only workgroup of the first panel just publishes
	double sum = 0;
	#pragma unroll
	for(int a = 0; a < OUTER_TILES; ++ a) {
		if(a < m) {
			#pragma unroll
			for(int i = 0; i < PER; ++ i)
				sum += vs[a][i] * s_z[a * NB + part + PARTS * i];
		}
	}
	sum += __shfl_xor(sum, 1);
	sum += __shfl_xor(sum, 2);
	sum += __shfl_xor(sum, 4);
	if(part == 0)
		z[jb * NB + c] -= sum;
}

void dense_backsolve(const double *M, int n_pad, int n, const double *p_invdiag, double *p_z, double *p_x, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	hipLaunchKernelGGL(dense_backsolve_init_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, stream, M, n_pad, n, p_z);
	for(int t1 = n_blocks; t1 > 0; t1 -= OUTER_TILES) {
		const int t0 = (t1 > OUTER_TILES)? t1 - OUTER_TILES : 0;
		hipLaunchKernelGGL(dense_backsolve_panel_kernel, dim3(t0 > 0? t0 : 1), dim3(512), 0, stream, M, n_pad, t0, t1, p_invdiag,
			p_z, p_x);
	}
}

} // namespace slampp
