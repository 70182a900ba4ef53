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
#include "plan.h" // dev_knob

namespace slampp {
#include "dense_device.inl"

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

#ifdef POTRF_VARIANTS // tools/bench_potrf.hip: timing of the two halves
template <bool b_chol, bool b_inverse, bool b_pipeline = true>
__global__ void __launch_bounds__(256)
potrf_diag_variant(double *M, int ld, int kb, int n, double *invL, int *p_flag)
{
	__shared__ double s_buf[POTRF_LDS_DOUBLES];
	potrf_diag_body<b_chol, b_inverse, b_pipeline>(M, ld, kb, n, invL, p_flag, s_buf);
}
#endif

// ---- panel solve: L21 = A21 inv(L11)^T ----
// With b_update_next the workgroup of the first row tile also applies this step's update to the next diagonal
// tile, A(kb+1,kb+1) -= L(kb+1,kb) L(kb+1,kb)^T, so that the next potrf can follow this launch directly (the rest of
// the 64-wide update rides in that potrf's launch).
__global__ void __launch_bounds__(512)
trsm_kernel(double *M, int ld, int kb, const double *invL, int b_update_next)
{
	__shared__ double s_buf[2 * NB * NB];
	trsm_tile_body8(M, ld, (kb + 1 + int(blockIdx.x)) * NB, kb * NB, invL, b_update_next && blockIdx.x == 0, s_buf, s_buf + NB * NB);
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

// ---- the same update for a 128 x 128 target (2 x 2 tiles) per workgroup ----
// A 64 x 64 job fetches two operand tiles per tile product: 8 flop per byte, and the K = 256 update of the trailing matrix --
// the bulk of the flops -- ran at 30-34 TFLOP/s inside the schedule (tools/dense_launch_trace.sh).  Four target tiles share
// their operands: the same 64 KB of LDS now hold a 128-row slab of each operand panel for 32 columns of K, and a step of
// 32 columns is two tile products' worth of matrix-core work per 64 KB fetched instead of one.  Wave w owns the target's
// columns 32 w .. 32 w + 31, all 128 rows: sixteen 16 x 16 accumulators, which start as the target itself (the products
// are subtracted by negating one operand fragment) so that nothing but the store follows the K loop.
// The targets are the lower triangle of the 2-tile grid over the tile rows / columns [c0, c0 + 2 T2); on its diagonal the
// upper-right tile is above the matrix's diagonal and is not stored.
#ifndef SLAMPP_WIDE_K
#define SLAMPP_WIDE_K 32 // (tools/bench_syrk.hip: 16 measured 1-2 % slower)
#endif
enum { WIDE = 2 * NB, WIDE_K = SLAMPP_WIDE_K };

__device__ __forceinline__ int lds_at_wide(int k, int row) { return k * WIDE + (row ^ ((k & 1) << 4)); }

struct TWideRegs {
	v2f64 v[WIDE * WIDE_K / 2 / 256];
};

// rows row0 .. row0 + 127 of the columns col0 .. col0 + 31: thread t moves rows 2 (t & 63), +1 of columns t >> 6, + 4, ...
__device__ __forceinline__ void fetch_wide(TWideRegs &t_regs, const double *M, int ld, int row0, int col0)
{
	const int r = (threadIdx.x & 63) * 2, c0 = threadIdx.x >> 6;
	#pragma unroll
	for(int i = 0; i < WIDE * WIDE_K / 2 / 256; ++ i)
		t_regs.v[i] = *reinterpret_cast<const v2f64*>(M + size_t(row0 + r) + size_t(col0 + c0 + 4 * i) * ld);
}

__device__ __forceinline__ void stage_wide(double *Ts, const TWideRegs &t_regs)
{
	const int r = (threadIdx.x & 63) * 2, c0 = threadIdx.x >> 6;
	#pragma unroll
	for(int i = 0; i < WIDE * WIDE_K / 2 / 256; ++ i)
		*reinterpret_cast<v2f64*>(Ts + lds_at_wide(c0 + 4 * i, r)) = t_regs.v[i];
}

__device__ __forceinline__ void syrk_wide_tile(double *M, int ld, int k0, int k1, int c0, int T2, int tile, double *Ps, double *Qs)
{
	// linear index -> (J, I), column by column of the 2-tile grid, rows J .. T2-1 (closed form on the reversed index)
	const int total = T2 * (T2 + 1) / 2, rev = total - 1 - tile;
	int m = int((sqrt(8.0 * double(rev) + 1.0) - 1.0) * 0.5);
	while((m + 1) * (m + 2) / 2 <= rev) ++ m;
	while(m * (m + 1) / 2 > rev) -- m;
	const int J = T2 - 1 - m, I = J + (m - (rev - m * (m + 1) / 2));
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; // (scalar: the column offsets of the target stay out of the vector registers)
	const int lo = lane & 15, hi = lane >> 4;
	const int row0 = (c0 + 2 * I) * NB, colq = (c0 + 2 * J) * NB;
	const bool b_diag = I == J;              // (workgroup-uniform) both operands are the same rows: one slab
	const int c_first = (b_diag && wave >= 2)? 4 : 0; // (wave-uniform) the tile above the diagonal: computed along (whatever that part of the array holds; one job in T2 / 2), never stored
	TWideRegs t_p, t_q;
	fetch_wide(t_p, M, ld, row0, k0 * NB);
	if(!b_diag)
		fetch_wide(t_q, M, ld, colq, k0 * NB);
	v4f64 acc[2][8];
	size_t n_tgt = size_t(row0 + lo) + size_t(colq + 32 * wave + hi) * ld; // the lane's first target element
	#pragma unroll
	for(int m2 = 0; m2 < 2; ++ m2) {
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg) {
			const double *p_col = M + n_tgt + size_t(16 * m2 + 4 * reg) * ld;
			#pragma unroll
			for(int c = 0; c < 8; ++ c)
				acc[m2][c][reg] = p_col[16 * c];
		}
	}
	const double *Qr = b_diag? Ps : Qs;
	const int n_steps = (k1 - k0) * (NB / WIDE_K);
	for(int s = 0; s < n_steps; ++ s) {
		if(s > 0)
			__syncthreads(); // the previous slabs have been consumed
		stage_wide(Ps, t_p);
		if(!b_diag)
			stage_wide(Qs, t_q);
		__syncthreads();
		if(s + 1 < n_steps) { // the next slabs travel while the matrix cores work on these
			fetch_wide(t_p, M, ld, row0, k0 * NB + (s + 1) * WIDE_K);
			if(!b_diag)
				fetch_wide(t_q, M, ld, colq, k0 * NB + (s + 1) * WIDE_K);
		}
		#pragma unroll
		for(int ks = 0; ks < WIDE_K / 4; ++ ks) {
			const int k = ks * 4 + hi;
			const double a0 = -Qr[lds_at_wide(k, 32 * wave + lo)], a1 = -Qr[lds_at_wide(k, 32 * wave + 16 + lo)];
			#pragma unroll
			for(int c = 0; c < 8; ++ c) {
				const double b = Ps[lds_at_wide(k, 16 * c + lo)];
				acc[0][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][c], 0, 0, 0);
				acc[1][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][c], 0, 0, 0);
			}
		}
	}
	// (the store addresses are formed here, from values the compiler cannot trace back: kept from before the loop they
	// were 34 spilled registers)
	int n_ld = ld;
	asm volatile("" : "+v"(n_tgt), "+s"(n_ld));
	#pragma unroll
	for(int m2 = 0; m2 < 2; ++ m2) {
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg) {
			double *p_col = M + n_tgt + size_t(16 * m2 + 4 * reg) * n_ld;
			#pragma unroll
			for(int c = 0; c < 8; ++ c) {
				if(c >= c_first)
					p_col[16 * c] = acc[m2][c][reg];
			}
		}
	}
}

__global__ void __launch_bounds__(256, 2)
syrk_wide_kernel(double *M, int ld, int k0, int k1, int c0, int T2)
{
	__shared__ double s_buf[2 * WIDE * WIDE_K];
	syrk_wide_tile(M, ld, k0, k1, c0, T2, int(blockIdx.x), s_buf, s_buf + WIDE * WIDE_K);
}

// workgroup 0 factors and inverts the diagonal tile kb; the others run tiles of up to three symmetric updates
// that do not depend on it (see the schedule in dense_cholesky): the single-workgroup step that every panel
// has to wait for gives the rest of the chip something to do
__global__ void __launch_bounds__(256, 2)
potrf_diag_kernel(double *M, int ld, int kb, int n, double *invL, int *p_flag, int n_blocks, TSyrkJob t_job_w, TSyrkJob t_job_a,
	TSyrkJob t_job_b, TSyrkJob t_job_c)
{
	__shared__ double s_buf[(int(POTRF_LDS_DOUBLES) > 2 * NB * NB)? int(POTRF_LDS_DOUBLES) : 2 * NB * NB];
	if(blockIdx.x == 0) {
		potrf_diag_body<true, true>(M, ld, kb, n, invL, p_flag, s_buf);
		return;
	}
	int idx = int(blockIdx.x) - 1;
	if(idx < t_job_w.n_tiles) { // 128 x 128 targets: c0 = the first tile of the 2-tile grid, c1 = its size
		syrk_wide_tile(M, ld, t_job_w.k0, t_job_w.k1, t_job_w.c0, t_job_w.c1, t_job_w.tile0 + idx, s_buf, s_buf + WIDE * WIDE_K);
		return;
	}
	idx -= t_job_w.n_tiles;
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

// Schedule.  The matrix is cut into outer panels of OUTER_TILES 64-wide tiles; panel b is factored by the chain
// potrf -> trsm per tile, the trsm also updating the next diagonal tile.  Every other update does not run after that
// chain but beside it.  As extra workgroups of the potrf launches (which would otherwise keep one CU busy and 255 idle):
//   every potrf but the first        carries  the rest of the tile before's 64-wide update of panel b (all but the
//                                             diagonal tile that tile's trsm has done; for tile 0 of the panel the tile
//                                             before is the last one of panel b - 1, whose 64-wide update ends here),
//   potrf of tile k >= 1 of panel b  carries  the 64-wide update of panel b + 1 by tile k - 1,
//   potrf of tile 0 of panel b       carries  the K = 256 update of panel b + 1 by panel b - 1,
//   every potrf of panel b           carries  a quarter of the K = 256 update of the panels >= b + 2 by panel b - 1 (the
//                                             bulk of the flops; as 128 x 128 targets while at least 96 tiles trail).
// Nothing separates two chains: the potrf -> trsm -> potrf sequence runs through the panel boundaries.  Launches on one
// stream serialize the writers of every target tile, and inside one launch no two workgroups share a target.
// (A second, lowest-priority stream for the K = 256 updates, two events per panel, was measured again in round 3 with
// this schedule: 3.75 against 3.49 ms at n = 6 000 -- while that launch runs, the chain's own launches wait for
// workgroup slots behind it, 19 -> 75 us per potrf launch -- DESIGN.md section 4.2.)
void dense_cholesky(double *M, int n_pad, int n, double *p_invdiag, int *p_flag, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	const int n_outer = (n_blocks + OUTER_TILES - 1) / OUTER_TILES;
	const int n_wide_min_tiles = dev_knob("SLAMPP_HIP_DEV_WIDE_MIN_TILES", 96); // (development: 2 puts every far update on the 128 x 128 jobs)
	for(int b = 0; b < n_outer; ++ b) {
		const int t0 = b * OUTER_TILES, t1 = (t0 + OUTER_TILES < n_blocks)? t0 + OUTER_TILES : n_blocks;
		const int u0 = t1, u1 = (u0 + OUTER_TILES < n_blocks)? u0 + OUTER_TILES : n_blocks; // panel b + 1
		const int v0 = u1;                                                                   // panel b + 2 onwards
		const int p0 = t0 - OUTER_TILES, p1 = t0;                                            // panel b - 1
		const int n_next_tiles = n_syrk_tiles(n_blocks, u0, u1);
		// the K = 256 update of the panels >= b + 2: 128 x 128 targets where the trailing matrix is large enough to give every
		// CU several of them (tools/bench_syrk.hip: 50.8 against 44.0 TFLOP/s at 180 trailing tiles, 47.1 / 44.9 at 100, even
		// at 84, 30 / 35 at 38 -- a job is then a quarter of the launch's length); with an odd number of trailing tiles the
		// first tile column stays with the 64 x 64 jobs
		const int n_far_T = (b > 0 && v0 < n_blocks)? n_blocks - v0 : 0;
		const bool b_wide = n_far_T >= n_wide_min_tiles && n_far_T >= 2;
		const int n_wide_T2 = b_wide? n_far_T / 2 : 0, n_wide_c0 = n_blocks - 2 * n_wide_T2;
		const int n_wide_tiles = n_wide_T2 * (n_wide_T2 + 1) / 2;
		const int n_far_c1 = b_wide? n_wide_c0 : n_blocks; // (the 64 x 64 jobs' share: nothing, or the odd first tile column)
		const int n_far_tiles = (n_far_T > 0)? n_syrk_tiles(n_blocks, v0, n_far_c1) : 0;
		for(int kb = t0; kb < t1; ++ kb) {
			const int k = kb - t0, m = t1 - t0;
			TSyrkJob t_inner = {0, 0, 0, 0, 0, 0}, t_near = {0, 0, 0, 0, 0, 0}, t_far = {0, 0, 0, 0, 0, 0}, t_wide = {0, 0, 0, 0, 0, 0};
			if(kb > 0) // tile kb - 1's 64-wide update of this panel (tile 0 = (kb, kb): done by that tile's trsm); for k = 0 it comes from the panel before
				t_inner = TSyrkJob{kb - 1, kb, kb, t1, 1, n_syrk_tiles(n_blocks, kb, t1) - 1};
			if(k > 0)
				t_near = TSyrkJob{kb - 1, kb, u0, u1, 0, n_next_tiles};
			else if(b > 0)
				t_near = TSyrkJob{p0, p1, u0, u1, 0, n_next_tiles};
			if(n_far_tiles > 0) {
				const int n_begin = int(int64_t(n_far_tiles) * k / m), n_end = int(int64_t(n_far_tiles) * (k + 1) / m);
				t_far = TSyrkJob{p0, p1, v0, n_far_c1, n_begin, n_end - n_begin};
			}
			if(n_wide_tiles > 0) {
				const int n_begin = int(int64_t(n_wide_tiles) * k / m), n_end = int(int64_t(n_wide_tiles) * (k + 1) / m);
				t_wide = TSyrkJob{p0, p1, n_wide_c0, n_wide_T2, n_begin, n_end - n_begin};
			}
			double *invL = p_invdiag + size_t(kb) * NB * NB;
			// (the longest jobs first -- K = 256 before K = 64 --: the workgroups are started in index order, and a launch ends
			// with its last job; 3.43 -> 3.26 ms at n = 6 000.  Sizing the four shares of the far update so that the launches
			// carry the same number of K tiles -- the launch of tile 0 also has the K = 256 update of the next panel --
			// changed nothing: 3.27)
			hipLaunchKernelGGL(potrf_diag_kernel, dim3(1 + t_wide.n_tiles + t_inner.n_tiles + t_near.n_tiles + t_far.n_tiles), dim3(256), 0, stream,
				M, n_pad, kb, n, invL, p_flag, n_blocks, t_wide, t_far, t_near, t_inner);
			const int n_below = n_blocks - kb - 1;
			if(n_below > 0)
				hipLaunchKernelGGL(trsm_kernel, dim3(n_below), dim3(512), 0, stream, M, n_pad, kb, invL, 1);
		}
	}
}

void dense_factor_panel(double *M, int n_pad, int n, int t0, int t1, double *p_invdiag, int *p_flag, hipStream_t stream)
{
	const int n_blocks = n_pad / NB;
	const TSyrkJob t_none = {0, 0, 0, 0, 0, 0};
	for(int kb = t0; kb < t1; ++ kb) {
		double *invL = p_invdiag + size_t(kb) * NB * NB;
		hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, stream, M, n_pad, kb, n, invL, p_flag, n_blocks, t_none, t_none, t_none, t_none);
		const int n_below = n_blocks - kb - 1;
		if(n_below > 0)
			hipLaunchKernelGGL(trsm_kernel, dim3(n_below), dim3(512), 0, stream, M, n_pad, kb, invL, 0);
		launch_syrk(M, n_pad, n_blocks, kb, kb + 1, kb + 1, t1, stream); // the rest of this panel's columns
	}
}

void dense_update_panels(double *M, int n_pad, int k0, int k1, int c0, int c1, hipStream_t stream)
{
	launch_syrk(M, n_pad, n_pad / NB, k0, k1, c0, c1, stream);
}

// ---- backward substitution x = L^-T y, right-looking ----
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
// A thread's eight values of a tile row are four 16-byte pairs, 2 part + 16 i: the eight threads of a row read whole
// 128-byte lines (as 8-byte loads at part + 8 i they touched every line twice: half the load instructions now).
__global__ void __launch_bounds__(512)
dense_backsolve_panel_kernel(const double *M, int ld, int t0, int t1, const double *p_invdiag, double *z, double *x, int n_first,
	const longlong2 *__restrict__ p_dst, double *p_w, double *p_x_out)
{
	// n_first >= 0: the first launch of a substitution (the last panel) takes z = y out of row ld-1 of the factor (columns
	// < n_first; zero beyond) instead of from z -- what a launch of its own used to copy.  p_dst (optional): where entry i of
	// the dense system goes in the solver's vectors (.x in w, .y in the caller's x; < 0 for padding): the publishing
	// workgroup stores there as well, and no scatter launch follows the substitution.
	enum { PARTS = 8, PER = NB / PARTS, N_PAIRS = OUTER_TILES * (OUTER_TILES - 1) / 2 };
	__shared__ double s_z[OUTER_TILES * NB]; // z of the panel
	__shared__ double s_x[OUTER_TILES * NB]; // x of the panel, tile by tile (an array of its own: nobody waits for the readers of z)
	const int t = threadIdx.x, c = t / PARTS, part = t % PARTS;
	const int jb = blockIdx.x;
	const int m = t1 - t0;
	const bool b_strip = jb < t0;
	double vi[OUTER_TILES][PER], vl[N_PAIRS][PER], vs[OUTER_TILES][PER];
	#pragma unroll
	for(int a = 0; a < OUTER_TILES; ++ a) {
		if(a < m) {
			const double *invL = p_invdiag + size_t(t0 + a) * NB * NB + c * NB + 2 * part;
			#pragma unroll
			for(int i = 0; i < PER; i += 2) {
				const v2f64 v = *reinterpret_cast<const v2f64*>(invL + PARTS * i);
				vi[a][i] = v.x;
				vi[a][i + 1] = v.y;
			}
			#pragma unroll
			for(int b = 0; b < a; ++ b) {
				const double *col = M + size_t((t0 + a) * NB + 2 * part) + size_t((t0 + b) * NB + c) * ld;
				#pragma unroll
				for(int i = 0; i < PER; i += 2) {
					const v2f64 v = *reinterpret_cast<const v2f64*>(col + PARTS * i);
					vl[a * (a - 1) / 2 + b][i] = v.x;
					vl[a * (a - 1) / 2 + b][i + 1] = v.y;
				}
			}
			if(b_strip) {
				const double *col = M + size_t((t0 + a) * NB + 2 * part) + size_t(jb * NB + c) * ld;
				#pragma unroll
				for(int i = 0; i < PER; i += 2) {
					const v2f64 v = *reinterpret_cast<const v2f64*>(col + PARTS * i);
					vs[a][i] = v.x;
					vs[a][i + 1] = v.y;
				}
			}
		}
	}
	const bool b_first = n_first >= 0;
	for(int i = t; i < m * NB; i += 512) {
		const int q = t0 * NB + i;
		s_z[i] = b_first? ((q < n_first)? M[size_t(ld - 1) + size_t(q) * ld] : 0.0) : z[q];
	}
	double z_in = 0; // the strip's own entry, requested with everything else
	if(b_strip && part == 0) {
		const int q = jb * NB + c;
		z_in = b_first? ((q < n_first)? M[size_t(ld - 1) + size_t(q) * ld] : 0.0) : z[q];
	}
	__syncthreads();
	#pragma unroll
	for(int a = OUTER_TILES - 1; a >= 0; -- a) {
		if(a < m) { // workgroup-uniform
			double *zk = s_z + a * NB;
			double sum = 0; // x_kb[c] = sum_r inv(L_kk)[r][c] z_kb[r]; PARTS lanes per entry
			#pragma unroll
			for(int i = 0; i < PER; ++ i)
				sum += vi[a][i] * zk[2 * part + PARTS * (i & ~1) + (i & 1)];
			sum += __shfl_xor(sum, 1);
			sum += __shfl_xor(sum, 2);
			sum += __shfl_xor(sum, 4);
			double *xk = s_x + a * NB;
			if(part == 0)
				xk[c] = sum;
			__syncthreads();
			#pragma unroll
			for(int b = 0; b < a; ++ b) { // z_j -= L(kb, j)^T x_kb inside the panel
				double upd = 0;
				#pragma unroll
				for(int i = 0; i < PER; ++ i)
					upd += vl[a * (a - 1) / 2 + b][i] * xk[2 * part + PARTS * (i & ~1) + (i & 1)];
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
		for(int i = t; i < m * NB; i += 512) {
			const double v = s_x[i];
			x[t0 * NB + i] = v;
			if(p_dst) {
				const longlong2 d = p_dst[t0 * NB + i];
				if(d.x >= 0) {
					p_w[d.x] = v;
					p_x_out[d.y] = v;
				}
			}
		}
	}
	if(!b_strip)
		return; // the only workgroup of the first panel just publishes
	double sum = 0;
	#pragma unroll
	for(int a = 0; a < OUTER_TILES; ++ a) {
		if(a < m) {
			#pragma unroll
			for(int i = 0; i < PER; ++ i)
				sum += vs[a][i] * s_x[a * NB + 2 * part + PARTS * (i & ~1) + (i & 1)];
		}
	}
	sum += __shfl_xor(sum, 1);
	sum += __shfl_xor(sum, 2);
	sum += __shfl_xor(sum, 4);
	if(part == 0)
		z[jb * NB + c] = z_in - sum;
}

void dense_backsolve(const double *M, int n_pad, int n, const double *p_invdiag, double *p_z, double *p_x, hipStream_t stream,
	const longlong2 *p_dst, double *p_w, double *p_x_out)
{
	const int n_blocks = n_pad / NB;
	for(int t1 = n_blocks; t1 > 0; t1 -= OUTER_TILES) {
		const int t0 = (t1 > OUTER_TILES)? t1 - OUTER_TILES : 0;
		hipLaunchKernelGGL(dense_backsolve_panel_kernel, dim3(t0 > 0? t0 : 1), dim3(512), 0, stream, M, n_pad, t0, t1, p_invdiag,
			p_z, p_x, (t1 == n_blocks)? n : -1, p_dst, p_w, p_x_out);
	}
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(dense_chol) // (the handle's bring-up thread loads this unit's code object: capi.hip)
