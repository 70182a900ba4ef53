// simt_kernel.hip -- wide stages of the sparse block factorization, one *lane* per task.
//
// The stages at the bottom of a nested-dissection elimination tree hold thousands of tasks (at 100 000 poses: 15 894
// leaf subtrees of 4-8 columns, then 7 513 / 3 675 / 1 931 single separator columns), and those tasks come in a handful
// of shapes: the same number of columns, blocks, row entries and update pairs, referring to the same relative
// operands (at 100 000 poses 158 shapes, the six most frequent cover 87 % of the leaves).  The wave-per-task kernel
// (subtree_kernel.hip) spends about 600 wave-instructions on every 6 x 6 column with 36 of 64 lanes busy, most of
// them cross-lane traffic of the in-wave Cholesky (v_readlane, ds_bpermute, LDS tiles); it is bound by instruction
// issue at 6 % of the HBM roof.  Here 64 tasks of one shape share a wave and every lane runs
// the plain scalar algorithm on its own task: Cholesky, inverse and the block products are straight-line FMAs in
// registers, no lane ever talks to another, and one wave-instruction serves 64 columns (about 2 000 FMAs per column
// and lane = 31 wave-instructions per column instead of 600).  What differs between the lanes is data: where the
// blocks live.  So the *program* of a shape (column / block / pair counts and operand indices) is read with scalar
// loads, and a per-lane table holds the offsets (of the task's own blocks, of its Lambda blocks, of its operands).
// Operands go through the caches in the factor's ordinary layout: a lane reads its own 288-byte blocks with 16-byte
// loads (consecutive instructions of a lane walk down the same lines), and a block written by a lane is re-read by
// the same lane a column or two later.
//
// Arithmetic as in the other factor kernels (left-looking, one writer per block, fixed summation order); results
// differ from theirs in the last bits only through the order of the sums inside a 6-term dot product.
// Own translation unit (see dense_tiles.hip for why).
#include <hip/hip_runtime.h>
#include "sparse_kernels.h"

namespace slampp {

__device__ __forceinline__ double simt_rsqrt(double x)
{
	double y = __builtin_amdgcn_rsq(x);
	const double h = 0.5 * x;
	y = y * (1.5 - h * y * y);
	y = y * (1.5 - h * y * y);
	return y;
}

// D consecutive doubles at p (16-byte aligned when D is even: block offsets are multiples of D * D)
template <int D>
__device__ __forceinline__ void load_column(const double *__restrict__ p, double (&v)[D])
{
	if(D % 2 == 0) {
		const double2 *p2 = reinterpret_cast<const double2*>(p);
		#pragma unroll
		for(int i = 0; i < D / 2; ++ i) {
			const double2 t = p2[i];
			v[2 * i] = t.x;
			v[2 * i + 1] = t.y;
		}
	} else {
		#pragma unroll
		for(int i = 0; i < D; ++ i)
			v[i] = p[i];
	}
}

template <int D>
__device__ __forceinline__ void load_block(const double *__restrict__ p, double (&v)[D][D]) // v[c][r] = p[r + c D]
{
	if(D % 2 == 0) {
		const double2 *p2 = reinterpret_cast<const double2*>(p);
		#pragma unroll
		for(int i = 0; i < D * D / 2; ++ i) {
			const double2 t = p2[i];
			v[(2 * i) / D][(2 * i) % D] = t.x;
			v[(2 * i + 1) / D][(2 * i + 1) % D] = t.y;
		}
	} else {
		#pragma unroll
		for(int i = 0; i < D * D; ++ i)
			v[i / D][i % D] = p[i];
	}
}

template <int D>
__device__ __forceinline__ void store_block(double *p, const double (&m)[D][D]) // m[r][q] -> p[r + q D]
{
	if(D % 2 == 0) {
		double2 *p2 = reinterpret_cast<double2*>(p);
		#pragma unroll
		for(int q = 0; q < D; ++ q) {
			#pragma unroll
			for(int r = 0; r < D; r += 2)
				p2[(r + q * D) / 2] = double2{m[r][q], m[r + 1][q]};
		}
	} else {
		#pragma unroll
		for(int q = 0; q < D; ++ q) {
			#pragma unroll
			for(int r = 0; r < D; ++ r)
				p[r + q * D] = m[r][q];
		}
	}
}

// rows [r0, r0 + D / 2) of a block: v[c][i] = p[r0 + i + c D] (two lanes per task: each takes half the rows)
template <int D>
__device__ __forceinline__ void load_half_rows(const double *__restrict__ p, int r0, double (&v)[D][D / 2])
{
	#pragma unroll
	for(int c = 0; c < D; ++ c) {
		#pragma unroll
		for(int i = 0; i < D / 2; ++ i)
			v[c][i] = p[r0 + i + c * D];
	}
}

// position of the next column's header in the program, given the position right behind this column's touch list
__device__ __forceinline__ int pc_next_column(const int32_t *P, int pc, int nb, int nr)
{
	pc += 2 * nr;
	for(int kb = 1; kb < nb; ++ kb)
		pc += 1 + 2 * P[pc];
	return pc;
}

template <int D, int W, int LPT, bool b_linv> // W = tasks per wave, LPT = lanes per task (1, or 2 for even D: the lanes of a pair compute the
// diagonal block of a column both, and each half the rows of the blocks below it -- no traffic between them); b_linv: inv(L_jj)
// goes to memory as well (36 of a column's ~100 doubles at C3; the solve itself no longer reads it: backward_simt_kernel)
__global__ void __launch_bounds__(64)
factor_simt_kernel(const TSimtChunk *__restrict__ chunks, const int32_t *__restrict__ prog, const long long *__restrict__ tab,
	const double *__restrict__ A, double *L, double *Linv, const double *__restrict__ b, double *w, int *p_flag,
	long long *p_timing, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; A += n_member * t_batch.a; L += n_member * t_batch.l; Linv = Linv? Linv + n_member * t_batch.linv : Linv; b += n_member * t_batch.b; w += n_member * t_batch.w; p_flag += n_member; } // (TBatch: sparse_kernels.h)

	enum { DD = D * D };
	extern __shared__ long long s_tab[]; // the chunk's table of per-lane offsets: fetched once, in one go, instead of a
	// trip to memory in front of every operand
	const TSimtChunk ch = chunks[blockIdx.x];
	{
		const int n_entries = (4 * prog[ch.prog_off] + prog[ch.prog_off + 1] + prog[ch.prog_off + 2] + prog[ch.prog_off + 3]) * W;
		for(int i = threadIdx.x; i < n_entries; i += 64)
			s_tab[i] = tab[ch.tab_off + i];
		__syncthreads();
	}
	if(int(threadIdx.x) >= W * LPT)
		return; // (no barrier below)
	const int n_half = (LPT == 2)? int(threadIdx.x) & 1 : 0; // which half of the rows below the diagonal this lane takes
	enum { DH = (LPT == 2)? D / 2 : D };
	const int r0 = n_half * DH;
	long long *p_tm = 0; // development aid (SLAMPP_HIP_STAGE_TIMING): clock samples of a wave in the middle of the grid
	int n_tm = 0;
	if(p_timing && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
		const unsigned long long n_tm_record = atomicAdd((unsigned long long*)p_timing, 1ull);
		p_tm = (n_tm_record < 4096)? p_timing + 1 + 32 * n_tm_record : 0; // the buffer holds 4096 launch records: later launches go unrecorded
		if(p_tm)
			p_tm[n_tm ++] = wall_clock64();
	}
#define SIMT_TICK() do { if(p_tm && n_tm < 32) p_tm[n_tm ++] = wall_clock64(); } while(0)
	const int32_t *P = prog + ch.prog_off;                 // wave-uniform: scalar loads
	const long long *T = s_tab + threadIdx.x / LPT;        // field f of this lane's task at T[W f]
	const int n_cols = P[0], n_blocks = P[1], n_ops = P[2];
	const int f_blk = 4 * n_cols, f_op = f_blk + n_blocks, f_y = f_op + n_ops;
	int pc = 4, blk0 = 0;
	bool b_bad = false;
	double f_touched = 0; // keeps the early loads alive
	for(int ci = 0; ci < n_cols; ++ ci) {
		const int nb = P[pc], nr = P[pc + 1], n_touch = P[pc + 2];
		pc += 3;
		pc += n_touch; // (the column's distinct operands, for a variant that requested their lines ahead: measured slower)
		// the Lambda blocks of the next column come from memory nobody has touched yet: one load per cache line of each goes
		// out now, a whole column of arithmetic ahead of their use
		if(ci + 1 < n_cols) {
			const int nb_next = P[pc_next_column(P, pc, nb, nr)];
			for(int kb = 0; kb < nb_next; ++ kb) {
				const long long enc = T[W * (f_blk + blk0 + nb + kb)];
				const double *p_src = A + ((enc < 0)? 0 : (enc >> 1));
				f_touched += p_src[0] + p_src[16] + p_src[DD - 1];
			}
		}
		const long long l_base = T[W * (4 * ci)], linv_off = T[W * (4 * ci + 1)], cs_new = T[W * (4 * ci + 2)],
			cs_src = T[W * (4 * ci + 3)];
		double a[D][D], y[D]; // a: lower triangle of the diagonal block
		{
			// the diagonal block of Lambda: its upper triangle is what the reference's solvers consume
			// (src/slam/LinearSolver_CholMod.cpp:57); element (r, q), r >= q, is stored element (q, r)
			const long long enc = T[W * (f_blk + blk0)];
			const double *src = A + ((enc < 0)? 0 : (enc >> 1));
			#pragma unroll
			for(int r = 0; r < D; ++ r) {
				double c[D];
				load_column<D>(src + r * D, c); // column r of the stored block: elements (0..D-1, r)
				#pragma unroll
				for(int q = 0; q < D; ++ q)
					a[r][q] = (q <= r && enc >= 0)? c[q] : 0.0;
			}
			load_column<D>(b + cs_src, y);
		}
		{ // blocks L(j,c) of block row j: update of the diagonal block and of the right-hand side; the offsets of entry
			// e + 1 are fetched while entry e is being worked on, and a block is requested whole before its first product
			long long off = 0, yoff = 0;
			if(nr > 0) {
				off = T[W * (f_op + P[pc])];
				yoff = T[W * (f_y + P[pc + 1])];
			}
			for(int e = 0; e < nr; ++ e) {
				pc += 2;
				double c[D][D], yc[D];
				load_block<D>(L + off, c);
				load_column<D>(w + yoff, yc);
				if(e + 1 < nr) {
					off = T[W * (f_op + P[pc])];
					yoff = T[W * (f_y + P[pc + 1])];
				}
				#pragma unroll
				for(int t = 0; t < D; ++ t) {
					#pragma unroll
					for(int r = 0; r < D; ++ r) {
						#pragma unroll
						for(int q = 0; q <= r; ++ q)
							a[r][q] -= c[t][r] * c[t][q];
						y[r] -= c[t][r] * yc[t];
					}
				}
			}
		}
		SIMT_TICK(); // Lambda, row entries
		// Cholesky of the diagonal block, in place; rd[k] = 1 / L(k,k)
		double rd[D];
		#pragma unroll
		for(int k = 0; k < D; ++ k) {
			double piv = a[k][k];
			const bool b_neg = !(piv > 0); // also catches NaN
			b_bad = b_bad || b_neg;
			piv = b_neg? 1.0 : piv;
			const double s = simt_rsqrt(piv);
			rd[k] = s;
			a[k][k] = piv * s;
			#pragma unroll
			for(int r = k + 1; r < D; ++ r)
				a[r][k] *= s;
			#pragma unroll
			for(int q = k + 1; q < D; ++ q) {
				#pragma unroll
				for(int r = q; r < D; ++ r)
					a[r][q] -= a[r][k] * a[q][k];
			}
		}
		// its inverse (lower triangular) and y_j = inv(L_jj) (b_j - sum L(j,c) y_c)
		double x[D][D];
		#pragma unroll
		for(int c = 0; c < D; ++ c) {
			#pragma unroll
			for(int r = 0; r < D; ++ r) {
				if(r < c)
					x[r][c] = 0.0;
				else if(r == c)
					x[r][c] = rd[r];
				else {
					double sum = 0;
					#pragma unroll
					for(int t = c; t < r; ++ t)
						sum += a[r][t] * x[t][c];
					x[r][c] = -sum * rd[r];
				}
			}
		}
		{
			double yn[D];
			#pragma unroll
			for(int r = 0; r < D; ++ r) {
				double sum = 0;
				#pragma unroll
				for(int t = 0; t <= r; ++ t)
					sum += x[r][t] * y[t];
				yn[r] = sum;
			}
			#pragma unroll
			for(int r = 0; r < D; ++ r) {
				if(n_half == 0)
					w[cs_new + r] = yn[r];
				#pragma unroll
				for(int q = r + 1; q < D; ++ q)
					a[r][q] = 0.0; // the factor block is stored whole, zeros above its diagonal
			}
			if(n_half == 0) { // (the lanes of a pair hold the same numbers)
				store_block<D>(L + l_base, a);
				if constexpr(b_linv)
					store_block<D>(Linv + linv_off, x);
			}
		}
		SIMT_TICK(); // diagonal block
		// sub-diagonal blocks: L(i,j) = (Lambda(i,j) - sum L(i,c) L(j,c)^T) inv(L_jj)^T; with two lanes per task each lane
		// its DH rows of the block (inv(L_jj) is in its own registers: x above)
		for(int kb = 1; kb < nb; ++ kb) {
			const int np = P[pc ++];
			const long long enc = T[W * (f_blk + blk0 + kb)];
			double acc[DH][D];
			{
				const double *src = A + ((enc < 0)? 0 : (enc >> 1));
				const bool b_trans = (enc & 1) != 0, b_have = enc >= 0;
				if constexpr(LPT == 1) {
					double v[D][D]; // v[c][r] = stored element (r, c)
					#pragma unroll
					for(int c = 0; c < D; ++ c)
						load_column<D>(src + c * D, v[c]);
					#pragma unroll
					for(int r = 0; r < DH; ++ r) {
						#pragma unroll
						for(int q = 0; q < D; ++ q)
							acc[r][q] = b_have? (b_trans? v[r][q] : v[q][r]) : 0.0;
					}
				} else {
					// element (r0 + i, q) of the block: stored at (r0 + i) + q D, or transposed at q + (r0 + i) D
					#pragma unroll
					for(int i = 0; i < DH; ++ i) {
						#pragma unroll
						for(int q = 0; q < D; ++ q) {
							const double f = src[b_trans? q + (r0 + i) * D : (r0 + i) + q * D];
							acc[i][q] = b_have? f : 0.0;
						}
					}
				}
			}
			{
				long long off_a = 0, off_b = 0;
				if(np > 0) {
					off_a = T[W * (f_op + P[pc])];
					off_b = T[W * (f_op + P[pc + 1])];
				}
				for(int e = 0; e < np; ++ e) {
					pc += 2;
					double ca[D][DH], cb[D][D]; // ca[t][i] = L(i,c) element (r0 + i, t)
					if constexpr(LPT == 1) {
						double full[D][D];
						load_block<D>(L + off_a, full);
						#pragma unroll
						for(int t = 0; t < D; ++ t) {
							#pragma unroll
							for(int i = 0; i < DH; ++ i)
								ca[t][i] = full[t][i];
						}
					} else
						load_half_rows<D>(L + off_a, r0, ca);
					load_block<D>(L + off_b, cb);
					if(e + 1 < np) {
						off_a = T[W * (f_op + P[pc])];
						off_b = T[W * (f_op + P[pc + 1])];
					}
					#pragma unroll
					for(int t = 0; t < D; ++ t) {
						#pragma unroll
						for(int i = 0; i < DH; ++ i) {
							#pragma unroll
							for(int q = 0; q < D; ++ q)
								acc[i][q] -= ca[t][i] * cb[t][q];
						}
					}
				}
			}
			// out = acc inv(L_jj)^T: out[i][q] = sum_{t <= q} acc[i][t] x[q][t] (x[r][c] = element (r, c) of the inverse)
			double *p_out = L + l_base + kb * DD;
			#pragma unroll
			for(int q = 0; q < D; ++ q) {
				double out[DH];
				#pragma unroll
				for(int i = 0; i < DH; ++ i) {
					double sum = 0;
					#pragma unroll
					for(int t = 0; t <= q; ++ t)
						sum += acc[i][t] * x[q][t];
					out[i] = sum;
				}
				if(DH % 2 == 0 && D % 2 == 0) {
					#pragma unroll
					for(int i = 0; i < DH; i += 2)
						*reinterpret_cast<double2*>(p_out + r0 + i + q * D) = double2{out[i], out[i + 1]};
				} else {
					#pragma unroll
					for(int i = 0; i < DH; ++ i)
						p_out[r0 + i + q * D] = out[i];
				}
			}
			SIMT_TICK(); // one sub-diagonal block
		}
		blk0 += nb;
		// (what a lane loads in the following columns is what the lane itself has stored, or what an earlier launch
		// has: program order of one thread, no fence)
	}
	if(b_bad || f_touched == 1.2345e301) // (never equal: the sum only has to be used)
		atomicOr(p_flag, 1);
}

bool launch_factor_simt(const TSimtChunk *chunks, int n_chunks, int n_width, int n_lds_bytes, const int32_t *prog, const int64_t *tab, int n_dim,
	const double *A, double *L, double *Linv, const double *b, double *w, int *p_flag, hipStream_t stream, long long *p_timing, const TBatch &t_batch)
{
	if(n_chunks <= 0)
		return true;
	const long long *t = reinterpret_cast<const long long*>(tab);
#define SIMT_LAUNCH(D_, W_, LPT_) do { if(Linv) hipLaunchKernelGGL((factor_simt_kernel<D_, W_, LPT_, true>), dim3(n_chunks, t_batch.n), dim3(64), n_lds_bytes, stream, chunks, prog, t, \
	A, L, Linv, b, w, p_flag, p_timing, t_batch); else hipLaunchKernelGGL((factor_simt_kernel<D_, W_, LPT_, false>), dim3(n_chunks, t_batch.n), dim3(64), n_lds_bytes, stream, chunks, prog, t, \
	A, L, Linv, b, w, p_flag, p_timing, t_batch); } while(0)
	// (two lanes per task where the block dimension is even and a wave holds at most 32 tasks)
#define SIMT_WIDTHS(D_) do { if(n_width == 16) { if(b_pairs && D_ % 2 == 0) SIMT_LAUNCH(D_, 16, (D_ % 2 == 0)? 2 : 1); else SIMT_LAUNCH(D_, 16, 1); } \
	else if(n_width == 32) { if(b_pairs && D_ % 2 == 0) SIMT_LAUNCH(D_, 32, (D_ % 2 == 0)? 2 : 1); else SIMT_LAUNCH(D_, 32, 1); } \
	else SIMT_LAUNCH(D_, 64, 1); } while(0)
	// Two lanes per task is a third more work per task (both lanes of a pair do the column's diagonal block) for half the
	// dependent chain of the blocks below it: it pays while the launch leaves the chip's SIMDs short of waves (C3: 497
	// waves on 1 024 SIMDs, 120 -> 100 us) and costs where they are full (a million poses: 716 -> 779 us).
	const int n_pairs_env = dev_knob("SLAMPP_HIP_DEV_SIMT_PAIRS", -1); // development aid (plan.h)
	const bool b_pairs = (n_pairs_env >= 0)? n_pairs_env != 0 : n_chunks <= 2048;
	switch(n_dim) {
	case 3:
		SIMT_WIDTHS(3);
		return true;
	case 6:
		SIMT_WIDTHS(6);
		return true;
	case 7:
		SIMT_WIDTHS(7);
		return true;
	default:
		return false;
	}
#undef SIMT_WIDTHS
#undef SIMT_LAUNCH
}

// Backward substitution of the same tasks, one lane per task: x_j = L_jj^-T (y_j - sum_i L(i,j)^T x_i), columns last to first.
// The wave-per-task kernel (backward_stage_kernel) pays a trip to memory per column with 36 lanes busy, 15 894 workgroups for
// C3's leaf subtrees (40 us); here a lane walks its task's columns with its own blocks, the next column's lines requested a
// column ahead, and solves with L_jj^T itself (six dependent steps, reciprocals taken before the chain) -- inv(L_jj) is not
// read, which is what lets the factorization stop writing it.  x of rows inside the task is what the lane stored a moment
// ago (program order of one thread), x of rows above it what earlier launches stored.
template <int D, int W>
__global__ void __launch_bounds__(64)
backward_simt_kernel(const TSimtChunk *__restrict__ chunks, const int32_t *__restrict__ prog, const long long *__restrict__ tab,
	const double *__restrict__ L, double *w, double *x_out, TBatch t_batch)
{	{ const int64_t n_member = blockIdx.y; L += n_member * t_batch.l; w += n_member * t_batch.w; x_out += n_member * t_batch.b; } // (TBatch: sparse_kernels.h)

	enum { DD = D * D };
	extern __shared__ long long s_tab[];
	const TSimtChunk ch = chunks[blockIdx.x];
	const int32_t *P = prog + ch.prog_off; // wave-uniform: scalar loads
	const int n_cols = P[0], n_below = P[1];
	{
		const int n_entries = (3 * n_cols + n_below) * W;
		for(int i = threadIdx.x; i < n_entries; i += 64)
			s_tab[i] = tab[ch.tab_off + i];
		__syncthreads();
	}
	if(int(threadIdx.x) >= W)
		return; // (no barrier below)
	const long long *T = s_tab + threadIdx.x;
	// x of the task's own columns stays in LDS (round 6): a column's x is what the columns before it in the task multiply their
	// blocks with, and through memory that was a store and a load of the same thread through L2 per column -- the longest
	// waits of a kernel that waits 84 % of its time (tools/pmc_waits.sh c3).  Rows outside the task (the separators above,
	// solved by earlier launches) still come from w.  P[2 + n_cols + b]: which of the task's columns the row of block b is, or -1.
	double *s_x = reinterpret_cast<double*>(s_tab + (3 * n_cols + n_below) * W) + threadIdx.x;
	const int32_t *p_local = P + 2 + n_cols;
	int n_blk_end = n_below;
	double f_touched = 0;
	for(int ci = n_cols - 1; ci >= 0; -- ci) {
		const int nb = P[2 + ci];
		const long long l_base = T[W * (3 * ci)], cs_new = T[W * (3 * ci + 1)], cs_src = T[W * (3 * ci + 2)];
		const int n_blk0 = n_blk_end - (nb - 1);
		double acc[D], a[D][D];
		load_column<D>(w + cs_new, acc); // y_j
		load_block<D>(L + l_base, a);    // a[q][r] = L_jj(r, q)
		// The blocks of the column after next: a load per cache line, two columns of arithmetic ahead of their use -- and BEHIND
		// this column's own first loads (round 6).  Loads come back in the order they were issued (one counter): requested in
		// front of the column's own loads, as they were, the touches -- misses all the way to HBM -- were what every column's own,
		// long-arrived data waited behind.  Two columns ahead, because the next column's own loads queue behind these.
		asm volatile("" ::: "memory");
		double f_touch_now = 0;
		if(ci > 1 || (ci == n_cols - 1 && ci > 0)) {
			const int c_first = (ci > 1)? ci - 2 : ci - 1, c_last = (ci == n_cols - 1)? ci - 1 : c_first; // (the first column handled touches both)
			for(int c = c_last; c >= c_first; -- c) {
				const int nb_next = P[2 + c];
				const double *p_next = L + T[W * (3 * c)];
				for(int kb = 0; kb < nb_next; ++ kb)
					f_touch_now += p_next[kb * DD] + p_next[kb * DD + 16] + p_next[kb * DD + DD - 1];
			}
		}
		asm volatile("" ::: "memory");
		double rd[D];
		#pragma unroll
		for(int q = 0; q < D; ++ q) {
			const double d = a[q][q];
			double r = __builtin_amdgcn_rcp(d);
			r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
			r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
			rd[q] = r;
		}
		for(int kb = 1; kb < nb; ++ kb) {
			const long long xcs = T[W * (3 * n_cols + n_blk0 + kb - 1)];
			const int n_local = p_local[n_blk0 + kb - 1]; // (wave-uniform)
			double c[D][D], xi[D];
			load_block<D>(L + l_base + kb * DD, c); // c[q][r] = L(i,j)(r, q)
			if(n_local >= 0) {
				#pragma unroll
				for(int r = 0; r < D; ++ r)
					xi[r] = s_x[(n_local * D + r) * W];
			} else
				load_column<D>(w + xcs, xi);
			#pragma unroll
			for(int q = 0; q < D; ++ q) {
				#pragma unroll
				for(int r = 0; r < D; ++ r)
					acc[q] -= c[q][r] * xi[r];
			}
		}
		double x[D];
		#pragma unroll
		for(int q = D - 1; q >= 0; -- q) {
			double sum = acc[q];
			#pragma unroll
			for(int r = q + 1; r < D; ++ r)
				sum -= a[q][r] * x[r];
			x[q] = sum * rd[q];
		}
		#pragma unroll
		for(int q = 0; q < D; ++ q)
			s_x[(ci * D + q) * W] = x[q];
		f_touched += f_touch_now; // (used here: nothing of this column waits for the touches)
		if(D % 2 == 0) {
			#pragma unroll
			for(int q = 0; q < D; q += 2) {
				*reinterpret_cast<double2*>(w + cs_new + q) = double2{x[q], x[q + 1]};
				*reinterpret_cast<double2*>(x_out + cs_src + q) = double2{x[q], x[q + 1]};
			}
		} else {
			#pragma unroll
			for(int q = 0; q < D; ++ q) {
				w[cs_new + q] = x[q];
				x_out[cs_src + q] = x[q];
			}
		}
		n_blk_end = n_blk0;
	}
	if(f_touched == 1.2345e301) // (never: the sum only has to be used)
		w[0] = f_touched;
}

bool launch_backward_simt(const TSimtChunk *chunks, int n_chunks, int n_width, int n_lds_bytes, const int32_t *prog, const int64_t *tab,
	int n_dim, const double *L, double *w, double *x_out, hipStream_t stream, const TBatch &t_batch)
{
	if(n_chunks <= 0)
		return true;
	const long long *t = reinterpret_cast<const long long*>(tab);
#define BWD_LAUNCH(D_, W_) hipLaunchKernelGGL((backward_simt_kernel<D_, W_>), dim3(n_chunks, t_batch.n), dim3(64), n_lds_bytes, stream, chunks, prog, t, L, w, x_out, t_batch)
#define BWD_WIDTHS(D_) do { if(n_width == 16) BWD_LAUNCH(D_, 16); else if(n_width == 32) BWD_LAUNCH(D_, 32); else BWD_LAUNCH(D_, 64); } while(0)
	switch(n_dim) {
	case 3:
		BWD_WIDTHS(3);
		return true;
	case 6:
		BWD_WIDTHS(6);
		return true;
	case 7:
		BWD_WIDTHS(7);
		return true;
	default:
		return false;
	}
#undef BWD_WIDTHS
#undef BWD_LAUNCH
}

// inv(L_jj) from L_jj, a thread per column (forward substitution on the identity)
template <int D>
__global__ void invert_diagonals_kernel(TDevPlan p, int64_t col_begin, int64_t col_end, const double *__restrict__ L, double *Linv)
{
	const int64_t c = col_begin + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(c >= col_end)
		return;
	const TColDesc &cd = p.cols[c];
	const double *a = L + p.blks[cd.k0].loff; // element (r, q) at r + q D
	double x[D][D]; // x[r][c]
	#pragma unroll
	for(int q = 0; q < D; ++ q) {
		#pragma unroll
		for(int r = 0; r < D; ++ r) {
			if(r < q)
				x[r][q] = 0.0;
			else {
				double sum = (r == q)? 1.0 : 0.0;
				#pragma unroll
				for(int t = q; t < r; ++ t)
					sum -= a[r + t * D] * x[t][q];
				x[r][q] = sum / a[r + r * D];
			}
		}
	}
	double *p_out = Linv + cd.linv_off;
	#pragma unroll
	for(int q = 0; q < D; ++ q) {
		#pragma unroll
		for(int r = 0; r < D; ++ r)
			p_out[r + q * D] = x[r][q];
	}
}

bool launch_invert_diagonals(const TDevPlan &p, int64_t col_begin, int64_t col_end, const double *L, double *Linv, hipStream_t stream)
{
	if(col_end <= col_begin)
		return true;
	const unsigned n_grid = unsigned((col_end - col_begin + 127) / 128);
	switch(p.uniform_dim) {
	case 3:
		hipLaunchKernelGGL((invert_diagonals_kernel<3>), dim3(n_grid), dim3(128), 0, stream, p, col_begin, col_end, L, Linv);
		return true;
	case 6:
		hipLaunchKernelGGL((invert_diagonals_kernel<6>), dim3(n_grid), dim3(128), 0, stream, p, col_begin, col_end, L, Linv);
		return true;
	case 7:
		hipLaunchKernelGGL((invert_diagonals_kernel<7>), dim3(n_grid), dim3(128), 0, stream, p, col_begin, col_end, L, Linv);
		return true;
	default:
		return false;
	}
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(simt_kernel) // (the handle's bring-up thread loads this unit's code object: capi.hip)
