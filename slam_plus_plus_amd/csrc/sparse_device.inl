// sparse_device.inl -- device helpers shared by the kernels of the sparse path (sparse_kernels.hip, subtree_kernel.hip);
// included inside namespace slampp.
#ifndef SLAMPP_SPARSE_DEVICE_INL
#define SLAMPP_SPARSE_DEVICE_INL // (included inside namespace slampp; the guard lets several of the .hip files be compiled as one translation unit)

static const int64_t PAIR_OFF_MASK = (int64_t(1) << 48) - 1; // pair.x = offset | position of the target block in its column << 48 | dim << 56
enum { Y_LANE0 = 56 }; // lanes 56.. carry the right-hand side of the column when its dimension is <= 7
enum { CHUNK = UP_CHUNK }; // blocks of a column whose partial sums live in LDS at a time (multi-wave kernel)

__device__ __forceinline__ void wave_sync()
{
	// LDS operations of one wave execute in order; this only stops the compiler from moving
	// LDS accesses across the point where lanes exchange data
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the update pairs e = first, first + step, ... of block (i,j):  sum_t L(i,c)[r,t] L(j,c)[q,t]
template <int D>
__device__ __forceinline__ double accumulate_pairs(const longlong2 *__restrict__ pairs, int64_t p0, int np,
	int first, int step, const double *L, int r, int q, int di, int dj)
{
	double acc = 0;
	for(int e = first; e < np; e += step) {
		const longlong2 pr = pairs[p0 + e];
		const double *a = L + (pr.x & PAIR_OFF_MASK) + r;
		const double *b = L + pr.y + q;
		if(D) {
			double av[D? D : 1], bv[D? D : 1];
			#pragma unroll
			for(int t = 0; t < D; ++ t) {
				av[t] = a[t * D];
				bv[t] = b[t * D];
			}
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				acc += av[t] * bv[t];
		} else {
			const int dc = int(pr.x >> 56);
			for(int t = 0; t < dc; ++ t)
				acc += a[t * di] * b[t * dj];
		}
	}
	return acc;
}

// the same for the diagonal block, driven by the row entries L(j,c) of block row j; lanes flagged
// b_y accumulate the right-hand side instead: sum_t y_c[t] L(j,c)[q,t]
template <int D>
__device__ __forceinline__ double accumulate_row(const TRowEnt *__restrict__ rents, int64_t r0, int nr,
	int first, int step, const double *L, const double *w, int r, int q, int dj, bool b_y)
{
	double acc = 0;
	for(int e = first; e < nr; e += step) {
		const TRowEnt en = rents[r0 + e];
		const double *b = L + en.off + q;
		const double *a = b_y? w + en.ycs : L + en.off + r;
		const int as = b_y? 1 : dj;
		if(D) {
			double av[D? D : 1], bv[D? D : 1];
			#pragma unroll
			for(int t = 0; t < D; ++ t) {
				av[t] = a[t * as];
				bv[t] = b[t * D];
			}
			#pragma unroll
			for(int t = 0; t < D; ++ t)
				acc += av[t] * bv[t];
		} else {
			for(int t = 0; t < en.dc; ++ t)
				acc += a[t * as] * b[t * dj];
		}
	}
	return acc;
}

__device__ __forceinline__ double lambda_element(const double *__restrict__ A, int64_t enc, int r, int q,
	int di, int dj, bool b_diag)
{
	if(enc < 0)
		return 0;
	// diagonal blocks: read the upper triangle, which is what the reference's solvers consume
	// (src/slam/LinearSolver_CholMod.cpp:57); off-diagonal: as stored or transposed
	const int64_t off = enc >> 1;
	return ((enc & 1) || b_diag)? A[off + q + int64_t(r) * dj] : A[off + r + int64_t(q) * di];
}

// wave 0, diagonal block: Cholesky of the d x d block held one element per lane (lane = r + q d),
// its inverse, the forward-substituted right-hand side; everything is written out
template <int D>
__device__ __forceinline__ void finish_diagonal(const TColDesc &cd, double a, double ay, int lane, int r, int q,
	bool b_act, bool b_y_inline, const TDevPlan &p, double *L, double *Linv, const double *b, double *w,
	int64_t loff, int *p_flag, double *s_linv, double *s_rdiag, double *s_tile0)
{
	const int dj = D? D : cd.dj;
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
		L[loff + lane] = a;
		s_tile0[r + 8 * q] = a;
	}
	wave_sync();
	// inverse of the lower-triangular L_jj: lane c computes column c by forward substitution
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
					sum += s_tile0[rr + 8 * t] * s_linv[t + 8 * cq];
				x = -sum * s_rdiag[rr];
			}
			s_linv[rr + 8 * cq] = x;
		}
	}
	wave_sync();
	if(b_act)
		Linv[cd.linv_off + lane] = s_linv[r + 8 * q];
	// y_j = inv(L_jj) (b_j - sum L(j,c) y_c); the bracket sits in ay of lanes Y_LANE0 + t
	if(b_y_inline) {
		const int yq = lane - Y_LANE0;
		double y = 0;
		for(int t = 0; t < dj; ++ t) {
			const double vt = __shfl(ay, Y_LANE0 + t);
			if(yq >= t)
				y += vt * s_linv[(yq & 7) + 8 * t];
		}
		if(yq >= 0 && yq < dj)
			w[cd.cs_new + yq] = y;
	}
}

// 1 / sqrt(x): hardware estimate + two Newton steps (full double precision to within an ulp or two;
// the IEEE sqrt + divide pair costs three times as many instructions, and this kernel is issue-bound)
__device__ __forceinline__ double rsqrt_newton(double x)
{
	double y = __builtin_amdgcn_rsq(x);
	const double h = 0.5 * x;
	y = y * (1.5 - h * y * y);
	y = y * (1.5 - h * y * y);
	return y;
}

__device__ __forceinline__ double read_lane(double v, int n_lane) // n_lane must be wave-uniform
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), n_lane);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), n_lane);
	return __hiloint2double(hi, lo);
}

// the same as finish_diagonal for a compile-time dimension: branch-free, the inverse stays in registers
// (lane c < D owns column c), broadcasts of single elements are v_readlane with constant lane numbers
template <int D>
__device__ __forceinline__ void finish_diagonal_fixed(const TColDesc &cd, double a, double ay, int lane, int r, int q,
	bool b_act, double *L, double *Linv, double *w, int64_t loff, int *p_flag, double *s_linv,
	double *p_l_copy = 0, double *p_y_copy = 0) // the copies: where a caller keeps the block / y_j in LDS as well
{
	a = (r < q)? 0.0 : a;
	bool b_bad = false;
	double rd[D]; // 1 / L(k,k)
	#pragma unroll
	for(int kk = 0; kk < D; ++ kk) {
		double piv = read_lane(a, kk + kk * D);
		const bool b_neg = !(piv > 0); // also catches NaN
		b_bad = b_bad || b_neg;
		piv = b_neg? 1.0 : piv;
		const double s = rsqrt_newton(piv);
		rd[kk] = s;
		const double lcol = a * s; // meaningful in lanes (., kk)
		const double lr = __shfl(lcol, r + kk * D);
		const double lq = __shfl(lcol, q + kk * D);
		const double upd = a - lr * lq;
		a = (q == kk)? ((r >= kk)? lcol : 0.0) : ((q > kk && r >= q)? upd : a);
	}
	if(b_bad && lane == 0)
		atomicOr(p_flag, 1);
	if(b_act) {
		L[loff + lane] = a;
		if(p_l_copy)
			p_l_copy[lane] = a;
	}
	// inverse of L_jj: x[rr] = element (rr, c) of the inverse in lane c
	const int c = lane;
	double x[D];
	#pragma unroll
	for(int rr = 0; rr < D; ++ rr) {
		double sum = 0;
		#pragma unroll
		for(int t = 0; t < rr; ++ t)
			sum += read_lane(a, rr + t * D) * x[t];
		x[rr] = (((rr == c)? 1.0 : 0.0) - sum) * rd[rr];
	}
	wave_sync(); // earlier readers of s_linv (previous column's blocks) are done
	if(lane < D) {
		#pragma unroll
		for(int rr = 0; rr < D; ++ rr)
			s_linv[rr + 8 * lane] = x[rr];
	}
	wave_sync();
	if(b_act)
		Linv[cd.linv_off + lane] = s_linv[r + 8 * q];
	// y_j = inv(L_jj) (b_j - sum L(j,c) y_c); the bracket sits in ay of lanes Y_LANE0 + t
	const int yq = lane - Y_LANE0;
	double y = 0;
	#pragma unroll
	for(int t = 0; t < D; ++ t) {
		const double vt = read_lane(ay, Y_LANE0 + t);
		const double li = s_linv[(yq & 7) + 8 * t];
		y += (yq >= t)? vt * li : 0.0;
	}
	if(yq >= 0 && yq < D) {
		w[cd.cs_new + yq] = y;
		if(p_y_copy)
			p_y_copy[yq] = y;
	}
}


// ---- a block column as rows (panel_kernel.hip, round 4) ------------------------------------------------------------------
// DPP row broadcast of a double: every lane gets the value of lane N of its own row of 16 lanes (row_newbcast, the one DPP
// control the 64-bit ALU takes).  The fused multiply-add with a broadcast operand is inline assembly (the compiler keeps a
// v_mov_b64_dpp + v_fmac_f64), and the two wait states a DPP read needs after a VALU write of the same register are the
// caller's to provide (b_guard: an s_nop 1 in front): the hazard recogniser does not look inside inline assembly.  The
// statements are volatile: they keep the order they are written in, so an unguarded one can never be moved in front of the
// guarded one that follows the write of its source (the scheduler was free to do that with the trailing columns' FMAs,
// which do not depend on each other -- advisor, round 4).
template <int N>
__device__ __forceinline__ double dpp16_bcast(double v)
{
	double r;
	asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
	return r;
}
template <int N, bool b_guard>
__device__ __forceinline__ void dpp16_fmac(double &r_acc, double src, double mul) // acc += (src of lane N of the row) * mul
{
	if constexpr(b_guard)
		asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(r_acc) : "v"(src), "v"(mul), "n"(N));
	else
		asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(r_acc) : "v"(src), "v"(mul), "n"(N));
}

// Cholesky of a block column held one scalar ROW per lane: lanes 0 .. D - 1 of every row of 16 lanes hold the rows of the
// diagonal block (four identical copies), the other lanes rows of the blocks below it -- and the right-hand side b_j^T,
// which is just one more row: the same column operations turn it into y_j^T (the forward substitution is free).  Step k:
// pivot from lane k, s = 1 / sqrt, a[k] *= s, a[q] -= a[k] * (a[k] of lane q) for q > k -- one v_fmac_f64_dpp each, no LDS,
// no scalar registers, no inverse of the diagonal block on the chain (the blocks below come out of the same steps, where
// the block-wise form first inverts L_jj and then multiplies: 1.7 - 2.0 + 0.5 - 1.1 us a tree level; this: ~0.35).
template <int D, int K, int Q>
__device__ __forceinline__ void column_rows_trailing(double (&a)[D], double nt)
{
	if constexpr(Q < D) {
		dpp16_fmac<Q, (Q == K + 1)>(a[Q], a[K], nt);
		column_rows_trailing<D, K, Q + 1>(a, nt);
	}
}
template <int D, int K>
__device__ __forceinline__ void column_rows_steps(double (&a)[D], double (&piv_raw)[D])
{
	if constexpr(K < D) {
		const double pr = dpp16_bcast<K>(a[K]);
		piv_raw[K] = pr; // looked at after the chain (a compare and select on it costs ~70 cycles per step)
		const double s = rsqrt_newton(__builtin_fmax(pr, 1e-300)); // not positive: finite garbage in a block nobody reads, flag raised by the caller
		const double nt = a[K] * -s;
		a[K] *= s;
		column_rows_trailing<D, K, K + 1>(a, nt);
		column_rows_steps<D, K + 1>(a, piv_raw);
	}
}

// inverse of a finished diagonal block (element per lane in a, lower triangle) by the calling wave: Linv and nothing else --
// off the factorization's chain, for the backward substitution and the covariances
template <int D>
__device__ __forceinline__ void invert_diagonal_fixed(double a, int lane, int r, int q, bool b_act, double *Linv, int64_t linv_off, double *s_linv)
{
	const int c = lane;
	double x[D];
	#pragma unroll
	for(int rr = 0; rr < D; ++ rr) {
		const double d = read_lane(a, rr + rr * D);
		double rd = __builtin_amdgcn_rcp(d);
		rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
		rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
		double sum = 0;
		#pragma unroll
		for(int t = 0; t < rr; ++ t)
			sum += read_lane(a, rr + t * D) * x[t];
		x[rr] = (((rr == c)? 1.0 : 0.0) - sum) * rd;
	}
	wave_sync();
	if(lane < D) {
		#pragma unroll
		for(int rr = 0; rr < D; ++ rr)
			s_linv[rr + 8 * lane] = x[rr];
	}
	wave_sync();
	if(b_act)
		Linv[linv_off + lane] = s_linv[r + 8 * q];
}

// dimension-8 columns have no spare lanes for the right-hand side: a second pass by lanes 0..7
__device__ __forceinline__ void finish_rhs_wide(const TColDesc &cd, int lane, const TDevPlan &p, const double *L,
	const double *b, double *w, const double *s_linv)
{
	const int dj = cd.dj, q = lane & 7;
	double ay = accumulate_row<0>(p.rents, cd.r0, cd.nr, lane >> 3, 8, L, w, 0, q, dj, true);
	ay += __shfl_xor(ay, 8);
	ay += __shfl_xor(ay, 16);
	ay += __shfl_xor(ay, 32);
	ay = ((q < dj)? b[cd.cs_src + q] : 0) - ay;
	double y = 0;
	for(int t = 0; t < dj; ++ t) {
		const double vt = __shfl(ay, t);
		if(q >= t)
			y += vt * s_linv[q + 8 * t];
	}
	if(lane < dj)
		w[cd.cs_new + lane] = y;
}

// L(i,j) = acc Linv^T through the wave's LDS tile
template <int D>
__device__ __forceinline__ void finish_offdiagonal(double acc, int lane, int r, int q, bool b_act, int dj,
	double *L, int64_t loff, double *s_tile, const double *s_linv, double *p_l_copy = 0)
{
	wave_sync(); // the previous use of the tile is over
	if(b_act)
		s_tile[r + 8 * q] = acc;
	wave_sync();
	double v = 0;
	if(D) {
		#pragma unroll
		for(int t = 0; t < D; ++ t)
			if(t <= q) v += s_tile[r + 8 * t] * s_linv[q + 8 * t];
	} else {
		for(int t = 0; t <= q; ++ t)
			v += s_tile[r + 8 * t] * s_linv[q + 8 * t];
	}
	if(b_act) {
		L[loff + lane] = v;
		if(p_l_copy)
			p_l_copy[lane] = v;
	}
}

// block a lane's element belongs to when all blocks of the column have dimension di x dj
struct TLaneMap {
	bool b_act;
	int r, q;
};

__device__ __forceinline__ TLaneMap lane_map(int lane, int di, int dj)
{
	TLaneMap m;
	m.b_act = lane < di * dj;
	m.r = m.b_act? lane % di : 0;
	m.q = m.b_act? lane / di : 0;
	return m;
}

// which sub-diagonal block of the column update pair g belongs to (blocks own consecutive pair ranges)
__device__ __forceinline__ int pair_block(const TBlkDesc *s_blk, int nb, int64_t g)
{
	int kb = 1;
	while(kb + 1 < nb && s_blk[kb + 1].p0 <= g)
		++ kb;
	return kb;
}

// one update pair: sum_t L(i,c)[r,t] L(j,c)[q,t]
template <int D>
__device__ __forceinline__ double pair_product(const longlong2 pr, const double *L, int r, int q, int di, int dj)
{
	const double *a = L + (pr.x & PAIR_OFF_MASK) + r;
	const double *b = L + pr.y + q;
	double sum = 0;
	if(D) {
		double av[D? D : 1], bv[D? D : 1];
		#pragma unroll
		for(int t = 0; t < D; ++ t) {
			av[t] = a[t * D];
			bv[t] = b[t * D];
		}
		#pragma unroll
		for(int t = 0; t < D; ++ t)
			sum += av[t] * bv[t];
	} else {
		const int dc = int(pr.x >> 56);
		for(int t = 0; t < dc; ++ t)
			sum += a[t * di] * b[t * dj];
	}
	return sum;
}

// one row entry for the diagonal block (or, in the y lanes, for the right-hand side)
template <int D>
__device__ __forceinline__ double row_product(const TRowEnt en, const double *L, const double *w, int r, int q,
	int dj, bool b_y)
{
	const double *b = L + en.off + q;
	const double *a = b_y? w + en.ycs : L + en.off + r;
	const int as = b_y? 1 : dj;
	double sum = 0;
	if(D) {
		double av[D? D : 1], bv[D? D : 1];
		#pragma unroll
		for(int t = 0; t < D; ++ t) {
			av[t] = a[t * as];
			bv[t] = b[t * D];
		}
		#pragma unroll
		for(int t = 0; t < D; ++ t)
			sum += av[t] * bv[t];
	} else {
		for(int t = 0; t < en.dc; ++ t)
			sum += a[t * as] * b[t * dj];
	}
	return sum;
}

// ---- operands in an LDS image of the factor (subtree_kernel.hip, panel_kernel.hip) ----
// sum_t a[r + t D] b[q + t D] for matrix lanes, sum_t y[t] b[q + t D] for the right-hand side lanes
template <int D>
__device__ __forceinline__ double row_product_image(const double *blk, const double *yv, int r, int q, bool b_y)
{
	const double *a = b_y? yv : blk + r;
	const int as = b_y? 1 : D;
	double av[D], bv[D];
	#pragma unroll
	for(int t = 0; t < D; ++ t) {
		av[t] = a[t * as];
		bv[t] = blk[q + t * D];
	}
	double sum = 0;
	#pragma unroll
	for(int t = 0; t < D; ++ t)
		sum += av[t] * bv[t];
	return sum;
}

template <int D>
__device__ __forceinline__ double pair_product_image(const double *a, const double *b, int r, int q)
{
	double av[D], bv[D];
	#pragma unroll
	for(int t = 0; t < D; ++ t) {
		av[t] = a[r + t * D];
		bv[t] = b[q + t * D];
	}
	double sum = 0;
	#pragma unroll
	for(int t = 0; t < D; ++ t)
		sum += av[t] * bv[t];
	return sum;
}


#endif // SLAMPP_SPARSE_DEVICE_INL
