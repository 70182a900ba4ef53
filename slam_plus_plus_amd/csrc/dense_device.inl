// dense_device.inl -- device helpers shared by dense_chol.hip and dense_tiles.hip (included inside namespace slampp):
// the MFMA tile product, LDS staging of 64 x 64 operand tiles, and the factorization + inverse of a diagonal tile.

typedef double v4f64 __attribute__((ext_vector_type(4)));

enum { NB = dense_NB };

// LDS operand tiles are stored [k][row] with leading dimension 64 and rows XOR-swizzled by 16 on odd k: the
// two k-groups a half-wave reads in one fragment load land on disjoint halves of the banks, and a 64 x 64
// operand pair takes 64 KB, so two workgroups (or one next to the diagonal-tile kernel) fit on a CU
__device__ __forceinline__ int lds_at(int k, int row) { return k * NB + (row ^ ((k & 1) << 4)); }

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

// ---- panel solve of one tile by eight waves ----
// L(i,j) = A(i,j) inv(L_jj)^T for the tile at (row0, col0); with b_diag the workgroup also applies its tile to the
// diagonal tile of its row, A(i,i) -= L(i,j) L(i,j)^T (the one update the next diagonal tile waits for).  512 threads:
// a 64 x 64 x 64 product is 64 MFMAs of 64 cycles per wave with four waves, 1.7 us on one CU whatever else is idle; the
// workgroup with the diagonal tile does two of them one after the other and is what the launch lasts as long as.
// Eight waves (column block wave & 3, row half wave >> 2) halve both.
__device__ __forceinline__ void load_tile8(double *Ts, const double *M, int ld, int row0, int col0)
{
	const int r = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
	v2f64 v[NB / 16];
	#pragma unroll
	for(int i = 0; i < NB / 16; ++ i)
		v[i] = *reinterpret_cast<const v2f64*>(M + size_t(row0 + r) + size_t(col0 + c0 + 16 * i) * ld);
	#pragma unroll
	for(int i = 0; i < NB / 16; ++ i)
		*reinterpret_cast<v2f64*>(Ts + lds_at(c0 + 16 * i, r)) = v[i];
}

// The two products of a panel solve are not full ones: inv(L_jj) is lower triangular (Q[c][k] = 0 for k > c: column block cb
// of the result needs the K blocks 0 .. cb only, 40 of the 64 block products of 16 x 16 x 16) and of the symmetric update only
// the 10 lower blocks are kept -- 160 matrix instructions instead of 256 each, and at the 105 clocks the fp64 matrix pipe
// takes per instruction (tools/micro/mfma_f64_peak.hip) that is 1.75 instead of 2.8 us on the four SIMDs of the one CU that a
// tile's workgroup runs on, for the launch the whole level waits for.  Balanced over the eight waves (w and w + 4 share a
// SIMD):
//   solve    wave w: row block w & 3, column blocks {0, 3} (w < 4) or {1, 2}: 1 + 4 = 2 + 3 = 5 block products each;
//   update   the lower blocks in the order (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) (3,1) | (3,2) (3,3): wave w takes block w
//            whole (16 instructions); the last two are split by K halves over the waves 0 .. 3 (8 more instructions: the two
//            waves of a SIMD have 24 + 16 = 40), and the halves meet in LDS.
template <int CB>
__device__ __forceinline__ void tri_block_product(const double *Ps, const double *Qs, int rb, int lo, int hi, v4f64 &r_acc)
{
	#pragma unroll
	for(int ks = 0; ks < 4 * (CB + 1); ++ ks) { // K blocks 0 .. CB
		const int k = ks * 4 + hi;
		r_acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Qs[lds_at(k, 16 * CB + lo)], Ps[lds_at(k, 16 * rb + lo)], r_acc, 0, 0, 0);
	}
}

// K steps [ks0, ks1) of the lower block (rb, cb) of T T^T, T in LDS as [k][row]
__device__ __forceinline__ void sym_block_product(const double *Ts, int rb, int cb, int ks0, int ks1, int lo, int hi, v4f64 &r_acc)
{
	#pragma unroll 8
	for(int ks = ks0; ks < ks1; ++ ks) {
		const int k = ks * 4 + hi;
		r_acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ts[lds_at(k, 16 * cb + lo)], Ts[lds_at(k, 16 * rb + lo)], r_acc, 0, 0, 0);
	}
}

__device__ __forceinline__ void trsm_tile_body8(double *M, int ld, int row0, int col0, const double *invL, bool b_diag,
	double *Ps, double *Qs)
{
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int lo = lane & 15, hi = lane >> 4, rb = wave & 3;
	const bool b_outer = wave < 4; // column blocks {0, 3}; the others {1, 2}
	const int cb0 = b_outer? 0 : 1, cb1 = b_outer? 3 : 2;
	load_tile8(Ps, M, ld, row0, col0);
	load_tile8(Qs, invL, NB, 0, 0);
	// the update's blocks: wave w the w-th lower block; (3,2) and (3,3) by halves of K on the waves 0, 1 and 2, 3
	const int n_blk_r = (wave < 1)? 0 : (wave < 3)? 1 : (wave < 6)? 2 : 3, n_blk_c = wave - n_blk_r * (n_blk_r + 1) / 2;
	double cv[4], cv_h[4];
	if(b_diag) { // (workgroup-uniform) the diagonal tile is requested now, its latency hides behind the two products
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			cv[reg] = M[size_t(row0 + 16 * n_blk_r + lo) + size_t(row0 + 16 * n_blk_c + hi + 4 * reg) * ld];
		if(wave < 4 && !(wave & 1)) { // waves 0 and 2 store the split blocks (3,2) and (3,3)
			#pragma unroll
			for(int reg = 0; reg < 4; ++ reg)
				cv_h[reg] = M[size_t(row0 + 48 + lo) + size_t(row0 + 16 * (2 + (wave >> 1)) + hi + 4 * reg) * ld];
		}
	}
	__syncthreads();
	v4f64 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
	if(b_outer) {
		tri_block_product<0>(Ps, Qs, rb, lo, hi, acc[0]);
		tri_block_product<3>(Ps, Qs, rb, lo, hi, acc[1]);
	} else {
		tri_block_product<1>(Ps, Qs, rb, lo, hi, acc[0]);
		tri_block_product<2>(Ps, Qs, rb, lo, hi, acc[1]);
	}
	// every thread has read its operands out of LDS; the tile in global memory can be overwritten
	#pragma unroll
	for(int reg = 0; reg < 4; ++ reg) {
		M[size_t(row0 + 16 * rb + lo) + size_t(col0 + 16 * cb0 + hi + 4 * reg) * ld] = acc[0][reg];
		M[size_t(row0 + 16 * rb + lo) + size_t(col0 + 16 * cb1 + hi + 4 * reg) * ld] = acc[1][reg];
	}
	if(!b_diag)
		return;
	__syncthreads(); // every wave is done with Ps
	#pragma unroll
	for(int reg = 0; reg < 4; ++ reg) {
		Ps[lds_at(16 * cb0 + hi + 4 * reg, 16 * rb + lo)] = acc[0][reg]; // L(i,j) as an operand: [k = column][row]
		Ps[lds_at(16 * cb1 + hi + 4 * reg, 16 * rb + lo)] = acc[1][reg];
	}
	__syncthreads();
	v4f64 upd = {0, 0, 0, 0}, upd_h = {0, 0, 0, 0};
	sym_block_product(Ps, n_blk_r, n_blk_c, 0, NB / 4, lo, hi, upd);
	if(wave < 4) {
		const int n_half = wave & 1; // block (3, 2 + (wave >> 1)), K half wave & 1
		sym_block_product(Ps, 3, 2 + (wave >> 1), n_half * (NB / 8), (n_half + 1) * (NB / 8), lo, hi, upd_h);
		if(n_half) {
			#pragma unroll
			for(int reg = 0; reg < 4; ++ reg)
				Qs[(wave >> 1) * 256 + reg * 64 + lane] = upd_h[reg]; // (Qs is free: the inverse was last read before the barriers above)
		}
	}
	#pragma unroll
	for(int reg = 0; reg < 4; ++ reg)
		M[size_t(row0 + 16 * n_blk_r + lo) + size_t(row0 + 16 * n_blk_c + hi + 4 * reg) * ld] = cv[reg] - upd[reg];
	__syncthreads();
	if(wave < 4 && !(wave & 1)) {
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg)
			M[size_t(row0 + 48 + lo) + size_t(row0 + 16 * (2 + (wave >> 1)) + hi + 4 * reg) * ld] =
				cv_h[reg] - (upd_h[reg] + Qs[(wave >> 1) * 256 + reg * 64 + lane]);
	}
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

// DPP row broadcast of a double: every lane gets the value of lane N of its own row of 16 lanes (gfx90a+ row_newbcast,
// the one DPP control the 64-bit ALU takes).  The fused form, acc += bcast_N(src) * mul in one v_fmac_f64_dpp, is not
// something the compiler forms by itself (it keeps a v_mov_b64_dpp + v_fmac_f64), hence the inline assembly; the hazard
// recogniser does not look inside inline assembly, so the two wait states a DPP read needs after a VALU write of the same
// register are the caller's to provide: b_guard puts an s_nop 1 in front (callers set it where src may be fresh).
template <int N>
__device__ __forceinline__ double dpp_row_bcast(double v)
{
	double r;
	asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
	return r;
}
template <int N, bool b_guard>
__device__ __forceinline__ void dpp_fmac_row_bcast(double &r_acc, double src, double mul)
{
	if constexpr(b_guard)
		asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(r_acc) : "v"(src), "v"(mul), "n"(N));
	else
		asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(r_acc) : "v"(src), "v"(mul), "n"(N));
}

// One column step of the 64 x 16 panel of a diagonal tile, and the ones after it (K = 0 .. 15, unrolled by recursion: the
// broadcast lanes are immediates).  a[] is the lane's own row of the panel (lane = row of the tile), d[] the row
// (lane & 15) of the panel's 16 x 16 diagonal block, which every row of 16 lanes keeps a copy of and updates along: that
// is what lets the multipliers a(c,k), c > k, come by DPP from lane c of the lane's own row instead of through scalar
// registers (two v_readlane_b32 and an FMA per trailing column on a wave that issues an instruction every ~5 cycles was
// the cost of a step; now it is two v_fmac_f64_dpp, one for a[], one for the copy).  The update is the L D L^T form,
// a(r,c) -= a(r,k) a(c,k) / d_k; the pivots go to r_mine (lane & 15 == k keeps d_k) for the scaling after the panel.
template <int K, int C>
__device__ __forceinline__ void potrf_panel_update(double (&a)[16], double (&d)[16], double ntd, double nta)
{
	if constexpr(C < 16) {
		dpp_fmac_row_bcast<C, false>(d[C], d[K], ntd);
		dpp_fmac_row_bcast<C, false>(a[C], d[K], nta);
		potrf_panel_update<K, C + 1>(a, d, ntd, nta);
	}
}
// 1 / max(d_k, tiny) for the pivot in lane K of every row of 16 lanes; the raw pivot goes to r_mine in that lane.  No
// compare-and-select here: a VALU compare into scalar registers and a select from them cost ~70 cycles on this chain
// (tools/bench_potrf.hip), the v_max_f64 costs 8.  A pivot that is not positive (a matrix that is not positive definite;
// or the right-hand side row, whose "pivot" is 1 - y^T y) thus becomes 1e-300 instead of the 1 it used to become:
// finite garbage in a row nobody reads either way, and the flag is raised from r_mine after the panel.
template <int K>
__device__ __forceinline__ double potrf_panel_pivot(const double (&d)[16], double &r_mine, int g)
{
	const double piv_raw = dpp_row_bcast<K>(d[K]);
	r_mine = (g == K)? piv_raw : r_mine;
	const double piv = __builtin_fmax(piv_raw, 1e-300);
	double rw = __builtin_amdgcn_rcp(piv);
	rw = __builtin_fma(__builtin_fma(-piv, rw, 1.0), rw, rw);
	rw = __builtin_fma(__builtin_fma(-piv, rw, 1.0), rw, rw);
	return rw;
}
template <int K>
__device__ __forceinline__ void potrf_panel_steps(double (&a)[16], double (&d)[16], double rw, double &r_mine, int g)
{
	const double ntd = -(d[K] * rw), nta = -(a[K] * rw);
	if constexpr(K < 15) {
		dpp_fmac_row_bcast<K + 1, true>(d[K + 1], d[K], ntd);
		const double rw_next = potrf_panel_pivot<K + 1>(d, r_mine, g); // the chain goes on from the next column; the others follow beside it
		dpp_fmac_row_bcast<K + 1, false>(a[K + 1], d[K], nta);
		potrf_panel_update<K, K + 2>(a, d, ntd, nta);
		potrf_panel_steps<K + 1>(a, d, rw_next, r_mine, g);
	}
}
template <int K>
__device__ __forceinline__ void potrf_panel_scale(double (&a)[16], double rs)
{
	a[K] *= dpp_row_bcast<K>(rs); // L(r, c0 + k) = a(r,k) / sqrt(d_k)
	if constexpr(K < 15)
		potrf_panel_scale<K + 1>(a, rs);
}

// ---- round 6: the panel of a diagonal tile as a pipeline of three waves ----
// The sixteen column steps of a 64 x 16 panel used to be walked by ONE wave that updated, at every step, the trailing columns
// of its own 64 rows (a[]) and of the copy of the 16 x 16 diagonal block every row of 16 lanes keeps (d[]): 2 (15 - k)
// v_fmac_f64_dpp at ~11.6 clocks each with one wave on the SIMD (tools/micro/mfma_f64_peak.hip) around a dependent chain of
// ~110 -- 244 clocks a step by the in-kernel stamps, 3 900 a panel, four panels a tile.  Only the diagonal block is on the
// chain.  Now wave 0 walks the chain on the diagonal block alone and publishes, step by step, the multipliers of the step
// (column k of the unit lower factor, negated: m_k(c) = -D(c,k) / d_k) and a step counter in LDS; wave 1 holds the rows below
// the block (a lane a row) and wave 2 the columns of the block's inverse (a lane a column), and both apply step k as soon as
// the counter says it is there: a(r,c) += a(r,k) m_k(c), y_j(r) += m_k(r) y_j(k) -- the same multipliers serve both (the
// inverse of the unit lower factor by forward substitution; the rows are scaled by 1 / sqrt(d) at the end, like the panel).
// LDS operations of one wave are performed in order, and the accesses are volatile (the compiler keeps their order): the
// counter is written behind the multipliers and read in front of them, no barrier inside the sixteen steps.  Wave 3 meanwhile
// applies the previous panel to the tiles further right, as waves 1 - 3 used to.
// (LDS pointers with their address space spelled out: a volatile access through a generic pointer is a flat instruction with
// the system-coherence bits set and a wait behind it -- 2 000 clocks a step instead of 130)
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
template <int K, int C>
__device__ __forceinline__ void potrf_chain_update(double (&d)[16], double ntd)
{
	if constexpr(C < 16) {
		dpp_fmac_row_bcast<C, false>(d[C], d[K], ntd);
		potrf_chain_update<K, C + 1>(d, ntd);
	}
}
// potrf_panel_pivot for a pivot that sits in lane K of every row of 16 lanes in `v` (other lanes: anything)
template <int K>
__device__ __forceinline__ double potrf_chain_pivot(double v, lds_f64 *s_piv)
{
	const double piv_raw = dpp_row_bcast<K>(v);
	s_piv[K] = piv_raw; // (kept for the scaling behind the chain: lane g reads pivot g back -- a select per step was sixteen masks in scalar registers and two instructions a step on the chain)
	double piv;
	asm volatile("v_max_f64 %0, %1, %2" : "=v"(piv) : "v"(piv_raw), "v"(1e-300)); // (one instruction: __builtin_fmax puts a canonicalizing v_max in front)
	double rw = __builtin_amdgcn_rcp(piv);
	rw = __builtin_fma(__builtin_fma(-piv, rw, 1.0), rw, rw);
	rw = __builtin_fma(__builtin_fma(-piv, rw, 1.0), rw, rw);
	return rw;
}
template <int K>
__device__ __forceinline__ void potrf_chain_steps(double (&d)[16], double rw, lds_f64 *s_piv, int g,
	volatile lds_f64 *s_mult, volatile lds_i32 *s_step)
{
	const double ntd = -(d[K] * rw); // lane g: -D(g, k) / d_k
	if constexpr(K < 15) {
		s_mult[K * 16 + g] = ntd; // published at once, by every row of 16 lanes (the same value to the same place: no mask to set up on the chain)
		*s_step = K + 1;
		// The next pivot is D(K+1,K+1) - D(K+1,K)^2 / d_K: lane K + 1 has all three in its own registers -- a plain FMA there (and
		// nonsense in the other lanes, not used) and ONE broadcast, instead of the broadcast update of column K + 1 followed by
		// the broadcast of its diagonal entry: a dependent DPP operation is 16 clocks plus its wait states, and the chain is
		// made of nothing but latencies (the column's update itself follows below, off the chain)
		const double t_diag = __builtin_fma(d[K], ntd, d[K + 1]);
		const double rw_next = potrf_chain_pivot<K + 1>(t_diag, s_piv);
		dpp_fmac_row_bcast<K + 1, true>(d[K + 1], d[K], ntd);
		potrf_chain_update<K, K + 2>(d, ntd);
		potrf_chain_steps<K + 1>(d, rw_next, s_piv, g, s_mult, s_step);
	} else
		*s_step = 16; // (the last column has nothing below it: the followers only need to know the chain is through)
}
template <int K>
__device__ __forceinline__ void potrf_chain_scale(double (&d)[16], double rs)
{
	d[K] *= dpp_row_bcast<K>(rs); // L(c0 + g, c0 + k) = D(g,k) / sqrt(d_k)
	if constexpr(K < 15)
		potrf_chain_scale<K + 1>(d, rs);
}
__device__ __forceinline__ void potrf_wait_step(volatile lds_i32 *s_step, int n_step)
{
	while(*s_step < n_step)
		__builtin_amdgcn_s_sleep(1);
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // (what is read next is read after the counter)
}
// The followers.  A follower is bound by what it issues -- 15 - K FMAs a step at ~10 clocks each with one wave on its SIMD,
// and the reads of the step's multipliers -- so the trip to LDS must not add to that: the multipliers of step K + 1 (and the
// counter, read FIRST: LDS serves a wave's requests in order, so a counter that says "there" vouches for what was read behind
// it) are requested before the arithmetic of step K; only if the counter was not there yet does the follower wait and ask again.
template <int K>
__device__ __forceinline__ void potrf_fetch_step(const lds_f64 *s_mult, volatile lds_i32 *s_step, int &r_n_flag, double (&m)[16])
{
	r_n_flag = *s_step;
	asm volatile("" ::: "memory"); // (the compiler keeps the reads below behind the counter's)
	#pragma unroll
	for(int c = K + 1; c < 16; ++ c)
		m[c] = s_mult[K * 16 + c];
	asm volatile("" ::: "memory");
}
// step K and the ones behind it; b_rows: the rows below the block (v = a row's entries of the panel: v[c] += v[K] m_K(c)),
// else the columns of the inverse (v = a column: v[r] += m_K(r) v[K]) -- the same arithmetic on different data
template <int K>
__device__ __forceinline__ void potrf_follow(double (&v)[16], double (&m)[16], int n_flag, const lds_f64 *s_mult, volatile lds_i32 *s_step)
{
	if constexpr(K < 15) {
		if(n_flag < K + 1) { // (wave-uniform) asked too early: wait for the step, then ask again
			potrf_wait_step(s_step, K + 1);
			potrf_fetch_step<K>(s_mult, s_step, n_flag, m);
		}
		double m_next[16];
		int n_flag_next = 16;
		if constexpr(K + 1 < 15)
			potrf_fetch_step<K + 1>(s_mult, s_step, n_flag_next, m_next);
		#pragma unroll
		for(int c = K + 1; c < 16; ++ c)
			v[c] = __builtin_fma(v[K], m[c], v[c]);
		// (the step's arithmetic stays here: left to itself the compiler sinks the FMAs of several steps below the last wait,
		// parks their operands in accumulator registers, and the followers finish 2 000 clocks behind the chain)
		asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
			"+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
		potrf_follow<K + 1>(v, m_next, n_flag_next, s_mult, s_step);
	}
}

#ifdef POTRF_STAMPS
__device__ long long g_potrf_stamps[64];
#endif
enum { PL = NB + 1, TL = NB / 2 + 1 };

// D[m][n] += sum_{k < K} A[m][k] B[k][n] for one 16 x 16 tile; A[m][k] = p_A[m * a_m + k * a_k], B[k][n] =
// p_B[k * b_k + n * b_n]; lane l holds D[(l >> 4) + 4 reg][l & 15] in acc[reg]
__device__ __forceinline__ v4f64 mfma_tile16(const double *p_A, int a_m, int a_k, const double *p_B, int b_k, int b_n,
	int K, int lane, v4f64 acc)
{
	const int lo = lane & 15, hi = lane >> 4;
	// (round 6: the operands of all K steps are requested before the first product -- K is 16 or 32 at every call site -- instead
	// of an LDS round trip in front of each of the four or eight dependent matrix instructions: 870 -> ~500 clocks for K = 16)
	double a[8], b[8];
	#pragma unroll
	for(int i = 0; i < 8; ++ i) {
		if(4 * i < K) {
			a[i] = p_A[lo * a_m + (4 * i + hi) * a_k];
			b[i] = p_B[(4 * i + hi) * b_k + lo * b_n];
		}
	}
	#pragma unroll
	for(int i = 0; i < 8; ++ i) {
		if(4 * i < K)
			acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc, 0, 0, 0);
	}
	return acc;
}

// trailing update inside the diagonal tile: S(ti, tj) -= P(ti) P(tj)^T with the 16-column panel at column c0;
// computed transposed (m = column of S, n = row) so that the read-modify-write runs along rows
__device__ __forceinline__ void potrf_update_tile(double *s_L, int c0, int ti, int tj, int lane)
{
	const int lo = lane & 15, hi = lane >> 4;
	// (the target's entries are requested with the operands and go in as the accumulator: one trip to LDS in front of the
	// four products instead of one before and one behind them)
	v4f64 acc;
	#pragma unroll
	for(int reg = 0; reg < 4; ++ reg)
		acc[reg] = -s_L[(16 * tj + hi + 4 * reg) * PL + 16 * ti + lo];
	acc = mfma_tile16(s_L + c0 * PL + 16 * tj, 1, PL, s_L + c0 * PL + 16 * ti, PL, 1, 16, lane, acc);
	#pragma unroll
	for(int reg = 0; reg < 4; ++ reg)
		s_L[(16 * tj + hi + 4 * reg) * PL + 16 * ti + lo] = -acc[reg];
}

// inverse of the 16 x 16 lower triangular diagonal block at b0: lane c < 16 of the calling wave solves column c.
// Right-looking: as soon as x[r] is known every later row takes its share, sum[r2] -= L(r2,r) x[r] -- fifteen independent
// FMAs -- so the dependent chain is 16 x (multiply, one FMA) instead of the 120 FMAs in a row of the row-by-row form
// (2 800 -> 1 000 cycles by the in-kernel stamps; the last block's inverse is on the tile's critical path)
__device__ __forceinline__ void potrf_invert_block16(const double *s_L, const double *s_rd, double *s_X, int b0, int lane)
{
	if(lane >= 16)
		return;
	const int c = lane;
	double sum[16];
	#pragma unroll
	for(int r = 0; r < 16; ++ r)
		sum[r] = (r == c)? 1.0 : 0.0;
	// (nothing is stored before the last load: the 16 + 120 LDS reads -- the same addresses in every lane -- are requested as
	// far ahead as the registers allow instead of a round trip per step behind a store they might alias)
	double rd[16], x[16];
	#pragma unroll
	for(int r = 0; r < 16; ++ r)
		rd[r] = s_rd[b0 + r];
	#pragma unroll
	for(int r = 0; r < 16; ++ r) {
		x[r] = sum[r] * rd[r];
		#pragma unroll
		for(int r2 = r + 1; r2 < 16; ++ r2)
			sum[r2] -= s_L[(b0 + r) * PL + b0 + r2] * x[r];
	}
	#pragma unroll
	for(int r = 0; r < 16; ++ r)
		s_X[(b0 + r) * PL + b0 + c] = x[r];
}

enum { POTRF_LDS_DOUBLES = 2 * NB * PL + (NB / 2) * TL + NB + 16 * 16 + 2 }; // (+ the multipliers of a panel's steps and the step counter: potrf_chain_steps)

template <bool b_chol, bool b_inverse, bool b_pipeline = true>
__device__ __forceinline__ void potrf_diag_body(double *M, int ld, int kb, int n, double *invL, int *p_flag, double *s_buf)
{
	double *s_L = s_buf;                  // the tile, [col][row]
	double *s_X = s_L + NB * PL;          // its inverse, [row][col]
	double *s_T = s_X + NB * PL;          // products L21 X11, [row][col]
	double *s_rd = s_T + (NB / 2) * TL;   // reciprocals of the diagonal of L
	lds_f64 *s_mult = (lds_f64*)(s_rd + NB);            // the multipliers of the panel's steps, [step][row of the diagonal block]
	volatile lds_i32 *s_step = (volatile lds_i32*)(s_rd + NB + 16 * 16); // steps of the panel published so far
	lds_f64 *s_rd_lds = (lds_f64*)s_rd;

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
#ifdef POTRF_STAMPS // tools/bench_potrf.hip: where the time of a tile goes (wave 0, lane 0)
#define POTRF_STAMP(i) do { if(t == 0) { g_potrf_stamps[2 * (i)] = clock64(); g_potrf_stamps[2 * (i) + 1] = wall_clock64(); } } while(0)
#define POTRF_STAMP_WAVE(i, w) do { if(t == 64 * (w)) { g_potrf_stamps[2 * (i)] = clock64(); g_potrf_stamps[2 * (i) + 1] = wall_clock64(); } } while(0)
#else
#define POTRF_STAMP(i) do {} while(0)
#define POTRF_STAMP_WAVE(i, w) do {} while(0)
#endif
	POTRF_STAMP(0);
	if(b_chol && b_pipeline) {
		for(int J = 0; J < NB / 16; ++ J) {
			const int c0 = 16 * J, n_below = NB - c0 - 16; // rows below the diagonal block
			if(t == 0)
				*s_step = 0;
			__syncthreads(); // (the counter is down before anybody looks at it; the tile updates of the step before are in)
			POTRF_STAMP(1 + 3 * J);
			if(wave == 0) {
				// the chain: the diagonal block alone (every row of 16 lanes a copy, as before: the DPP broadcasts stay inside a row)
				const int g = lane & 15;
				double d[16];
				#pragma unroll
				for(int c = 0; c < 16; ++ c)
					d[c] = s_L[(c0 + c) * PL + c0 + g];
				const double rw_first = potrf_chain_pivot<0>(d[0], s_rd_lds + c0); // (the raw pivots wait in s_rd for the scaling)
				potrf_chain_steps<0>(d, rw_first, s_rd_lds + c0, g, (volatile lds_f64*)s_mult, s_step);
				if(J == 0) POTRF_STAMP_WAVE(16, 0);
				const double mine_raw = s_rd_lds[c0 + g];
				const double mine = __builtin_fmax(mine_raw, 1e-300);
				double rs = __builtin_amdgcn_rsq(mine);
				const double h = 0.5 * mine;
				rs = rs * (1.5 - h * rs * rs);
				rs = rs * (1.5 - h * rs * rs);
				*(volatile lds_f64*)(s_rd_lds + c0 + g) = rs; // (every row of 16 lanes the same values to the same places: no mask)
				*s_step = 17; // (behind the reciprocal square roots: what the followers scale with)
				asm volatile("" ::: "memory"); // (published here, not below the scaling: the followers are waiting for it)
				b_bad = b_bad || (!(mine_raw > 0) && c0 + g < n - o); // (past n rides the right-hand side row)
				{
					// the block's own columns scaled: the reciprocal square roots come back from LDS, all sixteen in one trip (by DPP
					// it was sixteen broadcasts with their wait states: 1 150 clocks for this wave to leave the panel)
					double rs_c[16];
					#pragma unroll
					for(int c = 0; c < 16; ++ c)
						rs_c[c] = s_rd_lds[c0 + c];
					if(lane < 16) {
						#pragma unroll
						for(int c = 0; c < 16; ++ c)
							s_L[(c0 + c) * PL + c0 + g] = (g >= c)? d[c] * rs_c[c] : 0.0;
					}
				}
				if(J == 0) POTRF_STAMP_WAVE(17, 0);
			} else if(wave == 1) {
				if(lane < n_below) { // the rows below the block, a lane a row
					const int r = c0 + 16 + lane;
					double a[16];
					#pragma unroll
					for(int c = 0; c < 16; ++ c)
						a[c] = s_L[(c0 + c) * PL + r];
					{
						double m[16];
						int n_flag;
						potrf_fetch_step<0>(s_mult, s_step, n_flag, m);
						potrf_follow<0>(a, m, n_flag, s_mult, s_step);
					}
					if(J == 0) POTRF_STAMP_WAVE(18, 1);
					potrf_wait_step(s_step, 17);
					double rs_c[16]; // (all sixteen requested before the first store: behind a store the compiler waits for each)
					#pragma unroll
					for(int c = 0; c < 16; ++ c)
						rs_c[c] = s_rd_lds[c0 + c];
					#pragma unroll
					for(int c = 0; c < 16; ++ c)
						s_L[(c0 + c) * PL + r] = a[c] * rs_c[c];
					if(J == 0) POTRF_STAMP_WAVE(19, 1);
				}
			} else if(wave == 2) {
				if(b_inverse && lane < 16) { // the block's inverse, a lane a column: forward substitution with the same multipliers
					double y[16];
					#pragma unroll
					for(int r = 0; r < 16; ++ r)
						y[r] = (r == lane)? 1.0 : 0.0;
					{
						double m[16];
						int n_flag;
						potrf_fetch_step<0>(s_mult, s_step, n_flag, m);
						potrf_follow<0>(y, m, n_flag, s_mult, s_step);
					}
					if(J == 0) POTRF_STAMP_WAVE(20, 2);
					potrf_wait_step(s_step, 17);
					double rs_r[16];
					#pragma unroll
					for(int r = 0; r < 16; ++ r)
						rs_r[r] = s_rd_lds[c0 + r];
					#pragma unroll
					for(int r = 0; r < 16; ++ r)
						s_X[(c0 + r) * PL + c0 + lane] = y[r] * rs_r[r];
					if(J == 0) POTRF_STAMP_WAVE(21, 2);
				}
			} else if(J > 0) {
				// wave 3: the panel before this one applied to the tiles right of this panel's columns
				const int c_prev = c0 - 16;
				for(int tj = J + 1; tj < NB / 16; ++ tj) {
					for(int ti = tj; ti < NB / 16; ++ ti)
						potrf_update_tile(s_L, c_prev, ti, tj, lane);
				}
			}
			__syncthreads();
			POTRF_STAMP(2 + 3 * J);
			// the next panel's columns: tiles (ti, J + 1), ti = J + 1 .. 3, one per wave
			if(J + 1 + wave < NB / 16)
				potrf_update_tile(s_L, c0, J + 1 + wave, J + 1, lane);
			POTRF_STAMP(3 + 3 * J);
		}
		__syncthreads();
		POTRF_STAMP(13);
	} else if(b_chol) {
		for(int J = 0; J < NB / 16; ++ J) {
			const int c0 = 16 * J;
			if(wave == 0) {
				// (the sixteen column steps are one dependent chain -- pivot, reciprocal, update of the next column, next
				// pivot -- and one wave walks it: see potrf_panel_steps for what a step costs and why it is laid out so.
				// A pivot that is not positive raises the flag if it is inside the matrix: see potrf_panel_pivot)
				const int r = lane, g = lane & 15;
				double a[16], d[16];
				#pragma unroll
				for(int c = 0; c < 16; ++ c) {
					a[c] = s_L[(c0 + c) * PL + r];
					d[c] = s_L[(c0 + c) * PL + c0 + g];
				}
				double mine_raw = 1.0;
				const double rw_first = potrf_panel_pivot<0>(d, mine_raw, g);
				potrf_panel_steps<0>(a, d, rw_first, mine_raw, g);
				b_bad = b_bad || (!(mine_raw > 0) && c0 + g < n - o); // (past n rides the right-hand side row)
				const double mine = __builtin_fmax(mine_raw, 1e-300);
				// the reciprocal square roots of the sixteen pivots, one per lane of a row, beside the chain rather than on it
				double rs = __builtin_amdgcn_rsq(mine);
				const double h = 0.5 * mine;
				rs = rs * (1.5 - h * rs * rs);
				rs = rs * (1.5 - h * rs * rs);
				potrf_panel_scale<0>(a, rs);
				if(lane < 16)
					s_rd[c0 + lane] = rs;
				#pragma unroll
				for(int c = 0; c < 16; ++ c)
					s_L[(c0 + c) * PL + r] = (r >= c0 + c)? a[c] : 0.0;
			}
			POTRF_STAMP(1 + 3 * J);
			__syncthreads();
			POTRF_STAMP(2 + 3 * J);
			// the next panel's columns first: tiles (ti, J + 1), ti = J + 1 .. 3, one per wave
			if(J + 1 + wave < NB / 16)
				potrf_update_tile(s_L, c0, J + 1 + wave, J + 1, lane);
			__syncthreads();
			POTRF_STAMP(3 + 3 * J);
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
		POTRF_STAMP(13);
	} else if(b_inverse) {
		if(wave < NB / 16)
			potrf_invert_block16(s_L, s_rd, s_X, 16 * wave, lane);
		__syncthreads();
	}
	if(__any(b_bad) && lane == 0)
		atomicOr(p_flag, 1);
	#pragma unroll
	for(int i = 0; i < NB * NB / 256; ++ i) {
		const int e = t + 256 * i, r = e & 63, c = e >> 6;
		if(c <= r)
			M[size_t(o + r) + size_t(o + c) * ld] = s_L[c * PL + r];
	}
	POTRF_STAMP(14);
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

