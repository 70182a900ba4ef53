// dense_tiles.hip -- tile-sparse, level-scheduled factorization of the dense top (see dense_chol.h), and the
// identity diagonal of its alignment gaps.  Kept in its own translation unit: the register allocation of the
// kernels in dense_chol.hip was seen to change (260 instead of 204 registers for the diagonal-tile kernel, one
// workgroup per CU instead of two) when these kernels were compiled next to them.
#include <hip/hip_runtime.h>
#include "dense_chol.h"
#include "plan.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace slampp {

#include "dense_device.inl"

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

// ---- tile-sparse, level-scheduled variant (see dense_chol.h) ----
// target tile (ti, tj) -= sum over its source tile columns kt of L(ti, kt) L(tj, kt)^T
__device__ __forceinline__ void tile_update_body(double *M, int ld, const int4 t_tgt, const int *__restrict__ p_sources, double *Ps, double *Qs)
{
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

// the diagonal tiles of a level; behind them ride the updates of the level before that nothing in this level waits for
// (everything but the diagonal tiles of this level's columns): the launch every level has to wait for keeps one CU
// busy per tile, the riders give the rest of the chip its share of the level's work
__global__ void __launch_bounds__(256)
tile_potrf_kernel(double *M, int ld, int n, double *p_invdiag, int *p_flag, const int *__restrict__ p_tiles, int n_potrf,
	const int4 *__restrict__ p_riders, const int *__restrict__ p_sources)
{
	__shared__ double s_buf[(int(POTRF_LDS_DOUBLES) > 2 * NB * NB)? int(POTRF_LDS_DOUBLES) : 2 * NB * NB];
	if(int(blockIdx.x) >= n_potrf) {
		tile_update_body(M, ld, p_riders[int(blockIdx.x) - n_potrf], p_sources, s_buf, s_buf + NB * NB);
		return;
	}
	const int kb = p_tiles[blockIdx.x];
	potrf_diag_body<true, true>(M, ld, kb, n, p_invdiag + size_t(kb) * NB * NB, p_flag, s_buf);
}

// L(i,j) = A(i,j) inv(L_jj)^T; with t_pair.z set the workgroup also applies its tile to the diagonal tile of its row,
// A(i,i) -= L(i,j) L(i,j)^T -- the one update the next level's diagonal tile is waiting for, where this tile is its
// only source in this level (a chain of tile columns: the top of every separator tree)
__global__ void __launch_bounds__(512)
tile_trsm_kernel(double *M, int ld, const double *p_invdiag, const int4 *__restrict__ p_pairs)
{
	__shared__ double s_buf[2 * NB * NB];
	const int4 t_pair = p_pairs[blockIdx.x];
	trsm_tile_body8(M, ld, t_pair.x * NB, t_pair.y * NB, p_invdiag + size_t(t_pair.y) * NB * NB, t_pair.z != 0, s_buf, s_buf + NB * NB);
}

__global__ void __launch_bounds__(256)
tile_update_kernel(double *M, int ld, const int4 *__restrict__ p_targets, const int *__restrict__ p_sources)
{
	__shared__ double Ps[NB * NB];
	__shared__ double Qs[NB * NB];
	tile_update_body(M, ld, p_targets[blockIdx.x], p_sources, Ps, Qs);
}

void CTileSchedule::Free()
{
	if(d_potrf) (void)hipFree(d_potrf);
	if(d_trsm) (void)hipFree(d_trsm);
	if(d_tgt) (void)hipFree(d_tgt);
	if(d_src) (void)hipFree(d_src);
	if(d_back_diag) (void)hipFree(d_back_diag);
	if(d_back_carry) (void)hipFree(d_back_carry);
	if(d_riders) (void)hipFree(d_riders);
	d_potrf = 0; d_trsm = 0; d_tgt = 0; d_src = 0; d_back_diag = 0; d_back_carry = 0; d_riders = 0;
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
	std::vector<int4> trsm; // (row tile, column tile, also updates the diagonal tile of its row, -)
	std::vector<int4> tgt;  // per level: first the urgent targets (diagonal tiles of the next level's columns), then the others
	std::vector<int> src;
	level_potrf_ptr.assign(1, 0);
	level_trsm_ptr.assign(1, 0);
	level_tgt_ptr.assign(1, 0);
	level_urgent_end.clear();
	std::vector<int> tgt_of(size_t(T) * T, -1); // per level: index (in `found`) of the target record of a tile
	// updates nothing in the next level waits for ("riders": extra workgroups of the diagonal-tile launches): per target tile
	// its deadline -- the last diagonal launch it may ride in -- and its sources (level they become available after, column)
	struct TPending { int i1, i2, n_deadline; size_t n_done; std::vector<int2> sources; };
	std::vector<TPending> pending;
	std::vector<int> pending_of(size_t(T) * T, -1);
	for(int l = 0; l < n_levels; ++ l) {
		std::vector<int2> found;                 // targets of this level
		std::vector<std::vector<int> > sources;  // per target
		const size_t n_trsm0 = trsm.size();
		for(int j = 0; j < T; ++ j) {
			if(height[j] != l)
				continue;
			potrf.push_back(j);
			for(int i = j + 1; i < T; ++ i) {
				if(nz[size_t(i) + size_t(j) * T])
					trsm.push_back(int4{i, j, 0, 0});
			}
			for(int i2 = j + 1; i2 < T; ++ i2) {
				if(!nz[size_t(i2) + size_t(j) * T])
					continue;
				for(int i1 = i2; i1 < T; ++ i1) {
					if(!nz[size_t(i1) + size_t(j) * T])
						continue;
					int &r_idx = tgt_of[size_t(i1) + size_t(i2) * T];
					if(r_idx < 0) {
						r_idx = int(found.size());
						found.push_back(int2{i1, i2});
						sources.push_back(std::vector<int>());
					}
					sources[r_idx].push_back(j);
				}
			}
		}
		// what the next level's diagonal tiles wait for: with a single source, the panel solve of that source does it
		// on its way; with several, an update launch of its own; everything else rides in the next level's first launch
		std::vector<int> urgent, deferred;
		for(size_t k = 0; k < found.size(); ++ k) {
			const int i1 = found[k].x, i2 = found[k].y;
			tgt_of[size_t(i1) + size_t(i2) * T] = -1; // (for the next level)
			const bool b_next_diag = i1 == i2 && height[i1] == l + 1;
			if(b_next_diag && sources[k].size() == 1) {
				bool b_found = false;
				for(size_t p = n_trsm0; p < trsm.size() && !b_found; ++ p) {
					if(trsm[p].x == i1 && trsm[p].y == sources[k][0]) {
						trsm[p].z = 1;
						b_found = true;
					}
				}
				if(!b_found)
					return false; // (cannot happen: the source tile is a panel tile of this level)
			} else
				(b_next_diag? urgent : deferred).push_back(int(k));
		}
		for(size_t q = 0; q < urgent.size(); ++ q) {
			const int k = urgent[q];
			int4 t = {found[k].x, found[k].y, int(src.size()), 0};
			src.insert(src.end(), sources[k].begin(), sources[k].end());
			t.w = int(src.size());
			tgt.push_back(t);
		}
		level_urgent_end.push_back(int(tgt.size()));
		for(size_t q = 0; q < deferred.size(); ++ q) { // not wanted by the next level: to the pool of riders, scheduled below
			const int k = deferred[q], i1 = found[k].x, i2 = found[k].y;
			int &r_id = pending_of[size_t(i1) + size_t(i2) * T];
			if(r_id < 0) {
				r_id = int(pending.size());
				// a sub-diagonal tile is read by the panel solves of its column's level (launched after that level's diagonal
				// tiles: it may still ride with those), a diagonal tile by the diagonal launch of its level
				pending.push_back(TPending{i1, i2, (i1 == i2)? height[i2] - 1 : height[i2], 0, std::vector<int2>()});
			}
			for(size_t e = 0; e < sources[k].size(); ++ e)
				pending[r_id].sources.push_back(int2{l, sources[k][e]});
		}
		level_potrf_ptr.push_back(int(potrf.size()));
		level_trsm_ptr.push_back(int(trsm.size()));
		level_tgt_ptr.push_back(int(tgt.size()));
	}
	// The riders, launch by launch.  Until round 5 every update rode in the very next diagonal launch: at the Venice-like C4's
	// reduced system the first launches carried 300 - 600 jobs of several source tiles each and lasted 25 - 50 us where the
	// diagonal tile needs 13 (150 us of a 0.79 ms solve), while from the tenth level on the chip idled beside one tile.  Now an
	// update may ride in any diagonal launch between the level that produced its sources and its deadline: every launch
	// takes what is due, then the most urgent of the rest up to a budget of source tiles, at most TILE_RIDER_CHUNK sources
	// of a target at a time (a job lasts as long as its list) -- one job per target and launch, sources in level order: the
	// sums keep a fixed order.
	const int TILE_RIDER_CHUNK = dev_knob("SLAMPP_HIP_DEV_RIDER_CHUNK", 4), TILE_RIDER_BUDGET = dev_knob("SLAMPP_HIP_DEV_RIDER_BUDGET", 1024);
	std::vector<int4> riders;
	rider_ptr.assign(2, 0); // (no riders in the first level's launch)
	{
		std::vector<int> order(pending.size());
		for(size_t i = 0; i < order.size(); ++ i)
			order[i] = int(i);
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pending[a].n_deadline < pending[b].n_deadline; });
		for(int lv = 1; lv < n_levels; ++ lv) {
			int n_budget = TILE_RIDER_BUDGET;
			for(int n_pass = 0; n_pass < 2; ++ n_pass) { // what is due first, then by deadline
				for(size_t q = 0; q < order.size(); ++ q) {
					TPending &r_p = pending[order[q]];
					const bool b_due = r_p.n_deadline <= lv;
					if(b_due != !n_pass || r_p.n_done == r_p.sources.size() || r_p.n_done == size_t(-1))
						continue;
					size_t n_avail = 0; // sources of levels below this launch's, not applied yet
					while(r_p.n_done + n_avail < r_p.sources.size() && r_p.sources[r_p.n_done + n_avail].x < lv)
						++ n_avail;
					if(!n_avail)
						continue;
					if(!b_due) {
						if(n_budget <= 0)
							continue;
						n_avail = std::min<size_t>(n_avail, size_t(TILE_RIDER_CHUNK));
					}
					int4 t = {r_p.i1, r_p.i2, int(src.size()), 0};
					for(size_t e = 0; e < n_avail; ++ e)
						src.push_back(r_p.sources[r_p.n_done + e].y);
					t.w = int(src.size());
					riders.push_back(t);
					r_p.n_done += n_avail;
					n_budget -= int(n_avail);
				}
			}
			if(getenv("SLAMPP_HIP_PLAN_TIMING")) { // development aid: what rides where
				int n_src_here = 0;
				for(size_t i = size_t(rider_ptr.back()); i < riders.size(); ++ i)
					n_src_here += riders[i].w - riders[i].z;
				fprintf(stderr, "[tiles] level %2d: %3d diagonal tiles, %4d riders with %5d source tiles\n", lv,
					level_potrf_ptr[lv + 1] - level_potrf_ptr[lv], int(riders.size()) - rider_ptr.back(), n_src_here);
			}
			rider_ptr.push_back(int(riders.size()));
		}
		if(n_levels < 2)
			rider_ptr.assign(size_t(n_levels) + 1, 0);
		for(size_t i = 0; i < pending.size(); ++ i) {
			if(pending[i].n_done != pending[i].sources.size())
				return false; // (cannot happen: a source's level is below its target's deadline)
		}
	}
	// the backward substitution, top down.  The ancestors of a tile column (the rows of its nonzero tiles) all have
	// different heights (two of them are linked by fill), so a column gets at most one contribution per launch: from the
	// ancestor of height h + 1 by the launch of height h -- in the column's own workgroup if h is its own height, by a
	// carrying workgroup otherwise.  z_k starts as y_k, which lives in the last row of the factor: the first workgroup to
	// touch z_k takes it from there
	std::vector<int4> back_diag, back_carry;
	back_diag_ptr.assign(1, 0);
	back_carry_ptr.assign(1, 0);
	{
		std::vector<char> touched(T, 0);
		for(int h = n_max_height; h >= 0; -- h) {
			for(int k = 0; k < T; ++ k) {
				if(height[k] > h)
					continue;
				int n_up = -1; // the ancestor of height h + 1
				for(int j = k + 1; j < T && n_up < 0; ++ j) {
					if(nz[size_t(j) + size_t(k) * T] && height[j] == h + 1)
						n_up = j;
				}
				if(height[k] == h)
					back_diag.push_back(int4{k, n_up, !touched[k], 0});
				else if(n_up >= 0) {
					back_carry.push_back(int4{n_up, k, !touched[k], 0});
					touched[k] = 1;
				}
			}
			back_diag_ptr.push_back(int(back_diag.size()));
			back_carry_ptr.push_back(int(back_carry.size()));
		}
	}
	const size_t n_b4 = back_diag.size() * sizeof(int4), n_b5 = (back_carry.size() + 1) * sizeof(int4), n_b6 = (riders.size() + 1) * sizeof(int4);
	const size_t n_b0 = potrf.size() * sizeof(int), n_b1 = (trsm.size() + 1) * sizeof(int4),
		n_b2 = (tgt.size() + 1) * sizeof(int4), n_b3 = (src.size() + 1) * sizeof(int);
	if(hipMalloc((void**)&d_potrf, n_b0) != hipSuccess || hipMalloc((void**)&d_trsm, n_b1) != hipSuccess ||
	   hipMalloc((void**)&d_tgt, n_b2) != hipSuccess || hipMalloc((void**)&d_src, n_b3) != hipSuccess ||
	   hipMalloc((void**)&d_back_diag, n_b4) != hipSuccess || hipMalloc((void**)&d_back_carry, n_b5) != hipSuccess ||
	   hipMalloc((void**)&d_riders, n_b6) != hipSuccess) {
		(void)hipGetLastError();
		Free();
		return false;
	}
	bool b_ok = hipMemcpyAsync(d_potrf, potrf.data(), n_b0, hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!trsm.empty())
		b_ok = b_ok && hipMemcpyAsync(d_trsm, trsm.data(), trsm.size() * sizeof(int4), hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!tgt.empty())
		b_ok = b_ok && hipMemcpyAsync(d_tgt, tgt.data(), tgt.size() * sizeof(int4), hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!src.empty())
		b_ok = b_ok && hipMemcpyAsync(d_src, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!riders.empty())
		b_ok = b_ok && hipMemcpyAsync(d_riders, riders.data(), riders.size() * sizeof(int4), hipMemcpyHostToDevice, stream) == hipSuccess;
	b_ok = b_ok && hipMemcpyAsync(d_back_diag, back_diag.data(), n_b4, hipMemcpyHostToDevice, stream) == hipSuccess;
	if(!back_carry.empty())
		b_ok = b_ok && hipMemcpyAsync(d_back_carry, back_carry.data(), back_carry.size() * sizeof(int4), hipMemcpyHostToDevice, stream) == hipSuccess;
	b_ok = b_ok && hipStreamSynchronize(stream) == hipSuccess; // the host vectors live on this stack frame
	if(!b_ok) {
		(void)hipGetLastError();
		Free();
		return false;
	}
	n_bytes = n_b0 + n_b1 + n_b2 + n_b3 + n_b4 + n_b5 + n_b6;
	n_tiles = T;
	n_levels = n_max_height + 1;
	return true;
}

// zeroes the tiles of the schedule (the diagonal ones whole): what a step's assembly adds into; the other tiles are
// never written by anything that works from the schedule, so once zero they stay zero
__global__ void __launch_bounds__(256)
tile_zero_kernel(double *M, int ld, const int *__restrict__ p_potrf, int n_potrf, const int4 *__restrict__ p_trsm,
	const uint8_t *__restrict__ p_unit)
{
	int ti, tj;
	if(int(blockIdx.x) < n_potrf)
		ti = tj = p_potrf[blockIdx.x];
	else {
		const int4 t = p_trsm[int(blockIdx.x) - n_potrf];
		ti = t.x;
		tj = t.y;
	}
	const int r = (threadIdx.x & 31) * 2, c0 = threadIdx.x >> 5;
	#pragma unroll
	for(int i = 0; i < NB / 8; ++ i) {
		const int c = c0 + 8 * i;
		v2f64 v = {0, 0};
		if(ti == tj && p_unit && (c >> 1) == (r >> 1) && p_unit[tj * NB + c]) { // the identity of padding and gaps (own launches once)
			if(c & 1)
				v.y = 1.0;
			else
				v.x = 1.0;
		}
		*reinterpret_cast<v2f64*>(M + size_t(ti * NB + r) + size_t(tj * NB + c) * ld) = v;
	}
}

void tile_zero(const CTileSchedule &r_s, double *M, int n_pad, hipStream_t stream, const uint8_t *p_unit)
{
	const int n_potrf = r_s.level_potrf_ptr.back(), n_trsm = r_s.level_trsm_ptr.back();
	hipLaunchKernelGGL(tile_zero_kernel, dim3(n_potrf + n_trsm), dim3(256), 0, stream, M, n_pad, r_s.d_potrf, n_potrf, r_s.d_trsm, p_unit);
}

void tile_cholesky(const CTileSchedule &r_s, double *M, int n_pad, int n, double *p_invdiag, int *p_flag, hipStream_t stream)
{
	for(int l = 0; l < r_s.n_levels; ++ l) {
		const int p0 = r_s.level_potrf_ptr[l], p1 = r_s.level_potrf_ptr[l + 1];
		const int t0 = r_s.level_trsm_ptr[l], t1 = r_s.level_trsm_ptr[l + 1];
		const int g0 = r_s.level_tgt_ptr[l], gu = r_s.level_urgent_end[l];
		const int r0 = r_s.rider_ptr[l], r1 = r_s.rider_ptr[l + 1]; // the updates that ride in this level's diagonal launch
		if(p1 > p0 || r1 > r0) {
			hipLaunchKernelGGL(tile_potrf_kernel, dim3((p1 - p0) + (r1 - r0)), dim3(256), 0, stream, M, n_pad, n, p_invdiag, p_flag,
				r_s.d_potrf + p0, p1 - p0, r_s.d_riders + r0, r_s.d_src);
		}
		if(t1 > t0)
			hipLaunchKernelGGL(tile_trsm_kernel, dim3(t1 - t0), dim3(512), 0, stream, M, n_pad, p_invdiag, r_s.d_trsm + t0);
		if(gu > g0)
			hipLaunchKernelGGL(tile_update_kernel, dim3(gu - g0), dim3(256), 0, stream, M, n_pad, r_s.d_tgt + g0, r_s.d_src);
	}
}

// One launch of the backward substitution by levels (tile_backsolve).  Thread (c, part) = (t / 8, t % 8) holds the eight
// entries 2 part + 16 i + {0, 1} of column c of a tile, fetched as four 16-byte pairs: the eight threads of a column read
// whole 128-byte lines.  Everything a workgroup needs is requested before the first dependent step.
__global__ void __launch_bounds__(512)
tile_backsolve_kernel(const double *M, int ld, int n, const double *p_invdiag, double *z, double *x,
	const int4 *__restrict__ p_diag, int n_diag, const int4 *__restrict__ p_carry,
	const longlong2 *__restrict__ p_dst, double *p_w, double *p_x_out)
{
	enum { PARTS = 8, PER = NB / PARTS };
	__shared__ double s_x[NB];
	__shared__ double s_z[NB];
	const int t = threadIdx.x, c = t / PARTS, part = t % PARTS;
	const bool b_diag = int(blockIdx.x) < n_diag;
	const int4 rec = b_diag? p_diag[blockIdx.x] : p_carry[int(blockIdx.x) - n_diag];
	const int j = b_diag? rec.y : rec.x, k = b_diag? rec.x : rec.y; // z_k -= L(j,k)^T x_j (j < 0: nothing to subtract)
	double vl[PER], vi[PER];
	if(j >= 0) {
		const double *col = M + size_t(j * NB + 2 * part) + size_t(k * NB + c) * ld;
		#pragma unroll
		for(int i = 0; i < PER; i += 2) {
			const v2f64 v = *reinterpret_cast<const v2f64*>(col + PARTS * i);
			vl[i] = v.x;
			vl[i + 1] = v.y;
		}
	}
	if(b_diag) {
		const double *invL = p_invdiag + size_t(k) * NB * NB + c * NB + 2 * part;
		#pragma unroll
		for(int i = 0; i < PER; i += 2) {
			const v2f64 v = *reinterpret_cast<const v2f64*>(invL + PARTS * i);
			vi[i] = v.x;
			vi[i + 1] = v.y;
		}
	}
	double z_in = 0;
	if(t < NB) {
		const int q = k * NB + t;
		z_in = rec.z? ((q < n)? M[size_t(ld - 1) + size_t(q) * ld] : 0.0) : z[q];
		s_x[t] = (j >= 0)? x[j * NB + t] : 0.0;
	}
	__syncthreads();
	double upd = 0;
	if(j >= 0) {
		#pragma unroll
		for(int i = 0; i < PER; ++ i)
			upd += vl[i] * s_x[2 * part + PARTS * (i & ~1) + (i & 1)];
		upd += __shfl_xor(upd, 1);
		upd += __shfl_xor(upd, 2);
		upd += __shfl_xor(upd, 4);
	}
	// (c, part = 0) holds the sum of column c; the thread that fetched z_in[c] is thread c: hand over through LDS
	if(part == 0)
		s_z[c] = upd;
	__syncthreads();
	if(!b_diag) {
		if(t < NB)
			z[k * NB + t] = z_in - s_z[t];
		return;
	}
	if(t < NB)
		s_x[t] = z_in - s_z[t]; // z_k, complete (s_x is free: every read of x_j is behind the barrier above)
	__syncthreads();
	double sum = 0; // x_k[c] = sum_r inv(L_kk)[r][c] z_k[r]
	#pragma unroll
	for(int i = 0; i < PER; ++ i)
		sum += vi[i] * s_x[2 * part + PARTS * (i & ~1) + (i & 1)];
	sum += __shfl_xor(sum, 1);
	sum += __shfl_xor(sum, 2);
	sum += __shfl_xor(sum, 4);
	if(part == 0) {
		const int q = k * NB + c;
		x[q] = sum;
		if(p_dst) {
			const longlong2 d = p_dst[q];
			if(d.x >= 0) {
				p_w[d.x] = sum;
				p_x_out[d.y] = sum;
			}
		}
	}
}

void tile_backsolve(const CTileSchedule &r_s, const double *M, int n_pad, int n, const double *p_invdiag, double *p_z,
	double *p_x, hipStream_t stream, const longlong2 *p_dst, double *p_w, double *p_x_out)
{
	for(int q = 0; q < r_s.n_levels; ++ q) {
		const int d0 = r_s.back_diag_ptr[q], d1 = r_s.back_diag_ptr[q + 1];
		const int c0 = r_s.back_carry_ptr[q], c1 = r_s.back_carry_ptr[q + 1];
		if(d1 - d0 + c1 - c0 > 0) {
			hipLaunchKernelGGL(tile_backsolve_kernel, dim3((d1 - d0) + (c1 - c0)), dim3(512), 0, stream, M, n_pad, n, p_invdiag,
				p_z, p_x, r_s.d_back_diag + d0, d1 - d0, r_s.d_back_carry + c0, p_dst, p_w, p_x_out);
		}
	}
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(dense_tiles) // (the handle's bring-up thread loads this unit's code object: capi.hip)
