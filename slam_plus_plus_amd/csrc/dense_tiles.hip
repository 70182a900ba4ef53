// dense_tiles.hip -- tile-sparse, level-scheduled factorization of the dense top (see dense_chol.h), and the
// identity diagonal of its alignment gaps.  Kept in its own translation unit: the register allocation of the
// kernels in dense_chol.hip was seen to change (260 instead of 204 registers for the diagonal-tile kernel, one
// workgroup per CU instead of two) when these kernels were compiled next to them.
#include <hip/hip_runtime.h>
#include "dense_chol.h"
#include "plan.h"
#include <algorithm>

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

} // namespace slampp
