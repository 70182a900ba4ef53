// schur_marginals.hip -- block diagonal of the covariance Lambda^-1 of a BA system from its Schur complement:
//   cameras    Sigma_cc = blocks (c, c) of Z = S^-1
//   landmarks  Sigma_pp = C_p^-1 + sum over the cameras a, b observing p of W_a^T Z(a,b) W_b,   W_o = U_o C_p^-1
// which is what CSchurComplement_Marginals::Schur_Marginals computes
// (/root/reference/include/slam/BAMarginals.h:579-806: Dinv + (R^-T U Dinv)^T (R^-T U Dinv) with S = R^T R, and the
// diagonal blocks of S^-1 by the recursive formula).  Z is dense here (dense_inverse.hip): only its lower triangle
// and the whole of its diagonal 64 x 64 tiles are valid, element (i, j) is read as (max, min).
// The landmark kernel is a gather: k (k + 1) / 2 blocks of Z per landmark with k observations, DC column segments
// each; neighbouring landmarks see neighbouring cameras, so most of them come out of L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slampp {

// N contiguous doubles with 16-byte loads where the address allows it (blocks of 6 x 6 and 6 x 3 doubles start at
// multiples of 16 bytes; odd sizes end with one 8-byte load)
template <int N>
__device__ __forceinline__ void load_block(double (&r_dst)[N], const double *__restrict__ p_src)
{
	typedef double v2f64 __attribute__((ext_vector_type(2)));
	if((reinterpret_cast<uintptr_t>(p_src) & 15) == 0) {
		#pragma unroll
		for(int i = 0; i + 1 < N; i += 2) {
			const v2f64 v = *reinterpret_cast<const v2f64*>(p_src + i);
			r_dst[i] = v.x;
			r_dst[i + 1] = v.y;
		}
		if(N & 1)
			r_dst[N - 1] = p_src[N - 1];
	} else {
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			r_dst[i] = p_src[i];
	}
}

template <int DC>
__global__ void schur_cam_cov_kernel(int64_t nc, const double *__restrict__ Z, int ld, double *out)
{
	const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(gid >= nc * (DC * DC))
		return;
	const int64_t c = gid / (DC * DC);
	const int e = int(gid - c * (DC * DC)), r = e % DC, q = e / DC;
	const int hi = (r > q)? r : q, lo = (r > q)? q : r;
	out[gid] = Z[size_t(c * DC + hi) + size_t(c * DC + lo) * ld];
}

template <int DC, int DP>
__global__ void __launch_bounds__(128)
schur_point_cov_kernel(const int64_t *__restrict__ ptr, const int32_t *__restrict__ brow, int64_t nc, int64_t np,
	const double *__restrict__ W, const double *__restrict__ Cinv, const double *__restrict__ Z, int ld, double *out)
{
	const int64_t pt = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(pt >= np)
		return;
	const int64_t k0 = ptr[nc + pt], k1 = ptr[nc + pt + 1] - 1; // the last block of the column is C_p itself
	const int64_t o0 = k0 - ptr[nc] - pt;
	double cov[DP * DP];
	#pragma unroll
	for(int i = 0; i < DP * DP; ++ i)
		cov[i] = Cinv[pt * (DP * DP) + i];
	for(int64_t a = k0; a < k1; ++ a) {
		const int64_t ca = brow[a];
		double wa[DC * DP];
		load_block<DC * DP>(wa, W + (o0 + (a - k0)) * (DC * DP));
		for(int64_t b = k0; b <= a; ++ b) { // block rows ascend inside a column: cb <= ca, the lower triangle of Z
			const int64_t cb = brow[b];
			double wb[DC * DP], t[DC * DP];
			load_block<DC * DP>(wb, W + (o0 + (b - k0)) * (DC * DP));
			#pragma unroll
			for(int i = 0; i < DC * DP; ++ i)
				t[i] = 0;
			#pragma unroll
			for(int q = 0; q < DC; ++ q) {
				double zc[DC]; // column q of the block: contiguous below the diagonal block, (max, min) inside it
				if(a == b) {
					#pragma unroll
					for(int r = 0; r < DC; ++ r) {
						const int hi = (r > q)? r : q, lo = (r > q)? q : r;
						zc[r] = Z[size_t(ca * DC + hi) + size_t(ca * DC + lo) * ld];
					}
				} else
					load_block<DC>(zc, Z + size_t(ca * DC) + size_t(cb * DC + q) * ld);
				#pragma unroll
				for(int r = 0; r < DC; ++ r) {
					#pragma unroll
					for(int j = 0; j < DP; ++ j)
						t[r + j * DC] += zc[r] * wb[q + j * DC];
				}
			}
			#pragma unroll
			for(int j = 0; j < DP; ++ j) {
				#pragma unroll
				for(int i = 0; i < DP; ++ i) {
					double sum = 0;
					#pragma unroll
					for(int r = 0; r < DC; ++ r)
						sum += wa[r + i * DC] * t[r + j * DC];
					cov[i + j * DP] += sum;
					if(a != b)
						cov[j + i * DP] += sum; // the mirrored pair (b, a) contributes the transpose
				}
			}
		}
	}
	#pragma unroll
	for(int i = 0; i < DP * DP; ++ i)
		out[pt * (DP * DP) + i] = cov[i];
}

// ---- the same from the sparse inverse subset (sparse_inverse.hip): Z is laid out like the factor of the reduced system,
// in its elimination order; where a block sits and whether it is stored transposed comes from host-built tables ----
template <int DC>
__global__ void schur_cam_cov_sparse_kernel(int64_t nc, const int64_t *__restrict__ cam_zoff, const double *__restrict__ Z, double *out)
{
	const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(gid >= nc * (DC * DC))
		return;
	const int64_t c = gid / (DC * DC);
	out[gid] = Z[cam_zoff[c] + (gid - c * (DC * DC))]; // diagonal blocks are stored whole
}

template <int DC, int DP>
__global__ void __launch_bounds__(128)
schur_point_cov_sparse_kernel(const int64_t *__restrict__ ptr, int64_t nc, int64_t np, const int64_t *__restrict__ pair_ptr,
	const int64_t *__restrict__ pair_tab, const double *__restrict__ W, const double *__restrict__ Cinv,
	const double *__restrict__ Z, double *out)
{
	const int64_t pt = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(pt >= np)
		return;
	const int64_t k0 = ptr[nc + pt], k1 = ptr[nc + pt + 1] - 1;
	const int64_t o0 = k0 - ptr[nc] - pt;
	const int64_t *tab = pair_tab + pair_ptr[pt];
	double cov[DP * DP];
	#pragma unroll
	for(int i = 0; i < DP * DP; ++ i)
		cov[i] = Cinv[pt * (DP * DP) + i];
	for(int64_t a = k0; a < k1; ++ a) {
		double wa[DC * DP];
		load_block<DC * DP>(wa, W + (o0 + (a - k0)) * (DC * DP));
		for(int64_t b = k0; b <= a; ++ b) {
			const int64_t ia = a - k0, ib = b - k0;
			const int64_t ent = tab[ia * (ia + 1) / 2 + ib]; // block Z(cam_a, cam_b): offset * 2 + stored transposed
			const bool b_tr = ent & 1;
			double wb[DC * DP], zb[DC * DC], t[DC * DP];
			load_block<DC * DC>(zb, Z + (ent >> 1)); // a thread's scattered loads are paid per instruction: 16 bytes each
			load_block<DC * DP>(wb, W + (o0 + ib) * (DC * DP));
			#pragma unroll
			for(int i = 0; i < DC * DP; ++ i)
				t[i] = 0;
			#pragma unroll
			for(int q = 0; q < DC; ++ q) {
				#pragma unroll
				for(int r = 0; r < DC; ++ r) {
					const double z = b_tr? zb[q + r * DC] : zb[r + q * DC];
					#pragma unroll
					for(int j = 0; j < DP; ++ j)
						t[r + j * DC] += z * wb[q + j * DC];
				}
			}
			#pragma unroll
			for(int j = 0; j < DP; ++ j) {
				#pragma unroll
				for(int i = 0; i < DP; ++ i) {
					double sum = 0;
					#pragma unroll
					for(int r = 0; r < DC; ++ r)
						sum += wa[r + i * DC] * t[r + j * DC];
					cov[i + j * DP] += sum;
					if(a != b)
						cov[j + i * DP] += sum;
				}
			}
		}
	}
	#pragma unroll
	for(int i = 0; i < DP * DP; ++ i)
		out[pt * (DP * DP) + i] = cov[i];
}

template <int DC, int DP>
static void launch_sparse_t(int64_t nc, int64_t np, const int64_t *ptr, const int64_t *cam_zoff, const int64_t *pair_ptr,
	const int64_t *pair_tab, const double *W, const double *Cinv, const double *Z, double *cam_cov, double *point_cov,
	hipStream_t stream)
{
	if(cam_cov)
		hipLaunchKernelGGL((schur_cam_cov_sparse_kernel<DC>), dim3(unsigned((nc * DC * DC + 255) / 256)), dim3(256), 0, stream,
			nc, cam_zoff, Z, cam_cov);
	if(point_cov)
		hipLaunchKernelGGL((schur_point_cov_sparse_kernel<DC, DP>), dim3(unsigned((np + 127) / 128)), dim3(128), 0, stream,
			ptr, nc, np, pair_ptr, pair_tab, W, Cinv, Z, point_cov);
}

void schur_marginals_sparse_launch(int DC, int DP, int64_t nc, int64_t np, const int64_t *ptr, const int64_t *cam_zoff,
	const int64_t *pair_ptr, const int64_t *pair_tab, const double *W, const double *Cinv, const double *Z, double *cam_cov,
	double *point_cov, hipStream_t stream)
{
	if(DC == 6 && DP == 3)
		launch_sparse_t<6, 3>(nc, np, ptr, cam_zoff, pair_ptr, pair_tab, W, Cinv, Z, cam_cov, point_cov, stream);
	else if(DC == 7 && DP == 3)
		launch_sparse_t<7, 3>(nc, np, ptr, cam_zoff, pair_ptr, pair_tab, W, Cinv, Z, cam_cov, point_cov, stream);
	else
		launch_sparse_t<3, 2>(nc, np, ptr, cam_zoff, pair_ptr, pair_tab, W, Cinv, Z, cam_cov, point_cov, stream);
}

template <int DC, int DP>
static void launch_t(int64_t nc, int64_t np, const int64_t *ptr, const int32_t *brow, const double *W, const double *Cinv,
	const double *Z, int ld, double *cam_cov, double *point_cov, hipStream_t stream)
{
	if(cam_cov)
		hipLaunchKernelGGL((schur_cam_cov_kernel<DC>), dim3(unsigned((nc * DC * DC + 255) / 256)), dim3(256), 0, stream,
			nc, Z, ld, cam_cov);
	if(point_cov)
		hipLaunchKernelGGL((schur_point_cov_kernel<DC, DP>), dim3(unsigned((np + 127) / 128)), dim3(128), 0, stream,
			ptr, brow, nc, np, W, Cinv, Z, ld, point_cov);
}

void schur_marginals_launch(int DC, int DP, int64_t nc, int64_t np, const int64_t *ptr, const int32_t *brow, const double *W,
	const double *Cinv, const double *Z, int ld, double *cam_cov, double *point_cov, hipStream_t stream)
{
	if(DC == 6 && DP == 3)
		launch_t<6, 3>(nc, np, ptr, brow, W, Cinv, Z, ld, cam_cov, point_cov, stream);
	else if(DC == 7 && DP == 3)
		launch_t<7, 3>(nc, np, ptr, brow, W, Cinv, Z, ld, cam_cov, point_cov, stream);
	else
		launch_t<3, 2>(nc, np, ptr, brow, W, Cinv, Z, ld, cam_cov, point_cov, stream);
}

} // namespace slampp
