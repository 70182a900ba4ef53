// Dev microbenchmark: throughput of the trailing update alone, 64 x 64 jobs (syrk_kernel) against 128 x 128 jobs
// (syrk_wide_kernel), and that the two leave the same matrix.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../slam_plus_plus_amd/csrc/dense_chol.hip"
using namespace slampp;

__global__ void fill_kernel(double *M, size_t n)
{
	for(size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
		M[i] = double((i * 2654435761u) % 1000u) * 1e-3 - 0.5;
}

__global__ void diff_kernel(const double *A, const double *B, int ld, int c0_rows, double *p_max)
{
	// lower triangle (by tiles) of the region [c0_rows, ld)^2
	double f = 0;
	for(size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < size_t(ld) * ld; i += size_t(gridDim.x) * blockDim.x) {
		const int r = int(i % ld), c = int(i / ld);
		if(r >= c0_rows && c >= c0_rows && r / 64 >= c / 64)
			f = fmax(f, fabs(A[i] - B[i]));
	}
	atomicMax((unsigned long long*)p_max, (unsigned long long)__double_as_longlong(f)); // (non-negative doubles order like integers)
}

int main(int argc, char **argv)
{
	const int n_blocks = (argc > 1)? atoi(argv[1]) : 94; // 6016
	const int ld = n_blocks * 64;
	const size_t n_elems = size_t(ld) * ld;
	double *M, *M2, *p_max;
	if(hipMalloc(&M, sizeof(double) * n_elems) != hipSuccess || hipMalloc(&M2, sizeof(double) * n_elems) != hipSuccess) return 1;
	(void)hipMalloc(&p_max, 8);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for(int c0 = 8; c0 < n_blocks - 8; c0 += 16) {
		const int T = n_blocks - c0, T_even = T & ~1, c0_even = n_blocks - T_even, T2 = T_even / 2;
		for(int kw = 1; kw <= 4; kw *= 4) {
			float best[2] = {1e9f, 1e9f};
			for(int rep = 0; rep < 5; ++ rep) {
				hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, M, n_elems);
				(void)hipEventRecord(e0);
				launch_syrk(M, ld, n_blocks, 0, kw, c0_even, n_blocks, 0);
				(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
				float ms; (void)hipEventElapsedTime(&ms, e0, e1); if(ms < best[0]) best[0] = ms;
				hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, M2, n_elems);
				(void)hipEventRecord(e0);
				hipLaunchKernelGGL(syrk_wide_kernel, dim3(T2 * (T2 + 1) / 2), dim3(256), 0, 0, M2, ld, 0, kw, c0_even, T2);
				(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
				(void)hipEventElapsedTime(&ms, e0, e1); if(ms < best[1]) best[1] = ms;
			}
			(void)hipMemset(p_max, 0, 8);
			hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, M, M2, ld, c0_even * 64, p_max);
			double f_diff;
			(void)hipMemcpy(&f_diff, p_max, 8, hipMemcpyDeviceToHost);
			const double tiles = double(T_even) * (T_even + 1) / 2, flops = tiles * 2.0 * 64 * 64 * 64 * kw;
			printf("T=%3d K=%3d: 64x64 jobs %7.1f us %5.1f TFLOP/s | 128x128 jobs %7.1f us %5.1f TFLOP/s (%d jobs) | max diff %g\n", T_even, kw * 64,
				best[0] * 1e3, flops / best[0] / 1e9, best[1] * 1e3, flops / best[1] / 1e9, T2 * (T2 + 1) / 2, f_diff);
		}
	}
	return 0;
}
