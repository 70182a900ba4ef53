// bench_hipsolver_potrf.hip -- vendor cross-check of the dense factorization of the reduced camera system (SURVEY.md section
// 2.1: hipSOLVER "only as a cross-check"): hipsolverDnDpotrf (lower, fp64, in place, device memory) on an n x n symmetric
// positive definite matrix, median of the timed calls, next to the flop count csrc/dense_chol.hip is priced with (n^3 / 3).
// NEVER linked into libslampp_hip.so: a stand-alone program under tools/.  Reference counterpart of the factorization:
// src/slam/LinearSolver_Schur_GPU.cpp:759 (culaDevicePosv), include/slam/LinearSolver_Schur.h:1839-1853 (Eigen LLT).
//   hipcc --offload-arch=gfx950 -O3 tools/bench_hipsolver_potrf.hip -o tools/bin/bench_hipsolver_potrf -lhipsolver
//   tools/bin/bench_hipsolver_potrf 6000 12000
#include <hip/hip_runtime.h>
#include <hipsolver/hipsolver.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { if((x) != hipSuccess) { fprintf(stderr, "HIP error at line %d\n", __LINE__); return 1; } } while(0)
#define CHECKS(x) do { if((x) != HIPSOLVER_STATUS_SUCCESS) { fprintf(stderr, "hipSOLVER error at line %d\n", __LINE__); return 1; } } while(0)

__global__ void fill_spd_kernel(double *A, int n)
{
	// diagonally dominant, symmetric: element (i, j) = 1 / (1 + |i - j|), diagonal + n
	const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
	if(i >= size_t(n) * n)
		return;
	const int r = int(i % n), c = int(i / n);
	const int d = (r > c)? r - c : c - r;
	A[i] = 1.0 / (1.0 + d) + ((r == c)? double(n) : 0.0);
}

int main(int argc, char **argv)
{
	std::vector<int> sizes;
	for(int i = 1; i < argc; ++ i)
		sizes.push_back(atoi(argv[i]));
	if(sizes.empty()) {
		sizes.push_back(6000);
		sizes.push_back(12000);
	}
	hipsolverHandle_t h;
	CHECKS(hipsolverCreate(&h));
	hipStream_t stream;
	CHECK(hipStreamCreate(&stream));
	CHECKS(hipsolverSetStream(h, stream));
	printf("{\"what\": \"hipsolverDnDpotrf, lower, fp64, in place (vendor cross-check; not linked into libslampp_hip.so)\", \"results\": [");
	for(size_t s = 0; s < sizes.size(); ++ s) {
		const int n = sizes[s];
		double *A, *W;
		int *info, lwork = 0;
		CHECK(hipMalloc((void**)&A, size_t(n) * n * sizeof(double)));
		CHECK(hipMalloc((void**)&info, sizeof(int)));
		CHECKS(hipsolverDnDpotrf_bufferSize(h, HIPSOLVER_FILL_MODE_LOWER, n, A, n, &lwork));
		CHECK(hipMalloc((void**)&W, std::max(lwork, 1) * sizeof(double)));
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0));
		CHECK(hipEventCreate(&e1));
		std::vector<float> ms;
		int h_info = 0;
		for(int rep = 0; rep < 7; ++ rep) {
			hipLaunchKernelGGL(fill_spd_kernel, dim3(unsigned((size_t(n) * n + 255) / 256)), dim3(256), 0, stream, A, n);
			CHECK(hipEventRecord(e0, stream));
			CHECKS(hipsolverDnDpotrf(h, HIPSOLVER_FILL_MODE_LOWER, n, A, n, W, lwork, info));
			CHECK(hipEventRecord(e1, stream));
			CHECK(hipStreamSynchronize(stream));
			float f = 0;
			CHECK(hipEventElapsedTime(&f, e0, e1));
			if(rep >= 2) // two warm-up calls
				ms.push_back(f);
			CHECK(hipMemcpy(&h_info, info, sizeof(int), hipMemcpyDeviceToHost));
		}
		std::sort(ms.begin(), ms.end());
		const double f_ms = ms[ms.size() / 2], f_flops = double(n) * n * n / 3.0;
		printf("%s{\"n\": %d, \"ms\": %.4f, \"ms_min\": %.4f, \"TFLOP/s\": %.3f, \"frac_of_78.6\": %.4f, \"info\": %d}", s? ", " : "", n, f_ms, ms[0],
			f_flops / (f_ms * 1e-3) / 1e12, f_flops / (f_ms * 1e-3) / 1e12 / 78.6, h_info);
		(void)hipFree(A); (void)hipFree(W); (void)hipFree(info);
	}
	printf("]}\n");
	return 0;
}
