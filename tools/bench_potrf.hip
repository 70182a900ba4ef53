// Dev microbenchmark: where does potrf_diag_kernel spend its time? (build: hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define POTRF_VARIANTS 1
#include "../slam_plus_plus_amd/csrc/dense_chol.hip"
using namespace slampp;
int main()
{
	const int n = 64, ld = 64;
	std::vector<double> h(n * n);
	for(int i = 0; i < n; ++ i) for(int j = 0; j < n; ++ j) h[i + j * ld] = (i == j)? 100.0 + i : 1.0 / (1 + abs(i - j));
	double *M, *inv; int *flag;
	hipMalloc(&M, sizeof(double) * n * n); hipMalloc(&inv, sizeof(double) * n * n); hipMalloc(&flag, 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for(int variant = 0; variant < 4; ++ variant) {
		float best = 1e9;
		for(int rep = 0; rep < 20; ++ rep) {
			hipMemcpy(M, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
			hipEventRecord(e0);
			switch(variant) {
			case 0: hipLaunchKernelGGL((potrf_diag_variant<true, true>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			case 1: hipLaunchKernelGGL((potrf_diag_variant<true, false>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			case 2: hipLaunchKernelGGL((potrf_diag_variant<false, true>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			case 3: hipLaunchKernelGGL((potrf_diag_variant<false, false>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			}
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
		}
		printf("variant chol=%d inv=%d: %.2f us\n", variant < 2, variant % 2 == 0, best * 1e3);
	}
	return 0;
}
