// Dev microbenchmark: one panel-solve launch of the dense factorization (trsm_kernel) with one tile below the diagonal
// tile, with and without the fused update of the next diagonal tile; and an empty launch for the floor.
// build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../slam_plus_plus_amd/csrc/dense_chol.hip"
using namespace slampp;
__global__ void empty_kernel(double *p) { if(p == (double*)1) p[0] = 0; }
int main()
{
	const int n_blocks = 8, ld = n_blocks * 64;
	double *M, *inv;
	(void)hipMalloc(&M, sizeof(double) * ld * ld); (void)hipMalloc(&inv, sizeof(double) * 64 * 64);
	(void)hipMemset(M, 0, sizeof(double) * ld * ld); (void)hipMemset(inv, 0, sizeof(double) * 64 * 64);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for(int variant = 0; variant < 4; ++ variant) {
		float best = 1e9f, sum = 0;
		const int n_reps = 200;
		for(int rep = 0; rep < 20; ++ rep) {
			(void)hipEventRecord(e0);
			for(int i = 0; i < n_reps; ++ i) { // back to back on one stream: what a launch adds to a chain
				switch(variant) {
				case 0: hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(512), 0, 0, M); break;
				case 1: hipLaunchKernelGGL(trsm_kernel, dim3(1), dim3(512), 0, 0, M, ld, 0, inv, 0); break;
				case 2: hipLaunchKernelGGL(trsm_kernel, dim3(1), dim3(512), 0, 0, M, ld, 0, inv, 1); break;
				case 3: hipLaunchKernelGGL(trsm_kernel, dim3(7), dim3(512), 0, 0, M, ld, 0, inv, 1); break;
				}
			}
			(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms; sum += ms;
		}
		static const char *p_s_names[] = {"empty launch, 512 threads", "trsm, one tile", "trsm, one tile + update of the next diagonal tile", "trsm, seven tiles, the first with the update"};
		printf("%-60s %.2f us per launch (best of 20 x %d)\n", p_s_names[variant], best * 1e3 / n_reps, n_reps);
	}
	return 0;
}
