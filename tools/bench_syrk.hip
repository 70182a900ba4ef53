// Dev microbenchmark: throughput of the trailing update (syrk_kernel) alone. build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../slam_plus_plus_amd/csrc/dense_chol.hip"
using namespace slampp;
int main(int argc, char **argv)
{
	const int n_blocks = (argc > 1)? atoi(argv[1]) : 188; // 12032
	const int ld = n_blocks * 64;
	double *M;
	if(hipMalloc(&M, sizeof(double) * size_t(ld) * ld) != hipSuccess) return 1;
	(void)hipMemset(M, 0, sizeof(double) * size_t(ld) * ld);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for(int kw = 1; kw <= 8; kw *= 2) {
		const int c0 = 8;
		float best = 1e9;
		for(int rep = 0; rep < 5; ++ rep) {
			(void)hipEventRecord(e0);
			launch_syrk(M, ld, n_blocks, 0, kw, c0, n_blocks, 0);
			(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
		}
		const double T = n_blocks - c0, tiles = T * (T + 1) / 2, flops = tiles * 2.0 * 64 * 64 * 64 * kw;
		printf("T=%d K=%d: %.1f us, %.0f tiles, %.1f TFLOP/s, %.1f tiles/us\n", int(T), kw * 64, best * 1e3, tiles, flops / best / 1e9, tiles / (best * 1e3));
	}
	return 0;
}
