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
	for(int variant = 0; variant < 6; ++ variant) {
		float best = 1e9;
		for(int rep = 0; rep < 20; ++ rep) {
			hipMemcpy(M, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
			hipEventRecord(e0);
			switch(variant) {
			case 0: hipLaunchKernelGGL((potrf_diag_variant<true, true>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			case 1: hipLaunchKernelGGL((potrf_diag_variant<true, false>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			case 2: hipLaunchKernelGGL((potrf_diag_variant<false, true>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			case 3: hipLaunchKernelGGL((potrf_diag_variant<false, false>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			case 4: hipLaunchKernelGGL((potrf_diag_variant<true, true, false>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break; // round 5's one-wave panels
			case 5: hipLaunchKernelGGL((potrf_diag_variant<true, false, false>), dim3(1), dim3(256), 0, 0, M, ld, 0, n, inv, flag); break;
			}
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
		}
		printf("variant chol=%d inv=%d%s: %.2f us\n", variant < 2 || variant >= 4, variant % 2 == 0, (variant >= 4)? " (one-wave panels, round 5)" : "", best * 1e3);
#ifdef POTRF_STAMPS
		if(variant < 2 || variant >= 4) {
			long long st[64];
			hipMemcpyFromSymbol(st, HIP_SYMBOL(slampp::g_potrf_stamps), sizeof(st));
			for(int i = 1; i < 15; ++ i)
				printf("  stamp %2d: +%6lld cycles, +%6lld ns\n", i, st[2 * i] - st[2 * (i - 1)], (st[2 * i + 1] - st[2 * (i - 1) + 1]) * 10);
			if(variant < 2) { // the three waves of the first panel against its start (stamp 1), by the wall clock (the waves' own clocks are not one clock)
				const char *p_s_what[6] = {"chain through", "chain wave done", "rows: steps through", "rows done", "inverse: steps through", "inverse done"};
				for(int i = 16; i < 22; ++ i)
					printf("  panel 0, %-24s +%6lld ns\n", p_s_what[i - 16], (st[2 * i + 1] - st[2 * 1 + 1]) * 10);
			}
		}
#endif
		if(variant == 0 || variant == 4) { // against a host Cholesky, and inv * L = I
			std::vector<double> L(n * n), X(n * n), R = h;
			hipMemcpy(L.data(), M, sizeof(double) * n * n, hipMemcpyDeviceToHost);
			hipMemcpy(X.data(), inv, sizeof(double) * n * n, hipMemcpyDeviceToHost);
			for(int k = 0; k < n; ++ k) {
				R[k + k * ld] = sqrt(R[k + k * ld]);
				for(int i = k + 1; i < n; ++ i) R[i + k * ld] /= R[k + k * ld];
				for(int j = k + 1; j < n; ++ j) for(int i = j; i < n; ++ i) R[i + j * ld] -= R[i + k * ld] * R[j + k * ld];
			}
			double e_L = 0, e_X = 0;
			for(int j = 0; j < n; ++ j) for(int i = j; i < n; ++ i) e_L = fmax(e_L, fabs(L[i + j * ld] - R[i + j * ld]));
			for(int i = 0; i < n; ++ i) for(int j = 0; j <= i; ++ j) {
				double f = 0;
				for(int k = j; k <= i; ++ k) f += X[i + k * ld] * R[k + j * ld];
				e_X = fmax(e_X, fabs(f - (i == j)));
			}
			printf("  max |L - L_host| = %g, max |X L - I| = %g\n", e_L, e_X);
		}
	}
	return 0;
}
