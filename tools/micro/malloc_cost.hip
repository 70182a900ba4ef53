// Dev helper: what hipMalloc / hipFree / a pageable upload cost, by size, fresh and after frees (the analysis makes ~50 device arrays)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipStream_t s; hipStreamCreate(&s);
	void *w; hipMalloc(&w, 1 << 20); hipFree(w);
	const size_t sizes[] = {4096, 65536, 1 << 20, 8 << 20, 32 << 20, 128 << 20};
	for(int round = 0; round < 3; ++ round) {
		for(size_t n : sizes) {
			std::vector<char> host(n, 1);
			void *p[8];
			double t0 = now();
			for(int i = 0; i < 8; ++ i) hipMalloc(&p[i], n);
			double t1 = now();
			for(int i = 0; i < 8; ++ i) hipMemcpyAsync(p[i], host.data(), n, hipMemcpyHostToDevice, s);
			double t2 = now();
			hipStreamSynchronize(s);
			double t3 = now();
			for(int i = 0; i < 8; ++ i) hipFree(p[i]);
			double t4 = now();
			printf("round %d, %9zu B x 8: hipMalloc %7.3f ms each, pageable hipMemcpyAsync %7.3f ms each (+ sync %6.3f), hipFree %7.3f ms each\n",
				round, n, (t1 - t0) / 8, (t2 - t1) / 8, t3 - t2, (t4 - t3) / 8);
		}
	}
	return 0;
}
