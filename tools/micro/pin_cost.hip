// dev microbenchmark: what pinning N MB of host memory costs, by method
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <omp.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	size_t mb = argc > 1? atoi(argv[1]) : 58;
	size_t n = mb << 20;
	hipFree(0);
	void *d; hipMalloc(&d, n);
	for(int rep = 0; rep < 2; ++ rep) {
		double t0 = now();
		void *p; hipHostMalloc(&p, n, hipHostMallocDefault);
		double t1 = now();
		hipMemcpy(d, p, n, hipMemcpyHostToDevice);
		double t2 = now();
		hipHostFree(p);
		double t3 = now();
		printf("hipHostMalloc default %zu MB: alloc %.2f ms, h2d %.2f ms, free %.2f\n", mb, t1 - t0, t2 - t1, t3 - t2);
		t0 = now();
		hipHostMalloc(&p, n, hipHostMallocNonCoherent);
		t1 = now();
		hipMemcpy(d, p, n, hipMemcpyHostToDevice);
		t2 = now();
		hipHostFree(p);
		printf("hipHostMalloc noncoherent: alloc %.2f ms, h2d %.2f ms\n", t1 - t0, t2 - t1);
		t0 = now();
		hipHostMalloc(&p, n, hipHostMallocNumaUser);
		t1 = now();
		hipHostFree(p);
		printf("hipHostMalloc numauser: alloc %.2f ms\n", t1 - t0);
		for(int huge = 0; huge < 2; ++ huge) {
			for(int nt = 1; nt <= 16; nt *= 4) {
				t0 = now();
				void *q = aligned_alloc(size_t(2) << 20, n);
				if(huge) madvise(q, n, MADV_HUGEPAGE);
				#pragma omp parallel for num_threads(nt) schedule(static)
				for(long i = 0; i < long(n); i += 4096)
					((volatile char*)q)[i] = 0;
				t1 = now();
				hipError_t e = hipHostRegister(q, n, hipHostRegisterDefault);
				t2 = now();
				hipMemcpy(d, q, n, hipMemcpyHostToDevice);
				t3 = now();
				hipHostUnregister(q);
				double t4 = now();
				free(q);
				printf("malloc%s + touch(%d thr) %.2f ms, register %.2f ms (%s), h2d %.2f ms, unregister %.2f\n", huge? "+THP" : "", nt, t1 - t0, t2 - t1, hipGetErrorString(e), t3 - t2, t4 - t3);
			}
		}
	}
	return 0;
}
