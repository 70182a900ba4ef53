// dev helper (round 6): what the first launch of a kernel costs, by what came before it -- the first kernel of the code object,
// a second kernel of the same code object, a kernel whose attributes were asked for first (hipFuncGetAttributes)
// hipcc --offload-arch=gfx950 -O3 -o first_launch_cost first_launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int N>
__global__ void k(double *p, int n)
{
	double a[N];
	for(int i = 0; i < N; ++ i)
		a[i] = p[(threadIdx.x * N + i) % n];
	for(int r = 0; r < n; ++ r) {
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			a[i] = a[i] * a[(i + 1) % N] + r;
	}
	double s = 0;
	for(int i = 0; i < N; ++ i)
		s += a[i];
	p[threadIdx.x] = s;
}

static double now()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <int N>
static void first(const char *p_s_what, double *p, bool b_attr)
{
	double t0 = now();
	if(b_attr) {
		hipFuncAttributes t;
		(void)hipFuncGetAttributes(&t, (const void*)k<N>);
	}
	double t1 = now();
	hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, p, 1);
	double t2 = now();
	(void)hipDeviceSynchronize();
	double t3 = now();
	hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, p, 1);
	(void)hipDeviceSynchronize();
	double t4 = now();
	printf("%-40s attributes %7.3f ms, first launch %7.3f + sync %7.3f, second launch + sync %7.3f\n", p_s_what, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
}

int main()
{
	double t0 = now();
	double *p;
	(void)hipMalloc(&p, 1 << 20);
	(void)hipMemset(p, 0, 1 << 20);
	(void)hipDeviceSynchronize();
	printf("runtime up, first allocation: %.1f ms\n", now() - t0);
	first<8>("first kernel of the code object", p, false);
	first<16>("second kernel, same code object", p, false);
	first<24>("third kernel", p, false);
	first<32>("fourth kernel, attributes asked first", p, true);
	first<40>("fifth kernel, attributes asked first", p, true);
	first<48>("sixth kernel", p, false);
	hipStream_t s;
	t0 = now();
	(void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	double t1 = now();
	hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, s, p, 1);
	(void)hipStreamSynchronize(s);
	double t2 = now();
	hipLaunchKernelGGL(k<56>, dim3(1), dim3(64), 0, s, p, 1);
	(void)hipStreamSynchronize(s);
	double t3 = now();
	printf("stream creation %.3f ms, a known kernel on it + sync %.3f, a new kernel on it + sync %.3f\n", t1 - t0, t2 - t1, t3 - t2);
	for(int i = 0; i < 6; ++ i) {
		hipStream_t s2;
		t0 = now();
		(void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
		t1 = now();
		hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, s2, p, 1);
		(void)hipStreamSynchronize(s2);
		t2 = now();
		hipEvent_t e;
		(void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
		double t3b = now();
		double *q;
		(void)hipHostMalloc(&q, 1 << 20, hipHostMallocDefault);
		double t4 = now();
		(void)hipMemcpyAsync(p, q, 1 << 20, hipMemcpyHostToDevice, s2);
		(void)hipStreamSynchronize(s2);
		double t5 = now();
		printf("stream %d: creation %.3f ms, a known kernel on it + sync %.3f, event %.3f, 1 MB pinned %.3f, copy of it + sync %.3f\n", i + 2, t1 - t0, t2 - t1, t3b - t2, t4 - t3b, t5 - t4);
	}
	return 0;
}
