// Dev microbenchmark: what the fp64 matrix cores sustain with nothing but v_mfma_f64_16x16x4 in flight
// (N independent accumulators per wave, W waves per SIMD).  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// the same with an operand pair of its own per accumulator (no two MFMAs in a row read the same registers)
template <int N>
__global__ void __launch_bounds__(256) mfma_loop_distinct(double *out, int n_iter, double a0, double b0)
{
	v4f64 acc[N];
	double a[N], b[N];
	#pragma unroll
	for(int i = 0; i < N; ++ i) {
		acc[i] = v4f64{0, 0, 0, 0};
		a[i] = a0 + threadIdx.x + i;
		b[i] = b0 - i;
	}
	const long long n_c0 = clock64(), n_w0 = wall_clock64();
	for(int it = 0; it < n_iter; ++ it) {
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
	}
	double s = 0;
	#pragma unroll
	for(int i = 0; i < N; ++ i)
		s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	const long long n_c1 = clock64(), n_w1 = wall_clock64();
	if(blockIdx.x == 0 && threadIdx.x == 0) {
		((long long*)out)[64] = n_c1 - n_c0;
		((long long*)out)[65] = n_w1 - n_w0;
	}
	if(s == 12345.678)
		out[threadIdx.x] = s;
}

// the vector unit's v_fma_f64 for comparison: N independent accumulators per lane
template <int N>
__global__ void __launch_bounds__(256) fma_loop(double *out, int n_iter, double a0, double b0)
{
	double acc[N];
	#pragma unroll
	for(int i = 0; i < N; ++ i)
		acc[i] = threadIdx.x + i;
	const double a = a0 + 1e-9 * threadIdx.x, b = b0;
	for(int it = 0; it < n_iter; ++ it) {
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			acc[i] = __builtin_fma(acc[i], a, b);
	}
	double s = 0;
	#pragma unroll
	for(int i = 0; i < N; ++ i)
		s += acc[i];
	if(s == 12345.678)
		out[threadIdx.x] = s;
}

// v_fmac_f64_dpp row_newbcast: acc += (lane K of the row's value of src) * mul -- the broadcast form an outer product on the
// vector unit would be made of
template <int N>
__global__ void __launch_bounds__(256) fmac_dpp_loop(double *out, int n_iter, double a0, double b0)
{
	double acc[N];
	#pragma unroll
	for(int i = 0; i < N; ++ i)
		acc[i] = threadIdx.x + i;
	double a = a0 + 1e-9 * threadIdx.x, b = b0 * 1e-3;
	for(int it = 0; it < n_iter; ++ it) {
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a), "v"(b), "n"(i % 16));
	}
	double s = 0;
	#pragma unroll
	for(int i = 0; i < N; ++ i)
		s += acc[i];
	if(s == 12345.678)
		out[threadIdx.x] = s;
}

// both at once: the even waves of a workgroup issue matrix instructions, the odd ones v_fma_f64 -- do the two pipes add up?
template <int N>
__global__ void __launch_bounds__(512) mixed_loop(double *out, int n_iter_mfma, int n_iter_fma, double a0, double b0)
{
	const int wave = threadIdx.x >> 6;
	double s = 0;
	if(wave & 1) {
		double acc[2 * N];
		#pragma unroll
		for(int i = 0; i < 2 * N; ++ i)
			acc[i] = threadIdx.x + i;
		const double a = a0 + 1e-9 * threadIdx.x, b = b0;
		for(int it = 0; it < n_iter_fma; ++ it) {
			#pragma unroll
			for(int i = 0; i < 2 * N; ++ i)
				acc[i] = __builtin_fma(acc[i], a, b);
		}
		#pragma unroll
		for(int i = 0; i < 2 * N; ++ i)
			s += acc[i];
	} else {
		v4f64 acc[N];
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			acc[i] = v4f64{0, 0, 0, 0};
		const double a = a0 + threadIdx.x, b = b0;
		for(int it = 0; it < n_iter_mfma; ++ it) {
			#pragma unroll
			for(int i = 0; i < N; ++ i)
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
		}
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	}
	if(s == 12345.678)
		out[threadIdx.x] = s;
}

static void run_mixed(double *out, int n_wgs, int n_iter_mfma, int n_iter_fma)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	for(int rep = 0; rep < 5; ++ rep) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(mixed_loop<8>, dim3(n_wgs), dim3(512), 0, 0, out, n_iter_mfma, n_iter_fma, 0.999, 0.5);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
	}
	const double f_mfma = double(n_wgs) * 4 * n_iter_mfma * 8 * 2048.0, f_fma = double(n_wgs) * 4 * 64 * double(n_iter_fma) * 16 * 2.0;
	printf("mixed, %d workgroups of 4 matrix + 4 vector waves, %d / %d iterations: %.1f us, matrix %.1f + vector %.1f = %.1f TFLOP/s\n", n_wgs, n_iter_mfma,
		n_iter_fma, best * 1e3, f_mfma / best / 1e9, f_fma / best / 1e9, (f_mfma + f_fma) / best / 1e9);
}

template <int N, bool b_dpp = false>
static void run_fma(double *out, int n_wgs)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int n_iter = 16384;
	float best = 1e9f;
	for(int rep = 0; rep < 5; ++ rep) {
		(void)hipEventRecord(e0);
		if(b_dpp)
			hipLaunchKernelGGL(fmac_dpp_loop<N>, dim3(n_wgs), dim3(256), 0, 0, out, n_iter, 0.999, 0.5);
		else
			hipLaunchKernelGGL(fma_loop<N>, dim3(n_wgs), dim3(256), 0, 0, out, n_iter, 0.999, 0.5);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
	}
	const double flops = double(n_wgs) * 256 * n_iter * N * 2.0;
	printf("%s: %d accumulators per lane, %d workgroups of 4 waves: %.1f us, %.1f TFLOP/s, %.2f clocks at 2.4 GHz per instruction per SIMD\n", b_dpp? "v_fmac_f64_dpp row_newbcast" : "v_fma_f64", N, n_wgs,
		best * 1e3, flops / best / 1e9, best * 1e-3 * 2.4e9 / (double(n_wgs) * 4 * n_iter * N / 1024.0));
}

template <int N>
__global__ void __launch_bounds__(256) mfma_loop(double *out, int n_iter, double a0, double b0)
{
	v4f64 acc[N];
	#pragma unroll
	for(int i = 0; i < N; ++ i)
		acc[i] = v4f64{0, 0, 0, 0};
	double a = a0 + threadIdx.x, b = b0;
	const long long n_c0 = clock64(), n_w0 = wall_clock64();
	for(int it = 0; it < n_iter; ++ it) {
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
	}
	double s = 0;
	#pragma unroll
	for(int i = 0; i < N; ++ i)
		s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	const long long n_c1 = clock64(), n_w1 = wall_clock64(); // (after the sums: they wait for the last MFMA)
	if(blockIdx.x == 0 && threadIdx.x == 0) {
		((long long*)out)[64] = n_c1 - n_c0;
		((long long*)out)[65] = n_w1 - n_w0;
	}
	if(s == 12345.678)
		out[threadIdx.x] = s;
}

template <int N, bool b_distinct = false>
static void run(double *out, int n_wgs, const char *p_s_label)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int n_iter = 4096;
	float best = 1e9f;
	for(int rep = 0; rep < 5; ++ rep) {
		(void)hipEventRecord(e0);
		if(b_distinct)
			hipLaunchKernelGGL(mfma_loop_distinct<N>, dim3(n_wgs), dim3(256), 0, 0, out, n_iter, 1.0, 2.0);
		else
			hipLaunchKernelGGL(mfma_loop<N>, dim3(n_wgs), dim3(256), 0, 0, out, n_iter, 1.0, 2.0);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
	}
	long long st[2];
	(void)hipMemcpy(st, (long long*)out + 64, 16, hipMemcpyDeviceToHost);
	printf("  wave 0: %.1f shader clocks per MFMA slot (all waves of the SIMD), %.0f MHz held\n", double(st[0]) / (double(n_iter) * N * ((n_wgs + 255) / 256)), double(st[0]) / (double(st[1]) * 10e-3));
	const double flops = double(n_wgs) * 4 * n_iter * N * 2048.0;
	printf("%s: %d accumulators, %d workgroups of 4 waves: %.1f us, %.1f TFLOP/s, %.1f cycles at 2.4 GHz per MFMA per SIMD\n", p_s_label, N, n_wgs,
		best * 1e3, flops / best / 1e9, best * 1e-3 * 2.4e9 / (double(n_wgs) * 4 * n_iter * N / 1024.0));
}

int main()
{
	double *out; (void)hipMalloc(&out, 4096);
	run<1>(out, 256, "dependent chain, 1 wave/SIMD");
	run<4>(out, 256, "1 wave/SIMD");
	run<8>(out, 256, "1 wave/SIMD");
	run<16>(out, 256, "1 wave/SIMD");
	run<4>(out, 512, "2 waves/SIMD");
	run<8>(out, 512, "2 waves/SIMD");
	run<8>(out, 1024, "4 waves/SIMD");
	run<8, true>(out, 256, "distinct operands, 1 wave/SIMD");
	run<8, true>(out, 512, "distinct operands, 2 waves/SIMD");
	run<16, true>(out, 512, "distinct operands, 2 waves/SIMD");
	run<8, true>(out, 1024, "distinct operands, 4 waves/SIMD");
	run_fma<8>(out, 256);
	run_fma<8>(out, 512);
	run_fma<8>(out, 1024);
	run_fma<16>(out, 1024);
	run_fma<8>(out, 2048);
	run_fma<16, true>(out, 256);
	run_fma<16, true>(out, 512);
	run_fma<16, true>(out, 1024);
	run_fma<16, true>(out, 2048);
	// (iteration counts that take each kind of wave about the same time alone: 105 clocks x 8 against 5 clocks x 16)
	run_mixed(out, 256, 4096, 0);
	run_mixed(out, 256, 0, 40000);
	run_mixed(out, 256, 4096, 40000);
	run_mixed(out, 512, 4096, 40000);
	run_mixed(out, 512, 4096, 20000);
	run_mixed(out, 512, 4096, 80000);
	return 0;
}
