// Dev microbenchmark: what the fp64 matrix cores sustain with nothing but v_mfma_f64_16x16x4 in flight
// (N independent accumulators per wave, W waves per SIMD).  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

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

// ---- round 6: the matrix-core rows with their occupancy MEASURED.  Every workgroup records when it started and ended
// (wall_clock64, 100 MHz) and where it ran; the host counts how many were in flight at the middle of the launch.  The
// round-5 rows for "4 waves/SIMD" assumed that 1 024 workgroups of four waves are resident at once; they were not (two
// rounds of 512: wave 0's own clock said 54 clocks a slot where the wall clock said 105).  A launch's occupancy is pinned by
// its dynamic LDS request (160 KB a CU: a request of 160 / W KB lets W workgroups = W waves per SIMD in and no more).
struct TWgRecord { long long n_start, n_end, n_clocks; int n_hw_id, n_pad; };

enum { KIND_16x16x4 = 0, KIND_16x16x4_DISTINCT = 1, KIND_4x4x4 = 2 };

template <int N, int KIND>
__global__ void __launch_bounds__(256) mfma_loop(double *out, TWgRecord *p_rec, int n_iter, double a0, double b0)
{
	extern __shared__ double s_pad[];
	const long long n_w0 = wall_clock64(), n_c0 = clock64();
	double s = 0;
	if constexpr(KIND == KIND_4x4x4) {
		double acc[N], a[N], b[N];
		#pragma unroll
		for(int i = 0; i < N; ++ i) {
			acc[i] = 0;
			a[i] = a0 + threadIdx.x + i;
			b[i] = b0 - i;
		}
		for(int it = 0; it < n_iter; ++ it) {
			#pragma unroll
			for(int i = 0; i < N; ++ i)
				acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[i], acc[i], 0, 0, 0);
		}
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			s += acc[i];
	} else {
		v4f64 acc[N];
		double a[N], b[N];
		#pragma unroll
		for(int i = 0; i < N; ++ i) {
			acc[i] = v4f64{0, 0, 0, 0};
			a[i] = a0 + threadIdx.x + ((KIND == KIND_16x16x4_DISTINCT)? i : 0);
			b[i] = b0 - ((KIND == KIND_16x16x4_DISTINCT)? i : 0);
		}
		for(int it = 0; it < n_iter; ++ it) {
			#pragma unroll
			for(int i = 0; i < N; ++ i)
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
		}
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	}
	const long long n_c1 = clock64(), n_w1 = wall_clock64(); // (after the sums: they wait for the last matrix instruction)
	if(threadIdx.x == 0) {
		TWgRecord r;
		r.n_start = n_w0; r.n_end = n_w1; r.n_clocks = n_c1 - n_c0;
		r.n_hw_id = int(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))); // HW_ID (hwreg 4), all 32 bits
		r.n_pad = 0;
		p_rec[blockIdx.x] = r;
	}
	if(s == 12345.678)
		out[threadIdx.x] = s + s_pad[threadIdx.x];
}

template <int N, int KIND>
static void run(double *out, TWgRecord *d_rec, int n_waves_per_simd, const char *p_s_label)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int n_wgs = 256 * n_waves_per_simd, n_iter = (KIND == KIND_4x4x4)? 16384 : 4096;
	// the LDS request that lets n_waves_per_simd workgroups (of four waves) onto a CU and not one more
	const int n_lds = (160 * 1024 / n_waves_per_simd) / 1024 * 1024 - 1024;
	auto p_kernel = mfma_loop<N, KIND>;
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, n_lds);
	int n_occ = 0;
	(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n_occ, p_kernel, 256, n_lds);
	hipFuncAttributes t_attr;
	(void)hipFuncGetAttributes(&t_attr, reinterpret_cast<const void*>(p_kernel));
	float best = 1e9f;
	std::vector<TWgRecord> rec(n_wgs);
	for(int rep = 0; rep < 5; ++ rep) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(p_kernel, dim3(n_wgs), dim3(256), n_lds, 0, out, d_rec, n_iter, 1.0, 2.0);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		if(ms < best) {
			best = ms;
			(void)hipMemcpy(rec.data(), d_rec, sizeof(TWgRecord) * n_wgs, hipMemcpyDeviceToHost);
		}
	}
	// workgroups in flight at the middle of the launch, and the mean of their own shader-clock counts
	long long n_first = rec[0].n_start, n_last = rec[0].n_end;
	for(const TWgRecord &r : rec) { n_first = std::min(n_first, r.n_start); n_last = std::max(n_last, r.n_end); }
	const long long n_mid = (n_first + n_last) / 2;
	int n_resident = 0, n_min_clocks_wg = 0;
	double f_clocks = 0, f_wall_own = 0;
	for(const TWgRecord &r : rec) {
		n_resident += r.n_start <= n_mid && r.n_end >= n_mid;
		f_clocks += double(r.n_clocks) / n_wgs;
		f_wall_own += double(r.n_end - r.n_start) / n_wgs;
	}
	(void)n_min_clocks_wg;
	const double f_resident_per_simd = n_resident * 4.0 / 1024.0; // four waves a workgroup, 1 024 SIMDs
	const double f_flops_per_instr = (KIND == KIND_4x4x4)? 512.0 : 2048.0;
	const double f_instr_per_simd = double(n_wgs) * 4 * n_iter * N / 1024.0;
	const double f_mhz = f_clocks / (f_wall_own * 10e-3);
	// a SIMD's issue slot per matrix instruction, two ways: from the waves' own clocks (a wave's clocks / its instructions /
	// the waves that shared its SIMD) and from the launch's wall time at the clock the waves measured
	const double f_slot_own = f_clocks / (double(n_iter) * N) / f_resident_per_simd;
	const double f_slot_wall = best * 1e-3 * (f_mhz * 1e6) / f_instr_per_simd;
	printf("%-34s N=%2d  asked %d waves/SIMD, occupancy query %d WG/CU, %3d VGPRs: resident at mid-launch %4d WGs = %.2f waves/SIMD; "
		"%7.1f us, %5.1f TFLOP/s; clocks per instruction per SIMD: %6.1f by the waves' own clocks, %6.1f by the wall clock (%+.1f %%), %4.0f MHz\n",
		p_s_label, N, n_waves_per_simd, n_occ, int(t_attr.numRegs), n_resident, f_resident_per_simd, best * 1e3,
		double(n_wgs) * 4 * n_iter * N * f_flops_per_instr / best / 1e9, f_slot_own, f_slot_wall, 100 * (f_slot_wall / f_slot_own - 1), f_mhz);
}


// the shape a product kernel has: RA + RB operands in registers, RA x RB accumulators, every instruction another pair
template <int RA, int RB>
__global__ void __launch_bounds__(256) mfma_outer_loop(double *out, TWgRecord *p_rec, int n_iter, double a0, double b0)
{
	extern __shared__ double s_pad[];
	const long long n_w0 = wall_clock64(), n_c0 = clock64();
	double acc[RA][RB], a[RA], b[RB];
	#pragma unroll
	for(int i = 0; i < RA; ++ i)
		a[i] = a0 + threadIdx.x + i;
	#pragma unroll
	for(int j = 0; j < RB; ++ j)
		b[j] = b0 - j;
	#pragma unroll
	for(int i = 0; i < RA; ++ i)
		#pragma unroll
		for(int j = 0; j < RB; ++ j)
			acc[i][j] = 0;
	for(int it = 0; it < n_iter; ++ it) {
		#pragma unroll
		for(int i = 0; i < RA; ++ i)
			#pragma unroll
			for(int j = 0; j < RB; ++ j)
				acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
	}
	double s = 0;
	#pragma unroll
	for(int i = 0; i < RA; ++ i)
		#pragma unroll
		for(int j = 0; j < RB; ++ j)
			s += acc[i][j];
	const long long n_c1 = clock64(), n_w1 = wall_clock64();
	if(threadIdx.x == 0) {
		TWgRecord r;
		r.n_start = n_w0; r.n_end = n_w1; r.n_clocks = n_c1 - n_c0; r.n_hw_id = 0; r.n_pad = 0;
		p_rec[blockIdx.x] = r;
	}
	if(s == 12345.678)
		out[threadIdx.x] = s + s_pad[threadIdx.x];
}

template <int RA, int RB>
static void run_outer(double *out, TWgRecord *d_rec, int n_waves_per_simd)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int n_wgs = 256 * n_waves_per_simd, n_iter = 16384 * 8 / (RA * RB);
	const int n_lds = (160 * 1024 / n_waves_per_simd) / 1024 * 1024 - 1024;
	auto p_kernel = mfma_outer_loop<RA, RB>;
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, n_lds);
	hipFuncAttributes t_attr;
	(void)hipFuncGetAttributes(&t_attr, reinterpret_cast<const void*>(p_kernel));
	float best = 1e9f;
	for(int rep = 0; rep < 5; ++ rep) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(p_kernel, dim3(n_wgs), dim3(256), n_lds, 0, out, d_rec, n_iter, 1.0, 2.0);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
	}
	const double f_instr_per_simd = double(n_wgs) * 4 * double(n_iter) * RA * RB / 1024.0;
	printf("v_mfma_f64_4x4x4_4b, %d x %d accumulators from %d + %d operands, %d waves/SIMD, %3d VGPRs: %7.1f us, %5.1f TFLOP/s, %5.1f clocks per instruction per SIMD at 2.4 GHz\n",
		RA, RB, RA, RB, n_waves_per_simd, int(t_attr.numRegs), best * 1e3, double(n_wgs) * 4 * double(n_iter) * RA * RB * 512.0 / best / 1e9,
		best * 1e-3 * 2.4e9 / f_instr_per_simd);
}

int main()
{
	double *out; (void)hipMalloc(&out, 4096);
	TWgRecord *d_rec; (void)hipMalloc(&d_rec, sizeof(TWgRecord) * 8192);
	run<1, KIND_16x16x4>(out, d_rec, 1, "v_mfma_f64_16x16x4, one chain");
	run<4, KIND_16x16x4>(out, d_rec, 1, "v_mfma_f64_16x16x4");
	run<8, KIND_16x16x4>(out, d_rec, 1, "v_mfma_f64_16x16x4");
	run<16, KIND_16x16x4>(out, d_rec, 1, "v_mfma_f64_16x16x4");
	run<4, KIND_16x16x4>(out, d_rec, 2, "v_mfma_f64_16x16x4");
	run<8, KIND_16x16x4>(out, d_rec, 2, "v_mfma_f64_16x16x4");
	run<4, KIND_16x16x4>(out, d_rec, 4, "v_mfma_f64_16x16x4");
	run<8, KIND_16x16x4>(out, d_rec, 4, "v_mfma_f64_16x16x4");
	run<4, KIND_16x16x4>(out, d_rec, 8, "v_mfma_f64_16x16x4");
	run<8, KIND_16x16x4>(out, d_rec, 8, "v_mfma_f64_16x16x4");
	run<8, KIND_16x16x4_DISTINCT>(out, d_rec, 1, "16x16x4, distinct operands");
	run<8, KIND_16x16x4_DISTINCT>(out, d_rec, 2, "16x16x4, distinct operands");
	run<8, KIND_16x16x4_DISTINCT>(out, d_rec, 4, "16x16x4, distinct operands");
	run<8, KIND_16x16x4_DISTINCT>(out, d_rec, 8, "16x16x4, distinct operands");
	run<1, KIND_4x4x4>(out, d_rec, 1, "v_mfma_f64_4x4x4_4b, one chain");
	run<8, KIND_4x4x4>(out, d_rec, 1, "v_mfma_f64_4x4x4_4b");
	run<8, KIND_4x4x4>(out, d_rec, 2, "v_mfma_f64_4x4x4_4b");
	run<8, KIND_4x4x4>(out, d_rec, 4, "v_mfma_f64_4x4x4_4b");
	run<16, KIND_4x4x4>(out, d_rec, 4, "v_mfma_f64_4x4x4_4b");
	run<8, KIND_4x4x4>(out, d_rec, 8, "v_mfma_f64_4x4x4_4b");
	run_outer<2, 4>(out, d_rec, 1);
	run_outer<4, 4>(out, d_rec, 1);
	run_outer<4, 4>(out, d_rec, 2);
	run_outer<8, 8>(out, d_rec, 1);
	run_outer<8, 8>(out, d_rec, 2);
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
