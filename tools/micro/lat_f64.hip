// Dev microbenchmark: cycles per instruction of the sequences the 64 x 64 diagonal-tile panel is made of
// (build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/lat_f64 tools/micro/lat_f64.hip; one wave, s_memtime around 256 repeats)
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ void lat_kernel(double *p_out, long long *p_cycles, double x0)
{
	const int lane = threadIdx.x;
	double a = x0 + lane * 1e-3, b = 1.0 + lane * 1e-6, c = 0.5;
	long long t0, t1;
	// 0: dependent v_fma_f64
	t0 = clock64();
	REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
	t1 = clock64();
	if(lane == 0) p_cycles[0] = t1 - t0;
	// 1: independent v_fma_f64 (4 accumulators)
	double a0 = a, a1 = a + 1, a2 = a + 2, a3 = a + 3;
	t0 = clock64();
	REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
		: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
	t1 = clock64();
	if(lane == 0) p_cycles[1] = t1 - t0;
	a = a0 + a1 + a2 + a3;
	// 2: dependent v_rcp_f64
	t0 = clock64();
	REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
	t1 = clock64();
	if(lane == 0) p_cycles[2] = t1 - t0;
	// 3: readlane pair -> fma with the scalar (independent of one another: the fma accumulators differ)
	t0 = clock64();
	REP16(asm volatile("v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_fma_f64 %0, s[20:21], %6, %0\n"
		"v_readlane_b32 s22, %4, 5\n v_readlane_b32 s23, %5, 5\n v_fma_f64 %1, s[22:23], %6, %1\n"
		"v_readlane_b32 s24, %4, 7\n v_readlane_b32 s25, %5, 7\n v_fma_f64 %2, s[24:25], %6, %2\n"
		"v_readlane_b32 s26, %4, 9\n v_readlane_b32 s27, %5, 9\n v_fma_f64 %3, s[26:27], %6, %3"
		: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(__double2loint(b)), "v"(__double2hiint(b)), "v"(c)
		: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
	t1 = clock64();
	if(lane == 0) p_cycles[3] = t1 - t0; // 64 x (2 readlane + fma)
	// 4: the same with all readlanes first, then the fmas (8 + 4 per group)
	t0 = clock64();
	REP16(asm volatile("v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %4, 5\n v_readlane_b32 s23, %5, 5\n"
		"v_readlane_b32 s24, %4, 7\n v_readlane_b32 s25, %5, 7\n v_readlane_b32 s26, %4, 9\n v_readlane_b32 s27, %5, 9\n"
		"v_fma_f64 %0, s[20:21], %6, %0\n v_fma_f64 %1, s[22:23], %6, %1\n v_fma_f64 %2, s[24:25], %6, %2\n v_fma_f64 %3, s[26:27], %6, %3"
		: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(__double2loint(b)), "v"(__double2hiint(b)), "v"(c)
		: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
	t1 = clock64();
	if(lane == 0) p_cycles[4] = t1 - t0;
	// 5: chain readlane -> fma -> readlane of the result ... (the pivot hand-over)
	double u = a0;
	t0 = clock64();
	#pragma unroll
	for(int i = 0; i < 16; ++ i) {
		const int lo = __builtin_amdgcn_readlane(__double2loint(u), 3), hi = __builtin_amdgcn_readlane(__double2hiint(u), 3);
		u = __builtin_fma(__hiloint2double(hi, lo), c, u);
	}
	t1 = clock64();
	if(lane == 0) p_cycles[5] = t1 - t0;
	// 6: LDS broadcast instead of readlane: ds_write_b64 own value, ds_read_b64 of one address, fma (dependent chain)
	__shared__ double s_x[64];
	double v = a1;
	t0 = clock64();
	for(int i = 0; i < 16; ++ i) {
		s_x[lane] = v;
		__builtin_amdgcn_s_waitcnt(0xc07f);
		const double m = s_x[i];
		v = __builtin_fma(m, c, v);
	}
	t1 = clock64();
	if(lane == 0) p_cycles[6] = t1 - t0; // 16 round trips
	// 7: dependent rcp + two Newton steps + mul (the chain of one column step without the broadcasts)
	double pv = a2;
	t0 = clock64();
	for(int i = 0; i < 16; ++ i) {
		double rw = __builtin_amdgcn_rcp(pv);
		rw = __builtin_fma(__builtin_fma(-pv, rw, 1.0), rw, rw);
		rw = __builtin_fma(__builtin_fma(-pv, rw, 1.0), rw, rw);
		pv = __builtin_fma(pv, rw, c); // next "pivot" depends on it
	}
	t1 = clock64();
	if(lane == 0) p_cycles[7] = t1 - t0;
	// 8: one DPP move (row_shr:1) + fma, dependent
	int w = __double2loint(a3);
	t0 = clock64();
	REP64(w = __builtin_amdgcn_update_dpp(0, w, 0x111, 0xf, 0xf, false);)
	t1 = clock64();
	if(lane == 0) p_cycles[8] = t1 - t0;
	// 9: ds_bpermute dependent chain
	t0 = clock64();
	REP16(w = __builtin_amdgcn_ds_bpermute(12, w);)
	t1 = clock64();
	if(lane == 0) p_cycles[9] = t1 - t0;
	p_out[lane] = a + a0 + a1 + a2 + a3 + u + v + pv + w;
}

int main()
{
	double *p_out; long long *p_cycles;
	hipMalloc(&p_out, 64 * sizeof(double)); hipMalloc(&p_cycles, 16 * sizeof(long long));
	for(int rep = 0; rep < 2; ++ rep)
		hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(64), 0, 0, p_out, p_cycles, 1.5);
	long long h[16];
	hipMemcpy(h, p_cycles, sizeof(h), hipMemcpyDeviceToHost);
	const char *names[] = {"dependent v_fma_f64", "independent v_fma_f64", "dependent v_rcp_f64", "(2 readlane + fma), interleaved",
		"(2 readlane + fma), grouped by 4", "readlane -> fma -> readlane chain", "LDS write + broadcast read + fma chain",
		"rcp + 2 Newton + fma chain", "dependent DPP move", "dependent ds_bpermute"};
	const int counts[] = {64, 64, 64, 64, 64, 16, 16, 16, 64, 16};
	for(int i = 0; i < 10; ++ i)
		printf("%-42s %6lld cycles / %d = %.1f each\n", names[i], h[i], counts[i], double(h[i]) / counts[i]);
	return 0;
}
