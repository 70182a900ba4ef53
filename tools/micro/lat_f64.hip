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
	// 5: dependent v_mov_b64_dpp row_newbcast
	double u = a0;
	t0 = clock64();
	REP64(asm volatile("s_nop 1\n v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(u));)
	t1 = clock64();
	if(lane == 0) p_cycles[5] = t1 - t0;
	// 6: independent v_fmac_f64_dpp (4 accumulators)
	t0 = clock64();
	REP16(asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
		"v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf"
		: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
	t1 = clock64();
	if(lane == 0) p_cycles[6] = t1 - t0;
	// 7: dependent v_fmac_f64_dpp (the accumulator is the next one's broadcast source)
	double v = a1;
	t0 = clock64();
	REP64(asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v) : "v"(c));)
	t1 = clock64();
	if(lane == 0) p_cycles[7] = t1 - t0;
	// 8: the chain of one column step: bcast -> cmp/select -> rcp -> 2 Newton -> mul -> fmac_dpp (x16)
	double pv = a2 + 2.0, w = 0.0;
	t0 = clock64();
	REP16(asm volatile("s_nop 1\n v_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
		"v_max_f64 %1, %1, %1\n v_max_f64 %1, %1, %1\n"
		"v_rcp_f64 %2, %1\n v_fma_f64 %1, -%0, %2, 1.0\n v_fmac_f64 %2, %1, %2\n v_fma_f64 %1, -%0, %2, 1.0\n v_fmac_f64 %2, %1, %2\n"
		"v_mul_f64 %1, %0, -%2\n s_nop 1\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf"
		: "+v"(pv), "=&v"(w), "=&v"(u));)
	t1 = clock64();
	if(lane == 0) p_cycles[8] = t1 - t0;
	// 9: LDS write + poll from the same wave (round trip seen by a consumer is at least this)
	__shared__ double s_x[64];
	t0 = clock64();
	for(int i = 0; i < 16; ++ i) {
		((__attribute__((address_space(3))) volatile double*)s_x)[lane] = v + i;
		v += ((__attribute__((address_space(3))) volatile double*)s_x)[i];
	}
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
		"(2 readlane + fma), grouped by 4", "dependent v_mov_b64_dpp (+ s_nop 1)", "independent v_fmac_f64_dpp",
		"dependent v_fmac_f64_dpp (+ s_nop 1)", "column step chain (bcast..fmac_dpp)", "LDS volatile write + read round trip"};
	const int counts[] = {64, 64, 64, 64, 64, 64, 64, 64, 16, 16};
	for(int i = 0; i < 10; ++ i)
		printf("%-42s %6lld cycles / %d = %.1f each\n", names[i], h[i], counts[i], double(h[i]) / counts[i]);
	return 0;
}
