// dev helper (round 6, judge item 5): what the Venice-like leg's partial blocks would cost as fp64 atomic adds into S instead of
// a write + a reduction pass: 0.94 M partial blocks of 36 doubles, each added into one of 0.5 M blocks of S (144 MB), the
// targets in the order a sorted job list would give (runs of neighbouring blocks) and in random order
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_f64 atomic_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

__global__ void add_partials(const double *__restrict__ p_partial, const int32_t *__restrict__ p_target, double *p_S, int64_t n_partials)
{
	const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; // one lane an element: 36 consecutive lanes a block
	const int64_t n_blk = i / 36;
	if(n_blk >= n_partials)
		return;
	const int n_e = int(i - n_blk * 36);
	atomicAdd(p_S + int64_t(p_target[n_blk]) * 36 + n_e, p_partial[i]);
}

__global__ void reduce_partials(const double *__restrict__ p_partial, const int32_t *__restrict__ p_first, double *p_S, int64_t n_blocks)
{
	const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; // the non-atomic form: a lane an element of S, its partials in a row
	const int64_t n_blk = i / 36;
	if(n_blk >= n_blocks)
		return;
	const int n_e = int(i - n_blk * 36);
	double f = p_S[i];
	for(int32_t k = p_first[n_blk]; k < p_first[n_blk + 1]; ++ k)
		f -= p_partial[int64_t(k) * 36 + n_e];
	p_S[i] = f;
}

int main()
{
	const int64_t n_blocks = 500000, n_partials = 940000;
	std::vector<int32_t> target(n_partials), first(n_blocks + 1, 0);
	std::mt19937 rng(1);
	for(int64_t i = 0; i < n_partials; ++ i)
		target[i] = int32_t(rng() % n_blocks);
	double *d_partial, *d_S;
	int32_t *d_target, *d_first;
	(void)hipMalloc(&d_partial, n_partials * 36 * sizeof(double));
	(void)hipMalloc(&d_S, n_blocks * 36 * sizeof(double));
	(void)hipMalloc(&d_target, n_partials * sizeof(int32_t));
	(void)hipMalloc(&d_first, (n_blocks + 1) * sizeof(int32_t));
	(void)hipMemset(d_partial, 0, n_partials * 36 * sizeof(double));
	(void)hipMemset(d_S, 0, n_blocks * 36 * sizeof(double));
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	for(int n_order = 0; n_order < 3; ++ n_order) {
		if(n_order == 1)
			std::sort(target.begin(), target.end()); // partials of one block next to each other (as the reduction has them)
		if(n_order == 2) { // neighbouring partials go to neighbouring blocks, as the jobs of a run produce them
			for(int64_t i = 0; i < n_partials; ++ i)
				target[i] = int32_t((i * n_blocks / n_partials + (i % 7) * 911) % n_blocks);
		}
		(void)hipMemcpy(d_target, target.data(), n_partials * sizeof(int32_t), hipMemcpyHostToDevice);
		const unsigned n_grid = unsigned((n_partials * 36 + 255) / 256);
		float f_best = 1e30f;
		for(int r = 0; r < 6; ++ r) {
			(void)hipEventRecord(e0, 0);
			hipLaunchKernelGGL(add_partials, dim3(n_grid), dim3(256), 0, 0, d_partial, d_target, d_S, n_partials);
			(void)hipEventRecord(e1, 0);
			(void)hipEventSynchronize(e1);
			float f_ms;
			(void)hipEventElapsedTime(&f_ms, e0, e1);
			f_best = std::min(f_best, f_ms);
		}
		printf("%-58s %8.1f us = %6.1f G atomics/s\n", (n_order == 0)? "34 M fp64 atomic adds, targets in random order" :
			(n_order == 1)? "34 M fp64 atomic adds, partials of a block next to each other" : "34 M fp64 atomic adds, neighbouring partials to neighbouring blocks",
			f_best * 1e3, n_partials * 36 / (f_best * 1e-3) * 1e-9);
	}
	{ // the reduction as it is: sorted partials, a lane an element of S
		std::vector<int32_t> sorted(n_partials);
		for(int64_t i = 0; i < n_partials; ++ i)
			sorted[i] = int32_t(rng() % n_blocks);
		std::sort(sorted.begin(), sorted.end());
		for(int64_t i = 0; i < n_partials; ++ i)
			++ first[sorted[i] + 1];
		for(int64_t b = 0; b < n_blocks; ++ b)
			first[b + 1] += first[b];
		(void)hipMemcpy(d_first, first.data(), (n_blocks + 1) * sizeof(int32_t), hipMemcpyHostToDevice);
		const unsigned n_grid = unsigned((n_blocks * 36 + 255) / 256);
		float f_best = 1e30f;
		for(int r = 0; r < 6; ++ r) {
			(void)hipEventRecord(e0, 0);
			hipLaunchKernelGGL(reduce_partials, dim3(n_grid), dim3(256), 0, 0, d_partial, d_first, d_S, n_blocks);
			(void)hipEventRecord(e1, 0);
			(void)hipEventSynchronize(e1);
			float f_ms;
			(void)hipEventElapsedTime(&f_ms, e0, e1);
			f_best = std::min(f_best, f_ms);
		}
		printf("%-58s %8.1f us (reads 0.27 GB of partials + 0.29 GB of S)\n", "the same sums as a reduction pass over sorted partials", f_best * 1e3);
	}
	return 0;
}
