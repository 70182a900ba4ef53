// Dev helper: which lanes' operands meet in which lane's result of v_mfma_f64_4x4x4_4b (four blocks of D = A B, 4 x 4 x 4 each):
// one-hot a in lane la, one-hot b in lane lb -> the lane whose result is 1 (or none).  Prints the layout it infers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out)
{
	const int lane = threadIdx.x;
	for(int la = 0; la < 64; ++ la) {
		for(int lb = 0; lb < 64; ++ lb) {
			const double a = (lane == la)? 1.0 : 0.0, b = (lane == lb)? 1.0 : 0.0;
			const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
			const unsigned long long m = __ballot(d != 0.0);
			if(lane == 0)
				out[la * 64 + lb] = m? (__popcll(m) == 1? __ffsll((long long)m) - 1 : -2) : -1;
		}
	}
}
int main()
{
	int *d_out; (void)hipMalloc(&d_out, 4096 * sizeof(int));
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out);
	std::vector<int> out(4096);
	(void)hipMemcpy(out.data(), d_out, 4096 * sizeof(int), hipMemcpyDeviceToHost);
	// for every a lane: the b lanes it meets and where the product lands
	for(int la = 0; la < 64; ++ la) {
		printf("a lane %2d meets b lanes:", la);
		for(int lb = 0; lb < 64; ++ lb) {
			if(out[la * 64 + lb] >= 0)
				printf(" %d->d%d", lb, out[la * 64 + lb]);
			else if(out[la * 64 + lb] == -2)
				printf(" %d->many", lb);
		}
		printf("\n");
	}
	return 0;
}
