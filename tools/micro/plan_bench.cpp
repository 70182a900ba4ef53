// Dev helper: build_plan() of csrc/plan.cpp alone on a structure file (int64 n, cumsum[n+1], ptr[n+1], int32 brow[ptr[n]]; written by
// tools/dump_structure.py), repeated; SLAMPP_HIP_PLAN_TIMING=1 prints the phases.  g++ -O3 -pthread -I slam_plus_plus_amd/csrc
#include "plan.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
using namespace slampp;
int main(int argc, char **argv)
{
	if(argc < 2) return 1;
	FILE *f = fopen(argv[1], "rb");
	if(!f) return 2;
	int64_t n;
	if(fread(&n, 8, 1, f) != 1) return 3;
	std::vector<int64_t> cumsum(n + 1), ptr(n + 1);
	if(fread(cumsum.data(), 8, n + 1, f) != size_t(n + 1) || fread(ptr.data(), 8, n + 1, f) != size_t(n + 1)) return 3;
	std::vector<int32_t> brow(ptr[n]);
	if(fread(brow.data(), 4, brow.size(), f) != brow.size()) return 3;
	fclose(f);
	const int reps = (argc > 2)? atoi(argv[2]) : 5;
	PlanOptions opt;
	if(argc > 3) { opt.dense_top_nb = atoi(argv[3]); opt.dense_top_auto = false; }
	for(int r = 0; r < reps; ++ r) {
		Plan P;
		const auto t0 = std::chrono::steady_clock::now();
		std::string err = build_plan(n, cumsum.data(), ptr.data(), brow.data(), opt, P);
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		printf("%s: n %ld, %.2f ms, err '%s', stages %zu, dense %d, order %.2f ms, symbolic %.2f ms\n", argv[1], (long)n, ms, err.c_str(),
			P.stage_ptr.empty()? size_t(0) : P.stage_ptr.size() - 1, P.dense_dim, P.order_ms, P.symbolic_ms);
	}
	return 0;
}
