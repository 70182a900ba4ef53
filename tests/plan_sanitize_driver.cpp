// Test infrastructure (tests/test_plan_sanitizers.py): the host planner (csrc/plan.cpp: nested dissection with its threads, block
// symbolic factorization, schedule, plan search) on assorted graphs, built with AddressSanitizer + UBSan / ThreadSanitizer on the CPU.
#include "plan.h"
#include <cstdio>
#include <random>
#include <set>
using namespace slampp;
static void run(const char *name, int n, const std::vector<std::pair<int,int>> &edges, int d, int dense_nb)
{
	std::vector<std::set<int>> cols(n);
	for(int i = 0; i < n; ++ i) cols[i].insert(i);
	for(auto &e : edges) { int a = std::min(e.first, e.second), b = std::max(e.first, e.second); if(a != b) cols[b].insert(a); }
	std::vector<int64_t> cumsum(n + 1), ptr(n + 1, 0); std::vector<int32_t> brow;
	for(int i = 0; i <= n; ++ i) cumsum[i] = int64_t(i) * d;
	for(int c = 0; c < n; ++ c) { for(int r : cols[c]) brow.push_back(r); ptr[c + 1] = int64_t(brow.size()); }
	PlanOptions opt; if(dense_nb >= 0) { opt.dense_top_nb = dense_nb; } 
	Plan P;
	std::string err = build_plan(n, cumsum.data(), ptr.data(), brow.data(), opt, P);
	printf("%s: n %d err '%s' stages %zu dense %d\n", name, n, err.c_str(), P.stage_ptr.size() ? P.stage_ptr.size() - 1 : 0, P.dense_dim);
}
int main()
{
	std::mt19937 rng(5);
	{ // chain with loop closures
		int n = 12000; std::vector<std::pair<int,int>> e;
		for(int i = 1; i < n; ++ i) e.push_back({i - 1, i});
		for(int i = 60; i < n; i += 50) e.push_back({i, i - 26 - int(rng() % 30)});
		run("chain", n, e, 6, -1);
	}
	{ // grid
		int w = 60, n = w * w; std::vector<std::pair<int,int>> e;
		for(int y = 0; y < w; ++ y) for(int x = 0; x < w; ++ x) { if(x) e.push_back({y * w + x - 1, y * w + x}); if(y) e.push_back({(y - 1) * w + x, y * w + x}); }
		run("grid", n, e, 3, -1);
	}
	{ // random sparse + disconnected pieces + isolated vertices
		int n = 5000; std::vector<std::pair<int,int>> e;
		for(int i = 0; i < 9000; ++ i) { int a = rng() % 3000, b = rng() % 3000; e.push_back({a, b}); }
		for(int i = 3001; i < 4000; ++ i) e.push_back({i - 1, i});
		run("random+chain+isolated", n, e, 6, -1);
	}
	{ // band
		int n = 2000; std::vector<std::pair<int,int>> e;
		for(int k = 1; k <= 3; ++ k) for(int i = k; i < n; ++ i) e.push_back({i - k, i});
		run("band", n, e, 6, -1);
	}
	{ int n = 1; run("one", n, {}, 6, -1); }
	{ int n = 70; std::vector<std::pair<int,int>> e; for(int i = 0; i < n; ++ i) for(int j = 0; j < i; ++ j) e.push_back({j, i}); run("clique", n, e, 6, -1); }
	return 0;
}
