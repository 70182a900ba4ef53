// build: g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Islam_plus_plus_amd/csrc tools/plan_sanitize.cpp slam_plus_plus_amd/csrc/plan.cpp -o /tmp/plan_sanitize && /tmp/plan_sanitize   (clean on round 1: no reports)
// ASAN/UBSAN driver of the host planner (plan.cpp) on random block graphs: chains with loops, grids, BA-like, tiny
#include "plan.h"
#include <cstdio>
#include <random>
#include <set>
#include <algorithm>
using namespace slampp;
static void run(int n, const std::vector<std::pair<int,int>> &edges, const std::vector<int> &dims, PlanOptions opt, const char *name)
{
	std::vector<std::set<int>> colrows(n);
	for(int i = 0; i < n; ++ i) colrows[i].insert(i);
	for(auto &e : edges) { int r = std::min(e.first, e.second), c = std::max(e.first, e.second); if(r != c) colrows[c].insert(r); }
	std::vector<int64_t> cs(n + 1, 0), ptr(n + 1, 0);
	std::vector<int32_t> brow;
	for(int i = 0; i < n; ++ i) { cs[i + 1] = cs[i] + dims[i]; for(int r : colrows[i]) brow.push_back(r); ptr[i + 1] = int64_t(brow.size()); }
	Plan plan;
	std::string err = build_plan(n, cs.data(), ptr.data(), brow.data(), opt, plan);
	std::vector<char> nz;
	int T = plan.dense_dim? dense_top_tile_pattern(plan, nz) : 0;
	if(T) tile_symbolic(T, nz);
	printf("%-28s n=%d err='%s' lblocks=%zu stages=%zu dense=%d chain=%.0fus\n", name, n, err.c_str(), plan.lrow.size(),
		plan.stage_ptr.empty()? 0 : plan.stage_ptr.size() - 1, plan.dense_dim, err.empty()? plan_chain_estimate_us(plan) : 0.0);
}
int main()
{
	std::mt19937_64 rng(7);
	for(int rep = 0; rep < 3; ++ rep) {
		for(int n : {1, 2, 3, 17, 400, 5000}) {
			std::vector<std::pair<int,int>> e;
			for(int i = 1; i < n; ++ i) e.push_back({i - 1, i});
			for(int i = 0; i < n / 20; ++ i) { int a = rng() % n, b = rng() % n; e.push_back({a, b}); }
			std::vector<int> d6(n, 6), dmix(n);
			for(int i = 0; i < n; ++ i) dmix[i] = (rng() % 3 == 0)? 3 : 6;
			PlanOptions o;
			run(n, e, d6, o, "chain+random loops");
			run(n, e, dmix, o, "mixed dims");
			o.natural_order = true; o.dense_top_nb = 0;
			run(n, e, d6, o, "natural order");
			PlanOptions o2; o2.subtree_size = 1 + rng() % 20; o2.leaf_size = 1 + rng() % 8; o2.dense_top_nb = 4 + rng() % 30; o2.dense_top_auto = false;
			run(n, e, d6, o2, "random options");
		}
		for(int w : {2, 13, 40}) { // grids: big separators, dense top
			std::vector<std::pair<int,int>> e;
			for(int y = 0; y < w; ++ y) for(int x = 0; x < w; ++ x) { if(x + 1 < w) e.push_back({y * w + x, y * w + x + 1}); if(y + 1 < w) e.push_back({y * w + x, (y + 1) * w + x}); }
			PlanOptions o;
			run(w * w, e, std::vector<int>(w * w, 6), o, "grid");
			o.dense_top_align = 0;
			run(w * w, e, std::vector<int>(w * w, 3), o, "grid packed dense top");
			o.dense_top_max_dim = 256;
			run(w * w, e, std::vector<int>(w * w, 6), o, "grid capped dense top");
		}
		{ // reduced-camera-system-like: band with wrap-around
			int n = 300; std::vector<std::pair<int,int>> e;
			for(int i = 0; i < n; ++ i) for(int j = 1; j <= 3; ++ j) e.push_back({i, (i + 7 * j) % n});
			run(n, e, std::vector<int>(n, 6), PlanOptions(), "circulant band");
		}
		{ // complete graph
			int n = 40; std::vector<std::pair<int,int>> e;
			for(int i = 0; i < n; ++ i) for(int j = 0; j < i; ++ j) e.push_back({i, j});
			run(n, e, std::vector<int>(n, 7), PlanOptions(), "complete");
		}
		{ // disconnected + isolated vertices
			int n = 50; std::vector<std::pair<int,int>> e;
			for(int i = 1; i < 20; ++ i) e.push_back({i - 1, i});
			for(int i = 31; i < 45; ++ i) e.push_back({i - 1, i});
			run(n, e, std::vector<int>(n, 6), PlanOptions(), "disconnected");
		}
	}
	return 0;
}
