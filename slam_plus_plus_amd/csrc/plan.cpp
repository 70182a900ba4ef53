// plan.cpp -- ordering + symbolic factorization + schedule, see plan.h
#include "plan.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>

namespace slampp {

namespace {

double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// nested dissection on level structures
// ---------------------------------------------------------------------------------------------

// Sub-problems are independent (disjoint vertex sets, disjoint output ranges known up front), so the
// two halves of a large bisection run on separate threads down to a fixed depth: the ordering is the
// cold path's largest host cost and the hosts of MI355X boxes have cores to spare.
//
// Round 6: no allocation proportional to a subset.  A subset is a range [b, e) of ONE array of vertices (m_verts), which
// ends up being the ordering: a cut rearranges its range into lower part | upper part | separator, stably, through a
// scratch array of the same size, and recurses into the first two; the breadth-first queue of a subset is the same range
// of a third array, the difference array of the cut by number the same range of a fourth.  The recursion used to build
// four or five std::vectors per subset -- megabytes each at the top levels, so every one of them an mmap and a munmap,
// which take the address space's lock against the page faults of every other thread of the process (the other halves of
// the recursion, the rest of the analysis, the caller's own threads): C3's ordering was 12 ms in a quiet process and 22 in
// one whose heap had been used (the reference's own solver run before ours).  Same cuts, same ordering as before.
class CNestedDissection {
	const int32_t m_n;
	const std::vector<int64_t> &m_ptr;
	const std::vector<int32_t> &m_adj;
	const int m_leaf;
	const int m_n_balance_pct; // a separator must leave at least this share of the vertices on either side
	const bool m_b_other_bank; // the separator of a level cut: the narrower of its two banks (PlanOptions::nd_other_bank)
	const int m_n_cut_min_size, m_n_cut_max_sep; // development knobs of the cut by vertex number, read once (plan.h)
	std::vector<int32_t> m_set;    // id of the subset a vertex currently belongs to
	std::vector<int32_t> m_level;  // BFS level (valid for the subset being processed)
	std::vector<int32_t> &m_verts; // the subsets, range by range; in the end perm[new] = old
	std::vector<int32_t> m_scratch, m_queue, m_diff; // per-range work space (see above)
	std::atomic<int32_t> m_next_id;
	// halves of at least this many vertices each run side by side, down to this depth of the recursion (threads start in
	// ~0.1 ms on the hosts of MI355X boxes: a few thousand vertices are worth one; a small graph -- the 2-D-like ones of the
	// configs, a reduced camera system -- is ordered several times over by the plan search, side by side as well)
	const size_t parallel_min_size;
	const int parallel_max_depth;

public:
	CNestedDissection(int32_t n, const std::vector<int64_t> &ptr, const std::vector<int32_t> &adj,
		int leaf, int n_balance_pct, std::vector<int32_t> &out, bool b_other_bank = false)
		:m_n(n), m_ptr(ptr), m_adj(adj), m_leaf(std::max(leaf, 1)), m_n_balance_pct(std::min(std::max(n_balance_pct, 1), 49)),
		m_b_other_bank(b_other_bank),
		m_n_cut_min_size(dev_knob("SLAMPP_HIP_DEV_ND_MIN", 64)), m_n_cut_max_sep(dev_knob("SLAMPP_HIP_DEV_ND_SEP", 4)),
		m_set(n, -1), m_level(n, -1),
		m_verts(out), m_scratch(n), m_queue(n), m_diff(n, 0), m_next_id(0),
		parallel_min_size(size_t(std::max(dev_knob("SLAMPP_HIP_DEV_ND_PAR_MIN", (n <= 8192)? 768 : 2048), 16))),
		parallel_max_depth(dev_knob("SLAMPP_HIP_DEV_ND_PAR_DEPTH", (n <= 8192)? 2 : 6))
	{
		m_verts.resize(n);
		std::iota(m_verts.begin(), m_verts.end(), 0);
	}

	void Run()
	{
		Order(0, size_t(m_n), 0);
	}

private:
	// BFS inside subset `id` from `root`; fills p_queue with the visit order and m_level; returns #levels (r_n_queue: #visited)
	int32_t BFS(int32_t root, int32_t id, int32_t *p_queue, size_t &r_n_queue)
	{
		size_t n_tail = 0;
		p_queue[n_tail ++] = root;
		m_level[root] = 0;
		int32_t n_levels = 1;
		for(size_t h = 0; h < n_tail; ++ h) {
			const int32_t v = p_queue[h], lv = m_level[v];
			for(int64_t e = m_ptr[v]; e < m_ptr[v + 1]; ++ e) {
				const int32_t w = m_adj[e];
				if(m_set[w] == id && m_level[w] < 0) {
					m_level[w] = lv + 1;
					n_levels = lv + 2;
					p_queue[n_tail ++] = w;
				}
			}
		}
		r_n_queue = n_tail;
		return n_levels;
	}

	void Reset_Levels(size_t b, size_t e)
	{
		for(size_t k = b; k < e; ++ k)
			m_level[m_verts[k]] = -1;
	}

	int32_t Degree_In(int32_t v, int32_t id) const
	{
		int32_t d = 0;
		for(int64_t e = m_ptr[v]; e < m_ptr[v + 1]; ++ e)
			d += (m_set[m_adj[e]] == id);
		return d;
	}

	// the two parts of a cut, side by side where they are large
	void Order_Both(size_t b, size_t n_lower, size_t n_upper, int n_depth)
	{
		if(n_depth < parallel_max_depth && std::min(n_lower, n_upper) >= parallel_min_size) {
			std::thread other([&]() { Order(b + n_lower, b + n_lower + n_upper, n_depth + 1); });
			Order(b, b + n_lower, n_depth + 1);
			other.join();
		} else {
			Order(b, b + n_lower, n_depth + 1);
			Order(b + n_lower, b + n_lower + n_upper, n_depth + 1);
		}
	}

	// orders the (possibly disconnected) vertex subset m_verts[b, e) in place
	void Order(size_t b, size_t e, int n_depth)
	{
		if(b >= e)
			return;
		const int32_t id = m_next_id ++;
		for(size_t k = b; k < e; ++ k) {
			const int32_t v = m_verts[k];
			m_set[v] = id;
			m_level[v] = -1;
		}
		const size_t n_size = e - b;
		TIndexCut t_cut;
		const TIndexCut *p_cut = 0; // a cut by number that stands unless the level structure finds a narrower one
		if(Find_Index_Cut(b, e, id, t_cut)) {
			enum { n_small = 256 };
			if(n_size >= size_t(n_small) || t_cut.n_sep <= 2) {
				Apply_Index_Cut(b, e, id, n_depth, t_cut);
				return;
			}
			p_cut = &t_cut;
		}
		// split into connected components first (iteratively, flat storage: the landmark part
		// of a BA system has 500k single-vertex components), then order each one
		int32_t *p_queue = &m_queue[b];
		size_t n_queue = 0, n_comp_verts = 0;
		std::vector<size_t> comp_ptr(1, 0);
		for(size_t k = b; k < e && n_comp_verts < n_size; ++ k) {
			if(m_level[m_verts[k]] >= 0)
				continue; // already in an earlier component
			BFS(m_verts[k], id, p_queue, n_queue);
			if(n_queue == n_size) {
				Order_Connected(b, e, id, n_depth, p_cut); // the whole subset is one component (its traversal is in the queue)
				return;
			}
			std::copy(p_queue, p_queue + n_queue, m_scratch.begin() + b + n_comp_verts); // (the components one behind the other)
			n_comp_verts += n_queue;
			comp_ptr.push_back(n_comp_verts);
		}
		std::copy(m_scratch.begin() + b, m_scratch.begin() + e, m_verts.begin() + b);
		for(size_t c = 0; c + 1 < comp_ptr.size(); ++ c) {
			const size_t cb = b + comp_ptr[c], ce = b + comp_ptr[c + 1];
			if(ce - cb == 1)
				continue; // (in place already)
			const int32_t cid = m_next_id ++; // own id: the recursion must not see the other components
			for(size_t k = cb; k < ce; ++ k)
				m_set[m_verts[k]] = cid;
			Reset_Levels(cb, ce);
			BFS(m_verts[cb], cid, &m_queue[cb], n_queue);
			Order_Connected(cb, ce, cid, n_depth, 0);
		}
	}

	// A cut by vertex number, tried before any traversal (round 4: the ordering was the largest part of C3's first call --
	// four to five breadth-first passes per level of the recursion, and at the top levels of a chain-like graph a frontier
	// is one to three vertices: nothing for threads to share).  The vertices of a pose graph are numbered along the
	// trajectory: the vertices below some number against those above it are a cut, and where few edges cross it -- the
	// odometry edge and the loop closures that span it -- the lower endpoints of those edges are a separator.  Every
	// position of the middle third is priced in one pass over the subset's edges (no dependence between vertices) and
	// the narrowest taken, the one nearest the middle among equals.  It also turned out the better ordering, not only
	// the cheaper one: level structures from a peripheral vertex of a chain with loop closures give lopsided halves
	// (C3: elimination tree 30 -> 20 levels, 10 -> 8 stages, 0.373 -> 0.333 ms a solve; a chain with a loop closure every
	// ten poses: 804 -> 301 Mflop).  Taken only if it is narrow (max_sep vertices): a graph whose numbering says nothing
	// about its shape (a grid numbered row by row: the cut is a whole row) falls through to the level structures; and
	// below n_small vertices a cut of three or four only stands if it is narrower than what the level structure finds (the
	// reduced camera system of the band-visibility BA leg: same width either way, but the cut by number left tasks of
	// seven columns where the level structure leaves four, and its stages took 6 % longer).
	struct TIndexCut {
		size_t n_half; // position (in the subset) of the first vertex of the upper part
		int32_t n_sep; // vertices of the lower part with a neighbour in the upper part
	};

	bool Find_Index_Cut(size_t b, size_t e, int32_t id, TIndexCut &r_cut)
	{
		const int min_size = m_n_cut_min_size, max_sep = m_n_cut_max_sep;
		const size_t n_size = e - b;
		const int32_t *S = &m_verts[b];
		if(n_size < size_t(min_size) || n_size <= size_t(m_leaf) * 4 || !std::is_sorted(S, S + n_size))
			return false;
		// how many vertices below position p have a neighbour at or above it, for every p of the middle third: a vertex at
		// position k whose highest neighbour sits at position r counts for p in (k, r] -- a difference array (entry p - 1 of
		// the subset's range of m_diff stands for position p: p runs from 1 to the subset's size).  m_level is free here
		// (all -1): it holds the positions for the pass
		for(size_t k = 0; k < n_size; ++ k)
			m_level[S[k]] = int32_t(k);
		const size_t n_lo = n_size * 35 / 100, n_hi = n_size - n_lo; // candidates: n_lo < p <= n_hi
		int32_t *diff0 = &m_diff[b]; // diff0[p - 1]: position p = 1 .. n_size
		std::fill(diff0, diff0 + n_size, 0);
		for(size_t k = 0; k < n_hi; ++ k) {
			const int32_t v = S[k];
			int32_t r = -1;
			for(int64_t ed = m_ptr[v]; ed < m_ptr[v + 1]; ++ ed) {
				const int32_t w = m_adj[ed];
				if(w > v && m_set[w] == id)
					r = std::max(r, m_level[w]);
			}
			if(r > int32_t(k)) {
				++ diff0[k]; // position k + 1
				if(size_t(r) + 1 <= n_size)
					-- diff0[size_t(r)]; // position r + 1 (beyond the last position there is nothing to end)
			}
		}
		for(size_t k = 0; k < n_size; ++ k)
			m_level[S[k]] = -1;
		size_t n_half = 0;
		int32_t n_best = INT32_MAX, n_run = 0;
		for(size_t p_ = 1; p_ <= n_hi; ++ p_) {
			n_run += diff0[p_ - 1];
			if(p_ > n_lo) {
				const size_t n_off = (p_ > n_size / 2)? p_ - n_size / 2 : n_size / 2 - p_;
				const size_t n_best_off = (n_half > n_size / 2)? n_half - n_size / 2 : n_size / 2 - n_half;
				if(n_run < n_best || (n_run == n_best && n_off < n_best_off)) {
					n_best = n_run;
					n_half = p_;
				}
			}
		}
		if(n_best > max_sep || n_best <= 0)
			return false; // too wide for a cut found without looking at the graph's shape -- or no edge at all across it (two
			// components, or more: the traversal sorts them out)
		r_cut.n_half = n_half;
		r_cut.n_sep = n_best;
		return true;
	}

	void Apply_Index_Cut(size_t b, size_t e, int32_t id, int n_depth, const TIndexCut &r_cut)
	{
		const size_t n_half = r_cut.n_half;
		const int32_t n_mid = m_verts[b + n_half]; // lower: numbers below n_mid
		// lower | upper | separator through the scratch range (stable: every part stays in ascending order)
		size_t n_lower = 0, n_sep = 0;
		int32_t *p_out = &m_scratch[b], *p_sep = &m_queue[b]; // (the queue's range is free here)
		for(size_t k = b; k < b + n_half; ++ k) {
			const int32_t v = m_verts[k];
			bool b_sep = false;
			for(int64_t ed = m_ptr[v]; ed < m_ptr[v + 1] && !b_sep; ++ ed) {
				const int32_t w = m_adj[ed];
				b_sep = w >= n_mid && m_set[w] == id;
			}
			if(b_sep)
				p_sep[n_sep ++] = v;
			else
				p_out[n_lower ++] = v;
		}
		const size_t n_upper = (e - b) - n_half;
		std::copy(m_verts.begin() + b + n_half, m_verts.begin() + e, p_out + n_lower);
		std::copy(p_sep, p_sep + n_sep, p_out + n_lower + n_upper);
		std::copy(p_out, p_out + (e - b), m_verts.begin() + b);
		Order_Both(b, n_lower, n_upper, n_depth);
	}

	// Cuthill-McKee-like order of a small connected subset
	void Order_Leaf(size_t b, size_t e, int32_t id)
	{
		int32_t *p_queue = &m_queue[b];
		size_t n_queue = 0;
		Reset_Levels(b, e);
		BFS(m_verts[b], id, p_queue, n_queue);
		const int32_t far = p_queue[n_queue - 1];
		Reset_Levels(b, e);
		BFS(far, id, p_queue, n_queue);
		std::copy(p_queue, p_queue + n_queue, m_verts.begin() + b);
	}

	// on entry the subset's range of m_queue holds a BFS of it (levels set) from an arbitrary root
	void Order_Connected(size_t b, size_t e, int32_t id, int n_depth, const TIndexCut *p_cut)
	{
		const size_t n_size = e - b;
		int32_t *p_queue = &m_queue[b];
		size_t n_queue = n_size;
		if(n_size <= size_t(m_leaf)) {
			Order_Leaf(b, e, id);
			return;
		}
		// pseudo-peripheral root: repeat BFS from a minimum-degree vertex of the last level
		int32_t n_levels = m_level[p_queue[n_queue - 1]] + 1;
		for(int n_pass = 0; n_pass < 3; ++ n_pass) {
			const int32_t last = n_levels - 1;
			int32_t best = -1, best_deg = INT32_MAX;
			for(size_t k = n_queue; k > 0 && m_level[p_queue[k - 1]] == last; -- k) {
				const int32_t v = p_queue[k - 1], d = Degree_In(v, id);
				if(d < best_deg) {
					best_deg = d;
					best = v;
				}
			}
			Reset_Levels(b, e);
			const int32_t n_new = BFS(best, id, p_queue, n_queue);
			if(n_new <= n_levels) {
				n_levels = n_new;
				break;
			}
			n_levels = n_new;
		}
		if(n_levels < 3) { // clique-like: no level separates anything
			if(p_cut) {
				Reset_Levels(b, e);
				Apply_Index_Cut(b, e, id, n_depth, *p_cut);
				return;
			}
			std::copy(p_queue, p_queue + n_queue, m_verts.begin() + b);
			return;
		}
		// level sizes; pick the smallest level that leaves >= 1/4 of the vertices on either side,
		// the median level if there is none
		std::vector<int32_t> count(n_levels, 0);
		for(size_t k = b; k < e; ++ k)
			++ count[m_level[m_verts[k]]];
		const int64_t n_total = int64_t(n_size);
		int32_t m_best = -1;
		{
			int64_t below = 0;
			int64_t best_size = INT64_MAX, best_imbalance = INT64_MAX;
			for(int32_t l = 0; l < n_levels; ++ l) {
				const int64_t above = n_total - below - count[l];
				if(l > 0 && l < n_levels - 1 && below * 100 >= n_total * m_n_balance_pct && above * 100 >= n_total * m_n_balance_pct) {
					const int64_t imb = std::abs(below - above);
					if(count[l] < best_size || (count[l] == best_size && imb < best_imbalance)) {
						best_size = count[l];
						best_imbalance = imb;
						m_best = l;
					}
				}
				below += count[l];
			}
			if(m_best < 0) {
				below = 0;
				for(int32_t l = 0; l < n_levels; ++ l) {
					below += count[l];
					if(below * 2 >= n_total) {
						m_best = std::min(std::max(l, 1), n_levels - 2);
						break;
					}
				}
			}
		}
		// separator = vertices of level m with a neighbour in level m+1; the rest of level m joins the lower part.
		// The parts are formed in the scratch range (lower from its front, upper and separator marked and placed behind)
		// and only take the subset's place once the cut stands
		int32_t *p_lower = &m_scratch[b];
		size_t n_lower = 0, n_upper = 0, n_sep = 0;
		// (a first pass decides every vertex's part: 0 lower, 1 upper, 2 separator -- kept in m_diff's range, free here)
		int32_t *p_part = &m_diff[b];
		for(size_t k = b; k < e; ++ k) {
			const int32_t v = m_verts[k], lv = m_level[v];
			int n_part;
			if(lv < m_best)
				n_part = 0;
			else if(lv > m_best)
				n_part = 1;
			else {
				bool b_sep = false;
				for(int64_t ed = m_ptr[v]; ed < m_ptr[v + 1] && !b_sep; ++ ed) {
					const int32_t w = m_adj[ed];
					b_sep = (m_set[w] == id && m_level[w] == m_best + 1);
				}
				n_part = b_sep? 2 : 0;
			}
			p_part[k - b] = n_part;
			n_lower += n_part == 0;
			n_upper += n_part == 1;
			n_sep += n_part == 2;
		}
		bool b_sort_lower = false;
		if(m_b_other_bank) {
			// the other bank of the same cut: the vertices of level m + 1 with a neighbour in level m -- where they are fewer,
			// they are the separator, and all of level m stays below
			size_t n_sep2 = 0;
			for(size_t k = b; k < e; ++ k) {
				const int32_t v = m_verts[k];
				if(p_part[k - b] != 1 || m_level[v] != m_best + 1)
					continue;
				bool b_sep = false;
				for(int64_t ed = m_ptr[v]; ed < m_ptr[v + 1] && !b_sep; ++ ed) {
					const int32_t w = m_adj[ed];
					b_sep = (m_set[w] == id && m_level[w] == m_best);
				}
				if(b_sep) {
					p_part[k - b] = 3; // (a candidate of the other bank)
					++ n_sep2;
				}
			}
			if(n_sep2 < n_sep) {
				for(size_t k = b; k < e; ++ k) {
					int32_t &r_part = p_part[k - b];
					r_part = (r_part == 2)? 0 : (r_part == 3)? 2 : r_part; // the first bank joins the lower part, the other one is the separator
				}
				n_lower += n_sep;
				n_upper -= n_sep2;
				n_sep = n_sep2;
				b_sort_lower = true; // (subsets stay sorted by vertex number: the cut by number asks for it; the separator is sorted as well, as it was)
			} else {
				for(size_t k = b; k < e; ++ k) {
					if(p_part[k - b] == 3)
						p_part[k - b] = 1;
				}
			}
		}
		if(p_cut && int32_t(n_sep) > p_cut->n_sep) { // the cut by number is narrower
			Reset_Levels(b, e);
			Apply_Index_Cut(b, e, id, n_depth, *p_cut);
			return;
		}
		{
			size_t n_l = 0, n_u = n_lower, n_s = n_lower + n_upper;
			for(size_t k = b; k < e; ++ k) {
				const int32_t v = m_verts[k];
				const int n_part = p_part[k - b];
				p_lower[(n_part == 0)? n_l ++ : (n_part == 1)? n_u ++ : n_s ++] = v;
			}
			if(b_sort_lower) {
				std::sort(p_lower, p_lower + n_lower);
				std::sort(p_lower + n_lower + n_upper, p_lower + n_size);
			}
			std::copy(p_lower, p_lower + n_size, m_verts.begin() + b);
		}
		Order_Both(b, n_lower, n_upper, n_depth);
	}
};

} // anonymous namespace

void nested_dissection(int32_t n, const std::vector<int64_t> &adj_ptr, const std::vector<int32_t> &adj,
	int leaf_size, std::vector<int32_t> &perm, int n_balance_pct, bool b_other_bank)
{
	CNestedDissection nd(n, adj_ptr, adj, leaf_size, n_balance_pct, perm, b_other_bank);
	nd.Run();
}

// ---------------------------------------------------------------------------------------------
// symbolic factorization + schedule
// ---------------------------------------------------------------------------------------------

std::vector<int> tile_symbolic(int T, std::vector<char> &nz)
{
	for(int j = 0; j < T; ++ j) {
		for(int i2 = j + 1; i2 < T; ++ i2) {
			if(!nz[size_t(i2) + size_t(j) * T])
				continue;
			for(int i1 = i2; i1 < T; ++ i1) {
				if(nz[size_t(i1) + size_t(j) * T])
					nz[size_t(i1) + size_t(i2) * T] = 1;
			}
		}
	}
	std::vector<int> height(T, 0);
	for(int i = 0; i < T; ++ i) {
		int h = 0;
		for(int j = 0; j < i; ++ j) {
			if(nz[size_t(i) + size_t(j) * T] && height[j] + 1 > h)
				h = height[j] + 1;
		}
		height[i] = h;
	}
	return height;
}

int dense_top_tile_pattern(const Plan &P, std::vector<char> &nz)
{
	enum { NB = 64 };
	const int T = (P.dense_dim + 1 + NB - 1) / NB; // one more row: the right-hand side
	nz.assign(size_t(T) * T, 0);
	for(int32_t j = 0; j < P.n; ++ j) {
		if(P.dense_pos[j] < 0)
			continue;
		const int c0 = P.dense_pos[j] / NB, c1 = (P.dense_pos[j] + P.dim[j] - 1) / NB;
		for(int64_t k = P.lptr[j]; k < P.lptr[j + 1]; ++ k) {
			const int32_t i = P.lrow[k];
			if(P.dense_pos[i] < 0)
				continue;
			const int r0 = P.dense_pos[i] / NB, r1 = (P.dense_pos[i] + P.dim[i] - 1) / NB;
			for(int tr = r0; tr <= r1; ++ tr) {
				for(int tc = c0; tc <= c1; ++ tc)
					nz[size_t(std::max(tr, tc)) + size_t(std::min(tr, tc)) * T] = 1;
			}
		}
	}
	for(int j = 0; j < T; ++ j) {
		nz[size_t(j) + size_t(j) * T] = 1;
		nz[size_t(T - 1) + size_t(j) * T] = 1; // the right-hand side rides in the last row
	}
	return T;
}

double plan_chain_estimate_us(const Plan &P)
{
	// measured on MI355X: a separator column costs about 5 us plus 0.08 us per block product (8 us at the
	// 40 products of a pose-chain separator; 12 vector-memory instructions per product bound it) in the multi-wave
	// stage kernel, a quarter more in the one-wave kernel, the backward substitution about 3 us per column; a level
	// of the tile schedule 27 us (34 until the end of round 3), a tile of the dense schedule 26 us, four tiles of the dense
	// backward substitution 10 us
	double f_us = 0;
	const int n_stages = int(P.stage_ptr.size()) - 1;
	// (round 3, tasks as panels in LDS: a launch of the factorization about 4.5 us, of the backward substitution 3.5; inside a
	// task 2.3 us per level -- the columns of one level run side by side --, its block products spread over eight waves,
	// the backward substitution a microsecond per column, one after the other)
	for(int s = 0; s < n_stages; ++ s) {
		double f_longest = 0;
		for(int32_t t = P.stage_ptr[s]; t < P.stage_ptr[s + 1]; ++ t) {
			int64_t n_products = 0;
			int n_levels = 0, n_cols = int(P.task_ptr[t + 1] - P.task_ptr[t]);
			bool b_tall = false;
			for(int64_t c = P.task_ptr[t]; c < P.task_ptr[t + 1]; ++ c) {
				const int32_t j = P.task_cols[c];
				n_products += P.col_products[j] + (P.rptr[j + 1] - P.rptr[j]); // (col_products: the pairs of the column's off-diagonal blocks, pptr[lptr[j + 1]] - pptr[lptr[j] + 1])
				if(!P.col_sub.empty()) {
					n_levels = std::max(n_levels, int(P.col_sub[j]) + 1);
					b_tall = b_tall || P.col_sub[j] != 0;
				}
			}
			if(!b_tall)
				n_levels = n_cols; // a chain (or a leaf subtree): one column after the other
			const double f_task = 2.3 * n_levels + 0.01 * double(n_products) + 1.0 * n_cols;
			f_longest = std::max(f_longest, f_task);
		}
		const bool b_wide = P.stage_ptr[s + 1] - P.stage_ptr[s] > 1024;
		f_us += f_longest * (b_wide? 2.5 : 1.0) + 8.0;
	}
	if(P.dense_dim) {
		std::vector<char> nz;
		const int T = dense_top_tile_pattern(P, nz);
		const std::vector<int> height = tile_symbolic(T, nz);
		const int n_levels = *std::max_element(height.begin(), height.end()) + 1;
		f_us += std::min(27.0 * n_levels, 26.0 * T) + 10.0 * ((T + 3) / 4) + 60.0;
	}
	return f_us;
}

// The analysis in three parts, so that the plan search below can share what its candidates have in common: the block graph
// (once per structure), ordering + symbolic factorization (once per ordering), dense top + update lists + schedule (once
// per candidate).

namespace {

struct TBlockGraph {
	int32_t n;
	int64_t n_ablocks;
	std::vector<int64_t> gptr; // symmetric adjacency, no self loops
	std::vector<int32_t> gadj;
	std::vector<int64_t> aoff; // [n_ablocks + 1] offsets of the source blocks in the packed value array
	int64_t nnz_upper;
	double graph_ms;
};

std::string plan_block_graph(int64_t n_bcols, const int64_t *cumsum, const int64_t *bcol_ptr, const int32_t *brow, TBlockGraph &G)
{
	if(n_bcols <= 0 || n_bcols > INT32_MAX / 2)
		return "invalid number of block columns";
	const int32_t n = int32_t(n_bcols);
	const double t0 = now_ms();
	G.n = n;
	G.n_ablocks = bcol_ptr[n];
	std::vector<int64_t> &gptr = G.gptr;
	gptr.assign(n + 1, 0);
	for(int32_t c = 0; c < n; ++ c) {
		if(bcol_ptr[c + 1] < bcol_ptr[c])
			return "block column pointers are not monotonic";
		for(int64_t k = bcol_ptr[c]; k < bcol_ptr[c + 1]; ++ k) {
			const int32_t r = brow[k];
			if(r < 0 || r > c)
				return "block structure is not upper triangular";
			if(k > bcol_ptr[c] && brow[k - 1] >= r)
				return "block rows are not sorted inside a block column";
			if(r != c) {
				++ gptr[r + 1];
				++ gptr[c + 1];
			}
		}
		if(bcol_ptr[c + 1] == bcol_ptr[c] || brow[bcol_ptr[c + 1] - 1] != c)
			return "a diagonal block is missing";
	}
	for(int32_t c = 0; c < n; ++ c)
		gptr[c + 1] += gptr[c];
	std::vector<int32_t> &gadj = G.gadj;
	gadj.resize(gptr[n]);
	{
		std::vector<int64_t> fill(gptr.begin(), gptr.end() - 1);
		for(int32_t c = 0; c < n; ++ c) {
			for(int64_t k = bcol_ptr[c]; k < bcol_ptr[c + 1]; ++ k) {
				const int32_t r = brow[k];
				if(r != c) {
					gadj[fill[r] ++] = c;
					gadj[fill[c] ++] = r;
				}
			}
		}
	}
	G.aoff.assign(G.n_ablocks + 1, 0);
	G.nnz_upper = 0;
	for(int32_t c = 0; c < n; ++ c) {
		const int64_t w = cumsum[c + 1] - cumsum[c];
		if(w <= 0)
			return "block column of zero or negative width";
		for(int64_t k = bcol_ptr[c]; k < bcol_ptr[c + 1]; ++ k) {
			const int64_t h = cumsum[brow[k] + 1] - cumsum[brow[k]];
			G.aoff[k + 1] = G.aoff[k] + h * w;
			G.nnz_upper += (brow[k] == c)? w * (w + 1) / 2 : h * w;
		}
	}
	G.graph_ms = now_ms() - t0;
	if(getenv("SLAMPP_HIP_PLAN_TIMING"))
		fprintf(stderr, "[plan] graph          %6.2f ms\n", G.graph_ms);
	return std::string();
}

// ordering (opt.natural_order, leaf_size, nd_balance_pct, nd_other_bank) and the block symbolic factorization under it
std::string plan_order_symbolic(const TBlockGraph &G, const int64_t *cumsum, const int64_t *bcol_ptr,
	const int32_t *brow, const PlanOptions &opt, Plan &P)
{
	P = Plan();
	const int32_t n = G.n;
	const int64_t n_ablocks = G.n_ablocks;
	const std::vector<int64_t> &gptr = G.gptr, &aoff = G.aoff;
	const std::vector<int32_t> &gadj = G.gadj;
	P.n = n;
	P.nnz_upper = G.nnz_upper;
	double t0 = now_ms();

	// ---- ordering ----
	if(opt.natural_order) {
		P.perm.resize(n);
		std::iota(P.perm.begin(), P.perm.end(), 0);
	} else
		nested_dissection(n, gptr, gadj, opt.leaf_size, P.perm, opt.nd_balance_pct, opt.nd_other_bank != 0);
	if(int32_t(P.perm.size()) != n)
		return "internal error: ordering lost vertices";
	P.pinv.assign(n, -1);
	for(int32_t j = 0; j < n; ++ j)
		P.pinv[P.perm[j]] = j;
	for(int32_t j = 0; j < n; ++ j) {
		if(P.pinv[j] < 0)
			return "internal error: ordering is not a permutation";
	}
	P.dim.resize(n);
	P.cs_new.assign(n + 1, 0);
	P.cs_src.resize(n);
	P.max_dim = 0;
	for(int32_t j = 0; j < n; ++ j) {
		const int32_t o = P.perm[j];
		const int64_t d = cumsum[o + 1] - cumsum[o];
		if(d <= 0)
			return "block column of zero or negative width";
		P.dim[j] = int32_t(d);
		P.cs_src[j] = cumsum[o];
		P.cs_new[j + 1] = P.cs_new[j] + d;
		P.max_dim = std::max(P.max_dim, int32_t(d));
		P.uniform_dim = P.uniform_dim && d == (cumsum[1] - cumsum[0]);
	}
	double t1 = now_ms();
	P.order_ms = t1 - t0 + G.graph_ms;

	const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
	double t_phase = now_ms();
#define PLAN_PHASE(name) do { if(b_timing) { const double t_ = now_ms(); \
	fprintf(stderr, "[plan] %-12s %8.2f ms\n", name, t_ - t_phase); t_phase = t_; } } while(0)
	if(b_timing)
		fprintf(stderr, "[plan] %-12s %8.2f ms\n", "order", P.order_ms);

	// ---- permuted lower-triangular structure of Lambda ----
	struct TAEntry { int32_t row; int32_t trans; int64_t src; };
	std::vector<int64_t> acol_ptr(n + 1, 0);
	for(int32_t c = 0; c < n; ++ c) {
		for(int64_t k = bcol_ptr[c]; k < bcol_ptr[c + 1]; ++ k)
			++ acol_ptr[std::min(P.pinv[brow[k]], P.pinv[c]) + 1];
	}
	for(int32_t j = 0; j < n; ++ j)
		acol_ptr[j + 1] += acol_ptr[j];
	std::vector<TAEntry> aent(n_ablocks);
	{
		std::vector<int64_t> fill(acol_ptr.begin(), acol_ptr.end() - 1);
		for(int32_t c = 0; c < n; ++ c) {
			for(int64_t k = bcol_ptr[c]; k < bcol_ptr[c + 1]; ++ k) {
				const int32_t i0 = P.pinv[brow[k]], j0 = P.pinv[c];
				// stored block is Lambda(r, c); the lower-triangular slot (max, min) needs it
				// as is when new_row(r) > new_row(c), transposed otherwise
				TAEntry e;
				if(i0 >= j0) { e.row = i0; e.trans = 0; e.src = aoff[k]; aent[fill[j0] ++] = e; }
				else { e.row = j0; e.trans = 1; e.src = aoff[k]; aent[fill[i0] ++] = e; }
			}
		}
	}

	PLAN_PHASE("permute");
	// ---- column structure of L by merging children (symbolic Cholesky on the block graph) ----
	P.parent.assign(n, -1);
	P.lptr.assign(n + 1, 0);
	P.lrow.clear();
	P.lrow.reserve(size_t(n_ablocks) * 2);
	std::vector<int32_t> first_child(n, -1), next_sibling(n, -1);
	std::vector<int32_t> mark(n, -1), rows;
	std::vector<int64_t> pos(n, -1); // block id of row i in the column being built
	P.asrc.clear();
	P.atrans.clear();
	P.asrc.reserve(size_t(n_ablocks) * 2); // (with lrow: three arrays growing by doubling were a dozen reallocations of megabytes each)
	P.atrans.reserve(size_t(n_ablocks) * 2);
	for(int32_t j = 0; j < n; ++ j) {
		rows.clear();
		rows.push_back(j);
		mark[j] = j;
		for(int64_t e = acol_ptr[j]; e < acol_ptr[j + 1]; ++ e) {
			const int32_t i = aent[e].row;
			if(mark[i] != j) {
				mark[i] = j;
				rows.push_back(i);
			}
		}
		for(int32_t c = first_child[j]; c >= 0; c = next_sibling[c]) {
			for(int64_t k = P.lptr[c] + 1; k < P.lptr[c + 1]; ++ k) {
				const int32_t i = P.lrow[k];
				if(mark[i] != j) {
					mark[i] = j;
					rows.push_back(i);
				}
			}
		}
		std::sort(rows.begin() + 1, rows.end());
		const int64_t base = int64_t(P.lrow.size());
		for(size_t t = 0; t < rows.size(); ++ t) {
			pos[rows[t]] = base + int64_t(t);
			P.lrow.push_back(rows[t]);
			P.asrc.push_back(-1);
			P.atrans.push_back(0);
		}
		for(int64_t e = acol_ptr[j]; e < acol_ptr[j + 1]; ++ e) {
			const int64_t k = pos[aent[e].row];
			P.asrc[k] = aent[e].src;
			P.atrans[k] = aent[e].trans;
		}
		P.lptr[j + 1] = int64_t(P.lrow.size());
		if(rows.size() > 1) {
			const int32_t p = rows[1];
			P.parent[j] = p;
			next_sibling[j] = first_child[p];
			first_child[p] = j;
		}
	}
	const int64_t n_lblocks = int64_t(P.lrow.size());
	if(n_lblocks > INT32_MAX)
		return "factor has too many blocks";
	P.loff.assign(n_lblocks + 1, 0);
	P.blk_col.resize(n_lblocks);
	P.linv_off.assign(n + 1, 0);
	for(int32_t j = 0; j < n; ++ j) {
		const int64_t dj = P.dim[j];
		int64_t below = 0;
		for(int64_t k = P.lptr[j]; k < P.lptr[j + 1]; ++ k) {
			const int64_t di = P.dim[P.lrow[k]];
			P.loff[k + 1] = P.loff[k] + di * dj;
			P.blk_col[k] = j;
			if(k > P.lptr[j])
				below += di;
		}
		P.linv_off[j + 1] = P.linv_off[j] + dj * dj;
		P.l_nnz += dj * (dj + 1) / 2 + dj * below;
		for(int64_t t = 0; t < dj; ++ t) {
			const double c = double(dj - t + below);
			P.factor_flops += c * c;
		}
	}

	PLAN_PHASE("symbolic");
#undef PLAN_PHASE
	P.symbolic_ms = now_ms() - t1;
	return std::string();
}

// the update lists of a finished plan: L(i,j) = A(i,j) - sum over pairs L[pa] * L[pb]^T
std::string plan_pair_lists(Plan &P)
{
	const int32_t n = P.n;
	const int64_t n_lblocks = int64_t(P.lrow.size());
	std::vector<int32_t> tgt;
	std::vector<int32_t> ga, gb;
	{
		int64_t n_pairs_total = 0;
		for(int32_t c = 0; c < n; ++ c) {
			const int64_t m = P.lptr[c + 1] - P.lptr[c] - 1;
			if(P.dense_pos[c] < 0)
				n_pairs_total += m * (m + 1) / 2;
		}
		tgt.reserve(size_t(n_pairs_total));
		ga.reserve(size_t(n_pairs_total));
		gb.reserve(size_t(n_pairs_total));
	}
	for(int32_t c = 0; c < n; ++ c) {
		if(P.dense_pos[c] >= 0)
			continue; // updates among dense-top columns happen inside the dense factorization
		const int64_t kb0 = P.lptr[c] + 1, kb1 = P.lptr[c + 1];
		for(int64_t kb = kb0; kb < kb1; ++ kb) {
			const int32_t j = P.lrow[kb];
			int64_t t = P.lptr[j]; // walks down column j; rows of c (>= j) are a subset of rows of j
			const int64_t t_end = P.lptr[j + 1];
			for(int64_t ka = kb; ka < kb1; ++ ka) {
				const int32_t i = P.lrow[ka];
				while(t < t_end && P.lrow[t] < i)
					++ t;
				if(t == t_end || P.lrow[t] != i)
					return "internal error: symbolic structure is not closed under updates";
				tgt.push_back(int32_t(t));
				ga.push_back(int32_t(ka));
				gb.push_back(int32_t(kb));
			}
		}
	}
	const size_t n_pairs = tgt.size();
	P.pptr.assign(n_lblocks + 1, 0);
	for(size_t p = 0; p < n_pairs; ++ p)
		++ P.pptr[tgt[p] + 1];
	for(int64_t k = 0; k < n_lblocks; ++ k)
		P.pptr[k + 1] += P.pptr[k];
	P.pa.resize(n_pairs);
	P.pb.resize(n_pairs);
	std::vector<int64_t> fill(P.pptr.begin(), P.pptr.end() - 1);
	for(size_t p = 0; p < n_pairs; ++ p) { // stable: pairs of a block stay ordered by source column
		const int64_t d = fill[tgt[p]] ++;
		P.pa[d] = ga[p];
		P.pb[d] = gb[p];
	}
	return std::string();
}

// dense top (opt.dense_top_*), products per column (the update lists themselves if asked for), row lists, schedule
// (opt.subtree_size, task_*) of an ordered plan
std::string plan_finish(const PlanOptions &opt, Plan &P, bool b_pair_lists)
{
	const int32_t n = P.n;
	const double t1 = now_ms();
	const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
	double t_phase = t1;
#define PLAN_PHASE(name) do { if(b_timing) { const double t_ = now_ms(); \
	fprintf(stderr, "[plan] %-12s %8.2f ms\n", name, t_ - t_phase); t_phase = t_; } } while(0)

	// ---- dense top ----
	std::vector<char> in_dense(n, 0);
	P.dense_pos.assign(n, -1);
	P.dense_dim = 0;
	if(opt.dense_top_nb > 0) {
		int64_t n_dim = 0;
		for(int n_threshold = opt.dense_top_nb;; n_threshold = n_threshold * 3 / 2 + 1) {
			std::fill(in_dense.begin(), in_dense.end(), char(0));
			n_dim = 0;
			for(int32_t j = 0; j < n; ++ j) { // parents have larger indices: one ascending pass closes the set upwards
				if(P.lptr[j + 1] - P.lptr[j] >= n_threshold)
					in_dense[j] = 1;
				if(in_dense[j]) {
					n_dim += P.dim[j];
					if(P.parent[j] >= 0)
						in_dense[P.parent[j]] = 1;
				}
			}
			if(n_dim <= opt.dense_top_max_dim)
				break;
		}
		if(n_dim < opt.dense_top_min_dim)
			std::fill(in_dense.begin(), in_dense.end(), char(0));
		else {
			// the dense-top columns form a forest (restricted elimination tree); a chain that starts at one of its leaves
			// is independent of every other such chain until the two meet in a common ancestor.  Chains of at least one
			// tile start at a tile boundary (positions skipped that way get an identity diagonal), so that no 64 x 64
			// tile couples two of them
			std::vector<int32_t> n_dense_children(n, 0);
			for(int32_t j = 0; j < n; ++ j) {
				if(in_dense[j] && P.parent[j] >= 0)
					++ n_dense_children[P.parent[j]];
			}
			int32_t n_pos = 0;
			for(int32_t j = 0; j < n; ++ j) {
				if(!in_dense[j])
					continue;
				if(opt.dense_top_align > 1 && !n_dense_children[j]) { // a leaf of the forest: how long is its chain?
					int64_t n_chain = 0;
					for(int32_t c = j; c >= 0 && n_dense_children[c] <= 1; c = P.parent[c])
						n_chain += P.dim[c];
					if(n_chain >= opt.dense_top_align)
						n_pos = (n_pos + opt.dense_top_align - 1) / opt.dense_top_align * opt.dense_top_align;
				}
				P.dense_pos[j] = n_pos;
				n_pos += P.dim[j];
			}
			P.dense_dim = n_pos;
		}
	}

	PLAN_PHASE("dense top");
	// ---- update lists ----
	// column c contributes L(i,c) L(j,c)^T to block (i,j) for every pair of its sub-diagonal rows i >= j.  What the chain
	// model asks of them is how many products the off-diagonal blocks of a column receive -- counted here without the lists,
	// which only the plan that is kept needs (plan_pair_lists(); round 6: the lists were half of what a candidate cost)
	P.col_products.assign(n, 0);
	for(int32_t c = 0; c < n; ++ c) {
		if(in_dense[c])
			continue; // updates among dense-top columns happen inside the dense factorization
		const int64_t kb0 = P.lptr[c] + 1, kb1 = P.lptr[c + 1];
		for(int64_t kb = kb0; kb < kb1; ++ kb)
			P.col_products[P.lrow[kb]] += kb1 - kb - 1;
	}
	P.pptr.clear();
	P.pa.clear();
	P.pb.clear();
	if(b_pair_lists) {
		std::string s_err = plan_pair_lists(P);
		if(!s_err.empty())
			return s_err;
	}

	PLAN_PHASE("pairs");
	// ---- row lists ----
	{
		P.rptr.assign(n + 1, 0);
		for(int32_t j = 0; j < n; ++ j) {
			if(in_dense[j])
				continue; // blocks of dense-top columns exist only inside the dense factor
			for(int64_t k = P.lptr[j] + 1; k < P.lptr[j + 1]; ++ k)
				++ P.rptr[P.lrow[k] + 1];
		}
		for(int32_t j = 0; j < n; ++ j)
			P.rptr[j + 1] += P.rptr[j];
		P.rblk.resize(P.rptr[n]);
		std::vector<int64_t> fill(P.rptr.begin(), P.rptr.end() - 1);
		for(int32_t j = 0; j < n; ++ j) {
			if(in_dense[j])
				continue;
			for(int64_t k = P.lptr[j] + 1; k < P.lptr[j + 1]; ++ k)
				P.rblk[fill[P.lrow[k]] ++] = int32_t(k);
		}
	}

	PLAN_PHASE("rows");
	// ---- schedule ----
	{
		const int32_t T = std::max(opt.subtree_size, 1);
		// the forest that is eliminated block by block: dense-top columns removed
		std::vector<int32_t> par(n, -1);
		int32_t n_sched = 0;
		for(int32_t j = 0; j < n; ++ j) {
			if(!in_dense[j]) {
				++ n_sched;
				par[j] = (P.parent[j] >= 0 && !in_dense[P.parent[j]])? P.parent[j] : -1;
			}
		}
		std::vector<int32_t> sz(n, 1), height(n, 1);
		for(int32_t j = 0; j < n; ++ j) {
			const int32_t p = P.parent[j];
			if(p >= 0)
				height[p] = std::max(height[p], height[j] + 1);
			P.etree_height = std::max(P.etree_height, height[j]);
			if(!in_dense[j] && par[j] >= 0)
				sz[par[j]] += sz[j];
		}
		// task id of every column. bottom: whole subtrees of <= T columns; top: maximal chains
		std::vector<int32_t> task_of(n, -1), task_level;
		std::vector<int32_t> troot(n, -1);
		for(int32_t j = n - 1; j >= 0; -- j) {
			if(!in_dense[j] && sz[j] <= T) {
				const int32_t p = par[j];
				troot[j] = (p >= 0 && sz[p] <= T)? troot[p] : j;
			}
		}
		std::vector<int32_t> n_top_children(n, 0), n_bottom_children(n, 0), last_top_child(n, -1);
		for(int32_t j = 0; j < n; ++ j) {
			const int32_t p = par[j];
			if(!in_dense[j] && p >= 0) {
				if(troot[j] >= 0)
					++ n_bottom_children[p];
				else {
					++ n_top_children[p];
					last_top_child[p] = j;
				}
			}
		}
		int32_t n_tasks = 0;
		std::vector<int32_t> root_task(n, -1);
		P.col_sub.assign(n, 0);
		{
		for(int32_t j = 0; j < n; ++ j) {
			if(in_dense[j])
				continue;
			if(troot[j] >= 0) {
				int32_t &r = root_task[troot[j]];
				if(r < 0) {
					r = n_tasks ++;
					task_level.push_back(0);
				}
				task_of[j] = r;
			} else {
				// troot[j] < 0 implies sz[j] > T; children were visited before (they have smaller indices)
				if(n_top_children[j] == 1 && n_bottom_children[j] == 0) {
					task_of[j] = task_of[last_top_child[j]]; // extend the chain
				} else {
					task_of[j] = n_tasks ++;
					task_level.push_back(1);
				}
			}
		}
		// level of a top task = 1 + max level of the tasks of its columns' children
		for(int32_t j = 0; j < n; ++ j) {
			const int32_t p = par[j];
			if(!in_dense[j] && p >= 0 && task_of[p] != task_of[j]) {
				int32_t &lp = task_level[task_of[p]];
				lp = std::max(lp, task_level[task_of[j]] + 1);
			}
		}
		// a chain task's level may have been raised after its children were examined: iterate in
		// column order again until stable (columns ascend, so one extra pass suffices for trees,
		// but chains merge several columns; loop to be safe)
		for(bool b_changed = true; b_changed;) {
			b_changed = false;
			for(int32_t j = 0; j < n; ++ j) {
				const int32_t p = par[j];
				if(!in_dense[j] && p >= 0 && task_of[p] != task_of[j] &&
				   task_level[task_of[p]] < task_level[task_of[j]] + 1) {
					task_level[task_of[p]] = task_level[task_of[j]] + 1;
					b_changed = true;
				}
			}
		}
		}
		if(opt.task_height >= 2) {
			// Tall tasks, above the wide stages.  A stage with more tasks than the chip holds at once is bound by
			// throughput, and one wave per column serves it best; from the first stage with at most task_wide_min tasks on, a
			// stage is a launch on the critical path, and there every separator column gets a (stage, level inside the
			// stage): one level above the highest of its separator children, a new stage every task_height levels; columns of
			// one stage that are joined by tree edges form one task -- a slice of the elimination tree: its columns of one
			// level are independent of each other, everything below the slice was eliminated by earlier stages.  A slice that
			// would outgrow the panel kernel's capacities is cut: the column starts the next stage instead.
			int32_t n_levels = 0;
			for(int32_t t = 0; t < n_tasks; ++ t)
				n_levels = std::max(n_levels, task_level[t] + 1);
			std::vector<int32_t> level_tasks(n_levels + 1, 0);
			for(int32_t t = 0; t < n_tasks; ++ t)
				++ level_tasks[task_level[t]];
			int32_t n_first_tall = 1;
			while(n_first_tall < n_levels && level_tasks[n_first_tall] > opt.task_wide_min)
				++ n_first_tall;
			const int h = std::min(opt.task_height, 8);
			std::vector<int32_t> col_stage(n, 0), grp(n, -1), grp_cols(n, 0), grp_blocks(n, 0);
			std::vector<char> b_tall(n, 0);
			for(int32_t j = 0; j < n; ++ j) {
				if(!in_dense[j]) {
					col_stage[j] = task_level[task_of[j]];
					b_tall[j] = troot[j] < 0 && col_stage[j] >= n_first_tall;
				}
			}
			auto find = [&](int32_t v) { while(grp[v] != v) { grp[v] = grp[grp[v]]; v = grp[v]; } return v; };
			std::vector<int32_t> top_first_child(n, -1), top_next_sibling(n, -1);
			for(int32_t j = n - 1; j >= 0; -- j) { // (descending: the lists come out ascending)
				const int32_t p = par[j];
				if(!in_dense[j] && troot[j] < 0 && p >= 0) {
					top_next_sibling[j] = top_first_child[p];
					top_first_child[p] = j;
				}
			}
			for(int32_t j = 0; j < n; ++ j) {
				if(in_dense[j] || !b_tall[j])
					continue;
				int32_t s = n_first_tall, u = 0;
				for(int32_t c = top_first_child[j]; c >= 0; c = top_next_sibling[c]) {
					int32_t sc = col_stage[c], uc = b_tall[c]? P.col_sub[c] + 1 : h; // (a column of the stages below: the next stage at the earliest)
					if(uc >= h) {
						++ sc;
						uc = 0;
					}
					if(sc > s || (sc == s && uc > u)) {
						s = sc;
						u = uc;
					}
				}
				grp[j] = j;
				grp_cols[j] = 1;
				grp_blocks[j] = int32_t(P.lptr[j + 1] - P.lptr[j]);
				if(u > 0) { // some children are in this stage: join their slices, if the result still fits
					int64_t n_cols_total = 1, n_blocks_total = grp_blocks[j];
					for(int32_t c = top_first_child[j]; c >= 0; c = top_next_sibling[c]) {
						if(b_tall[c] && col_stage[c] == s) { // (distinct children are in distinct slices: a slice has one root)
							const int32_t g = find(c);
							n_cols_total += grp_cols[g];
							n_blocks_total += grp_blocks[g];
						}
					}
					if(n_cols_total > opt.task_max_cols || n_blocks_total > opt.task_max_blocks) {
						++ s;
						u = 0;
					} else {
						for(int32_t c = top_first_child[j]; c >= 0; c = top_next_sibling[c]) {
							if(b_tall[c] && col_stage[c] == s)
								grp[find(c)] = j;
						}
						grp_cols[j] = int32_t(n_cols_total);
						grp_blocks[j] = int32_t(n_blocks_total);
					}
				}
				col_stage[j] = s;
				P.col_sub[j] = u;
			}
			// the tasks again: those of the stages below as they were, one per slice above (ids ascend with the first column)
			std::vector<int32_t> new_task(n_tasks, -1), slice_task(n, -1), new_level;
			int32_t n_new_tasks = 0;
			for(int32_t j = 0; j < n; ++ j) {
				if(in_dense[j])
					continue;
				if(!b_tall[j]) {
					int32_t &r = new_task[task_of[j]];
					if(r < 0) {
						r = n_new_tasks ++;
						new_level.push_back(task_level[task_of[j]]);
					}
					task_of[j] = r;
				} else {
					int32_t &r = slice_task[find(j)];
					if(r < 0) {
						r = n_new_tasks ++;
						new_level.push_back(col_stage[j]);
					}
					task_of[j] = r;
				}
			}
			n_tasks = n_new_tasks;
			task_level.swap(new_level);
		}
		int32_t n_stages = 0;
		for(int32_t t = 0; t < n_tasks; ++ t)
			n_stages = std::max(n_stages, task_level[t] + 1);
		// renumber tasks by (stage, first column)
		std::vector<int32_t> stage_count(n_stages + 1, 0);
		for(int32_t t = 0; t < n_tasks; ++ t)
			++ stage_count[task_level[t] + 1];
		for(int32_t s = 0; s < n_stages; ++ s)
			stage_count[s + 1] += stage_count[s];
		P.stage_ptr.assign(stage_count.begin(), stage_count.end());
		std::vector<int32_t> new_id(n_tasks);
		{
			std::vector<int32_t> fill(stage_count.begin(), stage_count.end() - 1);
			for(int32_t t = 0; t < n_tasks; ++ t) // old ids ascend with the first column
				new_id[t] = fill[task_level[t]] ++;
		}
		P.task_ptr.assign(n_tasks + 1, 0);
		for(int32_t j = 0; j < n; ++ j)
			if(!in_dense[j]) ++ P.task_ptr[new_id[task_of[j]] + 1];
		for(int32_t t = 0; t < n_tasks; ++ t)
			P.task_ptr[t + 1] += P.task_ptr[t];
		P.task_cols.resize(n_sched);
		{
			std::vector<int64_t> fill(P.task_ptr.begin(), P.task_ptr.end() - 1);
			for(int32_t j = 0; j < n; ++ j)
				if(!in_dense[j]) P.task_cols[fill[new_id[task_of[j]]] ++] = j;
		}
	}
	PLAN_PHASE("schedule");
#undef PLAN_PHASE
	P.symbolic_ms += now_ms() - t1;
	return std::string();
}

} // anonymous namespace

namespace {

// The candidates of one plan search.  A candidate is (dense-top threshold, balance of the dissection, bank of its level
// cuts); candidates of one (balance, bank) share ordering and symbolic factorization, all share the block graph.  Round 6:
// the search used to build its up to 13 candidates one after the other, each from the caller's arrays (C1: 42 ms of a
// 42 ms analysis).  The decisions are the same ones in the same order -- Ensure() only makes sure that what the next
// decisions will look at has been built, the missing orderings side by side, then the missing candidates side by side.
class CPlanSearch {
public:
	struct TKey {
		int n_nb, n_balance, n_bank;
		bool operator <(const TKey &r_o) const
		{
			return (n_nb != r_o.n_nb)? n_nb < r_o.n_nb : (n_balance != r_o.n_balance)? n_balance < r_o.n_balance : n_bank < r_o.n_bank;
		}
	};
	struct TCandidate {
		Plan plan;
		std::string s_err;
		double f_us = 0;
		bool b_built = false;
	};

private:
	const TBlockGraph &m_r_graph;
	const int64_t *m_p_cumsum, *m_p_bcol_ptr;
	const int32_t *m_p_brow;
	const PlanOptions m_opt;
	struct TOrdered {
		Plan plan; // up to the symbolic factorization
		std::string s_err;
		bool b_built = false;
	};
	std::map<std::pair<int, int>, TOrdered> m_ordered; // (balance, bank); natural order: one entry
	std::map<TKey, TCandidate> m_cand;
	double m_f_order_ms, m_f_symbolic_ms; // wall clock of the search's phases

	std::pair<int, int> t_Ordering_Key(const TKey &r_k) const
	{
		return m_opt.natural_order? std::make_pair(0, 0) : std::make_pair(r_k.n_balance, r_k.n_bank);
	}

	template <class CJob>
	static void Run_Side_By_Side(size_t n_jobs, CJob job)
	{
		if(!n_jobs)
			return;
		std::vector<std::thread> threads;
		std::exception_ptr p_error;
		std::mutex t_mutex;
		auto guarded = [&](size_t i) {
			try {
				job(i);
			} catch(...) {
				std::lock_guard<std::mutex> t_lock(t_mutex);
				p_error = std::current_exception();
			}
		};
		try {
			for(size_t i = 1; i < n_jobs; ++ i)
				threads.emplace_back(guarded, i);
		} catch(...) { // no more threads: the rest on this one
			for(size_t i = threads.size() + 1; i < n_jobs; ++ i)
				guarded(i);
		}
		guarded(0);
		for(size_t t = 0; t < threads.size(); ++ t)
			threads[t].join();
		if(p_error)
			std::rethrow_exception(p_error);
	}

public:
	CPlanSearch(const TBlockGraph &r_graph, const int64_t *p_cumsum, const int64_t *p_bcol_ptr, const int32_t *p_brow,
		const PlanOptions &r_opt)
		:m_r_graph(r_graph), m_p_cumsum(p_cumsum), m_p_bcol_ptr(p_bcol_ptr), m_p_brow(p_brow), m_opt(r_opt),
		m_f_order_ms(0), m_f_symbolic_ms(0)
	{}

	double f_Order_ms() const { return m_f_order_ms; }
	double f_Symbolic_ms() const { return m_f_symbolic_ms; }

	// orderings (and symbolic factorizations) that are not there yet, side by side
	void Ensure_Orderings(const std::vector<TKey> &r_keys)
	{
		std::vector<std::pair<std::pair<int, int>, TOrdered*> > todo;
		for(const TKey &r_k : r_keys) {
			TOrdered &r_o = m_ordered[t_Ordering_Key(r_k)];
			if(!r_o.b_built) {
				r_o.b_built = true;
				todo.push_back(std::make_pair(t_Ordering_Key(r_k), &r_o));
			}
		}
		const double t0 = now_ms();
		Run_Side_By_Side(todo.size(), [&](size_t i) {
			PlanOptions t_opt = m_opt;
			t_opt.nd_balance_pct = todo[i].first.first;
			t_opt.nd_other_bank = todo[i].first.second;
			TOrdered &r_o = *todo[i].second;
			r_o.s_err = plan_order_symbolic(m_r_graph, m_p_cumsum, m_p_bcol_ptr, m_p_brow, t_opt, r_o.plan);
		});
		if(!todo.empty()) {
			m_f_order_ms += now_ms() - t0;
			if(getenv("SLAMPP_HIP_PLAN_TIMING"))
				fprintf(stderr, "[plan search] %zu orderings side by side %8.2f ms\n", todo.size(), now_ms() - t0);
		}
	}

	// candidates that are not there yet, side by side (their orderings first)
	void Ensure(const std::vector<TKey> &r_keys)
	{
		Ensure_Orderings(r_keys);
		std::vector<std::pair<TKey, TCandidate*> > todo;
		for(const TKey &r_k : r_keys) {
			TCandidate &r_c = m_cand[r_k];
			if(!r_c.b_built) {
				r_c.b_built = true;
				todo.push_back(std::make_pair(r_k, &r_c));
			}
		}
		const double t0 = now_ms();
		Run_Side_By_Side(todo.size(), [&](size_t i) {
			const TKey &r_k = todo[i].first;
			TCandidate &r_c = *todo[i].second;
			const TOrdered &r_o = m_ordered.find(t_Ordering_Key(r_k))->second; // (there since Ensure_Orderings(); find(): nothing is inserted under the threads' feet)
			if(r_k.n_nb < 4 && r_k.n_nb != m_opt.dense_top_nb) {
				r_c.s_err = "threshold too low";
				return;
			}
			if(!(r_c.s_err = r_o.s_err).empty())
				return;
			PlanOptions t_opt = m_opt;
			t_opt.dense_top_nb = r_k.n_nb;
			t_opt.nd_balance_pct = r_k.n_balance;
			t_opt.nd_other_bank = r_k.n_bank;
			r_c.plan = r_o.plan;
			if((r_c.s_err = plan_finish(t_opt, r_c.plan, false)).empty())
				r_c.f_us = plan_chain_estimate_us(r_c.plan);
		});
		if(!todo.empty()) {
			m_f_symbolic_ms += now_ms() - t0;
			if(getenv("SLAMPP_HIP_PLAN_TIMING"))
				fprintf(stderr, "[plan search] %zu candidates side by side %8.2f ms\n", todo.size(), now_ms() - t0);
		}
	}

	TCandidate &r_Get(const TKey &r_k)
	{
		Ensure(std::vector<TKey>(1, r_k));
		return m_cand[r_k];
	}
};

} // anonymous namespace

std::string build_plan(int64_t n_bcols, const int64_t *cumsum, const int64_t *bcol_ptr,
	const int32_t *brow, const PlanOptions &r_opt, Plan &P)
{
	PlanOptions opt = r_opt;
	// development aids (environment, with SLAMPP_HIP_DEV=1: plan.h): override the options of the same names
	opt.task_height = std::min(std::max(dev_knob("SLAMPP_HIP_DEV_TASK_HEIGHT", opt.task_height), 1), 8);
	opt.task_max_cols = std::min(std::max(dev_knob("SLAMPP_HIP_DEV_TASK_MAX_COLS", opt.task_max_cols), 1), 64);
	opt.task_max_blocks = std::min(std::max(dev_knob("SLAMPP_HIP_DEV_TASK_MAX_BLOCKS", opt.task_max_blocks), 1), 1024);
	const bool b_balance_knob = dev_knob_set("SLAMPP_HIP_DEV_ND_BALANCE");
	const int n_balance_knob = std::min(std::max(dev_knob("SLAMPP_HIP_DEV_ND_BALANCE", opt.nd_balance_pct), 1), 49);
	opt.nd_balance_pct = std::min(std::max(opt.nd_balance_pct, 1), 49);
	P = Plan();
	TBlockGraph t_graph;
	{
		std::string s_err = plan_block_graph(n_bcols, cumsum, bcol_ptr, brow, t_graph);
		if(!s_err.empty())
			return s_err;
	}
	typedef CPlanSearch::TKey TKey;
	const bool b_print = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
	// (the balance knob overrides the balance of every candidate, as it did when each candidate read it for itself)
	auto Key = [&](int n_nb, int n_balance, int n_bank) { TKey k = {n_nb, b_balance_knob? n_balance_knob : n_balance, n_bank}; return k; };
	CPlanSearch search(t_graph, cumsum, bcol_ptr, brow, opt);
	const TKey t_first = Key(opt.dense_top_nb, opt.nd_balance_pct, opt.nd_other_bank);
	const bool b_may_search = opt.dense_top_auto && opt.dense_top_nb > 0;
	// candidates: the balance first (25 / 35 / 45 % at the threshold asked for), then a lower and a higher threshold at the
	// balance that came out best -- the tile levels of the dense top are what the chain is made of (33 us each: a
	// diagonal tile, its panel, the updates), and how many there are depends on where the dissection cuts (measured,
	// reduced camera system of the Venice-like leg: 2.34 ms at 25 %, 1.63 ms at 45 %; the model said 2.27 and 1.40)
	const int n_balanced = std::max(opt.nd_balance_pct, 25);
	const bool b_small = n_bcols <= 20000; // a small graph (the 2-D-like ones of the configs, a reduced camera system): the whole grid
	if(b_may_search && !opt.natural_order && n_bcols <= 8192) {
		// whether there is a dense top is known once the first plan stands; a graph this small is ordered in a millisecond
		// or two, so the orderings a search would want are made beside the first one rather than after it
		std::vector<TKey> spec(1, t_first);
		for(int n_bank = opt.nd_other_bank; n_bank < 2; ++ n_bank) {
			for(int n_balance = n_balanced; n_balance <= 45; n_balance += 10)
				spec.push_back(Key(opt.dense_top_nb, n_balance, n_bank));
		}
		search.Ensure_Orderings(spec);
	}
	auto Finish = [&]() {
		const double t0 = now_ms();
		std::string s_err = plan_pair_lists(P); // (candidates are priced without their update lists)
		P.order_ms = search.f_Order_ms(); // (wall clock of the whole search: the time went into this analysis)
		P.symbolic_ms = search.f_Symbolic_ms() + (now_ms() - t0);
		if(b_print)
			fprintf(stderr, "[plan] %-12s %8.2f ms\n", "pairs", now_ms() - t0);
		return s_err;
	};
	{
		CPlanSearch::TCandidate &r_first = search.r_Get(t_first);
		if(!r_first.s_err.empty())
			return r_first.s_err;
		if(!opt.dense_top_auto && b_print) {
			fprintf(stderr, "[plan] dense_top_nb %d, balance %d %%: dense dim %d, chain estimate %.0f us (as asked for)\n", opt.dense_top_nb,
				opt.nd_balance_pct, r_first.plan.dense_dim, r_first.f_us);
		}
		if(!opt.dense_top_auto || !r_first.plan.dense_dim) { // the only plan (a pose chain at full size comes this way: no copy)
			std::swap(P, r_first.plan);
			return Finish();
		}
		P = r_first.plan; // (a copy: the search may look at the same candidate again under another name -- the balance knob)
	}
	// A dense top: a 2-D-like graph.  Where the line between block-by-block elimination and the dense factorization
	// is best drawn depends on the graph (Manhattan-like: lower, sphere-like: higher), and its separators are long
	// enough that balanced halves beat the smallest separator (the opposite of pose chains, whose separators are one
	// or two vertices and for which the default of 15 % is tuned).  Rebuild with 25 % as the base, then try a lower
	// and a higher threshold; keep what the chain model clearly prefers.
	double f_best = plan_chain_estimate_us(P);
	if(b_print) {
		fprintf(stderr, "[plan] dense_top_nb %d, balance %d %%: dense dim %d, chain estimate %.0f us\n", opt.dense_top_nb,
			opt.nd_balance_pct, P.dense_dim, f_best);
	}
	int n_best_balance = opt.nd_balance_pct, n_best_nb = opt.dense_top_nb;
	auto Try = [&](int n_nb, int n_balance, double f_margin, int n_bank = -1) {
		const TKey t_key = Key(n_nb, n_balance, (n_bank >= 0)? n_bank : opt.nd_other_bank);
		if(n_nb < 4)
			return;
		CPlanSearch::TCandidate &r_c = search.r_Get(t_key);
		if(!r_c.s_err.empty())
			return;
		const double f_us = r_c.f_us;
		if(b_print) {
			fprintf(stderr, "[plan] dense_top_nb %d, balance %d %%%s: dense dim %d, chain estimate %.0f us\n", n_nb, n_balance,
				t_key.n_bank? ", narrower bank" : "", r_c.plan.dense_dim, f_us);
		}
		if(r_c.plan.dense_dim && f_us < f_best * f_margin) {
			f_best = std::min(f_best * std::max(f_margin, 1.0), f_us);
			n_best_balance = n_balance;
			n_best_nb = n_nb;
			P = r_c.plan; // (a copy: the same candidate may be looked at again under another name -- natural order, the balance knob)
		}
	};
	// the balanced base replaces the first plan unless it is clearly worse; everything else must be clearly better (the
	// model is rough, and rougher for the heavy columns a higher threshold leaves to the block kernels)
	const int p_nb[3] = {opt.dense_top_nb, opt.dense_top_nb * 2 / 3, opt.dense_top_nb * 3 / 2};
	const double p_nb_margin[3] = {0.95, 0.95, 0.90};
	if(b_small) {
		std::vector<TKey> all;
		for(int i = 0; i < 3; ++ i) {
			for(int n_balance = n_balanced; n_balance <= 45; n_balance += 10) {
				if(p_nb[i] >= 4)
					all.push_back(Key(p_nb[i], n_balance, opt.nd_other_bank));
			}
		}
		search.Ensure(all);
	} else {
		std::vector<TKey> all;
		for(int n_balance = n_balanced; n_balance <= 45; n_balance += 10)
			all.push_back(Key(opt.dense_top_nb, n_balance, opt.nd_other_bank));
		search.Ensure(all);
	}
	if(n_balanced != opt.nd_balance_pct)
		Try(opt.dense_top_nb, n_balanced, 1.10);
	if(b_small) {
		for(int i = 0; i < 3; ++ i) {
			if(i > 0 && P.task_cols.empty())
				break; // everything is in the dense top already: the threshold no longer matters
			for(int n_balance = 25; n_balance <= 45; n_balance += 10) {
				if((i > 0 || n_balance > n_balanced) && n_balance >= n_balanced)
					Try(p_nb[i], n_balance, p_nb_margin[i]);
			}
		}
	} else { // a large one: the balance first, then the threshold at the balance that came out best
		for(int n_balance = 35; n_balance <= 45; n_balance += 10) {
			if(n_balance > n_balanced)
				Try(opt.dense_top_nb, n_balance, 0.95);
		}
		const int n_balance_chosen = n_best_balance;
		{
			std::vector<TKey> both;
			for(int i = 1; i < 3; ++ i) {
				if(p_nb[i] >= 4)
					both.push_back(Key(p_nb[i], n_balance_chosen, opt.nd_other_bank));
			}
			search.Ensure(both);
		}
		Try(p_nb[1], n_balance_chosen, p_nb_margin[1]);
		Try(p_nb[2], n_balance_chosen, p_nb_margin[2]);
	}
	// the other bank of the level cuts (PlanOptions::nd_other_bank): another family of orderings, priced at the threshold that
	// came out best and every balance (round 4: the Venice-like reduced camera system 1.12 -> 1.00 ms; the model keeps C1 and
	// C2 where they were, and so does the clock)
	if(!opt.nd_other_bank) {
		const int n_nb_chosen = n_best_nb;
		std::vector<TKey> banks;
		for(int n_balance = n_balanced; n_balance <= 45; n_balance += 10)
			banks.push_back(Key(n_nb_chosen, n_balance, 1));
		search.Ensure(banks);
		for(int n_balance = n_balanced; n_balance <= 45; n_balance += 10)
			Try(n_nb_chosen, n_balance, 0.95, 1);
	}
	return Finish();
}


} // namespace slampp
