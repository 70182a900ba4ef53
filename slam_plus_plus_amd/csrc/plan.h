// plan.h -- host-side symbolic analysis of the block-sparse Cholesky (ordering, elimination
// tree, factor structure, update lists, stage schedule).  Everything here works on the *block*
// graph (n_bcols nodes), never on scalars -- the reference's native solver does the same
// (/root/reference/include/slam/LinearSolver_UberBlock.h:285-310) while its CHOLMOD wrapper
// orders the 6x larger scalar graph (/root/reference/src/slam/LinearSolver_CholMod.cpp:294-300).
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

namespace slampp {

// Development knobs: integers read from the environment, and only when SLAMPP_HIP_DEV=1 is set there as well -- the
// environment of the host application must not be able to change the ordering or the plan of every handle by accident
// (until round 5 some of these were plain names: ND_MIN, ND_SEP).  Whether the knobs are on at all is read from the
// environment where a handle is created, configured or analyzed (dev_knobs_refresh(): the cold path) and remembered: the
// launch helpers of the warm path ask dev_knob() at every factorization, and without SLAMPP_HIP_DEV=1 -- every production
// process -- that is a load of a flag, no getenv() beside a host application that may be calling setenv() (advisor, round 5).
// With the knobs on, a knob's value is read at every use: the tools that sweep them set them between two analyses.
inline std::atomic<int> &dev_knobs_state() { static std::atomic<int> n_state(-1); return n_state; }
inline bool dev_knobs_refresh()
{
	const char *p = getenv("SLAMPP_HIP_DEV");
	const int n_on = p && atoi(p) != 0;
	dev_knobs_state().store(n_on, std::memory_order_relaxed);
	return n_on != 0;
}
inline bool dev_knobs_on()
{
	const int n_on = dev_knobs_state().load(std::memory_order_relaxed);
	return (n_on < 0)? dev_knobs_refresh() : n_on != 0;
}
inline bool dev_knob_set(const char *p_s_name) { return dev_knobs_on() && getenv(p_s_name) != 0; }
inline int dev_knob(const char *p_s_name, int n_default) { return dev_knob_set(p_s_name)? atoi(getenv(p_s_name)) : n_default; }


struct Plan {
	int32_t n = 0;                    // block columns
	int32_t max_dim = 0;
	bool uniform_dim = true;          // all block columns have the same dimension
	std::vector<int32_t> perm, pinv;  // perm[new] = old, pinv[old] = new
	std::vector<int32_t> dim;         // [n] dimension of block column, new order
	std::vector<int64_t> cs_new;      // [n+1] scalar offset in the permuted vector
	std::vector<int64_t> cs_src;      // [n] scalar offset of new column j in the caller's (original) vector
	std::vector<int32_t> parent;      // [n] elimination tree (new order), -1 = root
	// factor structure: block-CSC of L (lower), rows sorted, first block of a column = diagonal
	std::vector<int64_t> lptr;        // [n+1]
	std::vector<int32_t> lrow;        // [l_blocks]
	std::vector<int64_t> loff;        // [l_blocks+1] offset in the factor values
	std::vector<int64_t> asrc;        // [l_blocks] offset in the packed Lambda values or -1 (fill-in)
	std::vector<int32_t> atrans;      // [l_blocks] source block is stored transposed
	std::vector<int64_t> linv_off;    // [n+1] offset of inv(L_jj) in the linv array
	// update lists: L(i,j) = A(i,j) - sum over pairs L[pa] * L[pb]^T
	std::vector<int64_t> pptr;        // [l_blocks+1]
	std::vector<int32_t> pa, pb;      // [n_pairs] factor block ids
	std::vector<int64_t> col_products; // [n] pairs of the off-diagonal blocks of a column (what the chain model prices; there before the lists are)
	// row lists (forward substitution): blocks L(j,c), c < j, of every block row j
	std::vector<int64_t> rptr;        // [n+1]
	std::vector<int32_t> rblk;        // [n_row_entries] factor block ids
	std::vector<int32_t> blk_col;     // [l_blocks] column of every factor block
	// schedule: stage -> tasks -> columns; tasks of one stage are independent, a task's columns
	// are eliminated in order by one workgroup
	std::vector<int32_t> stage_ptr;   // [n_stages+1]
	std::vector<int64_t> task_ptr;    // [n_tasks+1]
	std::vector<int32_t> task_cols;   // [n - number of dense-top columns]
	// tall tasks (PlanOptions::task_height > 1): a separator task is a slice of the elimination tree up to task_height
	// levels high; col_sub[j] = the level of column j inside its task (columns of one level do not depend on each other,
	// a column's children in the task have lower levels).  All zero otherwise.
	std::vector<int32_t> col_sub;     // [n]
	// dense top: an upper set of the elimination tree (the big separators of 2-D-like graphs) is not
	// eliminated block by block; its Schur complement is assembled into one dense matrix and handed to
	// the dense MFMA Cholesky (dense_chol.hip).  dense_pos[j] = scalar offset of column j there, or -1.
	std::vector<int32_t> dense_pos;   // [n]
	int32_t dense_dim = 0;            // scalar dimension of the dense top, 0 = none
	// statistics
	int64_t l_nnz = 0;                // scalar nonzeros of L (lower, incl. diagonal)
	double factor_flops = 0;          // sum over scalar columns of (column count)^2  (CHOLMOD's "fl")
	int64_t nnz_upper = 0;            // scalar nonzeros of triu(Lambda)
	int32_t etree_height = 0;
	double order_ms = 0, symbolic_ms = 0;
};

struct PlanOptions {
	bool natural_order = false; // no fill-reducing ordering: the caller's order is kept (Factorize_PosDef_Blocky: the factor
	                            // goes back to a caller that has ordered the matrix itself)
	int leaf_size = 3;        // nested dissection stops at subgraphs of this many block columns (round 4: 4 -> 3, C3 0.306 -> 0.302 ms, C1 0.673 -> 0.668; 2 measures the same)
	int nd_balance_pct = 15;  // a separator must leave at least this share (percent) of the vertices on either side; small
	                          // separators beat balanced halves here: 15 is 5-15 % faster than 25 on pose chains of 30k-300k poses
	int nd_other_bank = 0;    // 1: where a level structure cuts, the separator is the narrower of the cut's two banks (the vertices of level
	                          // m with a neighbour in level m + 1, or those of level m + 1 with a neighbour in level m) instead of always
	                          // the first; a candidate of the plan search where there is a dense top (round 4: the Venice-like reduced
	                          // camera system 1.12 -> 1.00 ms; C1 and C2 are better off without, and the chain model says so)
	int subtree_size = 8;     // a subtree of at most this many columns is one sequential task (8: best from 2k to 100k poses)
	int task_height = 6;      // separator tasks above the leaf subtrees: 1 = maximal chains of single children (one tree level
	                          // per stage), 2 .. 8 = slices of the elimination tree up to that many levels high (a stage, i.e. a
	                          // launch, then covers that many levels: the launches of a chain-like graph are its critical path;
	                          // round 3, C3: 3 / 4 / 5 / 6 levels 0.374 / 0.366 / 0.366 / 0.373 ms; re-measured at the end of round 4
	                          // -- block columns as rows, hand-ups, the balanced tree of the cut by vertex number --: 3 / 4 / 5 / 6 / 8
	                          // 0.316 / 0.314 / 0.307 / 0.306 / 0.307 ms, the reduced camera system of the band leg 213 / 206 / 198 us at
	                          // 4 / 5 / 6, C5's 242 / 235 / 235; a slice is cut at the panel kernel's capacities anyway -- 8 columns,
	                          // 96 blocks: 10 / 96 and 12 / 128 leave the panel path, 6 / 64 is 7 % slower at C3)
	int task_wide_min = 1024; // ... above the wide stages: a stage with more tasks than this stays one level high (throughput, not latency)
	int task_max_cols = 8, task_max_blocks = 96; // what such a slice may hold (the panel kernel's capacities)
	int dense_top_nb = 24;    // columns with this many blocks or more (and their ancestors) form the dense top; 0 = off
	bool dense_top_auto = true; // when that gives a dense top, also try 16 and 36 and keep the plan whose estimated chain
	                            // of dependent launches is shortest (the caller did not ask for a specific threshold)
	int dense_top_max_dim = 12288; // cap on its scalar dimension (the threshold is raised until it fits)
	int dense_top_min_dim = 192;   // below this the dense top is not worth its launches
	int dense_top_align = 64;      // independent chains of dense-top columns start at multiples of this (tile) size, so that
	                               // the tile schedule of the dense factorization can run them side by side; 0 = packed
};

// structure of the tile factor of a matrix whose nonzero T x T tile pattern is r_nonzero (column-major flags, lower
// triangle): closes the pattern under elimination in place and returns, per tile column, its height in the tile
// elimination tree (columns of one height are independent of each other)
std::vector<int> tile_symbolic(int T, std::vector<char> &r_nonzero);

// tile pattern of the dense top of a plan (64 x 64 tiles, last tile row = right-hand side) and its padded tile count
int dense_top_tile_pattern(const Plan &plan, std::vector<char> &r_nonzero);

// rough model of the chain of dependent launches of one factor + solve, in microseconds
double plan_chain_estimate_us(const Plan &plan);

// returns empty string on success, else an error message
std::string build_plan(int64_t n_bcols, const int64_t *bcol_cumsum, const int64_t *bcol_ptr,
	const int32_t *brow_idx, const PlanOptions &opt, Plan &plan);

// fill-reducing, parallelism-exposing ordering of the block graph: nested dissection with
// BFS level-structure separators; perm[new] = old
void nested_dissection(int32_t n, const std::vector<int64_t> &adj_ptr, const std::vector<int32_t> &adj,
	int leaf_size, std::vector<int32_t> &perm, int n_balance_pct = 15, bool b_other_bank = false);

} // namespace slampp
