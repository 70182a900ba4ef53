// sparse_kernels.h -- device view of the elimination plan + kernel launchers (sparse path)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plan.h" // (dev_knob)

namespace slampp {

// K value sets of ONE structure factored and solved by the same launches (slampp_hip_factor_solve_batch_device_async; the
// reference's LM loop tries its damping values one after the other: NonlinearSolver_Lambda_LM.h:967-1001, 1660-1676): the
// kernels of the sparse block path take their member from blockIdx.y and step their base pointers by these strides (in
// doubles; the not-positive-definite flag is one int per member).  n = 1, strides 0: an ordinary solve.
struct TBatch {
	int n;
	int64_t a, l, linv, b, w, h; // Lambda's values, the factor, the inverses of its diagonal blocks, the caller's vector, the workspace, the hand-up buffer
};
inline TBatch t_No_Batch() { TBatch t = {1, 0, 0, 0, 0, 0, 0}; return t; }



// One record per block column / factor block / row entry, packed so that a kernel gets everything
// it needs about an item with one (wave-uniform, broadcast) load instead of a chain of dependent
// index loads -- the path is latency-bound, every dependent load on it costs about a microsecond.
struct TColDesc { // 64 B
	int64_t k0;        // first factor block of the column (the diagonal one)
	int32_t nb, dj;    // number of blocks, column dimension
	int64_t linv_off;  // offset of inv(L_jj)
	int64_t cs_new;    // scalar offset in the permuted workspace
	int64_t cs_src;    // scalar offset in the caller's vector
	int64_t r0;        // first row entry
	int32_t nr;        // number of row entries = blocks L(j,c), c < j
	int32_t np;        // number of update pairs of the sub-diagonal blocks of the column (contiguous)
	int64_t p0;        // first of them
};

struct TBlkDesc { // 32 B
	int64_t loff;      // offset of the block in the factor values
	int64_t asrc;      // (offset in Lambda values) * 2 + transposed, or -1
	int64_t p0;        // first update pair
	uint32_t np_di;    // number of pairs (low 24 bits) | row dimension << 24
	int32_t xcs;       // scalar offset of the block's row in the permuted workspace
};

struct TRowEnt { // 16 B: block L(j,c) of block row j
	int64_t off;       // offset of the block
	int32_t ycs;       // scalar offset of column c in the permuted workspace
	int32_t dc;        // dimension of column c
};

// dense top: one record per factor block (i,j) with j in the dense top, and one per dense-top column
struct TDenseBlk { // 64 B
	int64_t asrc;      // source in the Lambda values (encoded as in TBlkDesc) or -1
	int64_t p0;        // first update pair (pairs from columns outside the dense top only)
	int64_t dst;       // offset of element (0,0) of the block in the dense matrix
	int64_t r0;        // diagonal blocks: first row entry
	int32_t np, nr;    // number of pairs; number of row entries, -1 for off-diagonal blocks
	int32_t di, dj;
	int64_t cs_src;    // diagonal blocks: scalar offset of the column in the caller's vector
	int32_t pos;       // diagonal blocks: scalar offset of the column in the dense system
	int32_t pad;
};

struct TDenseCol { // 24 B
	int64_t cs_new, cs_src;
	int32_t pos, dj;
};

// ---- panel packages (panel_kernel.hip): everything a separator task needs, in one buffer ----
// 16-byte units: head (4) | columns (3 each) | factor blocks of the task = slots of its LDS image (2 each) | internal row
// entries and internal update pairs (4 per unit: operands that are slots of the image) | fresh entries (2 each: updates
// whose operands the stage right below produced, sorted by the wave that brings them in)
enum { PANEL_W = 8, PANEL_COLS = 8, PANEL_UNITS = 1024, PANEL_UPD_W = 4 };
struct TPanelHead { // 64 B
	int32_t n_cols, n_slots, n_units, n_int_rows; // (internal row entries: the internal pairs follow them, unit-aligned)
	int32_t ext_ptr[12];                          // wave v brings in the fresh entries ext_ptr[v] .. ext_ptr[v + 1]
	                                              // ext_ptr[10], [11]: blocks this task hands up to the next stage's tasks, units of their list
};
// Round 4: what a task owes the tasks of the NEXT stage -- the products of its own finished blocks that update their
// blocks -- it computes itself, out of its LDS image at the end of its walk, and hands up as ready-made blocks (one per
// pair of tasks and target block: D x D doubles, then D of the right-hand side's share for a diagonal block); the task
// above subtracts them from its image with one round of coalesced loads, where it used to fetch both operands of every
// such product from memory itself ("fresh" entries: 4 - 10 us of a stage at C3, 90 - 126 products per task in the reduced
// camera system of a band-visibility BA problem, whose stages also kept a launch of their own for those updates).
struct TPanelOut { // 16 B
	int32_t op0, onp;   // its operand pairs in the list behind the records: (slot of L(i,c)) | (slot of L(j,c)) << 16; for kind 1 (slot of L(j,c)) | (c's number in the task) << 16
	int64_t dst;        // where it goes in the hand-up buffer, | kind << 62 (1 = a diagonal block: the right-hand side's share follows it)
};
struct TPanelCol { // 48 B
	int64_t linv_off, cs_new, cs_src;
	int32_t slot0, nb;  // the column's blocks are the slots slot0 .. slot0 + nb - 1 (diagonal block first)
	int32_t ir0, inr;   // its internal row entries: (slot of L(j,c)) | (c's number in the task) << 16
	int32_t sub, pad;   // level of the column inside a tall task (Plan::col_sub): the package lists the columns level by level
};
struct TPanelSlot { // 32 B
	int64_t loff, asrc; // as in TBlkDesc
	int32_t ip0, inp;   // internal update pairs: (slot of L(i,c)) | (slot of L(j,c)) << 16
	int64_t pad;
};
struct TPanelExt { // 32 B
	int64_t a_off, b_off; // offsets of L(i,c), L(j,c) in the factor (row entries: both L(j,c)); kind 2 / 3: a_off = offset in the hand-up buffer
	int32_t ycs;          // row entries: scalar offset of y_c in the workspace
	uint16_t slot, kind;  // target slot; 0 = update pair, 1 = row entry of the diagonal block, 2 = a block handed up by a task of
	                      // the stage below (subtracted as it is), 3 = such a block for a diagonal block (with its right-hand side share)
	int32_t col;          // row entries: number of the target column in the task
	int32_t pad;
};
// the updates whose operands stages further down produced are applied before the panels run, one half-workgroup per
// factor block (L(block) = Lambda(block) - sum, y_j = b_j - sum for diagonal blocks) -- for the first panel stage by a
// launch of their own, for every later one inside the launch of the stage below (nothing there depends on them)
struct TUpdSlot { // 64 B
	int64_t loff, asrc;
	int64_t e0;           // first entry
	int32_t ne, kind;     // number of entries; 1 = diagonal block (row entries L(j,c), with the right-hand side), 0 = below it (pairs)
	int64_t cs_src, cs_new; // diagonal blocks: where b_j is read and y_j (so far) goes
	int64_t pad[2];
};
struct TUpdEnt { // 16 B
	int64_t a_off; // offset of L(i,c) (row entries: of L(j,c))
	int64_t b_off; // offset of L(j,c) (row entries: scalar offset of y_c in the workspace)
};
inline int panel_slot_cap(int n_dim) { return (n_dim == 6)? 96 : (n_dim == 7)? 72 : 256; }

// How one stage's panel launch is shaped (decided per stage by the host): waves per task -- 8 where a stage is a launch on
// the critical path (latency of one task), 4 where it holds more tasks than the chip takes at once (throughput: more
// workgroups per CU) -- and the capacities its LDS is laid out for, the largest of the stage's tasks: package units,
// blocks of the image, columns, columns of one level.
struct TPanelLaunch {
	int32_t n_waves, n_cap_units, n_cap_blk, n_cap_cols, n_cap_lvl;
	int32_t n_cap_out;     // units of the largest hand-up list of the stage's tasks
	int32_t b_from_lambda; // the tasks read their blocks from Lambda (and b): no update role has prepared Lambda - sum in the factor's
	                       // storage -- the first stage above a leaf stage, whose every update is a fresh one
};
enum { PANEL_UPD_BATCH = 8 };
// fresh products a wave has in flight: four at eight waves per task, eight at four (the same operand staging per workgroup;
// at two waves per task four again: the launch is crowded, and 4.6 KB less LDS per workgroup is a workgroup more per CU -- C3 141 -> 139 us;
// the first slice stage above the leaves brings in ~100 products per task, each batch a trip to L2)
inline __host__ __device__ constexpr int panel_fresh_batch(int n_waves) { return (n_waves == 4)? 8 : 4; }
// LDS of a panel launch, in doubles: offsets of the panel role's regions (package, image, vectors, inverses of the
// level's diagonal blocks, one tile per wave, operand staging of the fresh updates) and the total, which also covers
// the update role's staging (riders: the next stage's updates from further down, PANEL_UPD_W waves per factor block)
struct TPanelLds {
	int IMAGE, VEC, LINV, TILE, OPS, YV, OUT, TOTAL;
};
enum { PANEL_LDS_BUDGET = 156 * 1024 }; // dynamic LDS a panel launch may ask for (gfx950: 160 KB per CU; the kernel's static arrays take the rest)
inline __host__ __device__ TPanelLds panel_lds(int D, bool b_fused, const TPanelLaunch &c)
{
	const int DD = D * D, W = c.n_waves;
	TPanelLds l;
	l.IMAGE = 2 * c.n_cap_units;
	l.VEC = l.IMAGE + c.n_cap_blk * DD;
	l.LINV = l.VEC + c.n_cap_cols * 8;
	l.TILE = l.LINV + c.n_cap_cols * 64; // (block-wise walk: the inverses of a level's diagonal blocks; row-wise walk: every column's finished diagonal block)
	l.OPS = l.TILE + W * 64;
	l.YV = l.OPS + (b_fused? W * 2 * panel_fresh_batch(W) * DD : 0);
	l.OUT = l.YV + (b_fused? W * panel_fresh_batch(W) * 8 : 0);
	const int n_panel_end = l.OUT + 2 * c.n_cap_out;
	const int n_upd_end = b_fused? W * 2 * PANEL_UPD_BATCH * DD + W * PANEL_UPD_BATCH * 8 + W * 64 : 0;
	l.TOTAL = (n_panel_end > n_upd_end)? n_panel_end : n_upd_end;
	return l;
}

// one workgroup per package (pkg_off: their offsets in pkg, in 16-byte units; pkg is padded by 64 * PANEL_W units)
// (upd_slots: the blocks of the NEXT stage's panel tasks, whose updates from below this stage ride in this launch)
// (b_fused: the plan has such stages at all; without them the leaner kernel runs)
// (out_off: per task the offset of its hand-up list in pkg, or -1; H: the hand-up buffer)
bool launch_factor_panel(int n_dim, bool b_fused, bool b_rows, const TPanelLaunch &r_cfg, const longlong2 *pkg, const int64_t *pkg_off, const int64_t *out_off,
	int n_tasks, const TUpdSlot *upd_slots, int n_upd_slots, const TUpdEnt *upd_ents, const double *A, const double *b, double *L, double *Linv, double *w,
	double *H, int *p_flag, hipStream_t stream, long long *p_timing = 0, const TBatch &t_batch = t_No_Batch());
void launch_panel_update(int n_dim, const TUpdSlot *slots, int n_slots, const TUpdEnt *ents, const double *A, double *L,
	const double *b, double *w, hipStream_t stream, const TBatch &t_batch = t_No_Batch());

// capacities of the staged path of the separator kernel (blocks, row entries, update pairs of a column): near the root,
// and in the wide stages right above the leaves
enum { UP_CHUNK = 16, UP_NR = 128, UP_NP = 512, WIDE_CHUNK = 8, WIDE_NR = 32, WIDE_NP = 48 };
enum { PKG_SPECULATIVE = 512 }; // 16-byte units fetched before a package's size is known (one per thread of the kernel)

// units of a package: header (4) + nb block records (2 each) + ne = nr + np operand pairs (1 each) + their
// right-hand side offsets (4 per unit) + their target tags (16 per unit)
inline __host__ __device__ int package_units(int nb, int ne)
{
	return 4 + 2 * nb + ne + (ne + 3) / 4 + (ne + 15) / 16;
}

struct TDevPlan {
	const TColDesc *cols;      // [n] in *schedule* order: the columns of task t are cols[task_ptr[t] .. task_ptr[t+1])
	const TBlkDesc *blks;      // [l_blocks]
	const longlong2 *pairs;    // [n_pairs] x = offset of L(i,c) | position of the target block in its column << 48 | dim(c) << 56, y = offset of L(j,c)
	const TRowEnt *rents;      // [n_row_entries]
	const int64_t *task_ptr;   // [n_tasks+1]
	int uniform_dim;           // > 0: every block column has this dimension (3, 6, 7 get unrolled kernels)
	int64_t n_blks, n_pairs, n_rents; // lengths of the arrays above (bulk loads of a task's records stop there)
	// column packages of the upper stages (fixed block size only): per task the offset, in 16-byte units, of its first
	// column's package in pkg, or -1; a package is the column's TColDesc followed -- if it fits the staged path of
	// factor_stage_kernel -- by its records in the very layout that kernel keeps them in LDS, so that one coalesced
	// read brings the descriptor and the records together instead of one after the other.  The packages of a task's
	// columns follow each other; pkg is padded so that a read of PKG_SPECULATIVE units from any package start stays inside
	const longlong2 *pkg;
	const int64_t *task_pkg;
	const int32_t *task_map;   // bottom stages, or null: the tasks a launch of the wave-per-task kernel works on, when they are not
	                           // a contiguous range (the others of the stage went to the lane-per-task kernel): task = task_map[task_begin + blockIdx.x]
	long long *p_timing;       // development aid (SLAMPP_HIP_STAGE_TIMING): [0] = launches so far, then 32 clock
	                           // samples per launch of workgroup 0 of the multi-wave factor kernel; normally null
};

// lane-per-task kernel of the wide stages (simt_kernel.hip): 64 tasks of one shape per wave.  A chunk names the shape's
// program (int32 stream: n_cols, n_blocks, n_ops, n_y, then per column nb, nr, n_touch, the column's n_touch distinct operands, nr x (operand, y index), and per
// sub-diagonal block np, np x (operand a, operand b)) and the chunk's table of per-lane offsets, [field][width]: per column
// (offset of its first factor block, offset of inv(L_jj), scalar offset in the workspace, scalar offset in the caller's
// vector), per block its source in Lambda ((offset << 1) | transposed, or -1), per operand its offset in the factor,
// per y index the scalar offset of that column in the workspace.  Lanes beyond the chunk's tasks repeat its last task.
struct TSimtChunk { // 16 B
	int32_t prog_off;  // in int32 units
	int32_t n_tasks;   // real tasks in the chunk (<= 64)
	int64_t tab_off;   // in int64 units
};

// returns false if the block dimension has no such kernel
bool launch_factor_simt(const TSimtChunk *chunks, int n_chunks, int n_width /* tasks per wave: 16, 32 or 64 */,
	int n_lds_bytes /* the largest table of the chunks: fields x width x 8 */,
	const int32_t *prog, const int64_t *tab, int n_dim,
	const double *A, double *L, double *Linv /* null: inv(L_jj) is not stored */, const double *b, double *w, int *p_flag, hipStream_t stream,
	long long *p_timing = 0 /* development aid, as TDevPlan::p_timing */, const TBatch &t_batch = t_No_Batch());
// backward substitution of the same tasks, a lane per task (chunk programs: n_cols, number of sub-diagonal blocks, nb per
// column; tables: per column offset of its first factor block, scalar offsets in the workspace and in the caller's vector,
// then the workspace offset of every sub-diagonal block's row): x_j from L_jj^T directly, no inverse
bool launch_backward_simt(const TSimtChunk *chunks, int n_chunks, int n_width, int n_lds_bytes, const int32_t *prog, const int64_t *tab,
	int n_dim, const double *L, double *w, double *x_out, hipStream_t stream, const TBatch &t_batch = t_No_Batch());
// inv(L_jj) of the columns cols[col_begin .. col_end) (schedule order) from their factor blocks: for callers that need the
// inverses the lane-per-task factorization did not store (another right-hand side, covariances)
bool launch_invert_diagonals(const TDevPlan &p, int64_t col_begin, int64_t col_end, const double *L, double *Linv, hipStream_t stream);

// numeric factorization of one stage, with the forward substitution y = L^-1 b fused in
// (b is read at its original position, y written to the permuted workspace w)
void launch_factor_stage(const TDevPlan &p, const double *A, double *L, double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, bool b_bottom_stage, int *p_flag, hipStream_t stream, const TBatch &t_batch = t_No_Batch());
// bottom stages out of LDS (subtree_kernel.hip); returns false if the block dimension has no such kernel
bool launch_factor_subtree_image(const TDevPlan &p, const double *A, double *L, double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, int *p_flag, hipStream_t stream, const TBatch &t_batch = t_No_Batch());
// the wide stages right above the leaves: thousands of single separator columns whose operands other launches produced;
// the separator kernel with one wave per column and small capacities (needs the column packages of those tasks)
void launch_factor_wide(const TDevPlan &p, const double *A, double *L, double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, int *p_flag, hipStream_t stream, const TBatch &t_batch = t_No_Batch());
// stand-alone forward substitution (another right-hand side with a kept factor)
void launch_forward_stage(const TDevPlan &p, const double *L, const double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, hipStream_t stream);
// backward substitution x = L^-T y, scattering x to its original position
void launch_backward_stage(const TDevPlan &p, const double *L, const double *Linv, double *w,
	double *x_out, int task_begin, int n_tasks, hipStream_t stream, const TBatch &t_batch = t_No_Batch());

// dense top (see plan.h): Schur complement of the block-eliminated part onto the dense-top columns,
// written into the lower triangle of the dense matrix Dm (leading dimension ld, right-hand side in row ld-1)
void launch_dense_assemble(const TDevPlan &p, const TDenseBlk *blks, int n_blks, const double *A, const double *L,
	const double *b, const double *w, double *Dm, int ld, bool b_rhs_only, hipStream_t stream);
void launch_dense_gather_factor(const TDenseBlk *blks, const int64_t *loffs, int n_blks, const double *Dm, int ld, double *L, hipStream_t stream);
void launch_gather_values(const int64_t *p_map, int64_t n, const double *p_src, double *p_dst, hipStream_t stream);

} // namespace slampp
