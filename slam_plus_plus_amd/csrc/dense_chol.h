// dense_chol.h -- dense fp64 Cholesky of the reduced camera system on gfx950 (MFMA f64 16x16x4)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace slampp {

enum { dense_NB = 64 }; // panel width = tile size

// Padded dimension for an n x n system with one right-hand side riding along as an extra row.
inline int dense_padded_dim(int n) { return ((n + 1 + dense_NB - 1) / dense_NB) * dense_NB; }

// Writes identity on the padding diagonal (rows n .. n_pad-2), zeroes nothing else: the caller
// fills the lower triangle of rows < n and the right-hand side into row n_pad-1, columns < n.
void dense_prepare_padding(double *M, int n_pad, int n, hipStream_t stream);
// the same for a list of positions inside the matrix (device array)
void dense_prepare_gaps(double *M, int n_pad, const int32_t *p_positions_dev, int n_positions, hipStream_t stream);

// In-place lower Cholesky M = L L^T of the n_pad x n_pad column-major matrix (ld = n_pad); only the
// lower triangle is read or written.  Row n_pad-1 carries the right-hand side, so after the call
// L(n_pad-1, 0:n) = y^T = (L^-1 rhs)^T (forward substitution fused into the panel updates).
// p_invdiag: workspace (n_pad / 64) * 64 * 64 doubles, receives inv(L_kk) of every diagonal tile.
// Sets *p_flag |= 1 if a pivot of a row < n is not positive.
void dense_cholesky(double *M, int n_pad, int n, double *p_invdiag, int *p_flag, hipStream_t stream);

// The pieces of dense_cholesky() for a factorization whose outer panels (OUTER = 4 tiles = 256 columns) are spread over
// several devices (group.hip: panel b belongs to member b mod P):
//   dense_factor_panel   factors the tile columns [t0, t1) of M, which must be up to date with every earlier panel:
//                        per tile potrf + inverse, panel solve, update of the panel's remaining columns;
//   dense_update_panels  applies the finished tile columns [k0, k1) to the lower tiles of the tile columns [c0, c1).
enum { dense_OUTER_TILES = 4 };
void dense_factor_panel(double *M, int n_pad, int n, int t0, int t1, double *p_invdiag, int *p_flag, hipStream_t stream);
void dense_update_panels(double *M, int n_pad, int k0, int k1, int c0, int c1, hipStream_t stream);

// Tile-sparse variant for the dense top of the sparse path: the separators assembled into one dense matrix still
// form a tree, so many 64 x 64 tiles are structurally zero and tile columns in different subtrees are independent.
// The schedule (built once per structure from the set of nonzero tiles) groups the tile columns by their height in
// the tile elimination tree; a level is three launches (potrf of all its diagonal tiles, trsm of all their
// sub-diagonal tiles, one gather-update per target tile that any of them touches), so the dependent chain is as
// long as the tree is high, not as long as the matrix is wide.
struct CTileSchedule {
	int n_tiles;                       // tiles per dimension
	int n_levels;
	std::vector<int> level_potrf_ptr, level_trsm_ptr, level_tgt_ptr; // [n_levels + 1] each
	std::vector<int> level_urgent_end; // [n_levels] the level's targets up to here are the next level's diagonal tiles (an update launch of their own)
	std::vector<int> rider_ptr;        // [n_levels + 1] into d_riders: the updates that ride in a level's diagonal launch (any launch between their sources' level and their deadline: dense_tiles.hip)
	int4 *d_riders;                    // (row tile, column tile, first source, one past the last source)
	int *d_potrf;                      // tile columns, level by level
	int4 *d_trsm;                      // (row tile, column tile, 1 = the workgroup also updates the diagonal tile of its row, -)
	int4 *d_tgt;                       // (row tile, column tile, first source, one past the last source)
	int *d_src;                        // source tile columns of the targets
	// backward substitution by the same levels, top down (tile_backsolve): launch q solves the tile columns of height
	// n_levels - 1 - q ("diagonal" records) and carries the x of the launch before it to every column further down
	std::vector<int> back_diag_ptr, back_carry_ptr; // [n_levels + 1] each, by launch
	int4 *d_back_diag;                 // (tile column k, its ancestor of height + 1 or -1, 1 = z_k is still y, -)
	int4 *d_back_carry;                // (source tile column j, target tile column k: z_k -= L(j,k)^T x_j, 1 = z_k is still y, -)
	CTileSchedule() :n_tiles(0), n_levels(0), d_riders(0), d_potrf(0), d_trsm(0), d_tgt(0), d_src(0), d_back_diag(0), d_back_carry(0) {}
	~CTileSchedule() { Free(); }
	CTileSchedule(const CTileSchedule&) = delete;
	CTileSchedule &operator =(const CTileSchedule&) = delete;
	void Free();
	size_t n_Bytes() const { return n_bytes; }
	// p_nonzero: n_tiles x n_tiles flags, column-major, lower triangle (row >= col); the diagonal and the last
	// tile row (it carries the right-hand side) are always taken as nonzero.  Returns false if the device arrays
	// could not be set up (the caller then uses dense_cholesky)
	bool Build(int n_tiles, const std::vector<char> &r_nonzero, hipStream_t stream);
private:
	size_t n_bytes = 0;
};

// same contract as dense_cholesky(), on the tiles of the schedule only
void tile_cholesky(const CTileSchedule &r_schedule, double *M, int n_pad, int n, double *p_invdiag, int *p_flag,
	hipStream_t stream);

// x = L^-T y by the levels of the schedule, top down: same contract as dense_backsolve() below, for a factor made by
// tile_cholesky() with this schedule.  One launch per level; every nonzero tile of the factor is read once, by one
// workgroup; what a level's columns wait for is one 64 x 64 product with the inverse of their diagonal tile and, where
// the column's nearest ancestor was solved by the launch before, one with that ancestor's tile -- everything else has been
// carried into z by earlier launches, one workgroup (one writer, a fixed order of sums) per tile.
void tile_backsolve(const CTileSchedule &r_schedule, const double *M, int n_pad, int n, const double *p_invdiag, double *p_z,
	double *p_x, hipStream_t stream, const longlong2 *p_dst = 0, double *p_w = 0, double *p_x_out = 0);

// zeroes the schedule's tiles of M (every tile a factorization by the schedule or an assembly into its pattern writes)
// p_unit (optional, n_pad bytes): positions whose diagonal entry is 1 afterwards (padding, alignment gaps)
void tile_zero(const CTileSchedule &r_schedule, double *M, int n_pad, hipStream_t stream, const uint8_t *p_unit = 0);

// x = L^-T y with y taken from row n_pad-1 of the factor; p_z: workspace n_pad doubles; p_x: n_pad doubles, x in [0, n)
// p_dst (optional, n_pad entries): where entry i of x also goes -- p_w[p_dst[i].x] and p_x_out[p_dst[i].y]; .x < 0: nowhere
void dense_backsolve(const double *M, int n_pad, int n, const double *p_invdiag, double *p_z, double *p_x, hipStream_t stream,
	const longlong2 *p_dst = 0, double *p_w = 0, double *p_x_out = 0);

// y = L^-1 r for another right-hand side with a kept factor: r sits in row n_pad-1 (columns < n) and is
// replaced by y, ready for dense_backsolve
void dense_forwardsolve(double *M, int n_pad, const double *p_invdiag, hipStream_t stream);

// Inverse of the matrix from its factor: on entry M holds L (dense_cholesky) and p_invdiag the inverses of its
// diagonal tiles; on return M holds X = inv(L) (lower; diagonal tiles as full tiles with zeros above) and Z the
// inverse X^T X = inv(L L^T): its lower tiles and the whole of its diagonal tiles (Z also serves as the scratch of
// the first step).  2 n^3 / 3 flops in 2 log2(n / 64) + 2 launches.
void dense_inverse_from_factor(double *M, int n_pad, const double *p_invdiag, double *Z, hipStream_t stream);

} // namespace slampp
