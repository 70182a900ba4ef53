// sparse_inverse.h -- sparse inverse subset on the pattern of the block factor (sparse_inverse.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plan.h"

namespace slampp {

struct CSparseInverse;

// lists for a plan with one block size (3, 6, 7) and -- unless allowed -- no dense top, or with any mix of block sizes up
// to 8 and no dense top; 0 if the plan is not of that kind; throws
CSparseInverse *sparse_inverse_setup(const Plan &P, hipStream_t stream, bool b_allow_dense_top = false);
void sparse_inverse_destroy(CSparseInverse *p);
size_t sparse_inverse_bytes(const CSparseInverse *p);
// Z, laid out like the factor L: block (i,j) of (L L^T)^-1 for every block of L's pattern (diagonal blocks whole)
// (blocks among dense-top columns are read from p_dense_top_inverse, the inverse of the top's Schur complement: lower
// triangle + diagonal tiles valid, leading dimension n_dense_ld)
void sparse_inverse_enqueue(const CSparseInverse &r_inv, const Plan &P, const double *L, const double *Linv, double *Z,
	hipStream_t stream, const double *p_dense_top_inverse = 0, int n_dense_ld = 0);
// one d x d block per block column: Z + p_where[c], or for p_where[c] = -(position + 1) the dense top's inverse at it
void inverse_diag_blocks_launch(int64_t n, int d, const int64_t *p_where, const double *Z, const double *Zd, int ld, double *out,
	hipStream_t stream);
// ... mixed block sizes: block c has p_dim[c] rows, lives at Z + p_where[c] and goes to out + p_out_off[c]
void inverse_diag_blocks_any_launch(int64_t n, const int32_t *p_dim, const int64_t *p_where, const int64_t *p_out_off, const double *Z,
	double *out, hipStream_t stream);
// replaces the right-hand side row of a copy of the dense top's factor by an identity row
void dense_top_clear_rhs_row(double *M, int ld, hipStream_t stream);
// offset of the factor block (i, k), i >= k, in L / Z, or -1
int64_t plan_block_offset(const Plan &P, int32_t i, int32_t k);

} // namespace slampp
