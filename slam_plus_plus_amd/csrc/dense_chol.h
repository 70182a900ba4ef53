// dense_chol.h -- dense fp64 Cholesky of the reduced camera system on gfx950 (MFMA f64 16x16x4)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slampp {

enum { dense_NB = 64 }; // panel width = tile size

// Padded dimension for an n x n system with one right-hand side riding along as an extra row.
inline int dense_padded_dim(int n) { return ((n + 1 + dense_NB - 1) / dense_NB) * dense_NB; }

// Writes identity on the padding diagonal (rows n .. n_pad-2), zeroes nothing else: the caller
// fills the lower triangle of rows < n and the right-hand side into row n_pad-1, columns < n.
void dense_prepare_padding(double *M, int n_pad, int n, hipStream_t stream);

// In-place lower Cholesky M = L L^T of the n_pad x n_pad column-major matrix (ld = n_pad); only the
// lower triangle is read or written.  Row n_pad-1 carries the right-hand side, so after the call
// L(n_pad-1, 0:n) = y^T = (L^-1 rhs)^T (forward substitution fused into the panel updates).
// p_invdiag: workspace (n_pad / 64) * 64 * 64 doubles, receives inv(L_kk) of every diagonal tile.
// Sets *p_flag |= 1 if a pivot of a row < n is not positive.
void dense_cholesky(double *M, int n_pad, int n, double *p_invdiag, int *p_flag, hipStream_t stream);

// x = L^-T y with y taken from row n_pad-1 of the factor; p_z: workspace n_pad doubles; p_x: n_pad doubles, x in [0, n)
void dense_backsolve(const double *M, int n_pad, int n, const double *p_invdiag, double *p_z, double *p_x, hipStream_t stream);

// y = L^-1 r for another right-hand side with a kept factor: r sits in row n_pad-1 (columns < n) and is
// replaced by y, ready for dense_backsolve
void dense_forwardsolve(double *M, int n_pad, const double *p_invdiag, hipStream_t stream);

} // namespace slampp
