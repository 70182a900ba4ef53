// sparse_inverse.h -- sparse inverse subset on the pattern of the block factor (sparse_inverse.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plan.h"

namespace slampp {

struct CSparseInverse;

// lists for a plan with one block size (3, 6, 7) and no dense top; 0 if the plan is not of that kind; throws
CSparseInverse *sparse_inverse_setup(const Plan &P, hipStream_t stream);
void sparse_inverse_destroy(CSparseInverse *p);
size_t sparse_inverse_bytes(const CSparseInverse *p);
// Z, laid out like the factor L: block (i,j) of (L L^T)^-1 for every block of L's pattern (diagonal blocks whole)
void sparse_inverse_enqueue(const CSparseInverse &r_inv, const Plan &P, const double *L, const double *Linv, double *Z,
	hipStream_t stream);
// offset of the factor block (i, k), i >= k, in L / Z, or -1
int64_t plan_block_offset(const Plan &P, int32_t i, int32_t k);

} // namespace slampp
