// sparse_kernels.h -- device view of the elimination plan + kernel launchers (sparse path)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slampp {

struct TDevPlan {
	const int32_t *dim;        // [n] block dimension, new order
	const int64_t *cs_new;     // [n+1] scalar offset in the permuted workspace
	const int64_t *cs_src;     // [n] scalar offset in the caller's vector
	const int64_t *lptr;       // [n+1]
	const int32_t *lrow;       // [l_blocks]
	const int64_t *loff;       // [l_blocks+1]
	const int64_t *asrc;       // [l_blocks] (offset in Lambda values) * 2 + transposed, or -1
	const int64_t *linv_off;   // [n+1]
	const int64_t *pptr;       // [l_blocks+1]
	const longlong2 *pairs;    // [n_pairs] x = offset of L(i,c) | dim(c) << 56, y = offset of L(j,c)
	const int64_t *rptr;       // [n+1]
	const int64_t *roff;       // [n_row_entries] offset of L(j,c)
	const int32_t *rcol;       // [n_row_entries] c
	const int64_t *task_ptr;   // [n_tasks+1]
	const int32_t *task_cols;  // [n]
};

void launch_factor_stage(const TDevPlan &p, const double *A, double *L, double *Linv,
	int task_begin, int n_tasks, int n_waves, int *p_flag, hipStream_t stream);
void launch_forward_stage(const TDevPlan &p, const double *L, const double *Linv, const double *b,
	double *w, int task_begin, int n_tasks, hipStream_t stream);
void launch_backward_stage(const TDevPlan &p, const double *L, const double *Linv, double *w,
	double *x_out, int task_begin, int n_tasks, hipStream_t stream);

} // namespace slampp
