#pragma once
// Code objects are loaded at the first launch of any kernel in them (one per translation unit of this library: 0.4 ms for a
// small one, more for the template-heavy ones; tools/micro/first_launch_cost.hip -- later kernels of a loaded code object
// cost microseconds).  Every unit that holds kernels of the solve paths defines an empty kernel and a function that
// launches it; the handle's bring-up thread (capi.hip) calls them while the caller's thread is busy with the analysis.
#include <hip/hip_runtime.h>

#define SLAMPP_PRELOAD_UNIT(unit) \
	namespace slampp { \
	__global__ void preload_##unit##_kernel() {} \
	void preload_##unit(hipStream_t stream) { hipLaunchKernelGGL(preload_##unit##_kernel, dim3(1), dim3(1), 0, stream); } \
	}

namespace slampp {

void preload_simt_kernel(hipStream_t stream);
void preload_panel_kernel(hipStream_t stream);
void preload_sparse_kernels(hipStream_t stream);
void preload_subtree_kernel(hipStream_t stream);
void preload_dense_tiles(hipStream_t stream);
void preload_dense_chol(hipStream_t stream);
void preload_schur(hipStream_t stream);
void preload_schur_tiles(hipStream_t stream);

// the units of the sparse path first (a Schur-mode handle uses them for the reduced system as well)
inline void preload_device_code(hipStream_t stream)
{
	preload_simt_kernel(stream);
	preload_panel_kernel(stream);
	preload_sparse_kernels(stream);
	preload_subtree_kernel(stream);
	preload_dense_tiles(stream);
	preload_dense_chol(stream);
	preload_schur(stream);
	preload_schur_tiles(stream);
}

} // ~slampp
