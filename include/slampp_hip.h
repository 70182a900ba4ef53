/*
 * slampp_hip.h -- C ABI of libslampp_hip.so: an MI355X (gfx950) sparse block Cholesky /
 * Schur-complement solver for the normal equations  Lambda * dx = eta  of SLAM++.
 *
 * This is the drop-in boundary for the reference's linear-solver concept
 * (/root/reference/include/slam/LinearSolverTags.h:38-135).  The reference binds its solvers
 * as compile-time template parameters, so the binding a maintainer adds is the header class
 * include/slam/LinearSolver_HIP.h (CLinearSolver_HIP / CLinearSolver_Schur_HIP), which gathers
 * the blocks of a CUberBlockMatrix and forwards to the entry points below.  Each entry point
 * names the reference call it stands in for.
 *
 * Conventions
 *   - plain C types only; all functions return a status (SLAMPP_HIP_*), never throw, never exit;
 *   - Lambda is passed exactly as the reference stores it (BlockMatrixBase.h:380-503): block-CSC,
 *     only the upper triangle populated (LinearSolver_CholMod.cpp:57), block rows sorted inside a
 *     block column, each block dense column-major;  values are passed packed in that block order;
 *   - eta is overwritten with the solution (same contract as Solve_PosDef, LinearSolver_CholMod.h:186);
 *   - a handle is not thread-safe; different handles are independent (no process-global state).
 */
#ifndef SLAMPP_HIP_H_INCLUDED
#define SLAMPP_HIP_H_INCLUDED

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slampp_hip_solver slampp_hip_solver; /* opaque */

enum {
	SLAMPP_HIP_OK = 0,
	SLAMPP_HIP_NOT_POSDEF = 1,          /* reference: Solve_PosDef returns false (LinearSolver_CholMod.cpp:310-317) */
	SLAMPP_HIP_ERR_INVALID = -1,        /* bad argument / call order */
	SLAMPP_HIP_ERR_ALLOC = -2,          /* host or device out of memory (reference: std::bad_alloc) */
	SLAMPP_HIP_ERR_DEVICE = -3,         /* HIP runtime error (reference: std::runtime_error, LinearSolver_Schur.h:1196-1209) */
	SLAMPP_HIP_ERR_UNSUPPORTED = -4     /* an operation outside what this build handles for this structure (block columns wider than 8 are cut into pieces at analysis time and solved; see slampp_hip_factorize / slampp_hip_marginals for what they still need) */
};

enum {
	SLAMPP_HIP_MODE_SPARSE = 0,         /* pose-graph path: fill-reducing ordering + sparse block Cholesky */
	SLAMPP_HIP_MODE_SCHUR = 1           /* BA path: Schur complement on the last (landmark) block columns, dense reduced system */
};

/* phase wall-clock of the last factor_solve, ms; names follow the reference's timers
 * (NonlinearSolver_Lambda.h:250, LinearSolver_Schur.h:1896-1911) */
typedef struct slampp_hip_times {
	double order_ms, symbolic_ms, upload_ms, factor_ms, solve_ms, download_ms;  /* sparse path */
	double schur_ms, reduce_ms, cholsol_ms, backsubst_ms;                        /* Schur path */
	double total_ms;
} slampp_hip_times;

typedef struct slampp_hip_stats {
	int64_t n_bcols, n_blocks_upper, n_scalars, nnz_upper;  /* Lambda, as given */
	int64_t l_blocks, l_nnz;          /* block / scalar nonzeros of the factor under our ordering */
	double  factor_flops, solve_flops; /* CHOLMOD's convention: sum of squared column counts; 4*lnz */
	int64_t n_stages, n_tasks, etree_height, n_update_pairs;
	int64_t n_cams, n_points, n_observations, schur_dim;     /* Schur path, else 0 (sparse path: schur_dim = dimension of the dense top) */
	int64_t device_bytes;
	int64_t n_bottom_stages;          /* leading (wide) stages: stage 0 by the lane-per-task kernel, the others one wave per task */
} slampp_hip_stats;

/* lifecycle -- stands in for the solver object's ctor / dtor / Free_Memory()
 * (LinearSolver_CholMod.h:148-167).  device_id: HIP device ordinal.
 * SLAMPP_HIP_ERR_DEVICE without a usable device of that ordinal (there is no CPU fallback).  The handle's HIP streams are
 * created on a thread of the library's while the caller goes on to set_structure / analyze (milliseconds each in a process
 * that has made none yet); the first entry point that needs them waits for that thread and is the one to report
 * SLAMPP_HIP_ERR_DEVICE if they could not be made. */
int slampp_hip_create(slampp_hip_solver **pp_solver, int device_id);

/* One handle over several devices of this process (SURVEY.md section 8b proposed create(device_ids*, n_dev); section 8e:
 * BA shards by landmark).  The reference's nonlinear solvers call their linear solver from one thread of one process
 * (include/slam/NonlinearSolver_Base.h:344-346,400,438; NonlinearSolver_Lambda_LM.h:1543-1552), so this is how a BA
 * solve reaches GPUs 1..n-1 behind the unchanged CLinearSolver_* calls: in SCHUR mode slampp_hip_analyze cuts the
 * landmarks into n contiguous shards balanced by observation count (slampp_hip_landmark_shard), every device gets an
 * internal solver and a host thread of its own, slampp_hip_factor_solve / _schur_marginals / _solve_marginal_poses send
 * each member its landmark columns of Lambda straight from the caller's (pinned) arrays over its own PCIe link, the
 * partial reduced camera systems are summed by an all-reduce inside the library -- RCCL bound at run time (dlopen of
 * librccl.so, ncclCommInitAll over the device list, rings over xGMI), or a direct exchange through peer pointers where
 * RCCL is absent or a device is listed twice (option "group_exchange": 0 = that choice, 1 = RCCL or fail, 2 = peer
 * pointers; environment SLAMPP_HIP_GROUP_EXCHANGE=rccl|peer) -- and every member writes its landmarks' part of the
 * solution into the caller's vector.  In SPARSE mode (pose graphs: one elimination tree, nothing to shard) the handle
 * is a plain solver on p_device_ids[0].  n_devices = 1 is slampp_hip_create.  The entry points that take device
 * pointers (…_device, …_device_async, slampp_hip_sync) apply to one device and return SLAMPP_HIP_ERR_INVALID on a
 * handle that is solving with shards; slampp_hip_set_allreduce likewise (the exchange is the library's).  A device may
 * be listed more than once (that is how the 1-GPU test boxes run two members). */
int slampp_hip_create_multi(slampp_hip_solver **pp_solver, const int *p_device_ids, int n_devices);

/* what a handle made by slampp_hip_create_multi is doing: members in use (0 = not sharded: one device, sparse mode, or
 * not analyzed yet), the landmark range [p_point_bounds[r], p_point_bounds[r + 1]) of every member (n + 1 entries, may
 * be NULL) and the exchange in use ("rccl (<library>)" / "peer" / "none"; valid until the next analyze) */
int slampp_hip_group_info(const slampp_hip_solver *p_solver, int *p_member_num, int64_t *p_point_bounds, int n_max_members,
	const char **pp_s_exchange);

/* how many exchanges (all-reduces of the reduced camera system) the members of a device group have gone into since the
 * handle was made.  Failure agreement: before a collective is enqueued the members meet at a host barrier with their
 * status, for RCCL and peer pointers alike; when one of them failed on its way there (allocation, upload, device error)
 * NOBODY enqueues -- the call returns that member's error, this count does not move, the next solve runs normally.  The
 * option "group_fail_member" = r + 1 makes member r fail that way (0 = off; a test hook).  The reference has no
 * counterpart: it is one thread on one device (include/slam/NonlinearSolver_Lambda_LM.h:1543-1552). */
int slampp_hip_group_exchange_count(const slampp_hip_solver *p_solver, int64_t *p_n_enqueued);

/* the exchange of slampp_hip_create_multi by itself, for deployment checks and tests: one member per listed device,
 * every member fills n_count doubles with a pattern of its own, the buffers are summed twice through the members'
 * all-reduce (n_exchange as the option "group_exchange": 0 / 1 = RCCL / 2 = peer pointers; 1 also runs with a single
 * device: the RCCL calls themselves) and every member checks every entry.  p_s_exchange_out receives the name of the
 * exchange used (or what went wrong). */
int slampp_hip_group_exchange_selftest(const int *p_device_ids, int n_devices, int n_exchange, int64_t n_count,
	char *p_s_exchange_out, int n_max_chars);

/* the landmark shard a rank of a multi-GPU BA solve holds, host code only (no GPU, no handle): rank n_rank of n_world
 * keeps block columns [0, n_matrix_cut) -- the cameras -- and a contiguous range of the landmark block columns balanced
 * by observation count; first call with the three pointers NULL for the sizes.  The shard's packed values are values
 * [0, n_camera_values) followed by values [n_value_begin, n_value_end) of the full system, its right-hand side entries
 * [0, n_camera_scalars) followed by [n_scalar_begin, n_scalar_end); the camera blocks and the camera part of eta are
 * added by one rank only (option "shard_primary").  The same rule as slam_plus_plus_amd/sharding.py. */
typedef struct slampp_hip_shard_view {
	int64_t n_bcols, n_blocks;
	int64_t n_camera_values, n_value_begin, n_value_end;
	int64_t n_camera_scalars, n_scalar_begin, n_scalar_end;
	int64_t n_point_begin, n_point_end;
	int64_t *p_bcol_cumsum;   /* [n_bcols + 1] */
	int64_t *p_bcol_ptr;      /* [n_bcols + 1] */
	int32_t *p_brow_idx;      /* [n_blocks] */
} slampp_hip_shard_view;
int slampp_hip_landmark_shard(int64_t n_bcols, const int64_t *p_bcol_cumsum, const int64_t *p_bcol_ptr,
	const int32_t *p_brow_idx, int64_t n_matrix_cut, int n_rank, int n_world, slampp_hip_shard_view *p_view);

void slampp_hip_destroy(slampp_hip_solver *p_solver);
int slampp_hip_free_memory(slampp_hip_solver *p_solver);
const char *slampp_hip_last_error(const slampp_hip_solver *p_solver);

/* Options (19).  All but the ones marked (*) take effect at the next slampp_hip_analyze.
 *
 * ordering / schedule of the sparse block path
 *   "natural_order"      0 / 1: keep the caller's block order instead of nested dissection (default 0; what
 *                        Factorize_PosDef_Blocky needs: the reference factors the matrix as it is ordered)
 *   "leaf_size"          nested-dissection leaf, in block columns (default 3)
 *   "subtree_size"       most columns one wave eliminates sequentially at the bottom of the tree (default 8)
 *   "task_height"        1 .. 8 (default 6): above the wide stages a separator task is a slice of the elimination tree this
 *                        many levels high; a launch covers that many levels, the slice's blocks stay in LDS
 *   "dense_top_nb"       block columns with at least this many blocks, and their ancestors, are factored as one dense matrix on
 *                        the matrix cores (default 24, or 16 / 36 where a model of the dependent launch chain prefers that;
 *                        setting the option fixes the threshold; 0 = off)
 *   "dense_top_max_dim"  cap on its dimension (default 12288)
 *   "dense_top_min_dim"  below this dimension there is no dense top (default 192)
 * Schur mode
 *   "schur_sparse"       how the reduced camera system S is factored: -1 (default) = by the sparse block path when fewer than
 *                        15 % of its camera-camera blocks are nonzero, 0 = dense on the matrix cores, 1 = sparse (the reference
 *                        chooses at compile time: __SCHUR_USE_DENSE_SOLVER, include/slam/LinearSolver_Schur.h:48-55)
 *   "schur_tiles"        how S = A - U C^-1 U^T is assembled: -1 (default) = landmark by landmark (runs of landmarks seen by the
 *                        same cameras on the matrix cores, tiles of neighbouring landmarks in LDS) where that takes at least
 *                        half of the contributions, per-block contribution lists for the rest; 0 = lists only; 1 / 2 / 3 =
 *                        runs and tiles wherever possible / tiles only / runs of any length
 *   "schur_incremental"  0 / 1 / 2: keep the assembled S for slampp_hip_schur_set_changed_points; 1 = a solve with a list of
 *                        changed landmarks updates it when that is the shorter way, 2 = whenever a list is given
 *   "schur_fallback"     default 1: a structure the Schur kernels do not take (no landmark part, landmark-landmark blocks,
 *                        block sizes other than (6,3), (7,3), (3,2)) is solved by the sparse block path, as the reference
 *                        solves it (LinearSolver_Schur.h:1635-1638, 1721-1726); 0 = slampp_hip_analyze reports
 *                        SLAMPP_HIP_ERR_UNSUPPORTED / _INVALID instead
 *   "marginals_dense"    (*) 1 = slampp_hip_schur_marginals always inverts S densely; 0 (default) = a sparse inverse subset on
 *                        the factor's pattern when the solves factor S by the sparse block path
 * several GPUs
 *   "shard_primary"      process-per-GPU sharding: this rank adds A and eta_x (default 1)
 *   "shard_rank", "shard_world"  (*) optional: who this rank is among the ranks behind the all-reduce callback; lets them
 *                        exchange their block lists instead of an indicator over all camera pairs (limited to 16384 cameras)
 *   "group_exchange"     handles made by slampp_hip_create_multi, see there
 * host side
 *   "staging_ahead"      (*) 0 / 1: slampp_hip_analyze also brings up the pinned host staging of slampp_hip_host_staging on a
 *                        thread of its own next to the ordering (callers that hand over host arrays: the header class sets it)
 *   "assembly_groups"    (*) slampp_hip_assembly_create: the most vertices one group of the Lambda assembly takes (default: as
 *                        many as fit in LDS; 0 = no groups, one wave per block of Lambda)
 *   "profile"            (*) 0 / 1 / 2 / 3, see slampp_hip_get_profile
 *
 * Anything else set_option knows ("panel", "panel_rows", "panel_handup", "simt", "simt_width", "simt_stages",
 * "simt_backward", "wide_min_tasks", "nd_balance", "dense_nb", "dense_top_tiles", "schur_distributed", "group_fail_member")
 * is a development option: an alternative the defaults were measured against, or a test hook.  They are refused with
 * SLAMPP_HIP_ERR_INVALID unless the process runs with SLAMPP_HIP_DEV=1, and are described where they are implemented
 * (csrc/solver.h). */
int slampp_hip_set_option(slampp_hip_solver *p_solver, const char *p_s_name, int64_t n_value);

/* structure of Lambda -- stands in for what the reference's wrappers read through
 * n_BlockColumn_Num() / n_BlockColumn_Base() / n_BlockColumn_Block_Num() / n_Block_Row()
 * (BlockMatrix.h:343-392,407-431) and for p_BlockStructure_to_Sparse (BlockMatrix.cpp:3880-3958).
 * Calling it again means "the block structure changed" = Clear_SymbolicDecomposition()
 * (LinearSolverTags.h:112-120). */
int slampp_hip_set_structure(slampp_hip_solver *p_solver, int64_t n_bcols,
	const int64_t *p_bcol_cumsum /* [n_bcols+1] */, const int64_t *p_bcol_ptr /* [n_bcols+1] */,
	const int32_t *p_brow_idx /* [n_blocks] */);

/* ordering + symbolic analysis -- stands in for SymbolicDecomposition_Blocky()
 * (LinearSolver_CholMod.cpp:868-951, LinearSolver_UberBlock.h:256-310; Schur: LinearSolver_Schur.h:1566-1606).
 * SCHUR mode: block columns [n_matrix_cut, n_bcols) are the landmarks (the reference's guided
 * ordering, LinearSolver_Schur.cpp:771-838, puts them last); they must form a block-diagonal C. */
int slampp_hip_analyze(slampp_hip_solver *p_solver, int n_mode, int64_t n_matrix_cut);

/* numeric factorization + solve -- stands in for Solve_PosDef_Blocky(lambda, eta)
 * (LinearSolverTags.h:130-134; LinearSolver_CholMod.cpp:738-866; LinearSolver_Schur.h:1623-1935).
 * p_values: packed block values of Lambda (host); p_rhs_inout: eta on entry, dx on return (host).
 * Returns SLAMPP_HIP_NOT_POSDEF if a pivot is not positive. p_times may be NULL. */
int slampp_hip_factor_solve(slampp_hip_solver *p_solver, const double *p_values,
	double *p_rhs_inout, slampp_hip_times *p_times);

/* Pinned host staging owned by the library, for callers that have to gather Lambda's values anyway (the header class:
 * a CUberBlockMatrix keeps its blocks in pooled pages behind per-block pointers, BlockMatrixBase.h:321,374,449-453):
 * *pp_values receives room for the packed values, *pp_rhs for the right-hand side / solution (either may be NULL).
 * Passing exactly these pointers to slampp_hip_factor_solve / _factorize / _marginals / _schur_marginals /
 * _solve_marginal_poses makes the transfers single DMA copies without a staging pass; any other host array is moved
 * through the same staging in chunks (1 MB first, doubling up to 32 MB) by a few host threads while the previous chunk is on the bus.  Valid until
 * the next slampp_hip_set_structure / _free_memory / _destroy (call again after set_structure). */
int slampp_hip_host_staging(slampp_hip_solver *p_solver, double **pp_values, double **pp_rhs);

/* Sends values [n_first, n_first + n_count) of the staging on their way (a copy stream of the library's own) while
 * the caller is still gathering the rest: chunks must follow each other, n_first = 0 starts a new pass; the next call
 * that takes the staged values sends what is left and waits for all of it.  Enqueue-only. */
int slampp_hip_upload_values_async(slampp_hip_solver *p_solver, int64_t n_first, int64_t n_count);

/* same with both arrays already resident in device memory (HBM); used by bench.py so that the
 * timed region excludes PCIe.  p_rhs_inout_dev is overwritten with the solution. */
int slampp_hip_factor_solve_device(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_rhs_inout_dev, slampp_hip_times *p_times);

/* another right-hand side with the factor of the last factor_solve
 * (reference: cholmod_solve / cs_lsolve+cs_ltsolve on a kept factor, LinearSolver_CholMod.cpp:322-347) */
int slampp_hip_solve_again(slampp_hip_solver *p_solver, double *p_rhs_inout);

/* Numeric factorization only, the factor handed back to the host -- for CLinearSolverTag-style callers that keep
 * the factor themselves (the reference's Factorize_PosDef_Blocky, LinearSolver_CholMod.cpp:362-544, used by its
 * L / FastL solvers on a matrix they have ordered: set the option "natural_order" to 1 for that).  p_factor_out
 * receives l_values doubles: the lower factor L of the permuted Lambda, block-CSC over the CALLER's block columns as
 * slampp_hip_factor_structure describes it (perm, dim, lptr / lrow / loff), blocks column-major, the diagonal block
 * first in each column.  Like the reference's it takes any matrix: big separators are factored as one dense matrix on
 * the matrix cores (the "dense top") and their columns handed back in the same block layout; block columns wider than
 * 8 are factored in pieces and put together again (those only in the caller's own order: natural_order = 1).
 * Returns SLAMPP_HIP_NOT_POSDEF if a pivot is not positive. */
int slampp_hip_factorize(slampp_hip_solver *p_solver, const double *p_values, double *p_factor_out);
/* ... and the block structure of that factor (after slampp_hip_analyze, sparse mode): sizes through the first three
 * pointers, arrays where the pointers are not NULL -- perm[new] = old and dim[new] over the caller's n_bcols block
 * columns, lptr [n_bcols + 1], lrow / loff [l_blocks].  Stands in for the structure the reference's factor comes back
 * in, a CUberBlockMatrix (LinearSolver_CholMod.cpp:520-544: From_Sparse into r_factor). */
int slampp_hip_factor_structure(const slampp_hip_solver *p_solver, int64_t *p_n_bcols, int64_t *p_l_blocks, int64_t *p_l_values,
	int32_t *p_perm, int32_t *p_dim, int64_t *p_lptr, int32_t *p_lrow, int64_t *p_loff);

/* Schur mode, option "schur_incremental" = 1 (set before slampp_hip_analyze): the next factor_solve updates the reduced
 * camera system the previous factor_solve of this handle assembled instead of rebuilding it -- the reference's dog-leg
 * solver does the same from Omega = Lambda_new - Lambda_old (include/slam/NonlinearSolver_Lambda_DL.h:1025-1086, 2301-).
 * p_points (host): the landmarks, as indices among the landmark block columns, strictly increasing, whose blocks (C_p and
 * the U blocks of their observations) differ from the previous call's values; the camera-camera blocks and the
 * right-hand side may differ freely; n_points = 0 says that no landmark changed.  The contributions of the named
 * landmarks are exchanged (the old ones rebuilt from W = U C^-1 and C^-1, which the previous solve left on the device)
 * with fp64 atomic adds, so this path -- unlike every other one -- does not sum in a fixed order.  One-shot; falls back
 * to the full rebuild when there is nothing valid to update (first solve, a solve that was not positive definite, a
 * covariance call in between, landmark shards).  With the option on, the dense reduced system is kept in a second
 * n_pad^2 buffer (the factorization works in place). */
int slampp_hip_schur_set_changed_points(slampp_hip_solver *p_solver, const int64_t *p_points, int64_t n_points);

/* Schur mode only: solves for the landmarks alone, dl = C^-1 eta_l, and zeroes the camera part of the vector -- the
 * reference's CLinearSolver_Schur::Solve_PosDef_Blocky_MarginalPoses (include/slam/LinearSolver_Schur.h:1956-2143).
 * Returns SLAMPP_HIP_NOT_POSDEF if a landmark block is not positive definite (the reference inverts it regardless). */
int slampp_hip_solve_marginal_poses(slampp_hip_solver *p_solver, const double *p_values, double *p_rhs_inout);
int slampp_hip_solve_marginal_poses_device_async(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_rhs_inout_dev);

/* Sparse mode: block diagonal of the covariance matrix Lambda^-1 -- what the reference's nonlinear solvers get from
 * CMarginals::Calculate_DenseMarginals_Recurrent_FBS(margs, R, ordering, mpart_Diagonal) after ordering and factoring
 * Lambda once more on the host for the purpose (include/slam/NonlinearSolver_Lambda.h:700-760, "todo - reuse what the
 * linear solver calculated").  Numeric factorization, then the blocks of the inverse on the factor's pattern by the
 * recursion run from the root of the elimination tree downwards; p_block_diag receives one dim x dim column-major block
 * per block column, in the order of slampp_hip_set_structure.  Where the plan has a dense top, its part of the
 * inverse is the dense inverse of the top's Schur complement (matrix cores), and the recursion continues below it.
 * One block size (3, 6 or 7: unrolled kernels), or -- like the reference's, Marginals.h:1694, which takes any -- a mix of
 * block sizes up to 8 (poses and landmarks in one graph): block c is d_c x d_c, the blocks follow each other; that
 * one without a dense top (option dense_top_nb = 0).  Block columns wider than 8: SLAMPP_HIP_ERR_UNSUPPORTED. */
int slampp_hip_marginals(slampp_hip_solver *p_solver, const double *p_values, double *p_block_diag);
int slampp_hip_marginals_device_async(slampp_hip_solver *p_solver, const double *p_values_dev, double *p_block_diag_dev);

/* Schur mode only: block diagonal of the covariance matrix Lambda^-1 -- the reference's
 * CSchurComplement_Marginals::Schur_Marginals (include/slam/BAMarginals.h:579-806, called from
 * NonlinearSolver_Lambda_LM.h:1326 and NonlinearSolver_Lambda_DL.h:1640 with the Cholesky factor of the Schur
 * complement it has to compute for the purpose).  Here the reduced camera system is assembled, factored and inverted
 * on the device in one call: p_cam_cov receives n_cams blocks of dc x dc doubles (the diagonal blocks of S^-1; may be
 * NULL to skip them, as b_do_cam_marginals = false does), p_point_cov n_points blocks of dp x dp doubles
 * (C_p^-1 + W_p^T S^-1 W_p), column-major, in the block order of slampp_hip_set_structure.  When the solves
 * factor the reduced system by the sparse block path (option "schur_sparse"), the blocks of S^-1 come from a sparse
 * inverse subset on that factor's pattern (every camera pair that shares a landmark is in it); otherwise, or with
 * option "marginals_dense", S is inverted densely (2 x 8 n^2 bytes of device memory, n = n_cams dc).  With landmark
 * shards every rank calls it, passes its own values and receives the covariances of its own landmarks (the
 * all-reduce callback is invoked once, on the packed blocks or on the whole n_pad^2 buffer).  Returns
 * SLAMPP_HIP_NOT_POSDEF like the solve. */
int slampp_hip_schur_marginals(slampp_hip_solver *p_solver, const double *p_values, double *p_cam_cov, double *p_point_cov);
int slampp_hip_schur_marginals_device_async(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_cam_cov_dev, double *p_point_cov_dev);

/* enqueue-only variants for benchmarking / stream capture: no host synchronisation, no status
 * read-back; slampp_hip_sync() waits and returns OK / NOT_POSDEF / error for everything enqueued since the
 * previous slampp_hip_sync() (NOT_POSDEF if any of those factorizations was not positive definite) */
int slampp_hip_factor_solve_device_async(slampp_hip_solver *p_solver, const double *p_values_dev,
	double *p_rhs_inout_dev);
int slampp_hip_sync(slampp_hip_solver *p_solver);

/* K value sets of the analyzed structure in ONE pass of launches -- K damping values of one Lambda, K graphs of one shape --,
 * sparse mode: member k reads p_values_dev + k * n_values_stride and overwrites p_rhs_inout_dev + k * n_rhs_stride with its
 * solution (strides in doubles, at least the length of one member's array; 16-byte aligned bases and even strides let the
 * members share launches, anything else is solved one member after the other with the same result).  The chain of dependent
 * launches is as long as for one system and every launch K times as wide: a pose-graph solve fills a fraction of the
 * chip, K of them fill it.  Stands where the reference's LM loop re-damps and re-solves one value after the other
 * (NonlinearSolver_Lambda_LM.h:967-1001, 1660-1676); SURVEY section 8(e): pose graphs do not shard, "replicas only
 * (multiple independent problems / damping values per GPU)".  slampp_hip_sync_batch waits and reports per member:
 * p_status[k] = SLAMPP_HIP_OK or SLAMPP_HIP_NOT_POSDEF (a member that is not positive definite does not disturb the
 * others).  The handle's own factor (slampp_hip_solve_again, covariances) is not touched by a batch of more than one that
 * shares launches.  A batch that is solved member by member (a dense top, regrouped block columns, unaligned bases or odd
 * strides) goes through the handle's factor arrays: after more than one such member the handle has NO factor of its own
 * (slampp_hip_solve_again and the covariance calls return SLAMPP_HIP_ERR_INVALID until the next factorization); a batch of
 * one is an ordinary factorization, and its factor is dropped if slampp_hip_sync_batch reports the member not positive definite. */
#define SLAMPP_HIP_MAX_BATCH 64
int slampp_hip_factor_solve_batch_device_async(slampp_hip_solver *p_solver, int n_batch, const double *p_values_dev,
	int64_t n_values_stride, double *p_rhs_inout_dev, int64_t n_rhs_stride);
int slampp_hip_sync_batch(slampp_hip_solver *p_solver, int *p_status /* [n_batch] */, int n_batch);
void *slampp_hip_stream(slampp_hip_solver *p_solver); /* the hipStream_t every kernel is launched on */

int slampp_hip_get_stats(const slampp_hip_solver *p_solver, slampp_hip_stats *p_stats);

/* Schur mode, after the first solve: the same figures for the inner solver that factors the reduced camera system by the
 * sparse block path (its factor's nonzeros and flops under our ordering, stages, the dimension of its dense top in
 * schur_dim) -- what the roofline of the "reduced_sparse" phase is priced with; all zero when the reduced system is
 * factored densely (then slampp_hip_get_stats has n^3/3).  A handle over several devices answers for member 0 (the
 * reduced system is the same on every member). */
int slampp_hip_get_reduced_stats(const slampp_hip_solver *p_solver, slampp_hip_stats *p_stats);

/* device-side phase timing (the counterpart of the reference's __SCHUR_PROFILING / CTimerSampler
 * phase timers, LinearSolver_Schur.h:1681-1912, Timer.h:391): with option "profile" = 1 every phase
 * of factor_solve is bracketed by HIP events on the solver's stream; the totals are collected at
 * slampp_hip_sync().  Phases: factor_leaves, factor_rest (with "profile" = 2: factor_wide, factor_upper), forward, backward (sparse path);
 * schur_init, schur_tiles (the landmark-major assembly) and / or schur_points, schur_gather, schur_rhs (contribution lists),
 * reduced_sparse or dense_chol + dense_solve, backsubst (Schur path); a phase that launches nothing is left out.
 * An event pair costs microseconds of stream time: "profile" = 3 keeps only the phase of the kernel that moves most of a
 * step's bytes or flops (factor_leaves; schur_tiles / schur_gather; dense_chol) -- what a timed region should carry. */
typedef struct slampp_hip_phase_time {
	char name[32];
	int64_t n_count;      /* times the phase ran */
	double f_total_ms;    /* summed device time */
} slampp_hip_phase_time;
int slampp_hip_get_profile(slampp_hip_solver *p_solver, slampp_hip_phase_time *p_phases, int n_max_phases,
	int *p_phase_num, int b_reset);

/* The same totals under the names the reference prints with __SCHUR_PROFILING (LinearSolver_Schur.h:1681-1912, in its order):
 *   "reperm", "slice", "transpose"   always 0: the packed block-CSC with the cameras first is [A U; C] already
 *   "inverse + multiply + add"       C^-1, U C^-1, (U C^-1) U^T and A - ... are one pass here (schur_init + schur_tiles /
 *                                    schur_points + schur_gather); the reference's lines "inverse", "diag gemm", "scale",
 *                                    "gemm" and "add" added up are its counterpart
 *   "RHS prep"                       schur_rhs (zero where the landmark-major assembly forms the right-hand side on its way)
 *   "cholsol"                        reduced_sparse, or dense_chol + dense_solve
 *   "dy solve"                       backsubst
 * and, in sparse mode, CHOLMOD's two numeric phases (LinearSolver_CholMod.cpp:322-347): "factorize" (with the forward
 * substitution, which is fused into it) and "solve" (the backward one).  Nothing is reset by this call. */
int slampp_hip_get_profile_reference_names(slampp_hip_solver *p_solver, slampp_hip_phase_time *p_phases, int n_max_phases,
	int *p_phase_num);

/* Lambda and eta assembled on the device from per-edge Jacobians (first row after the
 * solve): replaces the host loop that computes every edge's Hessian blocks and the reduction that sums
 * them into Lambda (include/slam/BaseTypes_Binary.h:759-840 Calculate_Hessians_v2,
 * include/slam/NonlinearSolver_Lambda_Base.h:1634-1688 Refresh_Lambda, :1520-1580 the unary factor) for one
 * homogeneous set of binary edges, so that Lambda never visits the host between linearization and solve.
 *
 * create: after slampp_hip_set_structure().  Edge e joins block columns p_vertex0[e] and p_vertex1[e] of
 * Lambda (all vertex0 of one dimension d0, all vertex1 of one dimension d1, both <= 7) with an
 * n_residual_dim-vector residual (<= 8); Lambda must hold the block (min, max) of every edge.
 *
 * assemble: device arrays, edge-major, every matrix column-major as the reference's Eigen types are:
 *   p_J0_dev [n_edges][rd x d0], p_J1_dev [n_edges][rd x d1], p_sigma_inv_dev [n_edges][rd x rd],
 *   p_error_dev [n_edges][rd], p_weight_dev [n_edges] robust weights or NULL (= 1).
 * p_unary_factor (host, d x d column-major, or NULL) and p_unary_error (host, d, or NULL) anchor
 * block column n_unary_vertex: U^T U is added to its diagonal block and the error to its eta.
 * Writes the packed block values of Lambda (the layout slampp_hip_factor_solve_device reads) and eta;
 * with b_accumulate != 0 adds to what the two arrays hold instead (a second edge set of the same Lambda).
 * Sums run in a fixed order: results are bit-reproducible.  Enqueue-only, on the solver's stream. */
/* Levenberg-Marquardt damping of device-resident values (the ones slampp_hip_assemble_device_async wrote): adds
 * f_alpha to the diagonal of the diagonal blocks of block columns [n_first_vertex, n_last_vertex) -- the reference's
 * ApplyDamping(r_lambda, f_alpha, n_first_vertex, n_last_vertex), include/slam/NonlinearSolver_Lambda_LM.h:228-239,
 * which it runs on the host matrix between Refresh_Lambda and the linear solve.  A negative f_alpha takes it back
 * out (the LM loop re-damps the same Lambda).  Enqueue-only, on the solver's stream; needs set_structure only. */
int slampp_hip_apply_damping_device_async(slampp_hip_solver *p_solver, double *p_values_dev, double f_alpha,
	int64_t n_first_vertex, int64_t n_last_vertex);

typedef struct slampp_hip_assembly slampp_hip_assembly; /* opaque */
int slampp_hip_assembly_create(slampp_hip_solver *p_solver, slampp_hip_assembly **pp_assembly, int64_t n_edges,
	const int64_t *p_vertex0, const int64_t *p_vertex1, int n_residual_dim);
void slampp_hip_assembly_destroy(slampp_hip_assembly *p_assembly); /* in either order with its solver's destroy */
int slampp_hip_assemble_device_async(slampp_hip_assembly *p_assembly, const double *p_J0_dev, const double *p_J1_dev,
	const double *p_sigma_inv_dev, const double *p_error_dev, const double *p_weight_dev, int64_t n_unary_vertex,
	const double *p_unary_factor, const double *p_unary_error, double *p_values_dev, double *p_eta_dev, int b_accumulate);

/* Every edge type of a graph in one call, as the reference's Refresh_Lambda reduces them all
 * (include/slam/NonlinearSolver_Lambda_Base.h:1659-1688: one loop per edge pool of the typelist): n_sets homogeneous
 * edge sets of the same Lambda -- pose-pose and pose-landmark edges, say -- each with its own assembly handle and
 * device arrays as above.  Lambda and eta start from zero (b_accumulate != 0: from what the arrays hold), every set is
 * added in the order given (fixed order: bit-reproducible), the unary factor with the first.  Enqueue-only. */
typedef struct slampp_hip_edge_set {
	slampp_hip_assembly *p_assembly;
	const double *p_J0_dev, *p_J1_dev, *p_sigma_inv_dev, *p_error_dev, *p_weight_dev; /* as slampp_hip_assemble_device_async */
} slampp_hip_edge_set;
int slampp_hip_assemble_sets_device_async(const slampp_hip_edge_set *p_sets, int n_sets, int64_t n_unary_vertex,
	const double *p_unary_factor, const double *p_unary_error, double *p_values_dev, double *p_eta_dev, int b_accumulate);

/* Multi-GPU BA (new functionality, no reference counterpart -- SURVEY.md section 8e): every rank
 * holds a landmark shard (its own points + all cameras); the partial reduced camera systems
 * [S | rhs] are summed over ranks by this callback (RCCL all-reduce over xGMI) between the Schur
 * accumulation and the dense factorization.  p_dev: device pointer, n_count doubles, in place,
 * on stream p_hip_stream.  Return 0 on success.  NULL callback = single GPU.
 * What travels is not the dense n x n buffer but the 6x6 (7x7, 3x3) blocks of S that are nonzero on at
 * least one rank, plus the right-hand side: on the first step with a new callback the ranks agree on that
 * set through the same callback (one or two extra, synchronous calls: concatenated block lists when
 * "shard_rank" / "shard_world" are set, else an indicator over the lower triangle of the camera-block grid),
 * so every rank must register its callback before the same step. */
typedef int (*slampp_hip_allreduce_fn)(void *p_context, double *p_dev, size_t n_count, void *p_hip_stream);
int slampp_hip_set_allreduce(slampp_hip_solver *p_solver, slampp_hip_allreduce_fn p_fn, void *p_context);

/* test hook: copies the elimination plan (permutation, factor structure, update lists, schedule)
 * into caller-provided buffers so that tests can replay it on the CPU (oracle/plan_exec.c).
 * First call with all pointers NULL to get the sizes. */
typedef struct slampp_hip_plan_view {
	int64_t n_bcols, l_blocks, n_pairs, n_row_entries, n_stages, n_tasks, n_task_cols, l_values;
	int32_t *p_perm;        /* [n_bcols] perm[new] = old */
	int32_t *p_dim;         /* [n_bcols] block dimension in the new order */
	int64_t *p_lptr;        /* [n_bcols+1] */
	int32_t *p_lrow;        /* [l_blocks] block row (new order), first of every column = diagonal */
	int64_t *p_loff;        /* [l_blocks] offset of the block in the factor values */
	int64_t *p_asrc;        /* [l_blocks] offset in the packed Lambda values, -1 = fill-in; */
	int32_t *p_atrans;      /* [l_blocks] 1 = source block must be transposed */
	int64_t *p_pptr;        /* [l_blocks+1] update pairs of every block */
	int32_t *p_pa, *p_pb;   /* [n_pairs] factor block ids: L(i,c) and L(j,c) */
	int64_t *p_rptr;        /* [n_bcols+1] row lists (forward solve) */
	int32_t *p_rblk;        /* [n_row_entries] */
	int32_t *p_stage_ptr;   /* [n_stages+1] -> tasks */
	int64_t *p_task_ptr;    /* [n_tasks+1] -> task columns */
	int32_t *p_task_cols;   /* [n_task_cols] */
	int32_t *p_dense_pos;   /* [n_bcols] scalar offset of the column in the dense top, -1 = eliminated block by block */
	int64_t dense_dim;      /* scalar dimension of the dense top (0 = none) */
} slampp_hip_plan_view;
int slampp_hip_get_plan(const slampp_hip_solver *p_solver, slampp_hip_plan_view *p_view);

/* the same analysis without a solver or a GPU (host code only): lets the CPU-only test suite
 * check ordering, symbolic factorization and schedule.  n_leaf_size / n_subtree_size <= 0 = default. */
typedef struct slampp_hip_plan slampp_hip_plan;
int slampp_hip_plan_create(slampp_hip_plan **pp_plan, int64_t n_bcols, const int64_t *p_bcol_cumsum,
	const int64_t *p_bcol_ptr, const int32_t *p_brow_idx, int n_leaf_size, int n_subtree_size,
	int n_dense_top_nb /* < 0 = default */);
int slampp_hip_plan_get(const slampp_hip_plan *p_plan, slampp_hip_plan_view *p_view, slampp_hip_stats *p_stats);
void slampp_hip_plan_destroy(slampp_hip_plan *p_plan);

#ifdef __cplusplus
}
#endif

#endif /* SLAMPP_HIP_H_INCLUDED */
