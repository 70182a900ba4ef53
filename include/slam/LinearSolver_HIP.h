/*
 * LinearSolver_HIP.h -- the binding a SLAM++ maintainer adds: linear solver classes with exactly
 * the member functions SLAM++'s nonlinear solvers expect from their CLinearSolver template
 * argument, forwarding to the C ABI of libslampp_hip.so (include/slampp_hip.h).
 *
 *   CLinearSolver_HIP            stands where CLinearSolver_CholMod / _CSparse / _UberBlock do
 *                                (/root/reference/include/slam/LinearSolver_CholMod.h:78-300,
 *                                 LinearSolver_UberBlock.h:41-461)
 *   CLinearSolver_Schur_HIP<..>  stands where CLinearSolver_Schur<CBaseSolver, CAMatrixBlockSizes,
 *                                CSystem> does (LinearSolver_Schur.h:1423-2392)
 *
 * Contract (LinearSolverTags.h:38-135): typedef _Tag; default ctor; copy ctor / operator = that copy
 * the configuration but no state (the nonlinear solver stores its solver by value,
 * NonlinearSolver_Base.h:344-346,400,438); Free_Memory(); Solve_PosDef(lambda, eta) with eta
 * overwritten by the solution; for the blockwise tag also Clear_SymbolicDecomposition(),
 * SymbolicDecomposition_Blocky(lambda), Solve_PosDef_Blocky(lambda, eta).
 * Errors: false = not positive definite; std::bad_alloc on host/device OOM (CLinearSolver_Schur
 * catches it to fall back to a sparse solver, LinearSolver_Schur.h:1844-1853); std::runtime_error
 * for device errors (precedent LinearSolver_Schur.h:1196-1209).  Never exits.
 *
 * Lambda is read through public const accessors only (BlockMatrix.h:343-485): the block structure
 * is handed over once per structure change, the block values are gathered (OpenMP) into a packed
 * staging array on every call -- CUberBlockMatrix keeps its blocks in pooled pages behind
 * per-block pointers (BlockMatrixBase.h:321,374,449-453), so there is nothing contiguous to pass.
 *
 * Usage, as with any other solver (cf. src/slam_simple_example/Main.cpp:63-67):
 *     typedef CLinearSolver_HIP CLinearSolverType;
 *     CNonlinearSolver_Lambda<CSystemType, CLinearSolverType> solver(system);
 */
#pragma once
#ifndef __LINEAR_SOLVER_HIP_INCLUDED
#define __LINEAR_SOLVER_HIP_INCLUDED

#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>
#if defined(__x86_64__) && defined(__SSE2__)
#include <emmintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif // _OPENMP

#include "slam/LinearSolverTags.h"
#include "slam/BlockMatrix.h"
#include "slam/LinearSolver_Schur.h" // the primary template specialized at the end of this file, and its guided ordering helper
#include "slampp_hip.h"

/**
 *	@brief shared plumbing of the two solver classes: handle life cycle, structure + value hand-over
 */
class CLinearSolver_HIP_Base {
protected:
	slampp_hip_solver *m_p_solver; /**< @brief C ABI handle (owned; never copied) */
	int m_n_device; /**< @brief HIP device ordinal (configuration, copied) */
	std::vector<int> m_devices; /**< @brief HIP device ordinals of a multi-GPU solver (configuration, copied; empty = m_n_device alone) */
	std::vector<std::pair<std::string, int64_t> > m_options; /**< @brief tuning knobs of slampp_hip_set_option (configuration, copied) */
	bool m_b_structure_valid; /**< @brief ordering / symbolic analysis matches the last structure */
	std::vector<int64_t> m_cumsum, m_bcol_ptr; /**< @brief structure handed to the library */
	std::vector<int32_t> m_brow;
	std::vector<size_t> m_order; /**< @brief new block column -> old (empty = identity) */
	struct TGatherEntry { uint32_t n_col, n_blk; int64_t n_dest; int32_t n_rows, n_cols; uint32_t n_row; bool b_transpose; };
	std::vector<TGatherEntry> m_gather; /**< @brief where every block of lambda goes in the packed values */
	std::vector<uint32_t> m_col_block_num, m_col_width; /**< @brief blocks and width of every block column of lambda when it was analyzed (its own order) */
	double *m_p_values, *m_p_rhs; /**< @brief pinned staging owned by the library (slampp_hip_host_staging) */
	size_t m_n_value_num; /**< @brief number of packed values of lambda */
	slampp_hip_times m_t_times;

	/**
	 *	@brief number of threads for the gather / copy loops: copying is memory-bound well before all cores of a big host
	 *		are busy, and a container with a CPU quota below its visible core count (16 of 256 on the boxes this was
	 *		measured on) spends 0.8 s per parallel region with one spinning thread per visible core
	 */
	static int n_Copy_Thread_Num()
	{
#ifdef _OPENMP
		const int n_max = omp_get_max_threads();
		return (n_max < 16)? n_max : 16;
#else // _OPENMP
		return 1;
#endif // _OPENMP
	}

	/**
	 *	@brief copies n doubles to the staging without bringing the destination into the caches (x86-64: movnti)
	 */
	static inline void Copy_Streaming(double *p_dest, const double *p_src, int n)
	{
#if defined(__x86_64__) && defined(__SSE2__)
		for(int i = 0; i < n; ++ i) {
			long long n_bits;
			memcpy(&n_bits, p_src + i, sizeof(n_bits));
			_mm_stream_si64(reinterpret_cast<long long*>(p_dest + i), n_bits);
		}
#else
		for(int i = 0; i < n; ++ i)
			p_dest[i] = p_src[i];
#endif
	}

	static double f_Wall_Ms()
	{
		struct timespec t_now;
		clock_gettime(CLOCK_MONOTONIC, &t_now);
		return t_now.tv_sec * 1e3 + t_now.tv_nsec * 1e-6;
	}

	void Throw_On_Error(int n_result) const // throw(std::bad_alloc, std::runtime_error)
	{
		if(n_result == SLAMPP_HIP_ERR_ALLOC)
			throw std::bad_alloc();
		if(n_result < 0) {
			throw std::runtime_error(std::string("CLinearSolver_HIP: ") +
				(m_p_solver? slampp_hip_last_error(m_p_solver) : "no device / library handle"));
		}
	}

	void Require_Handle() // throw(std::bad_alloc, std::runtime_error)
	{
		if(!m_p_solver) {
			if(m_devices.size() > 1) // BA systems are cut into landmark shards, one per device; pose graphs use the first one
				Throw_On_Error(slampp_hip_create_multi(&m_p_solver, &m_devices[0], int(m_devices.size())));
			else
				Throw_On_Error(slampp_hip_create(&m_p_solver, m_n_device));
			Throw_On_Error(slampp_hip_set_option(m_p_solver, "staging_ahead", 1)); // lambda arrives in host memory: the pinned staging comes up beside the analysis
			for(size_t i = 0, n = m_options.size(); i < n; ++ i)
				Throw_On_Error(slampp_hip_set_option(m_p_solver, m_options[i].first.c_str(), m_options[i].second));
		}
	}

	/**
	 *	@brief hands the block structure of lambda (optionally symmetrically permuted by
	 *		p_order: new -> old, keeping the upper triangle) to the library and analyzes it
	 */
	bool Analyze(const CUberBlockMatrix &r_lambda, int n_mode, size_t n_matrix_cut,
		const std::vector<size_t> *p_order) // throw(std::bad_alloc, std::runtime_error)
	{
		_ASSERTE(r_lambda.b_SymmetricLayout());
		Require_Handle();
		const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0; // development aid: where the cold call's time goes
		const double f_t0 = b_timing? f_Wall_Ms() : 0;
		const size_t n = r_lambda.n_BlockColumn_Num();
		std::vector<size_t> inv_order(n);
		if(p_order) {
			m_order = *p_order;
			for(size_t i = 0; i < n; ++ i)
				inv_order[m_order[i]] = i;
		} else {
			m_order.clear();
			for(size_t i = 0; i < n; ++ i)
				inv_order[i] = i;
		}
		m_cumsum.resize(n + 1);
		m_cumsum[0] = 0;
		for(size_t i = 0; i < n; ++ i)
			m_cumsum[i + 1] = m_cumsum[i] + int64_t(r_lambda.n_BlockColumn_Column_Num(p_order? m_order[i] : i));
		m_col_block_num.resize(n);
		m_col_width.resize(n);
		bool b_prefix = !p_order;
		if(b_prefix) {
			// no permutation, and the block rows ascend inside a column (CUberBlockMatrix keeps them sorted): the upper
			// triangle is a prefix of every column, in place -- counted, offset and filled column-parallel
			m_bcol_ptr.assign(n + 1, 0);
			std::vector<int64_t> value_ptr(n + 1, 0);
			const int n_thread_num = n_Copy_Thread_Num();
			const long n_col_num = long(n);
			int n_unsorted = 0;
			// (threads from 16 384 block columns on only: a team woken for a few hundred columns costs more than the loop, and
			// under a CPU quota -- containers -- a burst of spinning threads can stall the process for a scheduler period:
			// FastL's small parts of R took 90 ms here instead of 0.05, round 5)
			#pragma omp parallel for schedule(static) num_threads(n_thread_num) reduction(+:n_unsorted) if(n_col_num >= 16384)
			for(long c = 0; c < n_col_num; ++ c) {
				const size_t m = r_lambda.n_BlockColumn_Block_Num(c);
				m_col_block_num[c] = uint32_t(m);
				m_col_width[c] = uint32_t(r_lambda.n_BlockColumn_Column_Num(c));
				size_t n_upper = 0, n_last_row = 0;
				int64_t n_height = 0;
				for(size_t j = 0; j < m; ++ j) {
					const size_t r = r_lambda.n_Block_Row(c, j);
					if(j && r <= n_last_row)
						++ n_unsorted;
					n_last_row = r;
					if(r <= size_t(c)) {
						if(n_upper != j)
							++ n_unsorted;
						++ n_upper;
						n_height += int64_t(m_cumsum[r + 1] - m_cumsum[r]); // symmetric layout
					}
				}
				m_bcol_ptr[c + 1] = int64_t(n_upper);
				value_ptr[c + 1] = n_height * int64_t(m_col_width[c]);
			}
			if(n_unsorted)
				b_prefix = false; // not the layout assumed: the general path below
			else {
				for(size_t i = 0; i < n; ++ i) {
					m_bcol_ptr[i + 1] += m_bcol_ptr[i];
					value_ptr[i + 1] += value_ptr[i];
				}
				const size_t n_block_num = size_t(m_bcol_ptr[n]);
				m_brow.resize(n_block_num);
				m_gather.resize(n_block_num);
				#pragma omp parallel for schedule(static) num_threads(n_thread_num) if(n_col_num >= 16384)
				for(long c = 0; c < n_col_num; ++ c) {
					int64_t n_dest = value_ptr[c];
					const int32_t n_width = int32_t(m_col_width[c]);
					for(int64_t k = m_bcol_ptr[c], j = 0; k < m_bcol_ptr[c + 1]; ++ k, ++ j) {
						const size_t r = r_lambda.n_Block_Row(c, size_t(j));
						TGatherEntry t;
						t.n_col = uint32_t(c);
						t.n_blk = uint32_t(j);
						t.n_row = uint32_t(r);
						t.n_rows = int32_t(m_cumsum[r + 1] - m_cumsum[r]);
						t.n_cols = n_width;
						t.b_transpose = false;
						t.n_dest = n_dest;
						n_dest += int64_t(t.n_rows) * n_width;
						m_brow[k] = int32_t(r);
						m_gather[k] = t;
					}
				}
				m_n_value_num = size_t(value_ptr[n]);
			}
		}
		if(!b_prefix) {
			// count the blocks of every destination column (upper triangle of the permuted matrix)
			m_bcol_ptr.assign(n + 1, 0);
			size_t n_block_num = 0;
			for(size_t c = 0; c < n; ++ c) {
				m_col_block_num[c] = uint32_t(r_lambda.n_BlockColumn_Block_Num(c));
				m_col_width[c] = uint32_t(r_lambda.n_BlockColumn_Column_Num(c));
				for(size_t j = 0, m = r_lambda.n_BlockColumn_Block_Num(c); j < m; ++ j) {
					const size_t r = r_lambda.n_Block_Row(c, j);
					if(r > c)
						continue; // only the upper triangle is used, as in the reference's solvers
					++ m_bcol_ptr[std::max(inv_order[r], inv_order[c]) + 1];
					++ n_block_num;
				}
			}
			for(size_t i = 0; i < n; ++ i)
				m_bcol_ptr[i + 1] += m_bcol_ptr[i];
			m_brow.resize(n_block_num);
			m_gather.resize(n_block_num);
			{
				// place, then sort the rows of every destination column
				std::vector<int64_t> fill(m_bcol_ptr.begin(), m_bcol_ptr.end() - 1);
				std::vector<std::pair<int32_t, TGatherEntry> > placed(n_block_num);
				for(size_t c = 0; c < n; ++ c) {
					for(size_t j = 0, m = r_lambda.n_BlockColumn_Block_Num(c); j < m; ++ j) {
						const size_t r = r_lambda.n_Block_Row(c, j);
						if(r > c)
							continue;
						const size_t nr = inv_order[r], nc = inv_order[c];
						TGatherEntry t;
						t.n_col = uint32_t(c);
						t.n_blk = uint32_t(j);
						t.n_row = uint32_t(r);
						t.n_rows = int32_t(r_lambda.n_BlockColumn_Column_Num(r)); // symmetric layout
						t.n_cols = int32_t(r_lambda.n_BlockColumn_Column_Num(c));
						t.b_transpose = nr > nc; // lands below the diagonal: store its transpose above
						t.n_dest = 0;
						const size_t n_dest_col = std::max(nr, nc), n_dest_row = std::min(nr, nc);
						placed[fill[n_dest_col] ++] = std::make_pair(int32_t(n_dest_row), t);
					}
				}
				int64_t n_value_num = 0;
				for(size_t c = 0; c < n; ++ c) {
					std::sort(placed.begin() + m_bcol_ptr[c], placed.begin() + m_bcol_ptr[c + 1],
						[](const std::pair<int32_t, TGatherEntry> &a, const std::pair<int32_t, TGatherEntry> &b) {
							return a.first < b.first; });
					for(int64_t k = m_bcol_ptr[c]; k < m_bcol_ptr[c + 1]; ++ k) {
						m_brow[k] = placed[k].first;
						m_gather[k] = placed[k].second;
						m_gather[k].n_dest = n_value_num;
						n_value_num += int64_t(m_gather[k].n_rows) * m_gather[k].n_cols;
					}
				}
				m_n_value_num = size_t(n_value_num);
			}
		}
		const double f_t1 = b_timing? f_Wall_Ms() : 0;
		int n_result = slampp_hip_set_structure(m_p_solver, int64_t(n), &m_cumsum[0], &m_bcol_ptr[0],
			m_brow.empty()? 0 : &m_brow[0]);
		if(n_result == SLAMPP_HIP_OK)
			n_result = slampp_hip_analyze(m_p_solver, n_mode, int64_t(n_matrix_cut));
		const double f_t2 = b_timing? f_Wall_Ms() : 0;
		if(n_result == SLAMPP_HIP_OK)
			n_result = slampp_hip_host_staging(m_p_solver, &m_p_values, &m_p_rhs);
		if(b_timing) {
			fprintf(stderr, "[header] structure of lambda %.2f ms, set_structure + analyze %.2f ms, pinned staging %.2f ms\n",
				f_t1 - f_t0, f_t2 - f_t1, f_Wall_Ms() - f_t2);
		}
		Throw_On_Error(n_result);
		m_b_structure_valid = true;
		return true;
	}

	/**
	 *	@brief tells whether the cached analysis can still apply to r_lambda: same number, widths and block counts of the
	 *		block columns (one pass over the columns); the block rows themselves are verified by Gather_Values() while
	 *		it copies, so that a changed pattern of the same shape is detected without a second pass over the blocks
	 *	@note The reference's cached solvers compare sizes only (LinearSolver_Schur.h:1627) and rely on the caller
	 *		announcing a change through Clear_SymbolicDecomposition(); its CHOLMOD Factorize_PosDef_Blocky() re-analyzes on
	 *		every call (LinearSolver_CholMod.cpp:396-423), and FastL relies on that (NonlinearSolver_FastL.h:2131, 2388).
	 */
	bool b_Structure_Matches(const CUberBlockMatrix &r_lambda) const
	{
		const size_t n = r_lambda.n_BlockColumn_Num();
		if(!m_b_structure_valid || !m_p_solver || n + 1 != m_cumsum.size() ||
		   size_t(m_cumsum.back()) != r_lambda.n_Column_Num() || m_col_block_num.size() != n)
			return false;
		for(size_t c = 0; c < n; ++ c) {
			if(r_lambda.n_BlockColumn_Block_Num(c) != m_col_block_num[c] || r_lambda.n_BlockColumn_Column_Num(c) != m_col_width[c])
				return false;
		}
		return true;
	}

	/**
	 *	@brief copies the block values of lambda into the packed array the library reads
	 *	@return Returns true on success, false if a block of lambda is not where the cached structure has it
	 *		(the block structure changed: the caller re-analyzes and gathers again).
	 *	@note b_Structure_Matches() must hold (the block counts of the columns bound the indices used here).
	 */
	bool Gather_Values(const CUberBlockMatrix &r_lambda) // throw(std::bad_alloc, std::runtime_error)
	{
		const long n_block_num = long(m_gather.size());
		const int n_thread_num = n_Copy_Thread_Num();
		double *p_values = m_p_values;
		int64_t n_chunk_values = int64_t(1) << 17; // chunk k is on the bus while chunk k + 1 is gathered: 1 MB first (the bus
		// waits for the first chunk: 16 MB were 0.3 ms of gathering before the first byte moved), doubling up to 16 MB
		long n_first = 0;
		int64_t n_sent = 0;
		while(n_first < n_block_num) {
			long n_last = n_first;
			const int64_t n_limit = m_gather[n_first].n_dest + n_chunk_values;
			n_chunk_values = std::min<int64_t>(n_chunk_values * 2, int64_t(2) << 20);
			if(int64_t(m_n_value_num) <= n_limit)
				n_last = n_block_num;
			else {
				long n_lo = n_first + 1, n_hi = n_block_num; // first entry at or beyond the limit (entries are sorted by n_dest)
				while(n_lo < n_hi) {
					const long n_mid = (n_lo + n_hi) / 2;
					if(m_gather[n_mid].n_dest < n_limit)
						n_lo = n_mid + 1;
					else
						n_hi = n_mid;
				}
				n_last = n_lo;
			}
			int n_mismatch = 0;
			const long n_prefetch_distance = 8;
			#pragma omp parallel for schedule(static) reduction(+:n_mismatch) num_threads(n_thread_num) if(n_block_num >= 16384 && n_last - n_first > 512)
			for(long k = n_first; k < n_last; ++ k) {
				const TGatherEntry &t = m_gather[k];
				if(r_lambda.n_Block_Row(t.n_col, t.n_blk) != t.n_row) {
					++ n_mismatch;
					continue;
				}
				CUberBlockMatrix::_TyConstMatrixXdRef block = r_lambda.t_Block_AtColumn(t.n_col, t.n_blk);
				const double *p_src = block.data();
				double *p_dest = p_values + t.n_dest;
				// (round 6: the blocks live in pool pages, a few cache lines each at addresses the hardware prefetcher cannot
				// guess: the block eight entries on is requested while this one is copied; the staging is written with
				// streaming stores -- it is read next by the DMA engine, not by this core)
				if(k + n_prefetch_distance < n_last) {
					const TGatherEntry &t_ahead = m_gather[k + n_prefetch_distance];
					const char *p_ahead = reinterpret_cast<const char*>(r_lambda.t_Block_AtColumn(t_ahead.n_col, t_ahead.n_blk).data());
					for(int n_byte = 0, n_bytes = t_ahead.n_rows * t_ahead.n_cols * int(sizeof(double)); n_byte < n_bytes; n_byte += 64)
						__builtin_prefetch(p_ahead + n_byte, 0, 0);
				}
				if(!t.b_transpose) {
					Copy_Streaming(p_dest, p_src, t.n_rows * t.n_cols);
				} else {
					for(int c = 0; c < t.n_cols; ++ c)
						for(int r = 0; r < t.n_rows; ++ r)
							p_dest[c + r * t.n_cols] = p_src[r + c * t.n_rows]; // dest is n_cols x n_rows
				}
			}
#if defined(__x86_64__) && defined(__SSE2__)
			_mm_sfence(); // (the streaming stores of this thread; the team's threads passed the loop's barrier)
#endif
			if(n_mismatch) {
				Throw_On_Error(slampp_hip_upload_values_async(m_p_solver, 0, 0)); // (forget the chunks sent so far)
				return false;
			}
			const int64_t n_end = (n_last < n_block_num)? m_gather[n_last].n_dest : int64_t(m_n_value_num);
			if(n_last < n_block_num) { // the last chunk goes with the call that consumes the values
				Throw_On_Error(slampp_hip_upload_values_async(m_p_solver, n_sent, n_end - n_sent));
				n_sent = n_end;
			}
			n_first = n_last;
		}
		return true;
	}

	/**
	 *	@brief b_Structure_Matches() and Gather_Values() in one: re-analyzes with analyze() if the structure changed
	 *	@return Returns true if the values of lambda are in the staging, false if analyze() left this object without an
	 *		analysis of its own (the Schur class hands a lambda that lost its landmark part to its sparse solver:
	 *		the caller then takes that route).
	 */
	template <class CAnalyze>
	bool Gather_Or_Reanalyze(const CUberBlockMatrix &r_lambda, CAnalyze analyze) // throw(std::bad_alloc, std::runtime_error)
	{
		if(!b_Structure_Matches(r_lambda) || !Gather_Values(r_lambda)) {
			analyze();
			if(!m_b_structure_valid)
				return false;
			if(!b_Structure_Matches(r_lambda) || !Gather_Values(r_lambda))
				throw std::runtime_error("CLinearSolver_HIP: lambda changed while it was being read");
		}
		return true;
	}

	/**
	 *	@brief solves with the values Gather_Values() has put into the staging; eta goes through the pinned
	 *		right-hand side staging (permuted by blocks if the analysis reordered lambda, cf. BlockMatrix.cpp:9291-9401)
	 */
	bool Solve_Gathered(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta,
		bool b_landmarks_only = false) // throw(std::bad_alloc, std::runtime_error)
	{
		_ASSERTE(size_t(r_eta.rows()) == r_lambda.n_Column_Num());
		const long n = long(m_cumsum.size()) - 1, n_scalar_num = long(r_eta.rows());
		if(size_t(n_scalar_num) != size_t(m_cumsum.back()))
			throw std::runtime_error("CLinearSolver_HIP: the right-hand side does not match lambda");
		const double *p_eta = &r_eta(0);
		const int n_thread_num = n_Copy_Thread_Num();
		if(!m_order.empty()) {
			#pragma omp parallel for schedule(static) num_threads(n_thread_num) if(n > 4096)
			for(long i = 0; i < n; ++ i) {
				const size_t n_src = r_lambda.n_BlockColumn_Base(m_order[i]);
				for(int64_t d = 0, w = m_cumsum[i + 1] - m_cumsum[i]; d < w; ++ d)
					m_p_rhs[size_t(m_cumsum[i] + d)] = p_eta[n_src + d];
			}
		} else {
			#pragma omp parallel for schedule(static) num_threads(n_thread_num) if(n_scalar_num > 65536)
			for(long i = 0; i < n_scalar_num; ++ i)
				m_p_rhs[i] = p_eta[i];
		}
		++ n_Solve_Counter();
		const int n_result = b_landmarks_only?
			slampp_hip_solve_marginal_poses(m_p_solver, m_p_values, m_p_rhs) :
			slampp_hip_factor_solve(m_p_solver, m_p_values, m_p_rhs, &m_t_times);
		if(n_result == SLAMPP_HIP_NOT_POSDEF)
			return false;
		Throw_On_Error(n_result);
		if(m_devices.size() > 1 && n_Shard_Num() > 1)
			++ n_Sharded_Solve_Counter();
		double *p_x = &r_eta(0);
		if(!m_order.empty()) {
			#pragma omp parallel for schedule(static) num_threads(n_thread_num) if(n > 4096)
			for(long i = 0; i < n; ++ i) {
				const size_t n_dst = r_lambda.n_BlockColumn_Base(m_order[i]);
				for(int64_t d = 0, w = m_cumsum[i + 1] - m_cumsum[i]; d < w; ++ d)
					p_x[n_dst + d] = m_p_rhs[size_t(m_cumsum[i] + d)];
			}
		} else {
			#pragma omp parallel for schedule(static) num_threads(n_thread_num) if(n_scalar_num > 65536)
			for(long i = 0; i < n_scalar_num; ++ i)
				p_x[i] = m_p_rhs[i];
		}
		return true;
	}

public:
	/**
	 *	@brief the device list of the environment: SLAMPP_HIP_DEVICES="0,1,2,3,4,5,6,7" makes every default-constructed
	 *		solver a multi-GPU one, so that an unchanged application (slam_plus_plus -us ...) shards its BA systems over the
	 *		node without a line of code; a single ordinal (SLAMPP_HIP_DEVICES="3") selects that device for every
	 *		default-constructed solver; empty if the variable is not set
	 */
	static std::vector<int> Devices_From_Environment()
	{
		std::vector<int> devices;
		const char *p_s_list = getenv("SLAMPP_HIP_DEVICES");
		while(p_s_list && *p_s_list) {
			char *p_s_end;
			const long n_device = strtol(p_s_list, &p_s_end, 10);
			if(p_s_end == p_s_list || n_device < 0 || n_device > 1023 || (*p_s_end && *p_s_end != ',')) {
				// not a list of device ordinals: refuse all of it rather than run on devices the user did not name
				fprintf(stderr, "warning: SLAMPP_HIP_DEVICES=\"%s\" is not a comma-separated list of HIP device ordinals: ignored\n",
					getenv("SLAMPP_HIP_DEVICES"));
				devices.clear();
				return devices;
			}
			devices.push_back(int(n_device));
			p_s_list = (*p_s_end == ',')? p_s_end + 1 : p_s_end;
		}
		return devices;
	}

	/** @brief default constructor; n_device = -1 takes device 0, or the device list of SLAMPP_HIP_DEVICES if that is set */
	inline CLinearSolver_HIP_Base(int n_device = -1)
		:m_p_solver(0), m_n_device((n_device < 0)? 0 : n_device), m_b_structure_valid(false), m_p_values(0), m_p_rhs(0), m_n_value_num(0)
	{
		if(n_device < 0) {
			m_devices = Devices_From_Environment();
			if(!m_devices.empty())
				m_n_device = m_devices[0];
			if(m_devices.size() < 2)
				m_devices.clear(); // one ordinal: a one-device handle on THAT device (until round 5 it was dropped: device 0)
		}
	}

	/** @brief a solver over several devices of this process: BA systems (Schur) are cut into landmark shards, one per device */
	inline CLinearSolver_HIP_Base(const std::vector<int> &r_devices)
		:m_p_solver(0), m_n_device(r_devices.empty()? 0 : r_devices[0]), m_devices(r_devices), m_b_structure_valid(false),
		m_p_values(0), m_p_rhs(0), m_n_value_num(0)
	{}

	/** @brief copy-constructor; copies the configuration, not the state */
	inline CLinearSolver_HIP_Base(const CLinearSolver_HIP_Base &r_other)
		:m_p_solver(0), m_n_device(r_other.m_n_device), m_devices(r_other.m_devices), m_options(r_other.m_options),
		m_b_structure_valid(false), m_p_values(0), m_p_rhs(0), m_n_value_num(0)
	{}

	inline ~CLinearSolver_HIP_Base()
	{
		if(m_p_solver)
			slampp_hip_destroy(m_p_solver);
	}

	/** @brief copy operator; copies the configuration, not the state */
	inline CLinearSolver_HIP_Base &operator =(const CLinearSolver_HIP_Base &r_other)
	{
		if(m_p_solver && (m_n_device != r_other.m_n_device || m_devices != r_other.m_devices))
			Free_Memory(); // the handle lives on the devices it was made for
		m_n_device = r_other.m_n_device;
		m_devices = r_other.m_devices;
		m_options = r_other.m_options;
		return *this;
	}

	/**
	 *	@brief sets a tuning knob of the library (see slampp_hip_set_option() in slampp_hip.h, e.g. "schur_sparse",
	 *		"dense_top_nb"); part of the configuration: survives Free_Memory() and is copied with the solver
	 *	@note This throws std::runtime_error on an unknown name or a value out of range.
	 */
	void Set_Option(const char *p_s_name, int64_t n_value) // throw(std::bad_alloc, std::runtime_error)
	{
		if(m_p_solver)
			Throw_On_Error(slampp_hip_set_option(m_p_solver, p_s_name, n_value));
		size_t i = 0;
		while(i < m_options.size() && m_options[i].first != p_s_name)
			++ i;
		if(i == m_options.size())
			m_options.push_back(std::make_pair(std::string(p_s_name), n_value));
		else
			m_options[i].second = n_value;
		m_b_structure_valid = false; // options take effect at the next analysis
	}

	/** @brief deletes memory for all the auxiliary buffers and matrices, host and device */
	void Free_Memory()
	{
		if(m_p_solver) {
			slampp_hip_destroy(m_p_solver);
			m_p_solver = 0;
		}
		m_b_structure_valid = false;
		{ std::vector<int64_t> e0, e1; m_cumsum.swap(e0); m_bcol_ptr.swap(e1); }
		{ std::vector<int32_t> e; m_brow.swap(e); }
		{ std::vector<TGatherEntry> e; m_gather.swap(e); }
		{ std::vector<uint32_t> e0, e1; m_col_block_num.swap(e0); m_col_width.swap(e1); }
		m_p_values = m_p_rhs = 0; // went with the handle
		m_n_value_num = 0;
	}

	/** @brief number of solves all instances have sent to the library so far (diagnostic: shows that the GPU path ran) */
	static size_t &n_Solve_Counter()
	{
		static size_t n_counter = 0;
		return n_counter;
	}

	/** @brief number of solves all instances have run as landmark shards on more than one member (diagnostic) */
	static size_t &n_Sharded_Solve_Counter()
	{
		static size_t n_counter = 0;
		return n_counter;
	}

	/** @brief HIP device ordinal this solver runs on (part of the configuration; the first one of a device list) */
	inline int n_Device() const
	{
		return m_n_device;
	}

	/** @brief the device list of a multi-GPU solver (empty = n_Device() alone) */
	inline const std::vector<int> &r_Devices() const
	{
		return m_devices;
	}

	/**
	 *	@brief members in use, their landmark ranges and the name of the exchange ("rccl (..)" / "peer") once a BA system
	 *		was analyzed on a device list; 0 members = not sharded (one device, a pose graph, nothing analyzed yet)
	 */
	size_t n_Shard_Num(std::vector<int64_t> *p_point_bounds = 0, std::string *p_s_exchange = 0) const
	{
		int n_member_num = 0;
		const char *p_s_name = "none";
		int64_t p_bounds[17] = {0};
		if(m_p_solver)
			slampp_hip_group_info(m_p_solver, &n_member_num, p_bounds, 16, &p_s_name);
		if(p_point_bounds)
			p_point_bounds->assign(p_bounds, p_bounds + ((n_member_num)? n_member_num + 1 : 0));
		if(p_s_exchange)
			*p_s_exchange = p_s_name;
		return size_t(n_member_num);
	}

	/** @brief clears the symbolic decomposition (the block structure of lambda is about to change) */
	inline void Clear_SymbolicDecomposition()
	{
		m_b_structure_valid = false;
	}

	/** @brief phase timing of the last solve (ms) */
	inline const slampp_hip_times &t_Last_Times() const
	{
		return m_t_times;
	}
};

/**
 *	@brief numeric factorization of a pre-ordered matrix, the factor handed back as a block matrix
 *		(what the reference's L / FastL solvers ask of their linear solver, LinearSolver_CholMod.cpp:362-544);
 *		its own library handle, configured to keep the caller's order and to factor every column block by block
 */
class CLinearSolver_HIP_Factorizer : public CLinearSolver_HIP_Base {
public:
	/**
	 *	@brief where the time of Factorize() goes, summed over the calls of the calling thread (all its instances: the nonlinear
	 *		solvers work on copies of the solver they are given): the analysis of a new structure (ordering is the
	 *		caller's; symbolic factorization, schedule, uploads of the plan), the gather of lambda's blocks, the library
	 *		call (values up, numeric factorization, factor down), and the scatter of the factor into r_factor
	 */
	struct TTimes {
		size_t n_calls, n_analyses;
		double f_analyze_ms, f_gather_ms, f_factorize_ms, f_scatter_ms;
	};
	static TTimes &t_Times() { static thread_local TTimes t = {0, 0, 0, 0, 0, 0}; return t; } /**< @brief per calling thread: solvers used from several threads do not share counters (advisor, round 5) */

protected:
	std::vector<int32_t> m_l_perm, m_l_dim, m_l_row; /**< @brief structure of the factor (slampp_hip_plan_view) */
	std::vector<int64_t> m_l_ptr, m_l_off;
	std::vector<double> m_l_values;

	static double f_Now_Ms()
	{
		timespec t;
		clock_gettime(CLOCK_MONOTONIC, &t);
		return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
	}

public:
	inline CLinearSolver_HIP_Factorizer(int n_device = 0)
		:CLinearSolver_HIP_Base(n_device)
	{
		Set_Option("natural_order", 1); // (big separators of the caller's order are factored on the matrix cores and handed back like the rest)
	}

	/**
	 *	@brief factorizes r_lambda (upper triangle read), writes R (or L = R^T) into r_factor at the given block offset
	 *	@return Returns true on success, false if r_lambda is not positive definite.
	 */
	bool Factorize(CUberBlockMatrix &r_factor, const CUberBlockMatrix &r_lambda,
		size_t n_dest_row_id, size_t n_dest_column_id, bool b_upper_factor) // throw(std::bad_alloc, std::runtime_error)
	{
		const size_t n = r_lambda.n_BlockColumn_Num();
		TTimes &r_times = t_Times();
		++ r_times.n_calls;
		const double f_t0 = f_Now_Ms();
		double f_analysis = 0;
		// the reference's counterpart is stateless (it re-analyzes on every call, LinearSolver_CholMod.cpp:396-423) and
		// FastL hands it different parts of R without announcing a new structure (NonlinearSolver_FastL.h:2131, 2388):
		// the cached analysis is reused only if every block of lambda is where it was (verified while gathering)
		Gather_Or_Reanalyze(r_lambda, [&]() {
			const double f_a0 = f_Now_Ms();
			++ r_times.n_analyses;
			Analyze(r_lambda, SLAMPP_HIP_MODE_SPARSE, 0, 0);
			int64_t n_bcols = 0, n_l_blocks = 0, n_l_values = 0;
			Throw_On_Error(slampp_hip_factor_structure(m_p_solver, &n_bcols, &n_l_blocks, &n_l_values, 0, 0, 0, 0, 0)); // sizes
			m_l_perm.resize(size_t(n_bcols)); m_l_dim.resize(size_t(n_bcols));
			m_l_ptr.resize(size_t(n_bcols) + 1);
			m_l_row.resize(size_t(n_l_blocks)); m_l_off.resize(size_t(n_l_blocks));
			m_l_values.resize(size_t(n_l_values));
			Throw_On_Error(slampp_hip_factor_structure(m_p_solver, &n_bcols, &n_l_blocks, &n_l_values, m_l_perm.empty()? 0 : &m_l_perm[0],
				m_l_dim.empty()? 0 : &m_l_dim[0], &m_l_ptr[0], m_l_row.empty()? 0 : &m_l_row[0], m_l_off.empty()? 0 : &m_l_off[0])); // contents
			for(size_t i = 0, m = m_l_perm.size(); i < m; ++ i) {
				if(size_t(m_l_perm[i]) != i)
					throw std::runtime_error("CLinearSolver_HIP: the factorization did not keep the caller's order");
			}
			f_analysis += f_Now_Ms() - f_a0;
		});
		if(m_l_ptr.size() != n + 1)
			throw std::runtime_error("CLinearSolver_HIP: the factor's structure does not match lambda");
		const double f_t1 = f_Now_Ms();
		const int n_result = slampp_hip_factorize(m_p_solver, m_p_values, m_l_values.empty()? 0 : &m_l_values[0]);
		const double f_t2 = f_Now_Ms();
		r_times.f_analyze_ms += f_analysis;
		r_times.f_gather_ms += f_t1 - f_t0 - f_analysis;
		r_times.f_factorize_ms += f_t2 - f_t1;
		if(n_result == SLAMPP_HIP_NOT_POSDEF)
			return false;
		Throw_On_Error(n_result);
		for(size_t j = 0; j < n; ++ j) {
			const int dj = m_l_dim[j];
			for(int64_t k = m_l_ptr[j]; k < m_l_ptr[j + 1]; ++ k) {
				const size_t i = size_t(m_l_row[size_t(k)]);
				const int di = m_l_dim[i];
				Eigen::Map<const Eigen::MatrixXd> t_L_ij(&m_l_values[size_t(m_l_off[size_t(k)])], di, dj); // L(i, j), i >= j
				if(b_upper_factor) {
					double *p_dest = r_factor.p_GetBlock_Log(n_dest_row_id + j, n_dest_column_id + i, dj, di, true, false);
					if(!p_dest)
						return false; // incorrect structure of r_factor
					Eigen::Map<Eigen::MatrixXd>(p_dest, dj, di) = t_L_ij.transpose(); // R(j, i)
				} else {
					double *p_dest = r_factor.p_GetBlock_Log(n_dest_row_id + i, n_dest_column_id + j, di, dj, true, false);
					if(!p_dest)
						return false;
					Eigen::Map<Eigen::MatrixXd>(p_dest, di, dj) = t_L_ij;
				}
			}
		}
		r_times.f_scatter_ms += f_Now_Ms() - f_t2;
		return true;
	}
};

/**
 *	@brief sparse block Cholesky on the GPU (fill-reducing ordering on the block graph, symbolic
 *		analysis cached across calls while the block structure is unchanged)
 */
class CLinearSolver_HIP : public CLinearSolver_HIP_Base {
protected:
	CLinearSolver_HIP_Factorizer m_factorizer; /**< @brief for Factorize_PosDef_Blocky() (own handle, natural order) */

public:
	typedef CBlockwiseLinearSolverTag _Tag; /**< @brief solver type tag */

	inline CLinearSolver_HIP(int n_device = -1)
		:CLinearSolver_HIP_Base(n_device), m_factorizer(n_Device()) // (the resolved device: the first of SLAMPP_HIP_DEVICES when n_device < 0)
	{}

	/**
	 *	@brief a solver over several devices: pose graphs run on the first (one elimination tree does not shard); a Schur
	 *		solver made from this one -- as the nonlinear solvers make theirs, NonlinearSolver_Base.h:400 -- shards BA systems
	 */
	inline CLinearSolver_HIP(const std::vector<int> &r_devices)
		:CLinearSolver_HIP_Base(r_devices), m_factorizer(n_Device())
	{}

	/**
	 *	@brief factorizes a pre-ordered block matrix, puts result in another block matrix; same contract as
	 *		CLinearSolver_CholMod::Factorize_PosDef_Blocky() (LinearSolver_CholMod.h:196-214)
	 *
	 *	@param[out] r_factor is destination for the factor (must contain structure but not any nonzero blocks)
	 *	@param[in] r_lambda is a pre-ordered block matrix to be factorized
	 *	@param[in] r_workspace is unused (the reference needs it for CUberBlockMatrix::From_Sparse())
	 *	@param[in] n_dest_row_id is id of block row where the factor should be put
	 *	@param[in] n_dest_column_id is id of block column where the factor should be put
	 *	@param[in] b_upper_factor is the factor flag (if set, factor is U (R), if not set, factor is L)
	 *
	 *	@return Returns true on success, false on failure (not pos def or incorrect structure of r_factor).
	 */
	bool Factorize_PosDef_Blocky(CUberBlockMatrix &r_factor, const CUberBlockMatrix &r_lambda,
		std::vector<size_t> &UNUSED(r_workspace), size_t n_dest_row_id = 0,
		size_t n_dest_column_id = 0, bool b_upper_factor = true) // throw(std::bad_alloc, std::runtime_error)
	{
		return m_factorizer.Factorize(r_factor, r_lambda, n_dest_row_id, n_dest_column_id, b_upper_factor);
	}

	/** @brief deletes memory for all the auxiliary buffers and matrices, host and device */
	void Free_Memory()
	{
		CLinearSolver_HIP_Base::Free_Memory();
		m_factorizer.Free_Memory();
	}

	/**
	 *	@brief solves linear system given by positive-definite matrix
	 *	@param[in] r_lambda is positive-definite matrix (symmetric layout, upper triangle stored)
	 *	@param[in,out] r_eta is the right-side vector, and is overwritten with the solution
	 *	@return Returns true on success, false on failure (not positive definite).
	 *	@note This function throws std::bad_alloc and std::runtime_error.
	 */
	bool Solve_PosDef(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta) // throw(std::bad_alloc, std::runtime_error)
	{
		Clear_SymbolicDecomposition();
		return Solve_PosDef_Blocky(r_lambda, r_eta);
	}

	/** @brief calculates ordering and symbolic decomposition of a block matrix, to be reused by Solve_PosDef_Blocky() */
	bool SymbolicDecomposition_Blocky(const CUberBlockMatrix &r_lambda) // throw(std::bad_alloc, std::runtime_error)
	{
		return Analyze(r_lambda, SLAMPP_HIP_MODE_SPARSE, 0, 0);
	}

	/**
	 *	@brief calculates block diagonal of the covariance matrix (the inverse of lambda): what the reference's solvers get
	 *		from CMarginals::Calculate_DenseMarginals_Recurrent_FBS(margs, R, ordering, mpart_Diagonal) after ordering and
	 *		factoring lambda once more for the purpose (NonlinearSolver_Lambda.h:696-760, "todo - reuse what the linear
	 *		solver calculated") -- here lambda is all that is needed
	 *
	 *	@param[out] r_marginals is filled with one diagonal block per block column of lambda, in lambda's order
	 *	@param[in] r_lambda is the system matrix (symmetric layout, upper triangle stored; one block size of 3, 6 or 7, or any
	 *		mix of block sizes up to 8 in a structure without big separators)
	 *
	 *	@return Returns true on success, false if lambda is not positive definite.
	 *	@note This function throws std::bad_alloc and std::runtime_error.
	 */
	bool Marginals(CUberBlockMatrix &r_marginals, const CUberBlockMatrix &r_lambda) // throw(std::bad_alloc, std::runtime_error)
	{
		Gather_Or_Reanalyze(r_lambda, [&]() { SymbolicDecomposition_Blocky(r_lambda); });
		const size_t n = r_lambda.n_BlockColumn_Num();
		size_t n_doubles = 0; // (any mix of block sizes: block i is d_i x d_i, the blocks follow each other)
		for(size_t i = 0; i < n; ++ i)
			n_doubles += r_lambda.n_BlockColumn_Column_Num(i) * r_lambda.n_BlockColumn_Column_Num(i);
		std::vector<double> cov(n_doubles);
		const int n_result = slampp_hip_marginals(m_p_solver, m_p_values, cov.empty()? 0 : &cov[0]);
		if(n_result == SLAMPP_HIP_NOT_POSDEF)
			return false;
		Throw_On_Error(n_result);
		r_marginals.Clear();
		size_t n_at = 0;
		for(size_t i = 0; i < n; ++ i) {
			const size_t d = r_lambda.n_BlockColumn_Column_Num(i);
			double *p_dest = r_marginals.p_GetBlock_Log(i, i, d, d, true, false);
			if(!p_dest)
				throw std::runtime_error("CLinearSolver_HIP: cannot write the marginals");
			std::copy(&cov[n_at], &cov[n_at + d * d], p_dest);
			n_at += d * d;
		}
		return true;
	}

	/** @brief solves, reusing the symbolic decomposition for as long as the block structure stays the same */
	bool Solve_PosDef_Blocky(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta) // throw(std::bad_alloc, std::runtime_error)
	{
		Gather_Or_Reanalyze(r_lambda, [&]() { SymbolicDecomposition_Blocky(r_lambda); });
		return Solve_Gathered(r_lambda, r_eta);
	}
};

/**
 *	@brief Schur-complement solver on the GPU with the public surface of CLinearSolver_Schur
 *
 *	@tparam CBaseSolver is the base linear solver type (accepted for signature parity; the reduced
 *		camera system is factorized densely on the GPU, as CLinearSolver_Schur's GPU build does with
 *		CLinearSolver_DenseGPU, LinearSolver_Schur.h:1427-1435)
 *	@tparam CAMatrixBlockSizes is list of the block sizes (unused: sizes are read from lambda)
 *	@tparam CSystem is the optimized system type (unused)
 */
template <class CBaseSolver = CLinearSolver_HIP, class CAMatrixBlockSizes = void, class CSystem = void>
class CLinearSolver_Schur_HIP : public CLinearSolver_HIP_Base {
public:
	typedef CBlockwiseLinearSolverTag _Tag; /**< @brief solver type tag */

protected:
	size_t m_n_matrix_cut; /**< @brief number of camera (reduced system) block columns */
	CLinearSolver_HIP m_sparse_fallback; /**< @brief used when lambda has no landmark part */

	/**
	 *	@brief the reference's guided ordering (LinearSolver_Schur.cpp:771-838): block columns of the
	 *		widest size first (cameras / poses), all others last (landmarks), both in stable order
	 *	@return Returns the number of camera block columns.
	 */
	static size_t n_Calculate_GuidedOrdering(std::vector<size_t> &r_order, const CUberBlockMatrix &r_lambda)
	{
		const size_t n = r_lambda.n_BlockColumn_Num();
		size_t n_pose_dim = 0;
		for(size_t i = 0; i < n; ++ i)
			n_pose_dim = std::max(n_pose_dim, r_lambda.n_BlockColumn_Column_Num(i));
		r_order.clear();
		r_order.reserve(n);
		for(size_t i = 0; i < n; ++ i) {
			if(r_lambda.n_BlockColumn_Column_Num(i) == n_pose_dim)
				r_order.push_back(i);
		}
		const size_t n_cut = r_order.size();
		for(size_t i = 0; i < n; ++ i) {
			if(r_lambda.n_BlockColumn_Column_Num(i) != n_pose_dim)
				r_order.push_back(i);
		}
		return n_cut;
	}

public:
	inline CLinearSolver_Schur_HIP(int n_device = -1)
		:CLinearSolver_HIP_Base(n_device), m_n_matrix_cut(size_t(-1)), m_sparse_fallback(n_device)
	{}

	/** @brief landmark shards on the listed devices (one process, one host thread per device inside the library) */
	inline CLinearSolver_Schur_HIP(const std::vector<int> &r_devices)
		:CLinearSolver_HIP_Base(r_devices), m_n_matrix_cut(size_t(-1)), m_sparse_fallback(r_devices.empty()? 0 : r_devices[0])
	{}

	/** @brief the reference's constructor signature (LinearSolver_Schur.h:1472): a base solver of another kind is unused */
	inline CLinearSolver_Schur_HIP(const CBaseSolver &UNUSED(r_solver), int n_device = -1)
		:CLinearSolver_HIP_Base(n_device), m_n_matrix_cut(size_t(-1)), m_sparse_fallback(n_device)
	{}

	/** @brief from a CLinearSolver_HIP-family base solver: its configuration (device, options) is taken over */
	struct TConfigTag {}; /**< @brief selects the constructor below */
	inline CLinearSolver_Schur_HIP(const CLinearSolver_HIP_Base &r_config, TConfigTag UNUSED(t_tag))
		:CLinearSolver_HIP_Base(r_config), m_n_matrix_cut(size_t(-1)), m_sparse_fallback(r_config.n_Device())
	{}

	inline CLinearSolver_Schur_HIP(const CLinearSolver_Schur_HIP &r_other)
		:CLinearSolver_HIP_Base(r_other), m_n_matrix_cut(size_t(-1)), m_sparse_fallback(r_other.m_sparse_fallback)
	{}

	inline CLinearSolver_Schur_HIP &operator =(const CLinearSolver_Schur_HIP &r_other)
	{
		CLinearSolver_HIP_Base::operator =(r_other);
		return *this;
	}

	void Free_Memory()
	{
		CLinearSolver_HIP_Base::Free_Memory();
		m_sparse_fallback.Free_Memory();
		m_n_matrix_cut = size_t(-1);
	}

	inline void Clear_SymbolicDecomposition()
	{
		CLinearSolver_HIP_Base::Clear_SymbolicDecomposition();
		m_sparse_fallback.Clear_SymbolicDecomposition();
		m_n_matrix_cut = size_t(-1);
	}

	bool Solve_PosDef(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta) // throw(std::bad_alloc, std::runtime_error)
	{
		SymbolicDecomposition_Blocky(r_lambda);
		return Solve_PosDef_Blocky(r_lambda, r_eta);
	}

	inline bool SymbolicDecomposition_Blocky(const CUberBlockMatrix &r_lambda) // throw(std::bad_alloc, std::runtime_error)
	{
		return SymbolicDecomposition_Blocky(r_lambda, false);
	}

	/** @brief calculates the Schur ordering (always the guided one here) and analyzes the structure */
	bool SymbolicDecomposition_Blocky(const CUberBlockMatrix &r_lambda, bool UNUSED(b_force_guided_ordering)) // throw(std::bad_alloc, std::runtime_error)
	{
		std::vector<size_t> order;
		m_n_matrix_cut = n_Calculate_GuidedOrdering(order, r_lambda);
		if(m_n_matrix_cut == 0 || m_n_matrix_cut == order.size()) {
			m_b_structure_valid = false;
			return m_sparse_fallback.SymbolicDecomposition_Blocky(r_lambda); // no landmark part (cf. LinearSolver_Schur.h:1635-1638)
		}
		bool b_identity = true;
		for(size_t i = 0; i < order.size() && b_identity; ++ i)
			b_identity = order[i] == i;
		return Analyze(r_lambda, SLAMPP_HIP_MODE_SCHUR, m_n_matrix_cut, b_identity? 0 : &order);
	}

	bool Solve_PosDef_Blocky(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta) // throw(std::bad_alloc, std::runtime_error)
	{
		if(m_n_matrix_cut == size_t(-1))
			SymbolicDecomposition_Blocky(r_lambda); // no ordering yet: calculate one (LinearSolver_Schur.h:1627-1628)
		if(m_n_matrix_cut == 0 || m_n_matrix_cut == r_lambda.n_BlockColumn_Num())
			return m_sparse_fallback.Solve_PosDef_Blocky(r_lambda, r_eta); // (checks its own cached structure)
		if(!Gather_Or_Reanalyze(r_lambda, [&]() { SymbolicDecomposition_Blocky(r_lambda); })) // nonconforming ordering: a new one
			return m_sparse_fallback.Solve_PosDef_Blocky(r_lambda, r_eta); // lambda lost its landmark part between two calls
		return Solve_Gathered(r_lambda, r_eta);
	}

	/**
	 *	@brief incremental Schur complement: tells the next Solve_PosDef_Blocky() that, since the previous one, only
	 *		the blocks of the given landmark vertices changed (the camera blocks and eta may have changed freely); the
	 *		reduced camera system is then updated instead of rebuilt -- what the reference's dog-leg solver does from
	 *		Omega = delta lambda (NonlinearSolver_Lambda_DL.h:1025-1086, 2301-).  Needs Set_Option("schur_incremental", 1).
	 *	@param[in] r_vertex_ids is the list of block columns of lambda (landmarks) whose blocks changed; may be empty
	 *	@note This throws std::runtime_error if there is no analyzed structure or a vertex is not a landmark.
	 */
	void Set_Changed_Landmarks(const std::vector<size_t> &r_vertex_ids) // throw(std::bad_alloc, std::runtime_error)
	{
		if(!this->m_p_solver || !this->m_b_structure_valid || m_n_matrix_cut == size_t(-1))
			throw std::runtime_error("CLinearSolver_Schur_HIP: Set_Changed_Landmarks() needs an analyzed structure");
		const size_t n = this->m_cumsum.size() - 1;
		std::vector<int64_t> points;
		points.reserve(r_vertex_ids.size());
		if(this->m_order.empty()) {
			for(size_t i = 0, m = r_vertex_ids.size(); i < m; ++ i)
				points.push_back(int64_t(r_vertex_ids[i]) - int64_t(m_n_matrix_cut));
		} else {
			std::vector<size_t> inv_order(n);
			for(size_t i = 0; i < n; ++ i)
				inv_order[this->m_order[i]] = i;
			for(size_t i = 0, m = r_vertex_ids.size(); i < m; ++ i) {
				if(r_vertex_ids[i] >= n)
					throw std::runtime_error("CLinearSolver_Schur_HIP: Set_Changed_Landmarks(): no such vertex");
				points.push_back(int64_t(inv_order[r_vertex_ids[i]]) - int64_t(m_n_matrix_cut));
			}
		}
		std::sort(points.begin(), points.end());
		points.erase(std::unique(points.begin(), points.end()), points.end());
		if(!points.empty() && (points.front() < 0 || points.back() >= int64_t(n - m_n_matrix_cut)))
			throw std::runtime_error("CLinearSolver_Schur_HIP: Set_Changed_Landmarks(): a vertex is not a landmark");
		this->Throw_On_Error(slampp_hip_schur_set_changed_points(this->m_p_solver, points.empty()? 0 : &points[0],
			int64_t(points.size())));
	}

	/**
	 *	@brief solves for the landmarks only (dl = C^-1 eta_l), the solution for the poses is zeroed out;
	 *		the counterpart of CLinearSolver_Schur::Solve_PosDef_Blocky_MarginalPoses() (LinearSolver_Schur.h:1956-2143)
	 *	@note Unlike the reference, this returns false if a landmark block is not positive definite.
	 */
	bool Solve_PosDef_Blocky_MarginalPoses(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta) // throw(std::bad_alloc, std::runtime_error)
	{
		if(m_n_matrix_cut == size_t(-1))
			SymbolicDecomposition_Blocky(r_lambda, true); // force guided, as the reference does
		if(m_n_matrix_cut != 0 && m_n_matrix_cut != r_lambda.n_BlockColumn_Num())
			Gather_Or_Reanalyze(r_lambda, [&]() { SymbolicDecomposition_Blocky(r_lambda, true); });
		if(m_n_matrix_cut == 0 || m_n_matrix_cut == r_lambda.n_BlockColumn_Num())
			throw std::runtime_error("CLinearSolver_Schur_HIP: no landmarks to marginalize the poses against");
		return Solve_Gathered(r_lambda, r_eta, true);
	}

	/**
	 *	@brief calculates block diagonal of the covariance matrix (the inverse of lambda); the counterpart of
	 *		CSchurComplement_Marginals::Schur_Marginals() (BAMarginals.h:579-806), which the reference's solvers
	 *		call with a Cholesky factor of the Schur complement they compute for the purpose
	 *		(NonlinearSolver_Lambda_DL.h:1590-1640) -- here lambda is all that is needed
	 *
	 *	@param[out] r_cam_cov is filled with camera marginals (one diagonal block per camera,
	 *		in the order the cameras have in lambda)
	 *	@param[in] b_do_cam_marginals is camera marginals flag (if not set, r_cam_cov is left empty)
	 *	@param[out] r_lm_cov is filled with landmark marginals (one diagonal block per landmark,
	 *		in the order the landmarks have in lambda)
	 *	@param[in] r_lambda is the system matrix (symmetric layout, upper triangle stored)
	 *
	 *	@return Returns true on success, false if lambda is not positive definite.
	 *	@note This function throws std::bad_alloc and std::runtime_error.
	 */
	bool Schur_Marginals(CUberBlockMatrix &r_cam_cov, bool b_do_cam_marginals, CUberBlockMatrix &r_lm_cov,
		const CUberBlockMatrix &r_lambda) // throw(std::bad_alloc, std::runtime_error)
	{
		if(m_n_matrix_cut == size_t(-1))
			SymbolicDecomposition_Blocky(r_lambda, true);
		if(m_n_matrix_cut != 0 && m_n_matrix_cut != r_lambda.n_BlockColumn_Num())
			Gather_Or_Reanalyze(r_lambda, [&]() { SymbolicDecomposition_Blocky(r_lambda, true); });
		const size_t n = r_lambda.n_BlockColumn_Num(), n_cut = m_n_matrix_cut;
		if(n_cut == 0 || n_cut == n)
			throw std::runtime_error("CLinearSolver_Schur_HIP: no landmarks, the system has no Schur complement");
		const size_t dc = size_t(m_cumsum[1] - m_cumsum[0]), dp = size_t(m_cumsum[n_cut + 1] - m_cumsum[n_cut]);
		std::vector<double> cams((b_do_cam_marginals)? n_cut * dc * dc : 0), lms((n - n_cut) * dp * dp);
		const int n_result = slampp_hip_schur_marginals(m_p_solver, m_p_values,
			(b_do_cam_marginals)? &cams[0] : 0, &lms[0]);
		if(n_result == SLAMPP_HIP_NOT_POSDEF)
			return false;
		Throw_On_Error(n_result);
		r_cam_cov.Clear();
		r_lm_cov.Clear();
		for(size_t i = 0; i < n_cut && b_do_cam_marginals; ++ i) {
			double *p_dest = r_cam_cov.p_GetBlock_Log(i, i, dc, dc, true, false);
			if(!p_dest)
				throw std::runtime_error("CLinearSolver_Schur_HIP: cannot write the camera marginals");
			std::copy(&cams[i * dc * dc], &cams[(i + 1) * dc * dc], p_dest);
		}
		for(size_t i = 0; i < n - n_cut; ++ i) {
			double *p_dest = r_lm_cov.p_GetBlock_Log(i, i, dp, dp, true, false);
			if(!p_dest)
				throw std::runtime_error("CLinearSolver_Schur_HIP: cannot write the landmark marginals");
			std::copy(&lms[i * dp * dp], &lms[(i + 1) * dp * dp], p_dest);
		}
		return true;
	}
};

/**
 *	@brief CLinearSolver_Schur with CLinearSolver_HIP as its base solver *is* the GPU Schur solver
 *
 *	The reference's nonlinear solvers hard-wire the type of their Schur solver to
 *	CLinearSolver_Schur<CLinearSolver, CAMatrixBlockSizes, CSystem> (NonlinearSolver_Base.h:345-346) and construct it
 *	from the linear solver they were given (:400, :438).  This partial specialization therefore makes
 *	CNonlinearSolver_Lambda_LM<CSystem, CLinearSolver_HIP> (BA with -us: NonlinearSolver_Lambda_LM.h:1543-1552),
 *	CNonlinearSolver_Lambda (:526-527, :624) and the dog-leg solver run their Schur path on the GPU with not a line
 *	of the reference changed: the reduction, the factorization of the reduced camera system and the
 *	back-substitution all happen in libslampp_hip.so, as with CLinearSolver_Schur_HIP used directly.
 */
template <class CAMatrixBlockSizes, class CSystem>
class CLinearSolver_Schur<CLinearSolver_HIP, CAMatrixBlockSizes, CSystem> :
	public CLinearSolver_Schur_HIP<CLinearSolver_HIP, CAMatrixBlockSizes, CSystem> {
public:
	typedef CLinearSolver_Schur_HIP<CLinearSolver_HIP, CAMatrixBlockSizes, CSystem> _TyBase; /**< @brief the implementation */
	typedef CBlockwiseLinearSolverTag _Tag; /**< @brief solver type tag */

	// the public types of the primary template (LinearSolver_Schur.h:1426-1449); the nonlinear solvers read _TyGOH
	// (NonlinearSolver_Lambda_LM.h:397, NonlinearSolver_Lambda_DL.h:266)
	typedef CLinearSolver_HIP _TyBaseSolver; /**< @brief name of the base linear solver */
	typedef typename CSystem::_TyVertexTypelist _TyVertexTypelist; /**< @brief list of vertex types */
	typedef typename CSystem::_TyEdgeTypelist _TyEdgeTypelist; /**< @brief list of edge types */
	typedef typename CSystem::_TyHessianMatrixBlockList _TyLambdaMatrixBlockSizes; /**< @brief possible block matrices, found in lambda and L */
	typedef CLinearSolver_HIP::_Tag _TyBaseSolverTag; /**< @brief linear solver tag */
	typedef schur_detail::CGuidedOrdering_Helper<_TyVertexTypelist, _TyEdgeTypelist> _TyGOH; /**< @brief guided ordering helper */

	/** @brief the reference's constructor (LinearSolver_Schur.h:1472-1474); device and options of r_solver are taken over */
	inline CLinearSolver_Schur(const CLinearSolver_HIP &r_solver)
		:_TyBase(r_solver, typename _TyBase::TConfigTag())
	{}

	inline CLinearSolver_Schur(const CLinearSolver_Schur &r_other)
		:_TyBase(r_other)
	{}

	inline CLinearSolver_Schur &operator =(const CLinearSolver_Schur &r_other)
	{
		_TyBase::operator =(r_other);
		return *this;
	}
};

#endif // !__LINEAR_SOLVER_HIP_INCLUDED
