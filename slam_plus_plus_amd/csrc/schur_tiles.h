// schur_tiles.h -- landmark-major assembly of the reduced camera system (schur_tiles.hip)
#pragma once
#include "solver.h"
#include <cstdint>
#include <vector>
#include <thread>
#include <memory>
#include <exception>

namespace slampp {

enum { SCHUR_TILE_SLOTS = 64, SCHUR_TILE_MAX_POINTS = 64 };

struct TRunJob { // one wave of the run kernel
	int32_t n_first, n_points; // landmarks [n_first, n_first + n_points) of the run list
	int32_t n_k, n_rb, n_cb;   // cameras of the run's landmarks; observation blocks of the rows and of the columns
	int32_t n_pad;
	int64_t n_pbase;           // first partial block
};

// the run tables on their way to the device beside the rest of the analysis (schur_tiles_build starts it, schur_tiles_join
// waits for it): the thread owns what it reads
struct TRunUpload {
	std::thread t;
	raw_vector<TRunJob> jobs;
	raw_vector<int32_t> run_lm, run_k;
	std::exception_ptr p_error;
	~TRunUpload() { if(t.joinable()) t.join(); }
};

struct CSchurTiles {
	CTrashList trash;        // work arrays of schur_tiles_build that nobody reads any more: the caller frees them when it suits (solver.h: TTrash)
	bool b_enabled = false;  // some landmarks go through the tiles
	bool b_hybrid = false;   // ... and some through the contribution lists (the x-lists below)
	int64_t n_tiles = 0, n_slots = 0, n_tile_points = 0, n_list_points = 0, n_tile_pairs = 0, n_all_pairs = 0, n_rb = 0, n_max_slots = 0, n_max_k = 0;
	int64_t n_run_points = 0, n_run_jobs[5][2][2] = {}, n_run_job_first[5][2][2] = {}; // jobs by (16-line tiles per side, row block == column block, some landmarks end before the run's list)
	CDevArray<TRunJob> d_run_jobs;
	CDevArray<int32_t> d_run_lm;        // landmarks of the runs, piece after piece
	CDevArray<int32_t> d_run_k;         // ... their own number of observations (a run's landmarks may end before its list does)
	int64_t n_prefix_points = 0;        // landmarks in runs whose lists are prefixes of the run's, not the run's
	CDevArray<int64_t> d_run_rec;       // ... and where their blocks start in the values
	CDevArray<int32_t> d_tile_ptr;      // [n_tiles + 1] into d_tile_lm
	CDevArray<int32_t> d_tile_lm;       // landmarks of the tiles, in processing order
	CDevArray<int64_t> d_tile_slot_ptr; // [n_tiles + 1] the tile's range of partial blocks
	CDevArray<int64_t> d_pair_ptr;      // [np + 1] into d_lm_slot (empty ranges for landmarks outside the tiles)
	CDevArray<uint8_t> d_lm_slot;       // tile-local slot of every camera pair of every landmark, pair (a <= b) at b (b + 1) / 2 + a
	CDevArray<double> d_P, d_R;         // partial blocks [n_slots][DC * DC], partial right-hand sides [n_slots][DC]
	CDevArray<int64_t> d_rb_ptr;        // [n_rb + 1] partial blocks of every block of S that has some
	CDevArray<int32_t> d_rb_part, d_rb_sb;
	// the landmarks left to the contribution lists
	int64_t n_xblocks = 0, n_xentries = 0, n_xobs = 0;
	CDevArray<int32_t> d_xpoints;       // the landmarks themselves (T.n_list_points of them)
	CDevArray<int64_t> d_xsb_ptr;       // [n_xblocks + 1]
	CDevArray<int32_t> d_xsb_map;       // block of S of every list
	CDevArray<int32_t> d_xent_a;
	CDevArray<int64_t> d_xent_uoff;
	CDevArray<int64_t> d_xcam_ptr;      // [nc + 1] their observations by camera
	CDevArray<int32_t> d_xcam_obs;
	size_t n_Bytes() const
	{
		return d_run_jobs.n_Bytes() + d_run_lm.n_Bytes() + d_run_k.n_Bytes() + d_run_rec.n_Bytes() + d_tile_ptr.n_Bytes() + d_tile_lm.n_Bytes() + d_tile_slot_ptr.n_Bytes() + d_pair_ptr.n_Bytes() +
			d_lm_slot.n_Bytes() + d_P.n_Bytes() + d_R.n_Bytes() + d_rb_ptr.n_Bytes() + d_rb_part.n_Bytes() + d_rb_sb.n_Bytes() +
			d_xsb_ptr.n_Bytes() + d_xsb_map.n_Bytes() + d_xent_a.n_Bytes() + d_xent_uoff.n_Bytes() + d_xcam_ptr.n_Bytes() +
			d_xcam_obs.n_Bytes() + d_xpoints.n_Bytes();
	}
	std::shared_ptr<TRunUpload> p_run_upload; // (the last member: destroyed -- joined -- before the arrays it fills)
};

// host analysis: n_mode -1 = runs and tiles when together they take at least half of the contributions, 0 = never,
// 1 = wherever possible, 2 = tiles only, 3 = runs only (of any length).  sb_row / sb_col: the blocks of S sorted by (col, row), as the contribution lists have them.
// waits for the run tables to be on the device (throws what the upload threw); before the first use of the tiles and before
// the analysis that built them returns
void schur_tiles_join(CSchurTiles &T);
void schur_tiles_build(CSchurTiles &T, int n_mode, int DC, int DP, int64_t nc, int64_t np, const int64_t *ptr, const int32_t *brow,
	const std::vector<int32_t> &sb_row, const std::vector<int32_t> &sb_col, int64_t n_ablocks, hipStream_t stream);

// S -= sum U C^-1 U^T and r -= sum U C^-1 l over the landmarks of the tiles; C^-1 (and W = U C^-1 if p_W) are stored for
// them when b_store is set.  S is the dense buffer (ld) or, with p_dst, the packed values of the inner solver (p_r its rhs).
void schur_tiles_enqueue(const CSchurTiles &T, int DC, int DP, const int64_t *ptr, int64_t nc, int64_t ubase, const double *A,
	const double *eta, int n, double *Cinv, double *p_W, bool b_store, const int32_t *sb_row, const int32_t *sb_col,
	double *S, int ld, const int64_t *p_dst, double *p_r, int *p_flag, hipStream_t stream);

} // namespace slampp
