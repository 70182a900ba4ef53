// schur.hip -- the BA path: Schur complement of the landmark block, dense reduced camera system,
// back-substitution.  Restates CLinearSolver_Schur::Solve_PosDef_Blocky, steps 2-13
// (/root/reference/include/slam/LinearSolver_Schur.h:1699-1886) for gfx950:
//
//   Lambda = | A U |    S  = A - U C^-1 U^T           (reference: SliceTo / InverseOf_BlockDiag /
//            | V C |    r  = x - U C^-1 l              two MultiplyToWith_FBS / AddTo_FBS / PreMultiply_Add)
//                       dx = S^-1 r                   (reference: Eigen LLT on the densified S)
//                       dl = C^-1 (l - U^T dx)        (reference: PostMultiply_Add + PreMultiply_Add)
//
// Landmarks are independent units (C is block diagonal), so every kernel but the dense factor is a
// stream over the observations: HBM-bound integer/fp64 work, no MFMA.
//   schur_point_inverse : C_p^-1, one thread per point                        (72 B in, 72 B out)
//   schur_obs_W         : W_o = U_o C_p^-1, one thread per observation         (144 B in, 144 B out)
//   schur_gather_S      : one wave per nonzero block of S sums its contributions U_b W_a^T from a
//                         precomputed list -- no atomics, bit-reproducible      (288 B per contribution)
//   schur_rhs           : r_c = x_c - sum W_o l_p, one wave per camera
//   dense_cholesky / dense_backsolve (dense_chol.hip)
//   schur_obs_t, schur_point_backsubst : dl
// Multi-GPU (landmark shards): the all-reduce callback sums [S | r] over the ranks right before the
// dense factorization; only the primary shard adds A and x.
#include "solver.h"
#include "dense_chol.h"
#include "sparse_inverse.h"
#include "schur_tiles.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

namespace slampp {

#include "schur_device.inl"

static double schur_wall_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct CSchurState {
	int DC, DP;
	int64_t nc, np, n_obs, n_ablocks, n_sblocks, n_entries;
	int N, Npad;
	CDevArray<int64_t> d_ptr;       // [n+1] block column pointers of Lambda
	CDevArray<int32_t> d_brow;      // [n_blocks]
	CDevArray<int32_t> d_obs_pt;    // [n_obs]
	CDevArray<int64_t> d_sb_ptr;    // [n_sblocks+1]
	CDevArray<int32_t> d_sb_row, d_sb_col;
	CDevArray<int32_t> d_ent_a;     // [n_entries] observation whose W is used
	CDevArray<int64_t> d_ent_uoff;  // [n_entries] offset of the U block of the other observation in the values
	CDevArray<int64_t> d_cam_ptr;   // [nc+1]
	CDevArray<int32_t> d_cam_obs;   // [n_obs] observations of every camera, ascending
	CDevArray<double> d_S, d_W, d_Cinv, d_t, d_invdiag, d_z, d_x;
	// multi-GPU: the all-reduce moves only the blocks of S that are nonzero on some rank
	std::vector<int32_t> h_blk_row, h_blk_col; // this rank's nonzero blocks of S (lower triangle; camera indices)
	slampp_hip_allreduce_fn p_union_fn;        // the callback the union below was agreed through
	void *p_union_context;
	bool b_union_dense;                        // too many cameras for the indicator exchange: reduce the whole buffer
	int64_t n_union;
	CDevArray<int32_t> d_un_row, d_un_col;
	CDevArray<double> d_pack;                  // [n_union DC^2 + N]
	std::vector<int32_t> h_un_row, h_un_col;   // the agreed list, sorted by (column, row)
	// sparse reduced system: S handed to the sparse block path as its own little Lambda
	bool b_reduced_decided, b_reduced_sparse;
	slampp_hip_solver *p_inner;
	int64_t n_in_blocks;
	CDevArray<int64_t> d_sb_dst, d_a_dst;      // where the blocks of S this rank computes / the camera blocks of Lambda sit
	                                           // in the packed upper block-CSC values of the inner solver
	CDevArray<double> d_in_buf;                // [values (n_in_blocks DC^2) | right-hand side (N)]: also what the ranks exchange
	// marginal covariances (own buffers: the factor the last solve left behind stays usable)
	CDevArray<double> d_m_S, d_m_Z, d_m_invdiag, d_m_zero;
	// ... through the sparse inverse subset when the reduced system is factored by the sparse block path
	CSparseInverse *p_sinv;
	bool b_sinv_tried;
	CDevArray<double> d_m_Zs;                  // laid out like the inner solver's factor
	CDevArray<int64_t> d_cam_zoff, d_pair_ptr, d_pair_tab;
	// incremental update of the reduced system (option "schur_incremental"): what the previous solve assembled stays, and
	// a solve that names the landmarks whose blocks changed exchanges their contributions only
	bool b_prev_valid = false;                 // the buffers below describe the values of the last solve
	CDevArray<double> d_A_prev;                // the camera-camera blocks of Lambda of the last solve
	CDevArray<double> d_S_unf;                 // dense reduced system: S as assembled (d_S is factored in place)
	CDevArray<int64_t> d_changed;              // landmarks named for the next solve
	int64_t n_changed = -1;                    // -1: none named (full rebuild)
	CSchurTiles tiles;                         // landmark-major assembly of S (schur_tiles.hip)
	CSchurState() :p_union_fn(0), p_union_context(0), b_union_dense(false), n_union(0), b_reduced_decided(false),
		b_reduced_sparse(false), p_inner(0), n_in_blocks(0), p_sinv(0), b_sinv_tried(false) {}
	~CSchurState();
};

CSchurState::~CSchurState()
{
	if(p_sinv)
		sparse_inverse_destroy(p_sinv);
	if(p_inner) {
		p_inner->stream = 0; // borrowed from the owning solver
		delete p_inner;
	}
}

void schur_destroy(CSchurState *p) { delete p; }

size_t schur_device_bytes(const CSchurState *p)
{
	return p->d_ptr.n_Bytes() + p->d_brow.n_Bytes() + p->d_obs_pt.n_Bytes() + p->d_sb_ptr.n_Bytes() +
		p->d_sb_row.n_Bytes() + p->d_sb_col.n_Bytes() + p->d_ent_a.n_Bytes() + p->d_ent_uoff.n_Bytes() +
		p->d_cam_ptr.n_Bytes() + p->d_cam_obs.n_Bytes() + p->d_S.n_Bytes() + p->d_W.n_Bytes() +
		p->d_un_row.n_Bytes() + p->d_un_col.n_Bytes() + p->d_pack.n_Bytes() + p->d_sb_dst.n_Bytes() + p->d_a_dst.n_Bytes() +
		p->d_in_buf.n_Bytes() + (p->p_inner? p->p_inner->n_Device_Bytes() : 0) + p->d_m_S.n_Bytes() + p->d_m_Z.n_Bytes() +
		p->d_m_invdiag.n_Bytes() + p->d_m_zero.n_Bytes() + p->d_m_Zs.n_Bytes() + p->d_cam_zoff.n_Bytes() +
		p->d_pair_ptr.n_Bytes() + p->d_pair_tab.n_Bytes() + sparse_inverse_bytes(p->p_sinv) +
		p->d_Cinv.n_Bytes() + p->d_t.n_Bytes() + p->d_invdiag.n_Bytes() + p->d_z.n_Bytes() + p->d_x.n_Bytes() +
		p->d_A_prev.n_Bytes() + p->d_S_unf.n_Bytes() + p->d_changed.n_Bytes() + p->tiles.n_Bytes();
}

void schur_invalidate_previous(CSchurState *p)
{
	if(p) {
		p->b_prev_valid = false;
		p->n_changed = -1;
	}
}

// names the landmarks whose blocks differ from the previous solve's (host list, strictly increasing)
void schur_set_changed_points(slampp_hip_solver &s, const int64_t *p_points, int64_t n_points)
{
	CSchurState &S = *s.p_schur;
	for(int64_t i = 0; i < n_points; ++ i) {
		if(p_points[i] < 0 || p_points[i] >= S.np || (i && p_points[i] <= p_points[i - 1]))
			throw std::invalid_argument("schur_set_changed_points: landmark indices must be strictly increasing and in range");
	}
	S.d_changed.Alloc(size_t(std::max<int64_t>(n_points, 1)));
	if(n_points)
		SLAMPP_HIP_CHECK(hipMemcpy(S.d_changed.p(), p_points, size_t(n_points) * sizeof(int64_t), hipMemcpyHostToDevice));
	S.n_changed = n_points;
}

// stats of the inner solver that factors the reduced camera system by the sparse block path; false while there is none
// (dense reduced system, or no solve has decided yet)
bool schur_reduced_stats(const CSchurState *p, slampp_hip_stats &st)
{
	return p && p->b_reduced_decided && p->b_reduced_sparse && p->p_inner && slampp_hip_get_stats(p->p_inner, &st) == SLAMPP_HIP_OK;
}

void schur_fill_stats(const CSchurState *p, slampp_hip_stats &st)
{
	st.n_cams = p->nc;
	st.n_points = p->np;
	st.n_observations = p->n_obs;
	st.schur_dim = p->N;
	st.n_update_pairs = p->n_entries;
	st.l_blocks = p->n_sblocks;
	const double n = double(p->N);
	st.factor_flops = n * n * n / 3.0 + n * (n - 1) / 2.0 + n; // dense Cholesky (slam_schur_orderings/Main.cpp:682)
	st.solve_flops = 2.0 * n * n;
}

// ---------------------------------------------------------------------------------------------
// host analysis
// ---------------------------------------------------------------------------------------------

CSchurState *schur_analyze(slampp_hip_solver &s)
{
	const int64_t n = int64_t(s.cumsum.size()) - 1, nc = s.n_matrix_cut, np = n - nc;
	const int64_t *cs = s.cumsum.data(), *ptr = s.bcol_ptr.data();
	const int32_t *brow = s.brow.data();
	const int64_t DC = cs[1] - cs[0], DP = cs[nc + 1] - cs[nc];
	{
		// the shape of the system, column by column -- on a few threads from 65 536 block columns on (round 6: two passes over
		// C5's two million columns were 5 - 7 ms on one); what is reported is what the serial passes reported: a block size out of
		// line first, then the first landmark that is wrong and how, then the first camera
		const int n_check_threads = (n >= 65536)? 4 : 1;
		struct TFirst { int64_t n_size, n_landmark, n_camera; int n_landmark_error; };
		std::vector<TFirst> first(size_t(n_check_threads), TFirst{-1, -1, -1, 0});
		auto Check = [&](int t) {
			TFirst &r_first = first[size_t(t)];
			for(int64_t c = n * t / n_check_threads, c1 = n * (t + 1) / n_check_threads; c < c1; ++ c) {
				if(r_first.n_size < 0 && cs[c + 1] - cs[c] != (c < nc? DC : DP))
					r_first.n_size = c;
				const bool b_no_diagonal = ptr[c + 1] == ptr[c] || brow[ptr[c + 1] - 1] != c;
				if(c < nc) {
					if(r_first.n_camera < 0 && b_no_diagonal)
						r_first.n_camera = c;
				} else if(r_first.n_landmark < 0) {
					if(b_no_diagonal) {
						r_first.n_landmark = c;
						r_first.n_landmark_error = 1;
					} else if(ptr[c + 1] - ptr[c] >= 2 && brow[ptr[c + 1] - 2] >= nc) {
						r_first.n_landmark = c;
						r_first.n_landmark_error = 2;
					}
				}
			}
		};
		std::vector<std::thread> threads;
		for(int t = 1; t < n_check_threads; ++ t)
			threads.emplace_back(Check, t);
		Check(0);
		for(size_t t = 0; t < threads.size(); ++ t)
			threads[t].join();
		for(int t = 0; t < n_check_threads; ++ t) {
			if(first[size_t(t)].n_size >= 0)
				throw std::domain_error("Schur path: cameras and landmarks must each have one block size");
		}
		if(!((DC == 6 && DP == 3) || (DC == 7 && DP == 3) || (DC == 3 && DP == 2)))
			throw std::domain_error("Schur path: supported (camera, landmark) block sizes are (6,3), (7,3), (3,2)");
		for(int t = 0; t < n_check_threads; ++ t) { // (the threads' ranges ascend: the first one with a complaint has the first landmark)
			if(first[size_t(t)].n_landmark_error == 1)
				throw std::invalid_argument("Schur path: a landmark has no diagonal block");
			if(first[size_t(t)].n_landmark_error == 2)
				throw std::domain_error("Schur path: landmark-landmark blocks present, C is not block diagonal");
		}
		for(int t = 0; t < n_check_threads; ++ t) {
			if(first[size_t(t)].n_camera >= 0)
				throw std::invalid_argument("Schur path: a camera has no diagonal block");
		}
	}
	if(nc * DC + 64 > INT32_MAX / 2)
		throw std::domain_error("Schur path: reduced system too large");

	CSchurState *p = new CSchurState();
	try {
		CSchurState &S = *p;
		S.DC = int(DC); S.DP = int(DP);
		S.nc = nc; S.np = np;
		S.n_ablocks = ptr[nc];
		S.n_obs = ptr[n] - ptr[nc] - np;
		S.N = int(nc * DC);
		S.Npad = dense_padded_dim(S.N);
		if(S.n_obs > INT32_MAX)
			throw std::domain_error("Schur path: too many observations");

		const bool b_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
		double t_phase = schur_wall_ms();
#define SCHUR_SETUP_PHASE(name) do { if(b_timing) { const double t_ = schur_wall_ms(); \
		fprintf(stderr, "[schur setup] %-20s %8.2f ms\n", name, t_ - t_phase); t_phase = t_; } } while(0)
		// (round 4: the passes over the landmarks run on a few threads, a range of landmarks each -- C5's two million landmarks
		// and eight million observations were 34 + 19 ms here on one core; the camera-major list is a counting sort with one
		// counter array per thread, so every observation still lands where the serial pass put it)
		const int n_setup_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), np / 65536)));
		auto For_Landmark_Ranges = [np, n_setup_workers](const std::function<void(int, int64_t, int64_t)> &r_work) {
			std::vector<std::thread> threads;
			for(int t = 0; t < n_setup_workers; ++ t) {
				const int64_t n_first = np * t / n_setup_workers, n_last = np * (t + 1) / n_setup_workers;
				if(t + 1 < n_setup_workers)
					threads.emplace_back(r_work, t, n_first, n_last);
				else
					r_work(t, n_first, n_last);
			}
			for(size_t t = 0; t < threads.size(); ++ t)
				threads[t].join();
		};
		raw_vector<int32_t> obs_pt(S.n_obs), obs_cam(S.n_obs); // (written in full by the pass below)
		std::vector<int64_t> cam_ptr(nc + 1, 0);
		std::vector<std::vector<int64_t> > cam_count(n_setup_workers, std::vector<int64_t>(size_t(nc), 0));
		For_Landmark_Ranges([&](int t, int64_t n_first, int64_t n_last) {
			std::vector<int64_t> &r_count = cam_count[t];
			for(int64_t pt = n_first; pt < n_last; ++ pt) {
				const int64_t c = nc + pt, o0 = ptr[c] - ptr[nc] - pt;
				for(int64_t k = ptr[c]; k < ptr[c + 1] - 1; ++ k) {
					const int64_t o = o0 + (k - ptr[c]);
					obs_pt[o] = int32_t(pt);
					obs_cam[o] = brow[k];
					++ r_count[brow[k]];
				}
			}
		});
		for(int64_t c = 0; c < nc; ++ c) { // per camera: where each thread's observations start (threads in landmark order)
			int64_t n_sum = cam_ptr[c];
			for(int t = 0; t < n_setup_workers; ++ t) {
				const int64_t n_here = cam_count[t][c];
				cam_count[t][c] = n_sum;
				n_sum += n_here;
			}
			cam_ptr[c + 1] = n_sum;
		}
		raw_vector<int32_t> cam_obs(S.n_obs);
		For_Landmark_Ranges([&](int t, int64_t n_first, int64_t n_last) {
			std::vector<int64_t> &r_fill = cam_count[t];
			const int64_t o_first = ptr[nc + n_first] - ptr[nc] - n_first, o_last = ptr[nc + n_last] - ptr[nc] - n_last;
			for(int64_t o = o_first; o < o_last; ++ o)
				cam_obs[r_fill[obs_cam[o]] ++] = int32_t(o);
		});
		SCHUR_SETUP_PHASE("observation lists");
		// what is ready goes to the device from here on, beside the rest of the analysis (round 6: 130 MB out of pageable
		// vectors at C5, 7 ms at the end of the analysis; the vectors are not written again, and joined before the final sync)
		std::exception_ptr p_early_upload_error;
		struct TJoinEarly { std::thread t; ~TJoinEarly() { if(t.joinable()) t.join(); } } t_early_upload;
		s.Join_Bringup(); // (a handle fresh from slampp_hip_create: its streams came up beside the checks and the observation lists -- solver.h)
		{
			const int n_device = s.n_device;
			hipStream_t st_early = s.stream;
			auto Early_Uploads = [&, n_device, st_early]() {
				try {
					SLAMPP_HIP_CHECK(hipSetDevice(n_device));
					S.d_ptr.Upload(s.bcol_ptr, st_early);
					S.d_brow.Upload(s.brow, st_early);
					S.d_obs_pt.Upload(obs_pt, st_early);
					S.d_cam_ptr.Upload(cam_ptr, st_early);
					S.d_cam_obs.Upload(cam_obs, st_early);
				} catch(...) {
					p_early_upload_error = std::current_exception();
				}
			};
			if(S.n_obs >= (int64_t(1) << 20))
				t_early_upload.t = std::thread(Early_Uploads);
			else
				Early_Uploads(); // (a small system: a thread's start-up is what it would save)
		}
		// contributions to S grouped by block (row = camera of b, col = camera of a, a <= b within a point)
		const int64_t ubase = S.n_ablocks * DC * DC;
		std::vector<int64_t> sb_ptr;
		std::vector<int32_t> sb_row, sb_col;
		raw_vector<int32_t> ent_a;    // (the contribution lists -- 5 M entries at the uniform-visibility C4 -- and the counters they are
		raw_vector<int64_t> ent_uoff; // placed with: mappings of the library's own on huge pages, solver.h)
		bool b_tiles_built = false;
		{
			int64_t n_entries = 0;
			for(int64_t pt = 0; pt < np; ++ pt) {
				const int64_t k = ptr[nc + pt + 1] - ptr[nc + pt] - 1;
				n_entries += k * (k + 1) / 2;
			}
			S.n_entries = n_entries;
			if(nc * nc <= (int64_t(1) << 26)) { // the dense key space
				// Which blocks of S exist: a bit per camera pair, a bitmap per thread over its range of landmarks, OR-ed at the end
				// (round 5; counting the contributions of every block was 19 ms on one core at C5 -- with atomic adds from eight,
				// every landmark of the band structure on the same few thousand counters, 88 ms --, and the counts are only needed
				// where the landmarks cannot be taken one by one: below)
				const int64_t n_words = (nc * nc + 63) / 64;
				std::vector<std::vector<uint64_t> > pair_bits(n_setup_workers, std::vector<uint64_t>(size_t(n_words), 0));
				For_Landmark_Ranges([&](int t, int64_t n_first, int64_t n_last) {
					std::vector<uint64_t> &r_bits = pair_bits[t];
					for(int64_t pt = n_first; pt < n_last; ++ pt) {
						const int64_t o0 = ptr[nc + pt] - ptr[nc] - pt, o1 = ptr[nc + pt + 1] - ptr[nc] - (pt + 1);
						for(int64_t a = o0; a < o1; ++ a) {
							const int64_t n_base = int64_t(obs_cam[a]) * nc;
							for(int64_t b = a; b < o1; ++ b) {
								const int64_t key = n_base + obs_cam[b];
								r_bits[size_t(key >> 6)] |= uint64_t(1) << (key & 63);
							}
						}
					}
				});
				for(int64_t w = 0; w < n_words; ++ w) {
					uint64_t n_word = 0;
					for(int t = 0; t < n_setup_workers; ++ t)
						n_word |= pair_bits[t][size_t(w)];
					for(; n_word; n_word &= n_word - 1) {
						const int64_t key = w * 64 + __builtin_ctzll(n_word);
						sb_col.push_back(int32_t(key / nc));
						sb_row.push_back(int32_t(key % nc));
					}
				}
				pair_bits.clear();
				SCHUR_SETUP_PHASE("blocks of S");
				// the blocks of S are known: can the landmarks be taken one by one (schur_tiles.hip)?  Then the per-block
				// contribution lists -- 12 bytes and a scattered write per contribution -- are not needed at all
				schur_tiles_build(S.tiles, s.n_schur_tiles, int(DC), int(DP), nc, np, ptr, brow, sb_row, sb_col, S.n_ablocks, s.stream);
				b_tiles_built = true;
				SCHUR_SETUP_PHASE("runs and tiles");
				if(!S.tiles.b_enabled) {
					// the lists of every block after all: a counting sort of the contributions by block.  Round 6: on the threads of
					// the passes above, a counter array per thread over its range of landmarks (one thread counted and placed the
					// five million contributions of the uniform-visibility C4 in 37 ms); every contribution lands where the serial
					// pass put it: inside a block in landmark order
					const int64_t n_keys = nc * nc;
					const int n_list_workers = int(std::max<int64_t>(1, std::min<int64_t>(n_setup_workers, (int64_t(1) << 26) / std::max<int64_t>(n_keys, 1)))); // (at most 512 MB of counters)
					auto For_List_Ranges = [&](int64_t n, const std::function<void(int, int64_t, int64_t)> &r_work) {
						std::vector<std::thread> threads;
						for(int t = 0; t < n_list_workers; ++ t) {
							const int64_t n_first = n * t / n_list_workers, n_last = n * (t + 1) / n_list_workers;
							if(t + 1 < n_list_workers)
								threads.emplace_back(r_work, t, n_first, n_last);
							else
								r_work(t, n_first, n_last);
						}
						for(size_t t = 0; t < threads.size(); ++ t)
							threads[t].join();
					};
					std::vector<raw_vector<int64_t> > cnt(n_list_workers);
					For_List_Ranges(np, [&](int t, int64_t n_first, int64_t n_last) {
						raw_vector<int64_t> &r_cnt = cnt[t];
						r_cnt.assign(size_t(n_keys), 0);
						for(int64_t pt = n_first; pt < n_last; ++ pt) {
							const int64_t o0 = ptr[nc + pt] - ptr[nc] - pt, o1 = ptr[nc + pt + 1] - ptr[nc] - (pt + 1);
							for(int64_t a = o0; a < o1; ++ a)
								for(int64_t b = a; b < o1; ++ b)
									++ r_cnt[int64_t(obs_cam[a]) * nc + obs_cam[b]];
						}
					});
					// where every thread's contributions to every block start: block by block, inside a block thread by thread --
					// the ranges of blocks first by themselves, then shifted by what the ranges before them hold
					std::vector<int64_t> range_total(n_list_workers, 0);
					std::vector<std::vector<int64_t> > range_starts(n_list_workers); // start of every nonzero block of the range
					For_List_Ranges(n_keys, [&](int r, int64_t n_first, int64_t n_last) {
						int64_t n_sum = 0;
						for(int64_t key = n_first; key < n_last; ++ key) {
							const int64_t n_before = n_sum;
							for(int t = 0; t < n_list_workers; ++ t) {
								const int64_t n_here = cnt[t][size_t(key)];
								cnt[t][size_t(key)] = n_sum;
								n_sum += n_here;
							}
							if(n_sum != n_before)
								range_starts[r].push_back(n_before);
						}
						range_total[r] = n_sum;
					});
					std::vector<int64_t> range_base(n_list_workers + 1, 0);
					for(int r = 0; r < n_list_workers; ++ r)
						range_base[r + 1] = range_base[r] + range_total[r];
					For_List_Ranges(n_keys, [&](int r, int64_t n_first, int64_t n_last) {
						if(!range_base[r])
							return;
						for(int t = 0; t < n_list_workers; ++ t) {
							int64_t *p_cnt = cnt[t].data();
							for(int64_t key = n_first; key < n_last; ++ key)
								p_cnt[key] += range_base[r];
						}
					});
					for(int r = 0; r < n_list_workers; ++ r) {
						for(size_t i = 0; i < range_starts[r].size(); ++ i)
							sb_ptr.push_back(range_starts[r][i] + range_base[r]);
					}
					sb_ptr.push_back(n_entries);
					if(sb_ptr.size() != sb_row.size() + 1 || range_base[n_list_workers] != n_entries)
						throw std::logic_error("reduced camera system: the block list and the contribution counts disagree");
					ent_a.resize(n_entries);
					ent_uoff.resize(n_entries);
					For_List_Ranges(np, [&](int t, int64_t n_first, int64_t n_last) {
						int64_t *p_fill = cnt[t].data();
						for(int64_t pt = n_first; pt < n_last; ++ pt) {
							const int64_t o0 = ptr[nc + pt] - ptr[nc] - pt, o1 = ptr[nc + pt + 1] - ptr[nc] - (pt + 1);
							for(int64_t a = o0; a < o1; ++ a)
								for(int64_t b = a; b < o1; ++ b) {
									const int64_t d = p_fill[int64_t(obs_cam[a]) * nc + obs_cam[b]] ++;
									ent_a[d] = int32_t(a);
									ent_uoff[d] = ubase + b * DC * DP + pt * DP * DP;
								}
						}
					});
				}
			} else { // comparison sort on (key, a, b)
				ent_a.resize(n_entries);
				ent_uoff.resize(n_entries);
				struct TE { int64_t key; int32_t a, b; };
				std::vector<TE> ents(n_entries);
				int64_t e = 0;
				for(int64_t pt = 0; pt < np; ++ pt) {
					const int64_t o0 = ptr[nc + pt] - ptr[nc] - pt, o1 = ptr[nc + pt + 1] - ptr[nc] - (pt + 1);
					for(int64_t a = o0; a < o1; ++ a)
						for(int64_t b = a; b < o1; ++ b) {
							ents[e].key = int64_t(obs_cam[a]) * nc + obs_cam[b];
							ents[e].a = int32_t(a);
							ents[e].b = int32_t(b);
							++ e;
						}
				}
				std::sort(ents.begin(), ents.end(), [](const TE &x, const TE &y) {
					return x.key < y.key || (x.key == y.key && x.a < y.a); });
				for(e = 0; e < n_entries; ++ e) {
					if(!e || ents[e].key != ents[e - 1].key) {
						sb_ptr.push_back(e);
						sb_col.push_back(int32_t(ents[e].key / nc));
						sb_row.push_back(int32_t(ents[e].key % nc));
					}
					ent_a[e] = ents[e].a;
					ent_uoff[e] = ubase + int64_t(ents[e].b) * DC * DP + int64_t(obs_pt[ents[e].b]) * DP * DP;
				}
				sb_ptr.push_back(n_entries);
			}
		}
		S.n_sblocks = int64_t(sb_row.size());
		S.h_blk_row = sb_row;
		S.h_blk_col = sb_col;
		for(int64_t c = 0; c < nc; ++ c) { // the camera-camera blocks of Lambda land in S too (transposed: lower triangle)
			for(int64_t k = ptr[c]; k < ptr[c + 1]; ++ k) {
				S.h_blk_row.push_back(int32_t(c));
				S.h_blk_col.push_back(brow[k]);
			}
		}

		SCHUR_SETUP_PHASE("contribution lists");
		hipStream_t st = s.stream;
		if(!b_tiles_built)
			schur_tiles_build(S.tiles, s.n_schur_tiles, int(DC), int(DP), nc, np, ptr, brow, sb_row, sb_col, S.n_ablocks, st);
		SCHUR_SETUP_PHASE("runs and tiles");
		S.d_sb_ptr.Upload(sb_ptr, st);
		S.d_sb_row.Upload(sb_row, st);
		S.d_sb_col.Upload(sb_col, st);
		S.d_ent_a.Upload(ent_a, st);
		S.d_ent_uoff.Upload(ent_uoff, st);
		if(t_early_upload.t.joinable())
			t_early_upload.t.join();
		if(p_early_upload_error)
			std::rethrow_exception(p_early_upload_error);
		S.d_W.Alloc(size_t(S.n_obs) * (DC * DP));
		S.d_Cinv.Alloc(size_t(np) * DP * DP);
		S.d_t.Alloc(size_t(S.n_obs) * DP);
		s.d_flag.Alloc(1);
		SLAMPP_HIP_CHECK(hipMemsetAsync(s.d_flag.p(), 0, sizeof(int), s.stream)); // sync() before the first factorization reads it
		schur_tiles_join(S.tiles); // (the run tables, uploaded beside everything since the runs were found)
		SLAMPP_HIP_CHECK(hipStreamSynchronize(st));
		SCHUR_SETUP_PHASE("uploads");
		if(S.n_obs >= (int64_t(1) << 20) || !S.tiles.trash.empty()) {
			// the observation lists are on the device: their memory (C5: 100 MB) goes back to the system on a thread behind the
			// analysis' return, not at the end of this scope (solver.h: TTrash, t_discard)
			s.Join_Discard();
			for(size_t i = 0; i < S.tiles.trash.size(); ++ i)
				s.analysis_trash.emplace_back(std::move(S.tiles.trash[i])); // (the tile analysis' hashes, sort items and orders)
			S.tiles.trash.clear();
			Discard_Later(s.analysis_trash, obs_pt); Discard_Later(s.analysis_trash, obs_cam); Discard_Later(s.analysis_trash, cam_obs);
			Discard_Later(s.analysis_trash, ent_a); Discard_Later(s.analysis_trash, ent_uoff);
			slampp_hip_solver *p_solver = &s;
			try {
				s.t_discard = std::thread([p_solver]() { p_solver->analysis_trash.clear(); host_pool_release(); });
			} catch(std::system_error&) {
				s.analysis_trash.clear();
				host_pool_release();
			}
		}
#undef SCHUR_SETUP_PHASE
	} catch(...) {
		delete p;
		throw;
	}
	return p;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

// S(lower) := A^T blocks, r := eta_x   (primary shard only; S was zeroed before).  The reduced system lives either in
// the dense buffer S (leading dimension ld, right-hand side in its last row) or, when p_dst is given, in the packed
// upper block-CSC values of the inner sparse solver (block k of Lambda goes to p_dst[k] as it is) and the vector p_r
template <int DC>
__global__ void schur_scatter_A_kernel(const int64_t *ptr, const int32_t *brow, int64_t nc,
	const double *__restrict__ A, const double *__restrict__ eta, double *S, int ld, int n,
	const int64_t *__restrict__ p_dst, double *p_r)
{
	const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const int64_t n_elems = ptr[nc] * DC * DC;
	if(gid < n_elems) {
		const int64_t k = gid / (DC * DC);
		const int e = int(gid - k * (DC * DC)), rr = e % DC, q = e / DC; // element (rr, q) of block k = Lambda(r, c), r <= c
		// column of block k: binary search in ptr[0..nc]
		int64_t lo = 0, hi = nc;
		while(hi - lo > 1) {
			const int64_t mid = (lo + hi) >> 1;
			if(ptr[mid] <= k) lo = mid; else hi = mid;
		}
		const int64_t c = lo, r = brow[k];
		if(p_dst)
			S[p_dst[k] + e] = A[gid]; // the upper block (r, c), as stored
		else
			S[size_t(c * DC + q) + size_t(r * DC + rr) * ld] = A[gid]; // lower block (c, r) = block^T
	}
	if(gid < n) {
		if(p_dst)
			p_r[gid] = eta[gid];
		else
			S[size_t(ld - 1) + size_t(gid) * ld] = eta[gid];
	}
}

template <int DC, int DP>
__global__ void schur_point_inverse_kernel(const int64_t *ptr, int64_t nc, int64_t np, int64_t ubase,
	const double *__restrict__ A, double *Cinv, int *p_flag, const int32_t *__restrict__ p_list = 0)
{
	const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(i >= np)
		return;
	const int64_t pt = p_list? int64_t(p_list[i]) : i; // (np: the length of the list, if there is one)
	const int64_t o1 = ptr[nc + pt + 1] - ptr[nc] - (pt + 1);
	const double *C = A + ubase + o1 * (DC * DP) + pt * (DP * DP);
	double c[DP * DP], inv[DP * DP];
	#pragma unroll
	for(int i = 0; i < DP * DP; ++ i)
		c[i] = C[i];
	if(!spd_inverse<DP>(c, inv))
		atomicOr(p_flag, 1);
	#pragma unroll
	for(int i = 0; i < DP * DP; ++ i)
		Cinv[pt * (DP * DP) + i] = inv[i];
}

template <int DC, int DP>
__global__ void schur_obs_W_kernel(int64_t n_obs, int64_t ubase, const int32_t *obs_pt,
	const double *__restrict__ A, const double *__restrict__ Cinv, double *W, const int32_t *__restrict__ p_list = 0)
{
	const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(i >= n_obs)
		return;
	const int64_t o = p_list? int64_t(p_list[i]) : i;
	const int64_t pt = obs_pt[o];
	const double *U = A + ubase + o * (DC * DP) + pt * (DP * DP);
	double u[DC * DP], ci[DP * DP];
	#pragma unroll
	for(int i = 0; i < DC * DP; ++ i)
		u[i] = U[i];
	#pragma unroll
	for(int i = 0; i < DP * DP; ++ i)
		ci[i] = Cinv[pt * (DP * DP) + i];
	#pragma unroll
	for(int q = 0; q < DP; ++ q)
		#pragma unroll
		for(int r = 0; r < DC; ++ r) {
			double t = 0;
			#pragma unroll
			for(int k = 0; k < DP; ++ k)
				t += u[r + k * DC] * ci[k + q * DP];
			W[o * (DC * DP) + r + q * DC] = t;
		}
}


// ---- incremental update of the reduced camera system ----
// Stands where the reference's dog-leg solver updates its Schur complement from Omega = Lambda_new - Lambda_old instead
// of recomputing it (include/slam/NonlinearSolver_Lambda_DL.h:1025-1086, 2301-): after a relinearization that moved a
// few landmarks, S_new = S_old + (A_new - A_old) - sum over the changed landmarks of (U C^-1 U^T)_new - (U C^-1 U^T)_old.
// The old contribution of a landmark is rebuilt from what the previous solve left on the device, W = U C^-1 and C^-1
// (U C^-1 U^T = W C W^T), so no copy of the old values is kept; the reduced right-hand side is recomputed in full (eta
// changes everywhere).  Several landmarks may touch one block of S at the same time: fp64 atomic adds, so the sum order --
// unlike everywhere else in this library -- is not fixed (the option is off by default).
template <int DC>
__global__ void schur_delta_A_kernel(const int64_t *ptr, const int32_t *brow, int64_t nc,
	const double *__restrict__ A, double *A_prev, const double *__restrict__ eta, double *S, int ld, int n,
	const int64_t *__restrict__ p_dst, double *p_r)
{
	const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const int64_t n_elems = ptr[nc] * DC * DC;
	if(gid < n_elems) {
		const int64_t k = gid / (DC * DC);
		const int e = int(gid - k * (DC * DC)), rr = e % DC, q = e / DC;
		int64_t lo = 0, hi = nc;
		while(hi - lo > 1) {
			const int64_t mid = (lo + hi) >> 1;
			if(ptr[mid] <= k) lo = mid; else hi = mid;
		}
		const int64_t c = lo, r = brow[k];
		const double f_new = A[gid], f_delta = f_new - A_prev[gid];
		A_prev[gid] = f_new;
		if(p_dst)
			S[p_dst[k] + e] += f_delta;
		else
			S[size_t(c * DC + q) + size_t(r * DC + rr) * ld] += f_delta;
	}
	if(gid < n) {
		if(p_dst)
			p_r[gid] = eta[gid];
		else
			S[size_t(ld - 1) + size_t(gid) * ld] = eta[gid];
	}
}

// The contributions of the changed landmarks exchanged, one wave per landmark.  Its observations are taken T = 32 at a
// time: for a tile of "row" observations b the products W_b C (old) and U_b C_new^-1 (the new W_b) go to LDS, then for
// every tile of "column" observations a up to it their U and (old) W blocks; every lane finds the block of S of one
// camera pair (a binary search: a dozen dependent trips to memory), then a lane per (pair, element) adds
// (W_b C W_a^T - U_b C_new^-1 U_a^T)(r, q) to S.  A landmark seen by at most 32 cameras -- nearly all of them -- is one
// tile against itself.  (Until round 3 a landmark with more than 24 observations was walked by the wave's first lane
// alone, k (k + 1) / 2 searches one after the other: 30 such landmarks among 5 000 changed ones of the Venice-like C4
// made the update 6.3 ms, against 1.0 ms for assembling everything again.)
template <int DC, int DP>
__global__ void __launch_bounds__(64)
schur_changed_points_kernel(const int64_t *__restrict__ changed, int64_t n_changed, const int64_t *ptr,
	const int32_t *brow, int64_t nc, int64_t ubase, const double *__restrict__ A, double *Cinv, double *W,
	int64_t n_sblocks, const int32_t *__restrict__ sb_row, const int32_t *__restrict__ sb_col, double *S, int ld,
	const int64_t *__restrict__ p_dst, int *p_flag)
{
	enum { T = 32, BLK = DC * DP, PMAX = T * T };
	__shared__ double s_U[2][T * BLK], s_W[2][T * BLK]; // [0]: the column tile, [1]: the row tile
	__shared__ double s_wc[T * BLK], s_wn[T * BLK];     // of the row tile
	__shared__ int64_t s_dst[PMAX];
	__shared__ int32_t s_cam[2][T];
	const int lane = threadIdx.x;
	const int64_t pt = changed[blockIdx.x];
	const int64_t k0 = ptr[nc + pt], o0 = k0 - ptr[nc] - pt, o1 = ptr[nc + pt + 1] - ptr[nc] - (pt + 1);
	const int k = int(o1 - o0);
	double c_old[DP * DP], ci_old[DP * DP], c_new[DP * DP], ci_new[DP * DP];
	#pragma unroll
	for(int t = 0; t < DP * DP; ++ t) {
		ci_old[t] = Cinv[pt * (DP * DP) + t];
		c_new[t] = A[ubase + o1 * BLK + pt * (DP * DP) + t];
	}
	spd_inverse<DP>(ci_old, c_old); // C of the previous solve
	if(!spd_inverse<DP>(c_new, ci_new) && lane == 0)
		atomicOr(p_flag, 1);
	const int n_tiles = (k + T - 1) / T;
	for(int tb = 0; tb < n_tiles; ++ tb) {
		const int b0 = tb * T, kb = (k - b0 < T)? k - b0 : T;
		__syncthreads(); // (the tile before is done with)
		for(int i = lane; i < kb * BLK; i += 64) { // (a landmark's observations follow one another)
			s_U[1][i] = A[ubase + (o0 + b0) * BLK + pt * (DP * DP) + i];
			s_W[1][i] = W[(o0 + b0) * BLK + i];
		}
		if(lane < kb)
			s_cam[1][lane] = brow[k0 + b0 + lane];
		__syncthreads();
		for(int i = lane; i < kb * BLK; i += 64) { // W_b C_old (old), U_b C^-1_new (the new W_b)
			const int ob = i / BLK, e = i - ob * BLK, r = e % DC, q = e / DC;
			double t_old = 0, t_new = 0;
			#pragma unroll
			for(int t = 0; t < DP; ++ t) {
				t_old += s_W[1][ob * BLK + r + t * DC] * c_old[t + q * DP];
				t_new += s_U[1][ob * BLK + r + t * DC] * ci_new[t + q * DP];
			}
			s_wc[i] = t_old;
			s_wn[i] = t_new;
		}
		for(int ta = 0; ta <= tb; ++ ta) {
			const int a0 = ta * T, ka = (ta == tb)? kb : T;
			const int n_side = (ta == tb)? 1 : 0; // the column tile: the row tile itself on the diagonal
			__syncthreads(); // (s_wc / s_wn are there; the column tile and the destinations before are done with)
			if(ta != tb) {
				for(int i = lane; i < ka * BLK; i += 64) {
					s_U[0][i] = A[ubase + (o0 + a0) * BLK + pt * (DP * DP) + i];
					s_W[0][i] = W[(o0 + a0) * BLK + i];
				}
				if(lane < ka)
					s_cam[0][lane] = brow[k0 + a0 + lane];
				__syncthreads();
			}
			const int n_pairs = (ta == tb)? kb * (kb + 1) / 2 : kb * ka;
			for(int p = lane; p < n_pairs; p += 64) { // pair (b >= a): the block (row cam_b, column cam_a) of S
				int a, b;
				if(ta == tb) { // at b (b + 1) / 2 + a
					b = int((sqrtf(8.0f * float(p) + 1.0f) - 1.0f) * 0.5f);
					while((b + 1) * (b + 2) / 2 <= p) ++ b;
					while(b * (b + 1) / 2 > p) -- b;
					a = p - b * (b + 1) / 2;
				} else {
					b = p / ka;
					a = p - b * ka;
				}
				const int64_t cam_a = s_cam[n_side][a], cam_b = s_cam[1][b];
				if(p_dst) { // binary search on the blocks' sorted keys column * nc + row
					const int64_t key = cam_a * nc + cam_b;
					int64_t lo = 0, hi = n_sblocks - 1;
					while(lo < hi) {
						const int64_t mid = (lo + hi) >> 1;
						if(int64_t(sb_col[mid]) * nc + sb_row[mid] < key) lo = mid + 1; else hi = mid;
					}
					s_dst[p] = p_dst[lo];
				} else
					s_dst[p] = cam_b * DC + cam_a * DC * int64_t(ld);
			}
			__syncthreads();
			for(int i = lane; i < n_pairs * DC * DC; i += 64) {
				const int p = i / (DC * DC), e = i - p * (DC * DC), q = e % DC, r = e / DC; // (q fastest: neighbouring lanes, neighbouring addresses)
				int a, b;
				if(ta == tb) {
					b = int((sqrtf(8.0f * float(p) + 1.0f) - 1.0f) * 0.5f);
					while((b + 1) * (b + 2) / 2 <= p) ++ b;
					while(b * (b + 1) / 2 > p) -- b;
					a = p - b * (b + 1) / 2;
				} else {
					b = p / ka;
					a = p - b * ka;
				}
				double f_old = 0, f_new = 0;
				#pragma unroll
				for(int t = 0; t < DP; ++ t) {
					f_old += s_wc[b * BLK + r + t * DC] * s_W[n_side][a * BLK + q + t * DC]; // (W_b C W_a^T)(r, q)
					f_new += s_wn[b * BLK + r + t * DC] * s_U[n_side][a * BLK + q + t * DC]; // (U_b C^-1 U_a^T)(r, q)
				}
				// S = A - sum U C^-1 U^T: the old term comes back, the new one goes
				const int64_t n_at = p_dst? s_dst[p] + q + r * DC : s_dst[p] + r + q * int64_t(ld);
				atomicAdd(S + n_at, f_old - f_new);
			}
		}
	}
	// every pair of the landmark has read the old W: now it is replaced
	__syncthreads();
	if(n_tiles == 1) {
		for(int i = lane; i < k * BLK; i += 64)
			W[o0 * BLK + i] = s_wn[i];
	} else {
		for(int i = lane; i < k * BLK; i += 64) {
			const int ob = i / BLK, e = i - ob * BLK, r = e % DC, q = e / DC;
			double t_new = 0;
			#pragma unroll
			for(int t = 0; t < DP; ++ t)
				t_new += A[ubase + (o0 + ob) * BLK + pt * (DP * DP) + r + t * DC] * ci_new[t + q * DP];
			W[o0 * BLK + i] = t_new;
		}
	}
	if(lane < DP * DP)
		Cinv[pt * (DP * DP) + lane] = ci_new[lane];
}

// one workgroup of W waves per nonzero block of S: S(row, col) -= sum_e U_b W_a^T; contribution e is
// summed by wave e mod W, the partial sums are combined in a fixed order (bit-reproducible).
// The kernel is bound by the address path, not by bytes, so it spends as few vector-memory
// instructions as it can: the entry indices are fetched 64 at a time (one coalesced load, then
// v_readlane), the two operand blocks of an entry with ONE load (lanes 0..17 U_b, lanes 18..35 W_a,
// both contiguous) into LDS, eight entries per batch; a batch is then one small GEMM on the matrix cores
// (v_mfma_f64_16x16x4: S block += [U_1 .. U_8] [W_1 .. W_8]^T, K = 24).  All eight requests of a batch are issued
// before the first LDS store: written as "load, store" per entry the compiler serialized them, eight memory
// latencies per batch (0.71 -> 0.50 ms at C4 once hoisted).  Measured and not kept: batches of 16 (slower), an
// XCD-contiguous block order, camera-major copies of U and W (gather 4 % faster, the scattered writes of the copy
// cost 0.24 ms).  144-byte operand blocks straddle 64-byte sectors: about 1.9x the algorithmic bytes are fetched.
template <int DC, int DP, int W>
__global__ void __launch_bounds__(64 * W)
schur_gather_S_kernel(int64_t n_sblocks, const int64_t *sb_ptr, const int32_t *sb_row, const int32_t *sb_col,
	const int32_t *ent_a, const int64_t *ent_uoff, const double *__restrict__ A, const double *__restrict__ W_,
	double *S, int ld, const int64_t *__restrict__ p_dst, const int32_t *__restrict__ sb_map)
{
	enum { BLK = DC * DP, BATCH = 8 };
	__shared__ double s_ops[W][BATCH][2 * BLK];
	__shared__ double s_part[W][64];
	const int64_t n_list = blockIdx.x; // list n_list serves block sb of S (the same, unless only some landmarks go through lists)
	const int64_t sb = sb_map? int64_t(sb_map[n_list]) : n_list;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const bool b_act = lane < DC * DC;
	const int r = b_act? lane % DC : 0, q = b_act? lane / DC : 0;
	// Every lane loads (lanes past the two operand blocks repeat the last element: same cache line, no branch around
	// the load): lanes 0 .. BLK-1 the block U_b at A + u, the others W_a at W_ + a BLK.  All of this is there to keep the
	// wave-instruction count of an entry down -- with eight waves per SIMD
	// resident the kernel is bound by instruction issue around its loads, not by the loads.
	const int lane_c = (lane < 2 * BLK)? lane : 2 * BLK - 1;
	const bool b_u = lane_c < BLK;
	const int64_t lane_off = b_u? lane_c : lane_c - BLK;
	typedef double v4f64 __attribute__((ext_vector_type(4)));
	v4f64 macc = {0, 0, 0, 0};
	// the MFMA fragment a lane reads for step q4 of a batch: entry, and offset of its element of U inside s_ops[wave]
	int frag_ent[BATCH * DP / 4], frag_off[BATCH * DP / 4];
	#pragma unroll
	for(int q4 = 0; q4 < BATCH * DP / 4; ++ q4) {
		const int kg = 4 * q4 + (lane >> 4), ent = kg / DP, t = kg % DP;
		const int m = lane & 15, mm = (m < DC)? m : 0;
		frag_ent[q4] = ent;
		frag_off[q4] = ent * (2 * BLK) + mm + t * DC;
	}
	const double *s_my = &s_ops[wave][0][0];
	const int64_t e0 = sb_ptr[n_list], e1 = sb_ptr[n_list + 1];
	// this wave's entries: e0 + wave, e0 + wave + W, ...; processed in chunks of 64
	for(int64_t base = e0 + wave; base < e1; base += int64_t(64) * W) {
		const int64_t my = base + int64_t(lane) * W;
		const int32_t my_a = (my < e1)? ent_a[my] : 0;
		const int64_t my_u = (my < e1)? ent_uoff[my] : 0;
		const int n_chunk = int(min(int64_t(64), (e1 - base + W - 1) / W));
		for(int i = 0; i < n_chunk; i += BATCH) {
			double v[BATCH]; // all requests of the batch first, then the LDS stores: one memory latency per batch, not eight
			#pragma unroll
			for(int j = 0; j < BATCH; ++ j) {
				const int idx = min(i + j, n_chunk - 1); // the tail re-reads the last entry, its product is skipped below
				const int32_t a = __builtin_amdgcn_readlane(my_a, idx);
				const int64_t u = (int64_t(__builtin_amdgcn_readlane(int(my_u >> 32), idx)) << 32) |
					uint32_t(__builtin_amdgcn_readlane(int(my_u), idx));
				const double *p_src = b_u? A + u : W_ + int64_t(a) * BLK; // two arrays, one load: the lane picks its base
				v[j] = p_src[lane_off];
			}
			if(lane < 2 * BLK) {
				#pragma unroll
				for(int j = 0; j < BATCH; ++ j)
					s_ops[wave][j][lane] = v[j];
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			// the batch is one small GEMM: S block (DC x DC) += [U_1 .. U_BATCH] [W_1 .. W_BATCH]^T with K = BATCH * DP,
			// four k at a time on the matrix cores (rows / columns >= DC of the 16 x 16 tile are don't-cares)
			const int n_left = n_chunk - i; // entries of the batch that exist
			#pragma unroll
			for(int q4 = 0; q4 < BATCH * DP / 4; ++ q4) {
				const double a = (frag_ent[q4] < n_left)? s_my[frag_off[q4]] : 0.0; // the tail repeats its last entry: not summed
				const double b = s_my[frag_off[q4] + BLK];
				macc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, macc, 0, 0, 0);
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	}
	// lane l holds the tile elements ((l >> 4) + 4 reg, l & 15): hand the DC x DC corner to the lanes that write it
	#pragma unroll
	for(int reg = 0; reg < 4; ++ reg) {
		const int row = (lane >> 4) + 4 * reg, col = lane & 15;
		if(row < DC && col < DC)
			s_part[wave][row + col * DC] = macc[reg];
	}
	__syncthreads();
	if(wave != 0)
		return;
	double acc = 0;
	if(b_act) {
		#pragma unroll
		for(int ww = 0; ww < W; ++ ww)
			acc += s_part[ww][lane];
	}
	if(b_act) { // dense: the lower block (row, col); packed: its transpose, the upper block (col, row) of the inner solver
		const size_t idx = p_dst? size_t(p_dst[sb]) + q + r * DC :
			size_t(int64_t(sb_row[sb]) * DC + r) + size_t(int64_t(sb_col[sb]) * DC + q) * ld;
		S[idx] -= acc;
	}
}

// one wave per camera: r_c -= sum over its observations of W_o l_p
template <int DC, int DP>
__global__ void __launch_bounds__(64)
schur_rhs_kernel(const int64_t *cam_ptr, const int32_t *cam_obs, const int32_t *obs_pt, int n,
	const double *__restrict__ W, const double *__restrict__ eta, double *S, int ld, double *p_r)
{
	const int64_t c = blockIdx.x;
	const int lane = threadIdx.x;
	double acc[DC];
	#pragma unroll
	for(int i = 0; i < DC; ++ i)
		acc[i] = 0;
	const int64_t e1 = cam_ptr[c + 1];
	for(int64_t e = cam_ptr[c] + lane; e < e1; e += 64) {
		const int64_t o = cam_obs[e];
		const double *Wo = W + o * (DC * DP), *l = eta + n + int64_t(obs_pt[o]) * DP;
		#pragma unroll
		for(int t = 0; t < DP; ++ t) {
			const double lt = l[t];
			#pragma unroll
			for(int i = 0; i < DC; ++ i)
				acc[i] += Wo[i + t * DC] * lt;
		}
	}
	#pragma unroll
	for(int i = 0; i < DC; ++ i) {
		#pragma unroll
		for(int m = 32; m >= 1; m >>= 1)
			acc[i] += __shfl_xor(acc[i], m);
	}
	if(lane == 0) {
		#pragma unroll
		for(int i = 0; i < DC; ++ i) {
			if(p_r)
				p_r[c * DC + i] -= acc[i];
			else
				S[size_t(ld - 1) + size_t(c * DC + i) * ld] -= acc[i];
		}
	}
}

// T_o = U_o^T dx_cam(o), one lane per observation.  The U blocks of 64 consecutive observations are one contiguous piece of
// the values (with the C blocks of the landmarks they belong to in between): the wave fetches that piece with coalesced
// loads into LDS and every lane reads its block from there -- a lane fetching its own 144 bytes touched 64 lines per
// load instruction (94 us at C4 for 288 MB; the back-substitution was a third of a C5-size step)
template <int DC, int DP>
__global__ void __launch_bounds__(64)
schur_obs_t_kernel(int64_t n_obs, int64_t ubase, int64_t nc, const int64_t *ptr, const int32_t *brow,
	const int32_t *obs_pt, const double *__restrict__ A, const double *__restrict__ dx, double *T)
{
	enum { BLK = DC * DP, PIECE = 64 * BLK + 24 * DP * DP, // room for 24 C blocks in between (64 observations of landmarks with three or more each)
		N_LOADS = (PIECE + 2 + 127) / 128 }; // 16-byte requests per lane that bring a whole piece (one double of slack at either end)
	typedef double v2f64 __attribute__((ext_vector_type(2)));
	__shared__ __attribute__((aligned(16))) double s_u[N_LOADS * 128];
	const int lane = threadIdx.x;
	const int64_t o_first = int64_t(blockIdx.x) * 64;
	const int64_t o = min(o_first + lane, n_obs - 1); // (the tail repeats the last observation)
	const int64_t pt = obs_pt[o];
	const int64_t n_off = o * BLK + pt * (DP * DP); // offset of U_o behind ubase
	const int64_t n_off_first = (int64_t(__builtin_amdgcn_readlane(int(n_off >> 32), 0)) << 32) | uint32_t(__builtin_amdgcn_readlane(int(n_off), 0));
	const int64_t n_off_last = (int64_t(__builtin_amdgcn_readlane(int(n_off >> 32), 63)) << 32) | uint32_t(__builtin_amdgcn_readlane(int(n_off), 63));
	const int64_t n_piece = n_off_last - n_off_first + BLK;
	const bool b_staged = n_piece <= PIECE && (reinterpret_cast<uintptr_t>(A) & 15) == 0; // (wave-uniform; many one- or two-camera landmarks in a row:
	// every lane fetches its own block -- and so where the caller's values start at an odd double: the 16-byte requests below
	// count on an even one; slampp_hip_factor_solve_device_async passes the caller's pointer through unchecked: advisor, round 5)
	// (round 5) the whole piece is requested at once, in 16-byte pieces -- up to eleven per lane in flight -- and the camera's dx
	// beside it: the loop of eight 8-byte requests, wait, store took up to three round trips a wave, and the gather of dx a
	// fourth behind them.  A landmark's blocks start at an odd double where an odd number of 3 x 3 blocks precedes them: the
	// piece is fetched from the even double at or before its first one (n_shift), which stays inside the values.
	const int n_shift = int((ubase + n_off_first) & 1);
	const int n_piece_al = (int(n_piece) + n_shift + 1) & ~1; // doubles fetched: from the even double before the piece to the one after it
	v2f64 v[N_LOADS];
	if(b_staged) {
		const double *p_src = A + (ubase + n_off_first - n_shift);
		#pragma unroll
		for(int u = 0; u < N_LOADS; ++ u) {
			const int e = 2 * (64 * u + lane);
			v[u] = *reinterpret_cast<const v2f64*>(p_src + ((e < n_piece_al)? e : 0)); // (past the end: the first pair again)
		}
	}
	const int64_t cam = brow[ptr[nc] + o + pt]; // block index of observation o = ptr[nc] + o + (number of C blocks before it)
	double x[DC];
	#pragma unroll
	for(int i = 0; i < DC; ++ i)
		x[i] = dx[cam * DC + i];
	if(b_staged) {
		#pragma unroll
		for(int u = 0; u < N_LOADS; ++ u) {
			const int e = 2 * (64 * u + lane);
			if(e < n_piece_al)
				*reinterpret_cast<v2f64*>(s_u + e) = v[u];
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const double *U = b_staged? s_u + (n_off - n_off_first) + n_shift : A + ubase + n_off;
	if(o_first + lane < n_obs) {
		#pragma unroll
		for(int t = 0; t < DP; ++ t) {
			double sum = 0;
			#pragma unroll
			for(int i = 0; i < DC; ++ i)
				sum += U[i + t * DC] * x[i];
			T[o * DP + t] = sum;
		}
	}
}

template <int DC, int DP>
__global__ void schur_point_backsubst_kernel(const int64_t *ptr, int64_t nc, int64_t np, int n,
	const double *__restrict__ Cinv, const double *__restrict__ T, const double *__restrict__ dx, double *out)
{
	const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(gid < np) {
		const int64_t pt = gid;
		const int64_t o0 = ptr[nc + pt] - ptr[nc] - pt, o1 = ptr[nc + pt + 1] - ptr[nc] - (pt + 1);
		double v[DP];
		#pragma unroll
		for(int t = 0; t < DP; ++ t)
			v[t] = out[n + pt * DP + t];
		for(int64_t o = o0; o < o1; ++ o) {
			#pragma unroll
			for(int t = 0; t < DP; ++ t)
				v[t] -= T[o * DP + t];
		}
		#pragma unroll
		for(int r = 0; r < DP; ++ r) {
			double s = 0;
			#pragma unroll
			for(int t = 0; t < DP; ++ t)
				s += Cinv[pt * (DP * DP) + r + t * DP] * v[t];
			out[n + pt * DP + r] = s;
		}
	}
	if(gid < n)
		out[gid] = dx[gid];
}

// multi-GPU exchange: copies the union blocks of S (and the right-hand side row) into / out of one contiguous buffer
__global__ void __launch_bounds__(64)
schur_pack_kernel(const int32_t *__restrict__ un_row, const int32_t *__restrict__ un_col, int64_t n_union, int dc,
	double *S, int ld, int n, double *pack, int b_unpack)
{
	const int64_t b = blockIdx.x;
	const int lane = threadIdx.x;
	double *p_s, *p_p;
	if(b < n_union) {
		if(lane >= dc * dc)
			return;
		p_s = S + size_t(un_row[b] * dc + lane % dc) + size_t(un_col[b] * dc + lane / dc) * ld;
		p_p = pack + b * dc * dc + lane;
	} else {
		const int64_t i = (b - n_union) * 64 + lane;
		if(i >= n)
			return;
		p_s = S + size_t(ld - 1) + size_t(i) * ld;
		p_p = pack + n_union * dc * dc + i;
	}
	if(b_unpack)
		*p_s = *p_p;
	else
		*p_p = *p_s;
}

// Agrees with the other ranks on the set of blocks to exchange, through the caller's sum all-reduce alone: with
// "shard_rank" / "shard_world" set, the ranks concatenate their block lists (each writes into its own slot of a
// zeroed buffer); without, every rank marks its blocks in an indicator over the lower triangle of the camera-block
// grid.  Either way every rank derives the same ordered list from the sum.  One-time, synchronous.
static void schur_agree_on_union(slampp_hip_solver &s, CSchurState &S)
{
	hipStream_t st = s.stream;
	S.p_union_fn = s.p_allreduce;
	S.p_union_context = s.p_allreduce_context;
	S.n_union = 0;
	S.h_un_row.clear();
	S.h_un_col.clear();
	S.b_reduced_decided = false; // the block list the sparse reduced system is built from may change
	const int64_t nc = S.nc;
	std::vector<int32_t> un_row, un_col;
	// this rank's blocks as sorted, unique keys col * nc + row
	std::vector<int64_t> own(S.h_blk_row.size());
	for(size_t i = 0; i < own.size(); ++ i)
		own[i] = int64_t(S.h_blk_col[i]) * nc + S.h_blk_row[i];
	std::sort(own.begin(), own.end());
	own.erase(std::unique(own.begin(), own.end()), own.end());
	const int n_world = s.n_shard_world, n_rank = s.n_shard_rank;
	S.b_union_dense = false;
	if(n_world > 0 && n_rank >= 0 && n_rank < n_world) {
		// the caller told us who we are: every rank writes its list into its own slot of a zeroed buffer and the sum
		// is the concatenation -- the exchange grows with the number of nonzero blocks, not with nc^2
		std::vector<double> len(size_t(n_world), 0.0);
		len[n_rank] = double(own.size());
		CDevArray<double> d_len;
		d_len.Alloc(size_t(n_world));
		SLAMPP_HIP_CHECK(hipMemcpyAsync(d_len.p(), len.data(), len.size() * sizeof(double), hipMemcpyHostToDevice, st));
		if(s.p_allreduce(s.p_allreduce_context, d_len.p(), len.size(), (void*)st) != 0)
			throw CDeviceError("all-reduce callback failed");
		SLAMPP_HIP_CHECK(hipMemcpyAsync(len.data(), d_len.p(), len.size() * sizeof(double), hipMemcpyDeviceToHost, st));
		SLAMPP_HIP_CHECK(hipStreamSynchronize(st));
		size_t n_total = 0, n_before = 0;
		for(int r = 0; r < n_world; ++ r) {
			if(r == n_rank)
				n_before = n_total;
			n_total += size_t(len[r]);
		}
		if(size_t(len[n_rank]) != own.size())
			throw std::invalid_argument("shard_rank / shard_world do not match the ranks behind the all-reduce callback");
		std::vector<double> all(n_total, 0.0);
		for(size_t i = 0; i < own.size(); ++ i)
			all[n_before + i] = double(own[i]); // exact: keys are below 2^53
		CDevArray<double> d_all;
		d_all.Alloc(n_total);
		SLAMPP_HIP_CHECK(hipMemcpyAsync(d_all.p(), all.data(), n_total * sizeof(double), hipMemcpyHostToDevice, st));
		if(s.p_allreduce(s.p_allreduce_context, d_all.p(), n_total, (void*)st) != 0)
			throw CDeviceError("all-reduce callback failed");
		SLAMPP_HIP_CHECK(hipMemcpyAsync(all.data(), d_all.p(), n_total * sizeof(double), hipMemcpyDeviceToHost, st));
		SLAMPP_HIP_CHECK(hipStreamSynchronize(st));
		std::vector<int64_t> keys(n_total);
		for(size_t i = 0; i < n_total; ++ i)
			keys[i] = int64_t(all[i]);
		std::sort(keys.begin(), keys.end());
		keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
		for(size_t i = 0; i < keys.size(); ++ i) {
			if(keys[i] < 0 || keys[i] >= nc * nc || keys[i] % nc < keys[i] / nc)
				throw std::invalid_argument("block-list exchange returned an impossible key: is the callback a sum over all ranks?");
			un_col.push_back(int32_t(keys[i] / nc));
			un_row.push_back(int32_t(keys[i] % nc));
		}
	} else {
		// ranks unknown: an indicator over the lower triangle of the camera-block grid, summed
		S.b_union_dense = S.nc > 16384; // the indicator would exceed a gigabyte
		if(S.b_union_dense)
			return;
		const int64_t n_tri = nc * (nc + 1) / 2;
		std::vector<double> ind(size_t(n_tri), 0.0);
		for(size_t i = 0; i < own.size(); ++ i) {
			const int64_t c = own[i] / nc, r = own[i] % nc;
			ind[size_t(c * nc - c * (c - 1) / 2 + (r - c))] = 1.0;
		}
		CDevArray<double> d_ind;
		d_ind.Alloc(size_t(n_tri));
		SLAMPP_HIP_CHECK(hipMemcpyAsync(d_ind.p(), ind.data(), size_t(n_tri) * sizeof(double), hipMemcpyHostToDevice, st));
		if(s.p_allreduce(s.p_allreduce_context, d_ind.p(), size_t(n_tri), (void*)st) != 0)
			throw CDeviceError("all-reduce callback failed");
		SLAMPP_HIP_CHECK(hipMemcpyAsync(ind.data(), d_ind.p(), size_t(n_tri) * sizeof(double), hipMemcpyDeviceToHost, st));
		SLAMPP_HIP_CHECK(hipStreamSynchronize(st));
		size_t k = 0;
		for(int64_t c = 0; c < nc; ++ c) {
			for(int64_t r = c; r < nc; ++ r, ++ k) {
				if(ind[k] > 0.5) {
					un_row.push_back(int32_t(r));
					un_col.push_back(int32_t(c));
				}
			}
		}
	}
	S.n_union = int64_t(un_row.size());
	S.h_un_row = un_row;
	S.h_un_col = un_col;
	S.d_un_row.Upload(un_row, st);
	S.d_un_col.Upload(un_col, st);
	SLAMPP_HIP_CHECK(hipStreamSynchronize(st)); // un_row / un_col live on this stack frame
}

// Landmark shards: a rank whose own landmarks gave a C_p that is not positive definite must not be the only one to
// return false (the others would carry on into the next collective without it).  The status travels with the data:
// that rank poisons the first entry of its partial reduced right-hand side before the exchange, and every rank looks
// at the sum afterwards.
__global__ void schur_flag_poison_kernel(const int *p_flag, double *p_rhs0)
{
	if(*p_flag)
		*p_rhs0 = __builtin_nan("");
}

__global__ void schur_flag_check_kernel(const double *p_rhs0, int *p_flag)
{
	const double f = *p_rhs0;
	if(f != f)
		atomicOr(p_flag, 1);
}

// Decides how the reduced camera system is factored and, for the sparse choice, builds the inner solver: S becomes
// a block matrix with one block column per camera whose structure is the block list every rank agreed on (or this
// rank's own list on a single GPU), analyzed once by the same ordering / symbolic / scheduling code as a pose graph.
static void schur_try_sparse_reduced(slampp_hip_solver &s, CSchurState &S);

static void schur_setup_reduced(slampp_hip_solver &s, CSchurState &S)
{
	schur_try_sparse_reduced(s, S);
	if(!S.b_reduced_sparse) { // the dense buffers are only needed now
		S.d_S.Alloc(size_t(S.Npad) * S.Npad);
		S.d_invdiag.Alloc(size_t(S.Npad / dense_NB) * dense_NB * dense_NB);
		S.d_z.Alloc(S.Npad);
		S.d_x.Alloc(S.Npad);
		if(s.p_allreduce && !S.b_union_dense)
			S.d_pack.Alloc(size_t(S.n_union) * S.DC * S.DC + size_t(S.N));
	}
	S.b_reduced_decided = true;
}

static void schur_try_sparse_reduced(slampp_hip_solver &s, CSchurState &S)
{
	S.b_reduced_sparse = false;
	if(S.p_sinv) { // lists of the previous inner solver
		sparse_inverse_destroy(S.p_sinv);
		S.p_sinv = 0;
	}
	S.b_sinv_tried = false;
	if(S.p_inner) {
		S.p_inner->stream = 0;
		delete S.p_inner;
		S.p_inner = 0;
	}
	if(s.n_schur_sparse == 0 || (s.p_allreduce && S.b_union_dense))
		return;
	std::vector<int32_t> rows, cols;
	if(s.p_allreduce) {
		rows = S.h_un_row;
		cols = S.h_un_col;
	} else {
		std::vector<int64_t> keys(S.h_blk_row.size());
		for(size_t i = 0; i < keys.size(); ++ i)
			keys[i] = int64_t(S.h_blk_col[i]) * S.nc + S.h_blk_row[i];
		std::sort(keys.begin(), keys.end());
		keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
		rows.resize(keys.size());
		cols.resize(keys.size());
		for(size_t i = 0; i < keys.size(); ++ i) {
			cols[i] = int32_t(keys[i] / S.nc);
			rows[i] = int32_t(keys[i] % S.nc);
		}
	}
	const int64_t nc = S.nc, n_list = int64_t(rows.size());
	const double f_fill = double(n_list) / (0.5 * double(nc) * double(nc + 1));
	// measured (Venice-like visibility, 300k landmarks): 3 % of the blocks nonzero 6.6 against 17.7 ms, 6 % (C4's Venice leg)
	// 3.6 against 5.2, 12 % 1.8 against 2.3, 20 % 1.74 against 1.69, every pair (uniform) 5.7 against 5.4
	if(s.n_schur_sparse < 0 && !(nc >= 128 && f_fill < 0.15))
		return; // dense: the MFMA factorization wins once S is effectively dense
	// upper block-CSC: the lower block (r, c) is the transpose of the upper block (c, r) in block column r
	std::vector<int64_t> cumsum(nc + 1), bcol_ptr(nc + 1, 0);
	for(int64_t c = 0; c <= nc; ++ c)
		cumsum[c] = c * S.DC;
	for(int64_t i = 0; i < n_list; ++ i)
		++ bcol_ptr[rows[i] + 1];
	for(int64_t c = 0; c < nc; ++ c)
		bcol_ptr[c + 1] += bcol_ptr[c];
	std::vector<int32_t> brow(n_list);
	std::vector<int64_t> in_off(n_list);
	{
		std::vector<int64_t> fill(bcol_ptr.begin(), bcol_ptr.end() - 1);
		for(int64_t i = 0; i < n_list; ++ i) { // the list is sorted by (col, row): within a block column the rows come out ascending
			const int64_t k = fill[rows[i]] ++;
			brow[k] = cols[i];
			in_off[i] = k * S.DC * S.DC;
		}
	}
	for(int64_t c = 0; c < nc; ++ c) {
		if(bcol_ptr[c + 1] == bcol_ptr[c] || brow[bcol_ptr[c + 1] - 1] != c)
			throw std::logic_error("reduced camera system: a camera has no diagonal block");
	}
	slampp_hip_solver *p_inner = new slampp_hip_solver();
	S.p_inner = p_inner;
	p_inner->n_device = s.n_device;
	p_inner->stream = s.stream; // borrowed
	p_inner->opt = s.opt;
	p_inner->n_simt = s.n_simt;
	p_inner->n_simt_width = s.n_simt_width;
	p_inner->n_simt_stages = s.n_simt_stages;
	p_inner->n_wide_min_tasks = s.n_wide_min_tasks;
	p_inner->n_panel_rows = s.n_panel_rows;
	p_inner->n_panel_handup = s.n_panel_handup;
	p_inner->n_dense_top_tiles = s.n_dense_top_tiles;
	// (a small system is all latency: round 1 cut its leaf subtrees to four columns for the wave-per-task kernel; as panels
	// -- eight waves per subtree, 2 us per column -- the default of eight is faster again: 0.231 -> 0.220 ms at 1000 cameras)
	p_inner->cumsum = cumsum;
	p_inner->bcol_ptr = bcol_ptr;
	p_inner->brow = brow;
	p_inner->n_values = n_list * S.DC * S.DC;
	p_inner->n_scalars = nc * S.DC;
	p_inner->b_has_structure = true;
	p_inner->n_mode = SLAMPP_HIP_MODE_SPARSE;
	p_inner->Analyze_Sparse();
	p_inner->b_analyzed = true;
	S.n_in_blocks = n_list;
	// where this rank's own blocks go: the list is sorted by (col, row)
	std::vector<int64_t> keys(n_list);
	for(int64_t i = 0; i < n_list; ++ i)
		keys[i] = int64_t(cols[i]) * nc + rows[i];
	const size_t n_own = S.h_blk_row.size() - size_t(S.n_ablocks); // h_blk = [blocks of the gather | camera blocks of Lambda]
	std::vector<int64_t> sb_dst(n_own), a_dst(S.n_ablocks);
	for(size_t i = 0; i < S.h_blk_row.size(); ++ i) {
		const int64_t key = int64_t(S.h_blk_col[i]) * nc + S.h_blk_row[i];
		const size_t k = size_t(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
		if(k == keys.size() || keys[k] != key)
			throw std::logic_error("reduced camera system: a block of this rank is missing from the agreed list");
		((i < n_own)? sb_dst[i] : a_dst[i - n_own]) = in_off[k];
	}
	S.d_sb_dst.Upload(sb_dst, s.stream);
	S.d_a_dst.Upload(a_dst, s.stream);
	S.d_in_buf.Alloc(size_t(n_list) * S.DC * S.DC + size_t(S.N));
	SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream)); // the host vectors live on this stack frame
	S.b_reduced_sparse = true;
}

// S -= U C^-1 U^T and r -= U C^-1 l for all landmarks, C^-1 (and W = U C^-1 for all observations, if b_store_W) left
// behind: S is the dense buffer (ld) or, with p_sb_dst, the packed values of the inner solver (p_r its right-hand side)
template <int DC, int DP>
static void schur_assemble_t(slampp_hip_solver &s, CSchurState &S, const double *A, const double *rhs, double *p_S, int ld,
	const int64_t *p_sb_dst, double *p_r, bool b_store_W)
{
	hipStream_t st = s.stream;
	const int n = S.N;
	const int64_t ubase = S.n_ablocks * DC * DC;
	// Landmarks go through the tiles of schur_tiles.hip (read once each), through the contribution lists, or -- when only
	// some of them fit the tiles -- both: the tiles take theirs, the lists the rest.
	const CSchurTiles &T = S.tiles;
	const bool b_tiles = T.b_enabled, b_lists_all = !b_tiles;
	// (a phase that launches nothing records no events: an event pair costs microseconds of stream time)
	if(b_lists_all || T.b_hybrid)
		s.Phase_Begin("schur_points");
	if(b_lists_all) {
		hipLaunchKernelGGL((schur_point_inverse_kernel<DC, DP>), dim3(unsigned((S.np + 255) / 256)), dim3(256), 0, st,
			S.d_ptr.p(), S.nc, S.np, ubase, A, S.d_Cinv.p(), s.d_flag.p(), (const int32_t*)0);
		hipLaunchKernelGGL((schur_obs_W_kernel<DC, DP>), dim3(unsigned((S.n_obs + 255) / 256)), dim3(256), 0, st,
			S.n_obs, ubase, S.d_obs_pt.p(), A, S.d_Cinv.p(), S.d_W.p(), (const int32_t*)0);
	} else if(T.b_hybrid) { // only the landmarks of the lists: the others get C^-1 (and W) where they are multiplied
		hipLaunchKernelGGL((schur_point_inverse_kernel<DC, DP>), dim3(unsigned((T.n_list_points + 255) / 256)), dim3(256), 0, st,
			S.d_ptr.p(), S.nc, T.n_list_points, ubase, A, S.d_Cinv.p(), s.d_flag.p(), T.d_xpoints.p());
		if(T.n_xobs)
			hipLaunchKernelGGL((schur_obs_W_kernel<DC, DP>), dim3(unsigned((T.n_xobs + 255) / 256)), dim3(256), 0, st,
				T.n_xobs, ubase, S.d_obs_pt.p(), A, S.d_Cinv.p(), S.d_W.p(), T.d_xcam_obs.p());
	}
	if(b_lists_all || T.b_hybrid)
		s.Phase_End();

	if(b_tiles) {
		s.Phase_Begin("schur_tiles");
		// (C^-1 of every landmark is needed by the back-substitution; W only if the next solve may be an update)
		schur_tiles_enqueue(T, DC, DP, S.d_ptr.p(), S.nc, ubase, A, rhs, n, S.d_Cinv.p(), b_store_W? S.d_W.p() : 0,
			true, S.d_sb_row.p(), S.d_sb_col.p(), p_S, ld, p_sb_dst, p_r, s.d_flag.p(), st);
		s.Phase_End();
	}

	if((b_lists_all? S.n_sblocks : T.n_xblocks) > 0) {
		s.Phase_Begin("schur_gather");
		const int64_t n_lists = b_lists_all? S.n_sblocks : T.n_xblocks, n_list_entries = b_lists_all? S.n_entries : T.n_xentries;
		const int64_t *p_list_ptr = b_lists_all? S.d_sb_ptr.p() : T.d_xsb_ptr.p();
		const int32_t *p_list_a = b_lists_all? S.d_ent_a.p() : T.d_xent_a.p(), *p_list_map = b_lists_all? 0 : T.d_xsb_map.p();
		const int64_t *p_list_uoff = b_lists_all? S.d_ent_uoff.p() : T.d_xent_uoff.p();
		if(n_lists > 0) {
			// one wave per block of S for short contribution lists (dense S: 500k blocks x 10 contributions),
			// 8 waves per block for long ones (C4's band structure: 4000 blocks x 1250 contributions: 1.18 -> 0.69 ms)
			if(n_list_entries > 256 * n_lists)
				hipLaunchKernelGGL((schur_gather_S_kernel<DC, DP, 8>), dim3(unsigned(n_lists)), dim3(512), 0, st,
					n_lists, p_list_ptr, S.d_sb_row.p(), S.d_sb_col.p(), p_list_a, p_list_uoff, A,
					S.d_W.p(), p_S, ld, p_sb_dst, p_list_map);
			else
				hipLaunchKernelGGL((schur_gather_S_kernel<DC, DP, 1>), dim3(unsigned(n_lists)), dim3(64), 0, st,
					n_lists, p_list_ptr, S.d_sb_row.p(), S.d_sb_col.p(), p_list_a, p_list_uoff, A,
					S.d_W.p(), p_S, ld, p_sb_dst, p_list_map);
		}
		s.Phase_End();
	}

	if(!S.tiles.b_enabled || S.tiles.b_hybrid) { // (the tiles bring the right-hand side's share of their landmarks themselves)
		s.Phase_Begin("schur_rhs");
		if(!S.tiles.b_enabled)
			hipLaunchKernelGGL((schur_rhs_kernel<DC, DP>), dim3(unsigned(S.nc)), dim3(64), 0, st,
				S.d_cam_ptr.p(), S.d_cam_obs.p(), S.d_obs_pt.p(), n, S.d_W.p(), rhs, p_S, ld, p_r);
		else
			hipLaunchKernelGGL((schur_rhs_kernel<DC, DP>), dim3(unsigned(S.nc)), dim3(64), 0, st,
				S.tiles.d_xcam_ptr.p(), S.tiles.d_xcam_obs.p(), S.d_obs_pt.p(), n, S.d_W.p(), rhs, p_S, ld, p_r);
		s.Phase_End();
	}
}

template <int DC, int DP>
static void schur_enqueue_t(slampp_hip_solver &s, CSchurState &S, const double *A, double *rhs)
{
	hipStream_t st = s.stream;
	const int ld = S.Npad, n = S.N;
	const int64_t ubase = S.n_ablocks * DC * DC;
	// first step (or a new all-reduce callback): the ranks agree on the blocks of S, then it is decided whether the
	// reduced system is assembled into the dense buffer or straight into the inner sparse solver's values
	if(s.p_allreduce && (S.p_union_fn != s.p_allreduce || S.p_union_context != s.p_allreduce_context))
		schur_agree_on_union(s, S);
	if(!S.b_reduced_decided)
		schur_setup_reduced(s, S);
	const bool b_sparse = S.b_reduced_sparse;
	const size_t n_in_values = size_t(S.n_in_blocks) * DC * DC;
	// option "schur_incremental": what this solve assembles is kept (the dense system in a buffer of its own: the
	// factorization works in place), and a solve that names the changed landmarks updates it instead of rebuilding it
	const bool b_keep = s.n_schur_incremental != 0 && !s.p_allreduce && s.b_shard_primary;
	// ... when that is the shorter way (option value 1; 2 = whenever a list is given): the update exchanges the listed
	// landmarks' contributions with atomic adds (C4: 0.44 ms a solve at 500 landmarks, 0.46 at 5 000, 0.51 at 15 000, 0.69
	// at 50 000) and rebuilds the reduced right-hand side; the landmark-major assembly rebuilds everything (0.55 ms a
	// solve), the contribution lists take a millisecond longer (there the update still wins at 50 000: 4.6 against 5.5 ms)
	const bool b_pays = s.n_schur_incremental >= 2 || S.n_changed * (S.tiles.b_enabled? 32 : 4) <= S.np;
	const bool b_update = b_keep && S.b_prev_valid && S.n_changed >= 0 && b_pays;
	if(b_keep) {
		S.d_A_prev.Alloc(size_t(S.n_ablocks) * DC * DC);
		if(!b_sparse)
			S.d_S_unf.Alloc(size_t(ld) * ld);
	}
	double *p_S = b_sparse? S.d_in_buf.p() : ((b_keep)? S.d_S_unf.p() : S.d_S.p()); // where the blocks go
	double *p_r = b_sparse? S.d_in_buf.p() + n_in_values : 0;   // the reduced right-hand side, if it is a vector of its own
	const int64_t *p_sb_dst = b_sparse? S.d_sb_dst.p() : 0, *p_a_dst = b_sparse? S.d_a_dst.p() : 0;

	if(b_update) {
		s.Phase_Begin("schur_update");
		const int64_t n_work = std::max<int64_t>(S.n_ablocks * DC * DC, n);
		hipLaunchKernelGGL((schur_delta_A_kernel<DC>), dim3(unsigned((n_work + 255) / 256)), dim3(256), 0, st,
			S.d_ptr.p(), S.d_brow.p(), S.nc, A, S.d_A_prev.p(), rhs, p_S, ld, n, p_a_dst, p_r);
		if(S.n_changed > 0)
			hipLaunchKernelGGL((schur_changed_points_kernel<DC, DP>), dim3(unsigned(S.n_changed)), dim3(64), 0, st,
				S.d_changed.p(), S.n_changed, S.d_ptr.p(), S.d_brow.p(), S.nc, ubase, A, S.d_Cinv.p(), S.d_W.p(), S.n_sblocks,
				S.d_sb_row.p(), S.d_sb_col.p(), p_S, ld, p_sb_dst, s.d_flag.p());
		s.Phase_End();
	} else {
	s.Phase_Begin("schur_init");
	if(b_sparse)
		SLAMPP_HIP_CHECK(hipMemsetAsync(p_S, 0, (n_in_values + size_t(n)) * sizeof(double), st));
	else {
		SLAMPP_HIP_CHECK(hipMemsetAsync(p_S, 0, size_t(ld) * ld * sizeof(double), st));
		dense_prepare_padding(p_S, ld, n, st);
	}
	if(s.b_shard_primary) {
		const int64_t n_work = std::max<int64_t>(S.n_ablocks * DC * DC, n);
		hipLaunchKernelGGL((schur_scatter_A_kernel<DC>), dim3(unsigned((n_work + 255) / 256)), dim3(256), 0, st,
			S.d_ptr.p(), S.d_brow.p(), S.nc, A, rhs, p_S, ld, n, p_a_dst, p_r);
	}
	if(b_keep)
		SLAMPP_HIP_CHECK(hipMemcpyAsync(S.d_A_prev.p(), A, size_t(S.n_ablocks) * DC * DC * sizeof(double), hipMemcpyDeviceToDevice, st));
	s.Phase_End();

	schur_assemble_t<DC, DP>(s, S, A, rhs, p_S, ld, p_sb_dst, p_r, b_keep);
	}
	S.b_prev_valid = b_keep; // (a solve that turns out not positive definite takes it back: slampp_hip_sync)
	S.n_changed = -1;        // the list serves one solve

	if(b_update) { // (a full assembly brings the reduced right-hand side itself)
		s.Phase_Begin("schur_rhs");
		hipLaunchKernelGGL((schur_rhs_kernel<DC, DP>), dim3(unsigned(S.nc)), dim3(64), 0, st,
			S.d_cam_ptr.p(), S.d_cam_obs.p(), S.d_obs_pt.p(), n, S.d_W.p(), rhs, p_S, ld, p_r);
		s.Phase_End();
	}

	// option "schur_distributed" (members of a device group, dense reduced system): reduce-scatter + joint factorization
	const bool b_distributed = s.p_allreduce && !b_sparse && s.n_schur_distributed != 0 && s.p_dense_factor != 0;
	if(b_distributed) {
		s.Phase_Begin("dense_chol_distributed");
		if(s.p_dense_factor(s.p_dense_factor_context, p_S, ld, n, S.d_invdiag.p(), s.d_flag.p(), (void*)st) != 0)
			throw CDeviceError("distributed factorization of the reduced camera system failed");
		s.Phase_End();
	} else if(s.p_allreduce) {
		s.Phase_Begin("allreduce");
		double *p_rhs0 = b_sparse? p_r : p_S + n; // dense: the right-hand side is row n of S
		hipLaunchKernelGGL(schur_flag_poison_kernel, dim3(1), dim3(1), 0, st, s.d_flag.p(), p_rhs0);
		if(b_sparse) {
			// the packed values and the right-hand side are one buffer, and every rank built it from the same block list
			if(s.p_allreduce(s.p_allreduce_context, p_S, n_in_values + size_t(n), (void*)st) != 0)
				throw CDeviceError("all-reduce callback failed");
		} else if(S.b_union_dense) {
			// only the lower triangle and the rhs row carry data; the callback sums the whole buffer
			if(s.p_allreduce(s.p_allreduce_context, p_S, size_t(ld) * ld, (void*)st) != 0)
				throw CDeviceError("all-reduce callback failed");
		} else {
			const unsigned n_grid = unsigned(S.n_union + (n + 63) / 64);
			const size_t n_count = size_t(S.n_union) * DC * DC + size_t(n);
			hipLaunchKernelGGL(schur_pack_kernel, dim3(n_grid), dim3(64), 0, st, S.d_un_row.p(), S.d_un_col.p(), S.n_union, DC,
				p_S, ld, n, S.d_pack.p(), 0);
			if(s.p_allreduce(s.p_allreduce_context, S.d_pack.p(), n_count, (void*)st) != 0)
				throw CDeviceError("all-reduce callback failed");
			hipLaunchKernelGGL(schur_pack_kernel, dim3(n_grid), dim3(64), 0, st, S.d_un_row.p(), S.d_un_col.p(), S.n_union, DC,
				p_S, ld, n, S.d_pack.p(), 1);
		}
		hipLaunchKernelGGL(schur_flag_check_kernel, dim3(1), dim3(1), 0, st, p_rhs0, s.d_flag.p());
		s.Phase_End();
	}

	const double *p_dx;
	if(b_sparse) {
		s.Phase_Begin("reduced_sparse");
		S.p_inner->p_flag_shared = s.d_flag.p();
		S.p_inner->n_panel_rows = s.n_panel_rows; // (read at every launch)
		S.p_inner->Enqueue_Sparse(p_S, p_r, true); // p_r: the reduced right-hand side on entry, dx on return
		s.Phase_End();
		p_dx = p_r;
	} else {
		if(!b_distributed) {
			s.Phase_Begin("dense_chol");
			if(b_keep) { // the assembled system stays as it is; its copy is factored
				SLAMPP_HIP_CHECK(hipMemcpyAsync(S.d_S.p(), p_S, size_t(ld) * ld * sizeof(double), hipMemcpyDeviceToDevice, st));
				p_S = S.d_S.p();
			}
			dense_cholesky(p_S, ld, n, S.d_invdiag.p(), s.d_flag.p(), st);
			s.Phase_End();
		}
		s.Phase_Begin("dense_solve");
		dense_backsolve(p_S, ld, n, S.d_invdiag.p(), S.d_z.p(), S.d_x.p(), st);
		s.Phase_End();
		p_dx = S.d_x.p();
	}

	s.Phase_Begin("backsubst");
	hipLaunchKernelGGL((schur_obs_t_kernel<DC, DP>), dim3(unsigned((S.n_obs + 63) / 64)), dim3(64), 0, st,
		S.n_obs, ubase, S.nc, S.d_ptr.p(), S.d_brow.p(), S.d_obs_pt.p(), A, p_dx, S.d_t.p());
	const int64_t n_work = std::max<int64_t>(S.np, n);
	hipLaunchKernelGGL((schur_point_backsubst_kernel<DC, DP>), dim3(unsigned((n_work + 255) / 256)), dim3(256), 0, st,
		S.d_ptr.p(), S.nc, S.np, n, S.d_Cinv.p(), S.d_t.p(), p_dx, rhs);
	s.Phase_End();
	SLAMPP_HIP_CHECK(hipGetLastError());
}

// dl_p = C_p^-1 l_p, dx = 0
template <int DP>
__global__ void schur_landmarks_only_kernel(int64_t np, int n, const double *__restrict__ Cinv, double *out)
{
	const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(gid < np) {
		double v[DP];
		#pragma unroll
		for(int t = 0; t < DP; ++ t)
			v[t] = out[n + gid * DP + t];
		#pragma unroll
		for(int r = 0; r < DP; ++ r) {
			double sum = 0;
			#pragma unroll
			for(int t = 0; t < DP; ++ t)
				sum += Cinv[gid * (DP * DP) + r + t * DP] * v[t];
			out[n + gid * DP + r] = sum;
		}
	}
	if(gid < n)
		out[gid] = 0.0;
}

template <int DC, int DP>
static void schur_enqueue_marginal_t(slampp_hip_solver &s, CSchurState &S, const double *A, double *rhs)
{
	hipStream_t st = s.stream;
	const int64_t ubase = S.n_ablocks * DC * DC;
	s.Phase_Begin("landmarks_only");
	hipLaunchKernelGGL((schur_point_inverse_kernel<DC, DP>), dim3(unsigned((S.np + 255) / 256)), dim3(256), 0, st,
		S.d_ptr.p(), S.nc, S.np, ubase, A, S.d_Cinv.p(), s.d_flag.p());
	const int64_t n_work = std::max<int64_t>(S.np, S.N);
	hipLaunchKernelGGL((schur_landmarks_only_kernel<DP>), dim3(unsigned((n_work + 255) / 256)), dim3(256), 0, st,
		S.np, S.N, S.d_Cinv.p(), rhs);
	s.Phase_End();
	SLAMPP_HIP_CHECK(hipGetLastError());
}

void schur_marginals_launch(int DC, int DP, int64_t nc, int64_t np, const int64_t *ptr, const int32_t *brow, const double *W,
	const double *Cinv, const double *Z, int ld, double *cam_cov, double *point_cov, hipStream_t stream); // schur_marginals.hip

// Block diagonal of Lambda^-1 (see schur_marginals.hip): the reduced system is assembled into a dense buffer of its
// own whatever way the solves factor it, factored, inverted on the matrix cores, then gathered per landmark.
// Landmark shards: S is summed over the ranks as a whole buffer (the padding diagonal comes back as the number
// of ranks, which its decoupled rows do not mind); every rank then writes the covariances of its own landmarks.
// lists of the sparse inverse subset and the tables that say where the blocks the covariances need sit in it; false if
// the inner solver's plan is not of the kind sparse_inverse_setup takes (then the dense inverse is used)
static bool schur_setup_sparse_marginals(slampp_hip_solver &s, CSchurState &S)
{
	if(S.b_sinv_tried)
		return S.p_sinv != 0;
	S.b_sinv_tried = true;
	const Plan &P = S.p_inner->plan;
	if(P.max_dim != S.DC)
		return false;
	S.p_sinv = sparse_inverse_setup(P, s.stream);
	if(!S.p_sinv)
		return false;
	const int64_t nc = S.nc, np = S.np;
	const int64_t *ptr = s.bcol_ptr.data();
	const int32_t *brow = s.brow.data();
	std::vector<int64_t> cam_zoff(nc), pair_ptr(np + 1, 0);
	for(int64_t c = 0; c < nc; ++ c)
		cam_zoff[c] = P.loff[P.lptr[P.pinv[c]]];
	for(int64_t pt = 0; pt < np; ++ pt) {
		const int64_t k = ptr[nc + pt + 1] - ptr[nc + pt] - 1;
		pair_ptr[pt + 1] = pair_ptr[pt] + k * (k + 1) / 2;
	}
	std::vector<int64_t> pair_tab((size_t(pair_ptr[np])));
	for(int64_t pt = 0; pt < np; ++ pt) {
		const int64_t k0 = ptr[nc + pt], k = ptr[nc + pt + 1] - k0 - 1;
		int64_t *tab = pair_tab.data() + pair_ptr[pt];
		for(int64_t a = 0; a < k; ++ a) {
			const int32_t pa = P.pinv[brow[k0 + a]];
			for(int64_t b = 0; b <= a; ++ b) {
				const int32_t pb = P.pinv[brow[k0 + b]];
				const int64_t off = plan_block_offset(P, std::max(pa, pb), std::min(pa, pb));
				if(off < 0)
					throw std::logic_error("covariances: a camera pair that shares a landmark is not a block of the reduced system's factor");
				tab[a * (a + 1) / 2 + b] = off * 2 + (pa < pb); // Z(cam_a, cam_b) is the stored block, or its transpose
			}
		}
	}
	S.d_cam_zoff.Upload(cam_zoff, s.stream);
	S.d_pair_ptr.Upload(pair_ptr, s.stream);
	S.d_pair_tab.Upload(pair_tab, s.stream);
	S.d_m_Zs.Alloc(size_t(P.loff.back()));
	if(!S.d_m_zero.p()) {
		const size_t n_zero = size_t(S.N) + size_t(S.np) * S.DP; // a whole right-hand side of zeros (the assembly reads the landmarks' part too)
		S.d_m_zero.Alloc(n_zero);
		SLAMPP_HIP_CHECK(hipMemsetAsync(S.d_m_zero.p(), 0, n_zero * sizeof(double), s.stream));
	}
	SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream)); // the tables live on this stack frame
	return true;
}

// the covariances when the reduced system is factored by the sparse block path: the same assembly as a solve (into the
// inner solver's packed values, summed over the ranks), its factorization, then the blocks of S^-1 on the factor's
// pattern (sparse_inverse.hip) instead of a dense inverse -- the cost of a second factorization, and no n^2 memory
template <int DC, int DP>
static void schur_enqueue_marginals_sparse_t(slampp_hip_solver &s, CSchurState &S, const double *A, double *cam_cov,
	double *point_cov)
{
	hipStream_t st = s.stream;
	const int n = S.N;
	const size_t n_in_values = size_t(S.n_in_blocks) * DC * DC;
	double *p_S = S.d_in_buf.p(), *p_r = S.d_in_buf.p() + n_in_values;
	s.Phase_Begin("marginals_assemble");
	SLAMPP_HIP_CHECK(hipMemsetAsync(p_S, 0, (n_in_values + size_t(n)) * sizeof(double), st));
	if(s.b_shard_primary) {
		const int64_t n_work = std::max<int64_t>(S.n_ablocks * DC * DC, n);
		hipLaunchKernelGGL((schur_scatter_A_kernel<DC>), dim3(unsigned((n_work + 255) / 256)), dim3(256), 0, st,
			S.d_ptr.p(), S.d_brow.p(), S.nc, A, S.d_m_zero.p(), p_S, S.Npad, n, S.d_a_dst.p(), p_r);
	}
	s.Phase_End();
	// (the covariance gather below reads W of every observation and C^-1 of every landmark)
	schur_assemble_t<DC, DP>(s, S, A, S.d_m_zero.p(), p_S, S.Npad, S.d_sb_dst.p(), p_r, true);
	if(s.p_allreduce) {
		s.Phase_Begin("allreduce");
		hipLaunchKernelGGL(schur_flag_poison_kernel, dim3(1), dim3(1), 0, st, s.d_flag.p(), p_r);
		if(s.p_allreduce(s.p_allreduce_context, p_S, n_in_values + size_t(n), (void*)st) != 0)
			throw CDeviceError("all-reduce callback failed");
		hipLaunchKernelGGL(schur_flag_check_kernel, dim3(1), dim3(1), 0, st, p_r, s.d_flag.p());
		s.Phase_End();
	}
	s.Phase_Begin("marginals_factor");
	S.p_inner->p_flag_shared = s.d_flag.p();
	S.p_inner->b_leaf_linv_wanted = true; // (the inverse subset multiplies by inv(L_jj) of every column)
	S.p_inner->Enqueue_Sparse(p_S, p_r, true, true); // numeric factorization only
	S.p_inner->Ensure_Leaf_Inverses();
	s.Phase_End();
	s.Phase_Begin("marginals_inverse");
	sparse_inverse_enqueue(*S.p_sinv, S.p_inner->plan, S.p_inner->d_L.p(), S.p_inner->d_Linv.p(), S.d_m_Zs.p(), st);
	s.Phase_End();
	s.Phase_Begin("marginals_gather");
	schur_marginals_sparse_launch(DC, DP, S.nc, S.np, S.d_ptr.p(), S.d_cam_zoff.p(), S.d_pair_ptr.p(), S.d_pair_tab.p(),
		S.d_W.p(), S.d_Cinv.p(), S.d_m_Zs.p(), cam_cov, point_cov, st);
	s.Phase_End();
	SLAMPP_HIP_CHECK(hipGetLastError());
}

template <int DC, int DP>
static void schur_enqueue_marginals_t(slampp_hip_solver &s, CSchurState &S, const double *A, double *cam_cov, double *point_cov)
{
	hipStream_t st = s.stream;
	const int ld = S.Npad, n = S.N;
	// decided as for a solve: with the sparse reduced system the covariances go through the sparse inverse subset
	if(s.p_allreduce && (S.p_union_fn != s.p_allreduce || S.p_union_context != s.p_allreduce_context))
		schur_agree_on_union(s, S);
	if(!S.b_reduced_decided)
		schur_setup_reduced(s, S);
	if(S.b_reduced_sparse && s.n_marginals_dense == 0 && schur_setup_sparse_marginals(s, S)) {
		schur_enqueue_marginals_sparse_t<DC, DP>(s, S, A, cam_cov, point_cov);
		return;
	}
	if(!S.d_m_S.p()) {
		S.d_m_S.Alloc(size_t(ld) * ld);
		S.d_m_Z.Alloc(size_t(ld) * ld);
		S.d_m_invdiag.Alloc(size_t(ld / dense_NB) * dense_NB * dense_NB);
		const size_t n_zero = size_t(n) + size_t(S.np) * DP; // a whole right-hand side of zeros
		S.d_m_zero.Alloc(n_zero);
		SLAMPP_HIP_CHECK(hipMemsetAsync(S.d_m_zero.p(), 0, n_zero * sizeof(double), st));
	}
	double *p_S = S.d_m_S.p();
	s.Phase_Begin("marginals_assemble");
	SLAMPP_HIP_CHECK(hipMemsetAsync(p_S, 0, size_t(ld) * ld * sizeof(double), st));
	dense_prepare_padding(p_S, ld, n, st);
	if(s.b_shard_primary) {
		const int64_t n_work = std::max<int64_t>(S.n_ablocks * DC * DC, n);
		hipLaunchKernelGGL((schur_scatter_A_kernel<DC>), dim3(unsigned((n_work + 255) / 256)), dim3(256), 0, st,
			S.d_ptr.p(), S.d_brow.p(), S.nc, A, S.d_m_zero.p(), p_S, ld, n, (const int64_t*)0, (double*)0);
	}
	s.Phase_End();
	schur_assemble_t<DC, DP>(s, S, A, S.d_m_zero.p(), p_S, ld, (const int64_t*)0, (double*)0, true);
	if(s.p_allreduce) {
		s.Phase_Begin("allreduce");
		hipLaunchKernelGGL(schur_flag_poison_kernel, dim3(1), dim3(1), 0, st, s.d_flag.p(), p_S + n);
		if(s.p_allreduce(s.p_allreduce_context, p_S, size_t(ld) * ld, (void*)st) != 0)
			throw CDeviceError("all-reduce callback failed");
		hipLaunchKernelGGL(schur_flag_check_kernel, dim3(1), dim3(1), 0, st, p_S + n, s.d_flag.p());
		s.Phase_End();
	}
	s.Phase_Begin("marginals_factor");
	dense_cholesky(p_S, ld, n, S.d_m_invdiag.p(), s.d_flag.p(), st);
	s.Phase_End();
	s.Phase_Begin("marginals_inverse");
	dense_inverse_from_factor(p_S, ld, S.d_m_invdiag.p(), S.d_m_Z.p(), st);
	s.Phase_End();
	s.Phase_Begin("marginals_gather");
	schur_marginals_launch(DC, DP, S.nc, S.np, S.d_ptr.p(), S.d_brow.p(), S.d_W.p(), S.d_Cinv.p(), S.d_m_Z.p(), ld,
		cam_cov, point_cov, st);
	s.Phase_End();
	SLAMPP_HIP_CHECK(hipGetLastError());
}

void schur_enqueue_marginals(slampp_hip_solver &s, const double *p_values_dev, double *p_cam_cov_dev, double *p_point_cov_dev)
{
	schur_invalidate_previous(s.p_schur); // C^-1, W (and the packed reduced system) are recomputed from these values
	CSchurState &S = *s.p_schur;
	if(S.DC == 6 && S.DP == 3)
		schur_enqueue_marginals_t<6, 3>(s, S, p_values_dev, p_cam_cov_dev, p_point_cov_dev);
	else if(S.DC == 7 && S.DP == 3)
		schur_enqueue_marginals_t<7, 3>(s, S, p_values_dev, p_cam_cov_dev, p_point_cov_dev);
	else
		schur_enqueue_marginals_t<3, 2>(s, S, p_values_dev, p_cam_cov_dev, p_point_cov_dev);
}

// the reference's Solve_PosDef_Blocky_MarginalPoses (LinearSolver_Schur.h:1956-2143): the landmarks' block of the
// system alone, dl = C^-1 eta_l, the pose part of the solution zeroed
void schur_enqueue_marginal_poses(slampp_hip_solver &s, const double *p_values_dev, double *p_rhs_dev)
{
	schur_invalidate_previous(s.p_schur);
	CSchurState &S = *s.p_schur;
	if(S.DC == 6 && S.DP == 3)
		schur_enqueue_marginal_t<6, 3>(s, S, p_values_dev, p_rhs_dev);
	else if(S.DC == 7 && S.DP == 3)
		schur_enqueue_marginal_t<7, 3>(s, S, p_values_dev, p_rhs_dev);
	else
		schur_enqueue_marginal_t<3, 2>(s, S, p_values_dev, p_rhs_dev);
}

void schur_enqueue(slampp_hip_solver &s, const double *p_values_dev, double *p_rhs_dev)
{
	CSchurState &S = *s.p_schur;
	if(S.DC == 6 && S.DP == 3)
		schur_enqueue_t<6, 3>(s, S, p_values_dev, p_rhs_dev);
	else if(S.DC == 7 && S.DP == 3)
		schur_enqueue_t<7, 3>(s, S, p_values_dev, p_rhs_dev);
	else
		schur_enqueue_t<3, 2>(s, S, p_values_dev, p_rhs_dev);
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(schur) // (the handle's bring-up thread loads this unit's code object: capi.hip)
