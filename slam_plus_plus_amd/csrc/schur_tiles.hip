// schur_tiles.hip -- landmark-major assembly of the reduced camera system S = A - U C^-1 U^T, r = x - U C^-1 l
// (steps 4-9 of CLinearSolver_Schur::Solve_PosDef_Blocky, /root/reference/include/slam/LinearSolver_Schur.h:1730-1830:
// InverseOf_BlockDiag, the two MultiplyToWith_FBS products, AddTo_FBS, PreMultiply_Add).
//
// The contribution lists of schur.hip read 288 bytes per (landmark, camera pair): a landmark seen by k cameras is read
// k (k + 1) / 2 times -- 1.44 GB at C4 for 0.58 GB of blocks, and the list kernel ran at the HBM traffic those re-reads
// generate.  Here every landmark is read once, by one of two kernels the host analysis (schur_tiles_build) picks:
//   runs   landmarks seen by exactly the same cameras (at least four of them): their products all land in the same blocks
//          of S, so a wave keeps the sums in matrix-core accumulators for a whole piece of the run (schur_run_kernel);
//   tiles  the other landmarks, sorted by their first two cameras and cut where they touch more than SCHUR_TILE_SLOTS
//          distinct blocks of S: a wave keeps a tile's blocks (and the right-hand sides of its cameras) in LDS and adds
//          every landmark's k (k + 1) / 2 products U_b W_a^T into them, one lane per (camera pair, row) (schur_tile_kernel).
// Either way the landmark's record [U_1 .. U_k | C | l] is loaded, C inverted, W = U C^-1 formed on the fly; the sums go
// to a partial array, and a last kernel adds up the partial blocks of every block of S in list order: no atomics on
// memory, the sum order is fixed by the analysis (bit-reproducible, like the lists).
// Landmarks with more cameras than a tile has room for that are in no run, and tiles whose landmarks share too little
// (fewer than three contributions per block: random visibility), stay with the contribution lists.
// SLAMPP_HIP_DEV_TILE_POINTS / SLAMPP_HIP_DEV_RUN_PIECE (environment, with SLAMPP_HIP_DEV=1: plan.h) are development knobs for the landmarks per tile / per run piece.
#include "schur_tiles.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>
#include <functional>
#include <chrono>

namespace slampp {

#include "schur_device.inl"

constexpr int tile_max_k(int n_cap)
{
	int k = 1;
	while((k + 1) * (k + 2) / 2 <= n_cap)
		++ k;
	return k;
}

__device__ __forceinline__ int64_t readlane64(int64_t v, int n_lane)
{
	return (int64_t(__builtin_amdgcn_readlane(int(v >> 32), n_lane)) << 32) | uint32_t(__builtin_amdgcn_readlane(int(v), n_lane));
}

__device__ __forceinline__ double readlane_f64(double v, int n_lane)
{
	return __longlong_as_double(readlane64(__double_as_longlong(v), n_lane));
}

__device__ __forceinline__ void wave_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int DC, int DP, int CAP, int NL>
__global__ void __launch_bounds__(64)
schur_tile_kernel(const int32_t *__restrict__ tile_ptr, const int32_t *__restrict__ tile_lm,
	const int64_t *__restrict__ tile_slot_ptr, const int64_t *__restrict__ pair_ptr, const uint8_t *__restrict__ lm_slot,
	const int64_t *__restrict__ ptr, int64_t nc, int64_t ubase, const double *__restrict__ A, const double *__restrict__ eta,
	int n, double *Cinv, double *W, int b_store, double *P, double *R, int *p_flag, int n_max_slots)
{
	enum { BLK = DC * DP, BB = DC * DC, MAXK = tile_max_k(CAP), GROUP = 4,
		GCAP = (MAXK * BLK > 320)? (MAXK * BLK + 63) / 64 * 64 : 320 };
	// LDS sized by the launch for the largest tile (a tile of ten blocks needs 3 kB, one of 64 blocks 21 kB: the waves a
	// CU holds hide the latency of the landmark loads)
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];
	double *s_u = s_dyn, *s_w = s_u + GCAP;
	double *s_acc = s_w + GCAP, *s_r = s_acc + n_max_slots * BB;
	uint8_t *s_slot = (uint8_t*)(s_r + n_max_slots * DC), *s_tri = s_slot + 64;
	const int lane = threadIdx.x;
	const int64_t tile = blockIdx.x;
	const int lm0 = tile_ptr[tile], lm1 = tile_ptr[tile + 1];
	const int64_t slot0 = tile_slot_ptr[tile];
	const int n_slots = int(tile_slot_ptr[tile + 1] - slot0);
	for(int e = lane; e < n_slots * BB; e += 64)
		s_acc[e] = 0;
	for(int e = lane; e < n_slots * DC; e += 64)
		s_r[e] = 0;
	{
		// pair j = b (b + 1) / 2 + a of a landmark's cameras a <= b
		int b = 0;
		while((b + 1) * (b + 2) / 2 <= lane)
			++ b;
		s_tri[lane] = uint8_t((lane - b * (b + 1) / 2) | (b << 4));
	}
	const int64_t ptr_nc = ptr[nc];
	bool b_bad = false;
	for(int base = lm0; base < lm1; base += 64) {
		const int n_chunk = min(64, lm1 - base);
		// lane L prepares landmark base + L: where its record sits, C^-1, l
		int my_k = 0;
		int64_t my_rec = 0, my_pp = 0, my_o0 = 0;
		double ci[DP * DP], l[DP];
		#pragma unroll
		for(int i = 0; i < DP * DP; ++ i)
			ci[i] = 0;
		#pragma unroll
		for(int i = 0; i < DP; ++ i)
			l[i] = 0;
		if(lane < n_chunk) {
			const int64_t pt = tile_lm[base + lane];
			const int64_t k0 = ptr[nc + pt], k1 = ptr[nc + pt + 1];
			my_k = int(k1 - k0 - 1);
			my_o0 = k0 - ptr_nc - pt;
			my_rec = ubase + my_o0 * BLK + pt * (DP * DP);
			my_pp = pair_ptr[pt];
			double c[DP * DP];
			const double *C = A + my_rec + int64_t(my_k) * BLK;
			#pragma unroll
			for(int i = 0; i < DP * DP; ++ i)
				c[i] = C[i];
			if(!spd_inverse<DP>(c, ci))
				b_bad = true;
			#pragma unroll
			for(int i = 0; i < DP; ++ i)
				l[i] = eta[n + pt * DP + i];
			if(b_store) {
				#pragma unroll
				for(int i = 0; i < DP * DP; ++ i)
					Cinv[pt * (DP * DP) + i] = ci[i];
			}
		}
		// Landmarks are taken GROUP at a time: all their blocks are requested together, and the next group's while this
		// one is multiplied (a landmark's products take a fraction of a memory latency and a CU holds ten of these waves:
		// with one landmark in flight per wave the kernel ran at 1.5 TB/s).  The requests are straight-line code -- every
		// lane of every slot loads, lanes and slots with nothing to fetch re-read the group's first address -- because the
		// compiler puts a full s_waitcnt in front of a load it finds behind a branch.
		double v[GROUP][NL];
		uint8_t n_my_slot;
		int n_group, n_next = 0; // landmarks of the group in flight; first landmark after it
		auto Request = [&](int n_first) {
			const int i0 = min(n_first, n_chunk - 1);
			const int64_t rec0 = readlane64(my_rec, i0);
			int n_doubles = 0, n_group_pairs = 0;
			n_group = 0;
			#pragma unroll
			for(int g = 0; g < GROUP; ++ g) {
				const int ig = min(n_first + g, n_chunk - 1);
				const int k = __builtin_amdgcn_readlane(my_k, ig);
				const int64_t rec = readlane64(my_rec, ig);
				const bool b_in = n_group == g && n_first + g < n_chunk &&
					(g == 0 || (n_doubles + k * BLK <= GCAP && n_group_pairs + k * (k + 1) / 2 <= 64));
				#pragma unroll
				for(int m = 0; m < NL; ++ m)
					v[g][m] = A[(b_in && lane + 64 * m < k * BLK)? rec + lane + 64 * m : rec0];
				n_doubles += b_in? k * BLK : 0;
				n_group_pairs += b_in? k * (k + 1) / 2 : 0;
				n_group += b_in;
			}
			// the slots of consecutive landmarks of a tile are consecutive
			n_my_slot = lm_slot[readlane64(my_pp, i0) + ((lane < n_group_pairs)? lane : 0)];
			n_next = n_first + n_group;
		};
		Request(0);
		for(int i = 0; i < n_chunk;) {
			const int n_cur_group = n_group;
			{
				int n_off = 0, n_slot_num = 0;
				#pragma unroll
				for(int g = 0; g < GROUP; ++ g) {
					const int k = (g < n_cur_group)? __builtin_amdgcn_readlane(my_k, min(i + g, n_chunk - 1)) : 0;
					#pragma unroll
					for(int m = 0; m < NL; ++ m) {
						if(lane + 64 * m < k * BLK)
							s_u[n_off + lane + 64 * m] = v[g][m];
					}
					n_off += k * BLK;
					n_slot_num += k * (k + 1) / 2;
				}
				if(lane < n_slot_num)
					s_slot[lane] = n_my_slot;
			}
			wave_lds_fence();
			Request(n_next); // (past the chunk's end: an empty group, its loads re-read one address)
			int n_off = 0, n_pair_off = 0;
			for(int g = 0; g < n_cur_group; ++ g) {
				const int k = __builtin_amdgcn_readlane(my_k, i + g);
				const int n_pairs = k * (k + 1) / 2;
				const double *s_ug = s_u + n_off;
				double *s_wg = s_w + n_off;
				const uint8_t *s_slotg = s_slot + n_pair_off;
				double c_i[DP * DP], l_i[DP];
				#pragma unroll
				for(int t = 0; t < DP * DP; ++ t)
					c_i[t] = readlane_f64(ci[t], i + g);
				#pragma unroll
				for(int t = 0; t < DP; ++ t)
					l_i[t] = readlane_f64(l[t], i + g);
				const int64_t o0 = readlane64(my_o0, i + g);
				// W_o = U_o C^-1 and the right-hand side's share W_o l: one lane per (observation, row) -- ten observations of a
				// seven-dimensional camera are 70 rows: a second round (until round 4 the rows past 64 were left out: a wrong S
				// for Sim(3) landmarks seen by exactly ten cameras that went through a tile)
				for(int n_row0 = 0; n_row0 < k * DC; n_row0 += 64) {
					const int n_row = n_row0 + lane;
					if(n_row >= k * DC)
						continue;
					const int o = n_row / DC, r = n_row - o * DC;
					double u[DP], rv = 0;
					#pragma unroll
					for(int t = 0; t < DP; ++ t)
						u[t] = s_ug[o * BLK + r + t * DC];
					#pragma unroll
					for(int q = 0; q < DP; ++ q) {
						double w = 0;
						#pragma unroll
						for(int t = 0; t < DP; ++ t)
							w += u[t] * c_i[t + q * DP];
						s_wg[o * BLK + r + q * DC] = w;
						if(W)
							W[(o0 + o) * BLK + r + q * DC] = w;
						rv += w * l_i[q];
					}
					const int n_slot = s_slotg[o * (o + 3) / 2]; // pair (o, o)
					atomicAdd(&s_r[n_slot * DC + r], rv); // (cameras are distinct inside a landmark: no two lanes meet)
				}
				wave_lds_fence();
				// the products U_b W_a^T: one lane per (pair, row of the block)
				for(int n_first = 0; n_first < n_pairs * DC; n_first += 64) {
					const int n_task = n_first + lane;
					if(n_task < n_pairs * DC) {
						const int j = n_task / DC, r = n_task - j * DC;
						const int n_tri = s_tri[j], a = n_tri & 15, b = n_tri >> 4;
						double u[DP];
						#pragma unroll
						for(int t = 0; t < DP; ++ t)
							u[t] = s_ug[b * BLK + r + t * DC];
						double *p_acc = &s_acc[int(s_slotg[j]) * BB + r];
						const double *p_w = &s_wg[a * BLK];
						#pragma unroll
						for(int q = 0; q < DC; ++ q) {
							double f = 0;
							#pragma unroll
							for(int t = 0; t < DP; ++ t)
								f += u[t] * p_w[q + t * DC];
							atomicAdd(p_acc + q * DC, f);
						}
					}
				}
				n_off += k * BLK;
				n_pair_off += n_pairs;
			}
			wave_lds_fence(); // the next group overwrites the operands
			i += n_cur_group;
		}
	}
	if(b_bad)
		atomicOr(p_flag, 1);
	for(int e = lane; e < n_slots * BB; e += 64)
		P[slot0 * BB + e] = s_acc[e];
	for(int e = lane; e < n_slots * DC; e += 64)
		R[slot0 * DC + e] = s_r[e];
}

// ---- runs: landmarks seen by exactly the same cameras ----
// A run's products all land in the same k (k + 1) / 2 blocks of S, so they never leave the registers: for every landmark
// the wave multiplies [U_1; ..; U_k] (6k x 3) by [W_1; ..; W_k]^T on the matrix cores (v_mfma_f64_16x16x4, one landmark =
// one K step, the fourth k idle; the tiles above the diagonal are skipped) and only the sums over the run's landmarks
// are written, as partial blocks.  Observations are handled OB = 64 / DC at a time: a job is (piece of a run of at most
// 64 landmarks, block of OB row observations, block of OB column observations); up to OB cameras that is one job per
// piece, a landmark seen by 30 cameras takes six.  The right-hand side's share U C^-1 l is one FMA per landmark and
// row tile on the vector unit, reduced over the three k lanes at the end (diagonal jobs).
// The blocks of GROUP landmarks are fetched with coalesced loads, the next group's while this one is multiplied,
// and go through LDS into the fragment layout (lane = (row or column, k)).  Loading the fragments straight from memory
// -- eight 8-byte gathers per landmark over its 576-byte record -- cost 176 L1 accesses per landmark and ran at
// 216 us for C4; coalesced it is ten.
// b_quad (round 5): FOUR landmarks per matrix-core step instead of one.  The K dimension of v_mfma_f64_16x16x4 held the
// three coordinates of ONE landmark (and an idle fourth slot): per landmark the wave read C^-1, formed W's column and issued
// a full set of tiles, 131 vector instructions beside 10 matrix ones.  With K = four landmarks -- lane (m16, kk) works on
// landmark kk of a quad, and the three coordinates are three rounds of tiles, S += sum_j U(:, j) W(:, j)^T -- no slot is idle
// (9 matrix instructions per tile set and quad instead of 12), the lanes of different kk no longer read the same U values
// four times, and C^-1, C^-1 l and the masks are fetched once per quad and lane.
template <int DC, int DP, int NT, bool b_diag, bool b_prefix, bool b_quad = false> // b_prefix: some landmarks of the job end before the run's list does
__global__ void __launch_bounds__(64)
schur_run_kernel(const TRunJob *__restrict__ jobs, const int32_t *__restrict__ run_lm, const int64_t *__restrict__ run_rec,
	const int32_t *__restrict__ run_k, int64_t ubase, const double *__restrict__ A, const double *__restrict__ eta, int n, double *Cinv, double *W, int b_store,
	double *P, double *R, int *p_flag)
{
	enum { BLK = DC * DP, BB = DC * DC, OB = 64 / DC, OBT = (NT * 16 / DC < OB)? NT * 16 / DC : OB, // observations per block here
		SEG = OBT * BLK, NLD = (SEG + 63) / 64,
		GROUP = (NT <= 2)? 8 : (b_diag? 4 : 2) }; // (off-diagonal jobs of three or four tiles a side keep 16 accumulator tiles: with four landmarks in
	// flight the requested values were parked in accumulator registers one by one, a wait behind every load -- round 4)
	typedef double v4f64 __attribute__((ext_vector_type(4)));
	__shared__ double s_ci[64 * DP * DP];
	__shared__ double s_z[64 * DP];
	__shared__ double s_u[GROUP][b_diag? SEG : 2 * SEG]; // [blocks of the row observations | of the column observations, if they are others]
	__shared__ int s_kown[b_quad? 64 : 1];       // b_quad: every landmark's own number of observations (0 past the piece's end) ...
	__shared__ int64_t s_o0[b_quad? 64 : 1];     // ... and its first observation (where its W goes)
	static_assert(!b_quad || GROUP % 4 == 0, "quads of landmarks");
	const int lane = threadIdx.x, kk = lane >> 4, m16 = lane & 15;
	const TRunJob job = jobs[blockIdx.x];
	const int k = job.n_k, n_rb = job.n_rb, n_cb = job.n_cb, n_points_all = job.n_points;
	int n_points = min(n_points_all, 64), n_sub_first = job.n_first; // (round 5) a job takes its landmarks 64 at a time -- lane L prepares landmark L of a sub-piece -- and keeps its sums across sub-pieces: longer pieces, fewer partial blocks
	const int n_len_a = min(OB, k - n_rb * OB) * BLK, n_len_b = min(OB, k - n_cb * OB) * BLK; // doubles of the two segments
	// what this lane feeds the matrix cores: element (row, kk) of the U rows, element (kk, column) of the W columns
	int n_off_a[NT], n_off_b[NT], n_obs_a[NT], n_obs_b[NT];
	bool b_a[NT], b_b[NT];
	#pragma unroll
	for(int t = 0; t < NT; ++ t) {
		const int n_line = t * 16 + m16, o = n_line / DC, e = n_line - o * DC;
		b_a[t] = o < OB && n_rb * OB + o < k && (b_quad || kk < DP);
		n_off_a[t] = b_a[t]? o * BLK + (b_quad? 0 : kk * DC) + e : 0; // (b_quad: element (row, 0); the lane reads all three columns)
		b_b[t] = o < OB && n_cb * OB + o < k && (b_quad || kk < DP);
		n_off_b[t] = (b_b[t]? o * BLK + e : 0) + (b_diag? 0 : SEG);
		n_obs_a[t] = n_rb * OB + o;
		n_obs_b[t] = n_cb * OB + o;
	}
	const int kc = (kk < DP)? kk : 0;
	// lane L prepares landmark L of the piece: where its blocks are (the host wrote that down), C^-1 and C^-1 l
	// (round 4: a run may also hold landmarks whose camera list is a PREFIX of the run's -- tracks born at the same camera and
	// lost at different ones --: my_k is the landmark's own number of observations, what lies beyond it reads as zero)
	int64_t my_rec = 0, my_pt = 0;
	int my_k = k;
	v4f64 acc[NT][NT];
	double racc[NT];
	#pragma unroll
	for(int rt = 0; rt < NT; ++ rt) {
		racc[rt] = 0;
		#pragma unroll
		for(int ct = 0; ct < NT; ++ ct)
			acc[rt][ct] = v4f64{0, 0, 0, 0};
	}
	const bool b_store_w = W != 0 && b_diag;
	// requests of a group: straight-line code (lanes past a segment's end and slots past the piece's end re-read a valid
	// address), so that all of them are in flight together
	double va[GROUP][NLD], vb[b_diag? 1 : GROUP][NLD];
	const int64_t n_seg_a = int64_t(n_rb) * OB * BLK, n_seg_b = int64_t(n_cb) * OB * BLK;
	auto Request = [&](int n_first) {
		#pragma unroll
		for(int g = 0; g < GROUP; ++ g) {
			// (b_prefix: a landmark may end before the run's list does.  Its loads stay inside its own record -- U blocks and the C
			// block behind them -- and what they bring from beyond its last observation is never multiplied: the operands are
			// masked where they are read out of LDS.  A select on a value just requested, tried first, made the compiler wait
			// for every load in turn: 24 round trips per group instead of one.)
			const int n_lm = min(n_first + g, n_points - 1);
			const double *p_rec = A + readlane64(my_rec, n_lm);
			const int n_last = b_prefix? __builtin_amdgcn_readlane(my_k, n_lm) * BLK + DP * DP - 1 : 0x7fffffff;
			#pragma unroll
			for(int m = 0; m < NLD; ++ m) {
				va[g][m] = p_rec[min(int(n_seg_a) + ((lane + 64 * m < n_len_a)? lane + 64 * m : 0), n_last)];
				if(!b_diag)
					vb[g][m] = p_rec[min(int(n_seg_b) + ((lane + 64 * m < n_len_b)? lane + 64 * m : 0), n_last)];
			}
		}
	};
	bool b_bad = false;
	for(int n_sub = 0; n_sub < n_points_all; n_sub += 64) {
	n_points = min(n_points_all - n_sub, 64);
	n_sub_first = job.n_first + n_sub;
	my_rec = 0; my_pt = 0; my_k = k;
	if(lane < n_points) {
		my_pt = run_lm[n_sub_first + lane];
		my_rec = run_rec[n_sub_first + lane];
		my_k = run_k[n_sub_first + lane];
	}
	if(n_sub)
		wave_lds_fence(); // the last group of the sub-piece before has been read: s_ci, s_z (and the quads' tables) can go
	Request(0); // (flies while the landmark blocks are inverted)
	if(lane < n_points) {
		const int64_t pt = my_pt;
		double c[DP * DP], ci[DP * DP];
		const double *C = A + my_rec + int64_t(my_k) * BLK;
		#pragma unroll
		for(int i = 0; i < DP * DP; ++ i)
			c[i] = C[i];
		if(!spd_inverse<DP>(c, ci))
			b_bad = true;
		#pragma unroll
		for(int i = 0; i < DP * DP; ++ i)
			s_ci[lane * (DP * DP) + i] = ci[i];
		#pragma unroll
		for(int t = 0; t < DP; ++ t) {
			double z = 0;
			#pragma unroll
			for(int i = 0; i < DP; ++ i)
				z += ci[t + i * DP] * eta[n + pt * DP + i];
			s_z[lane * DP + t] = z; // C^-1 l
		}
		if(b_store && n_rb == 0 && n_cb == 0) {
			#pragma unroll
			for(int i = 0; i < DP * DP; ++ i)
				Cinv[pt * (DP * DP) + i] = ci[i];
		}
	}
	if constexpr(b_quad) {
		s_kown[lane] = (lane < n_points)? my_k : 0;
		s_o0[lane] = (lane < n_points)? (my_rec - ubase - my_pt * (DP * DP)) / BLK : 0;
	}
	for(int p0 = 0; p0 < n_points; p0 += GROUP) {
		wave_lds_fence(); // the previous group's fragments have been read (and, the first time, s_ci / s_z written)
		#pragma unroll
		for(int g = 0; g < GROUP; ++ g) {
			#pragma unroll
			for(int m = 0; m < NLD; ++ m) {
				if(lane + 64 * m < n_len_a)
					s_u[g][lane + 64 * m] = va[g][m];
				if(!b_diag && lane + 64 * m < n_len_b)
					s_u[g][(b_diag? 0 : SEG) + lane + 64 * m] = vb[b_diag? 0 : g][m];
			}
		}
		wave_lds_fence();
		Request(p0 + GROUP); // (past the end: re-reads the last landmark)
		if constexpr(b_quad) {
			#pragma unroll
			for(int q0 = 0; q0 < GROUP; q0 += 4) {
				if(p0 + q0 >= n_points) // (wave-uniform: the whole quad lies past the piece's end)
					break;
				const int g = q0 + kk;                        // this lane's landmark of the quad
				const int p = min(p0 + g, n_points - 1);
				const int n_k_lane = (p0 + g < n_points)? (b_prefix? s_kown[p] : k) : 0; // observations of this lane's landmark (0: no landmark)
				double ci[DP * DP], z[DP];
				#pragma unroll
				for(int i = 0; i < DP * DP; ++ i)
					ci[i] = s_ci[p * (DP * DP) + i];
				#pragma unroll
				for(int j = 0; j < DP; ++ j)
					z[j] = s_z[p * DP + j];
				const double *su = &s_u[0][0] + g * (b_diag? SEG : 2 * SEG);
				double ua[NT][DP], wb[NT][DP];
				bool b_wb[NT];
				#pragma unroll
				for(int t = 0; t < NT; ++ t) {
					const bool b_row = b_a[t] && n_obs_a[t] < n_k_lane;
					double u[DP], v[DP];
					#pragma unroll
					for(int j = 0; j < DP; ++ j) {
						u[j] = su[n_off_a[t] + j * DC];
						ua[t][j] = b_row? u[j] : 0.0;
						racc[t] += ua[t][j] * z[j];
					}
					#pragma unroll
					for(int j = 0; j < DP; ++ j)
						v[j] = b_diag? u[j] : su[n_off_b[t] + j * DC];
					b_wb[t] = b_b[t] && n_obs_b[t] < n_k_lane;
					#pragma unroll
					for(int jj = 0; jj < DP; ++ jj) {
						double w = 0;
						#pragma unroll
						for(int i = 0; i < DP; ++ i)
							w += v[i] * ci[i + jj * DP]; // W(q, jj) = sum_i U(q, i) C^-1(i, jj)
						wb[t][jj] = b_wb[t]? w : 0.0;
					}
				}
				// (b_prefix: the longest landmark of the quad says how many row tiles are not all zeros)
				int n_k_quad = k;
				if(b_prefix) {
					n_k_quad = 0;
					#pragma unroll
					for(int i = 0; i < 4; ++ i)
						n_k_quad = max(n_k_quad, (p0 + q0 + i < n_points)? __builtin_amdgcn_readlane(my_k, min(p0 + q0 + i, n_points - 1)) : 0);
				}
				const int n_rt_own = b_prefix? (min(n_k_quad - n_rb * OB, int(OB)) * DC + 15) / 16 : NT;
				#pragma unroll
				for(int j = 0; j < DP; ++ j) {
					#pragma unroll
					for(int rt = 0; rt < NT; ++ rt) {
						if(b_prefix && rt >= n_rt_own) // (wave-uniform)
							continue;
						#pragma unroll
						for(int ct = 0; ct < NT; ++ ct) {
							if(ct <= rt || !b_diag)
								acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[rt][j], wb[ct][j], acc[rt][ct], 0, 0, 0);
						}
					}
				}
				if(b_store_w) {
					const int64_t o0 = s_o0[p];
					#pragma unroll
					for(int t = 0; t < NT; ++ t) {
						if(b_wb[t]) {
							const int n_line = t * 16 + m16, o = n_line / DC, q = n_line - o * DC;
							#pragma unroll
							for(int jj = 0; jj < DP; ++ jj)
								W[(o0 + n_obs_b[t]) * BLK + q + jj * DC] = wb[t][jj];
						}
					}
				}
			}
		} else {
			#pragma unroll
			for(int g = 0; g < GROUP; ++ g) {
				const int p = min(p0 + g, n_points - 1);
				const bool b_live = p0 + g < n_points;
				const int n_k_own = b_prefix? __builtin_amdgcn_readlane(my_k, p) : k; // observations of this landmark
				double cik[DP], wb[NT], ua[NT];
				#pragma unroll
				for(int j = 0; j < DP; ++ j)
					cik[j] = s_ci[p * (DP * DP) + j + kc * DP];
				const double z = s_z[p * DP + kc];
				#pragma unroll
				for(int t = 0; t < NT; ++ t) {
					const double u = s_u[g][n_off_a[t]];
					ua[t] = (b_a[t] && b_live && (!b_prefix || n_obs_a[t] < n_k_own))? u : 0.0;
					racc[t] += ua[t] * z;
					double w = 0;
					#pragma unroll
					for(int j = 0; j < DP; ++ j)
						w += s_u[g][n_off_b[t] + j * DC] * cik[j]; // W(q, kk) = sum_j U(q, j) C^-1(j, kk)
					wb[t] = (b_b[t] && (!b_prefix || n_obs_b[t] < n_k_own))? w : 0.0;
				}
				// (b_prefix: a landmark that ends inside the row block fills only the first tiles of rows -- and, on the diagonal, of
				// columns --: the others would multiply zeros, and these kernels spend 27 - 36 % of the fp64 matrix peak as it is)
				const int n_rt_own = b_prefix? (min(n_k_own - n_rb * OB, int(OB)) * DC + 15) / 16 : NT;
				#pragma unroll
				for(int rt = 0; rt < NT; ++ rt) {
					if(b_prefix && rt >= n_rt_own) // (wave-uniform)
						continue;
					#pragma unroll
					for(int ct = 0; ct < NT; ++ ct) {
						if(ct <= rt || !b_diag)
							acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[rt], wb[ct], acc[rt][ct], 0, 0, 0);
					}
				}
				if(b_store_w && b_live) {
					const int64_t o0 = (readlane64(my_rec, p) - ubase - readlane64(my_pt, p) * (DP * DP)) / BLK; // first observation of the landmark
					#pragma unroll
					for(int t = 0; t < NT; ++ t) {
						if(b_b[t] && n_obs_b[t] < n_k_own) {
							const int n_line = t * 16 + m16, o = n_line / DC, q = n_line - o * DC;
							W[(o0 + n_obs_b[t]) * BLK + q + kk * DC] = wb[t];
						}
					}
				}
			}
		}
	}
	} // (sub-pieces)
	if(b_bad)
		atomicOr(p_flag, 1);
	// lane l holds the elements ((l >> 4) + 4 reg, l & 15) of every 16 x 16 tile
	const int n_kb_c = min(OB, k - n_cb * OB); // observations of the column block
	#pragma unroll
	for(int rt = 0; rt < NT; ++ rt) {
		#pragma unroll
		for(int reg = 0; reg < 4; ++ reg) {
			const int n_row = rt * 16 + 4 * reg + kk, ob = n_row / DC, r = n_row - ob * DC;
			if(ob >= OB || n_rb * OB + ob >= k)
				continue;
			#pragma unroll
			for(int ct = 0; ct < NT; ++ ct) {
				const int n_col = ct * 16 + m16, oa = n_col / DC, q = n_col - oa * DC;
				if(oa >= OB || n_cb * OB + oa >= k || (b_diag && oa > ob))
					continue;
				const int64_t n_slot = job.n_pbase + (b_diag? ob * (ob + 1) / 2 + oa : ob * n_kb_c + oa);
				P[n_slot * BB + r + q * DC] = acc[rt][ct][reg];
			}
		}
	}
	if(b_diag) { // the right-hand side's share of row (rt, m16): the sum over the k lanes
		#pragma unroll
		for(int rt = 0; rt < NT; ++ rt) {
			double f = racc[rt];
			f += __shfl_down(f, 32);
			f += __shfl_down(f, 16);
			const int n_row = rt * 16 + m16, ob = n_row / DC, r = n_row - ob * DC;
			if(kk == 0 && ob < OB && n_rb * OB + ob < k)
				R[(job.n_pbase + ob * (ob + 3) / 2) * DC + r] = f;
		}
	}
}

// one wave per block of S that has partial blocks: S -= their sum, in list order; the diagonal blocks also bring the
// partial right-hand sides of their camera (lanes DC^2 .. DC^2 + DC - 1)
template <int DC>
__global__ void __launch_bounds__(64)
schur_tile_reduce_kernel(const int64_t *__restrict__ rb_ptr, const int32_t *__restrict__ rb_part,
	const int32_t *__restrict__ rb_sb, const int32_t *__restrict__ sb_row, const int32_t *__restrict__ sb_col,
	const double *__restrict__ P, const double *__restrict__ R, double *S, int ld, const int64_t *__restrict__ p_dst, double *p_r)
{
	enum { BB = DC * DC };
	const int64_t i = blockIdx.x;
	const int lane = threadIdx.x;
	const int64_t sb = rb_sb[i];
	const int64_t n_row = sb_row[sb], n_col = sb_col[sb];
	if constexpr(DC == 6) {
		// (round 5) 16 bytes a lane: 18 lanes hold a partial block, so one request brings THREE of them (lanes 0 .. 53) and the three
		// partial right-hand sides of a diagonal block beside them (lanes 54 .. 62, 3 lanes each): a third of the requests of the
		// one-element-a-lane form below, and three times the bytes in flight.  Every lane group adds up every third partial block
		// of the list, in list order; the three sums meet at the end: still a fixed order.
		typedef double v2f64 __attribute__((ext_vector_type(2)));
		const bool b_blk = lane < 54, b_rhs_lane = lane >= 54 && lane < 63 && n_row == n_col;
		const int g = b_blk? lane / 18 : (lane - 54) / 3, q = b_blk? lane - 18 * g : (lane - 54) - 3 * g; // partial block of the request, pair of elements
		const double *p_src = b_blk? P + 2 * q : R + 2 * ((lane < 63)? q : 0);
		const int n_stride = b_blk? BB : DC;
		v2f64 f = {0, 0};
		for(int64_t e0 = rb_ptr[i], e1 = rb_ptr[i + 1]; e0 < e1; e0 += 64) {
			const int n_here = int(min(int64_t(64), e1 - e0));
			const int n_my = rb_part[e0 + min(lane, n_here - 1)];
			for(int j0 = 0; j0 < n_here; j0 += 24) {
				v2f64 v[8];
				#pragma unroll
				for(int u = 0; u < 8; ++ u) {
					const int j = j0 + 3 * u;
					const int i0 = __builtin_amdgcn_readlane(n_my, min(j, n_here - 1)), i1 = __builtin_amdgcn_readlane(n_my, min(j + 1, n_here - 1)),
						i2 = __builtin_amdgcn_readlane(n_my, min(j + 2, n_here - 1));
					const int n_idx = (g == 0)? i0 : (g == 1)? i1 : i2;
					v[u] = *reinterpret_cast<const v2f64*>(p_src + int64_t(n_idx) * n_stride);
				}
				#pragma unroll
				for(int u = 0; u < 8; ++ u) {
					if(j0 + 3 * u + g < n_here)
						f += v[u];
				}
			}
		}
		// lane q of the first group collects the other two (the right-hand side lanes likewise, three lanes apart)
		const int n_from1 = b_blk? lane + 18 : lane + 3, n_from2 = b_blk? lane + 36 : lane + 6;
		const double f1x = __shfl(f.x, n_from1 & 63), f1y = __shfl(f.y, n_from1 & 63), f2x = __shfl(f.x, n_from2 & 63), f2y = __shfl(f.y, n_from2 & 63);
		f.x = (f.x + f1x) + f2x;
		f.y = (f.y + f1y) + f2y;
		if(lane < 18) {
			#pragma unroll
			for(int h = 0; h < 2; ++ h) {
				const int n_el = 2 * lane + h, r = n_el % DC, qc = n_el / DC;
				const size_t idx = p_dst? size_t(p_dst[sb]) + qc + r * DC : size_t(n_row * DC + r) + size_t(n_col * DC + qc) * ld;
				S[idx] -= h? f.y : f.x;
			}
		} else if(b_rhs_lane && lane < 57) {
			#pragma unroll
			for(int h = 0; h < 2; ++ h) {
				const int64_t c = n_row * DC + 2 * (lane - 54) + h;
				if(p_r)
					p_r[c] -= h? f.y : f.x;
				else
					S[size_t(ld - 1) + size_t(c) * ld] -= h? f.y : f.x;
			}
		}
		return;
	}
	const bool b_block = lane < BB, b_rhs = lane >= BB && lane < BB + DC && n_row == n_col;
	// the list's indices 64 at a time with one coalesced load, then eight partial blocks in flight at once (one after
	// the other a block with 36 partials -- Venice-like visibility -- was 36 dependent round trips: 156 us for that kernel);
	// the sum is still taken in list order
	const int n_el = b_block? lane : (b_rhs? lane - BB : 0); // (the other lanes re-read a valid element)
	const double *p_src = b_block? P : R;
	const int n_stride = b_block? BB : DC;
	double f = 0;
	for(int64_t e0 = rb_ptr[i], e1 = rb_ptr[i + 1]; e0 < e1; e0 += 64) {
		const int n_here = int(min(int64_t(64), e1 - e0));
		const int n_my = rb_part[e0 + min(lane, n_here - 1)];
		for(int j0 = 0; j0 < n_here; j0 += 8) {
			double v[8];
			#pragma unroll
			for(int u = 0; u < 8; ++ u)
				v[u] = p_src[int64_t(__builtin_amdgcn_readlane(n_my, min(j0 + u, n_here - 1))) * n_stride + n_el];
			#pragma unroll
			for(int u = 0; u < 8; ++ u)
				f += (j0 + u < n_here)? v[u] : 0.0;
		}
	}
	if(b_block) {
		const int r = lane % DC, q = lane / DC;
		const size_t idx = p_dst? size_t(p_dst[sb]) + q + r * DC : size_t(n_row * DC + r) + size_t(n_col * DC + q) * ld;
		S[idx] -= f;
	} else if(b_rhs) {
		const int64_t c = n_row * DC + (lane - BB);
		if(p_r)
			p_r[c] -= f;
		else
			S[size_t(ld - 1) + size_t(c) * ld] -= f;
	}
}

template <int DC, int DP>
static void tiles_enqueue_t(const CSchurTiles &T, const int64_t *ptr, int64_t nc, int64_t ubase, const double *A, const double *eta,
	int n, double *Cinv, double *p_W, bool b_store, const int32_t *sb_row, const int32_t *sb_col, double *S, int ld,
	const int64_t *p_dst, double *p_r, int *p_flag, hipStream_t stream)
{
	{
		const TRunJob *p_jobs = T.d_run_jobs.p();
		const bool b_quad = !dev_knob_set("SLAMPP_HIP_DEV_NO_QUAD_RUNS"); // (development: the one-landmark-per-step form, for A/B timing)
		// (the off-diagonal jobs of three and four tiles a side stage two landmarks; with four staged and quads they need 286
		// registers, one wave per SIMD instead of two: Venice-like C4 1.513 against 1.502 ms, not kept)
		const bool b_quad_wide = !dev_knob_set("SLAMPP_HIP_DEV_NO_QUAD_WIDE"); // (quads also for the diagonal jobs of three and four tiles a side, four landmarks staged at a time: Venice-like C4 1.545 -> 1.486 ms; the off-diagonal ones stage two and stay as they were)
#define LAUNCH_RUNS_Q(NT, DIAG, PFX, QUAD) hipLaunchKernelGGL((schur_run_kernel<DC, DP, NT, DIAG != 0, PFX != 0, QUAD>), \
			dim3(unsigned(T.n_run_jobs[NT][DIAG][PFX])), dim3(64), 0, stream, p_jobs + T.n_run_job_first[NT][DIAG][PFX], T.d_run_lm.p(), \
			T.d_run_rec.p(), T.d_run_k.p(), ubase, A, eta, n, Cinv, p_W, int(b_store), T.d_P.p(), T.d_R.p(), p_flag)
		// (quads of landmarks wherever a job stages a multiple of four landmarks at a time: all but the off-diagonal jobs of
		// three and four tiles a side, which stage two)
#define LAUNCH_RUNS(NT, DIAG, PFX) do { if(T.n_run_jobs[NT][DIAG][PFX]) { if((NT <= 2 || (DIAG && b_quad_wide)) && b_quad) LAUNCH_RUNS_Q(NT, DIAG, PFX, (NT <= 2 || DIAG)); \
			else LAUNCH_RUNS_Q(NT, DIAG, PFX, false); } } while(0)
#define LAUNCH_RUNS_D(NT, DIAG) do { LAUNCH_RUNS(NT, DIAG, 0); LAUNCH_RUNS(NT, DIAG, 1); } while(0)
		LAUNCH_RUNS_D(1, 1);
		LAUNCH_RUNS_D(2, 1);
		LAUNCH_RUNS_D(3, 1);
		LAUNCH_RUNS_D(4, 1);
		LAUNCH_RUNS_D(1, 0);
		LAUNCH_RUNS_D(2, 0);
		LAUNCH_RUNS_D(3, 0);
		LAUNCH_RUNS_D(4, 0);
#undef LAUNCH_RUNS_D
#undef LAUNCH_RUNS
#undef LAUNCH_RUNS_Q
	}
	if(T.n_tiles) {
	enum { BLK = DC * DP, MAXK = tile_max_k(SCHUR_TILE_SLOTS), OPS = (MAXK * BLK > 320)? (MAXK * BLK + 63) / 64 * 64 : 320,
		NL_MAX = (MAXK * BLK + 63) / 64 };
	const size_t n_lds = (2 * OPS + size_t(T.n_max_slots) * (DC * DC + DC)) * sizeof(double) + 128;
	const int n_loads = int((T.n_max_k * BLK + 63) / 64); // 64-lane loads that fetch the blocks of the largest landmark
#define LAUNCH_TILES(NL) hipLaunchKernelGGL((schur_tile_kernel<DC, DP, SCHUR_TILE_SLOTS, (NL <= NL_MAX)? NL : NL_MAX>), \
		dim3(unsigned(T.n_tiles)), dim3(64), n_lds, stream, T.d_tile_ptr.p(), T.d_tile_lm.p(), T.d_tile_slot_ptr.p(), \
		T.d_pair_ptr.p(), T.d_lm_slot.p(), ptr, nc, ubase, A, eta, n, Cinv, p_W, int(b_store), T.d_P.p(), T.d_R.p(), p_flag, \
		int(T.n_max_slots))
	if(n_loads <= 1)
		LAUNCH_TILES(1);
	else if(n_loads == 2)
		LAUNCH_TILES(2);
	else if(n_loads == 3)
		LAUNCH_TILES(3);
	else
		LAUNCH_TILES(4);
#undef LAUNCH_TILES
	}
	if(T.n_rb) // (landmarks nobody observes have no blocks of S to reduce)
		hipLaunchKernelGGL((schur_tile_reduce_kernel<DC>), dim3(unsigned(T.n_rb)), dim3(64), 0, stream,
		T.d_rb_ptr.p(), T.d_rb_part.p(), T.d_rb_sb.p(), sb_row, sb_col, T.d_P.p(), T.d_R.p(), S, ld, p_dst, p_r);
}

void schur_tiles_join(CSchurTiles &T)
{
	if(!T.p_run_upload)
		return;
	std::shared_ptr<TRunUpload> p_up;
	p_up.swap(T.p_run_upload);
	if(p_up->t.joinable())
		p_up->t.join();
	if(p_up->p_error)
		std::rethrow_exception(p_up->p_error);
}

void schur_tiles_enqueue(const CSchurTiles &T, int DC, int DP, const int64_t *ptr, int64_t nc, int64_t ubase, const double *A,
	const double *eta, int n, double *Cinv, double *p_W, bool b_store, const int32_t *sb_row, const int32_t *sb_col,
	double *S, int ld, const int64_t *p_dst, double *p_r, int *p_flag, hipStream_t stream)
{
	if(DC == 6 && DP == 3)
		tiles_enqueue_t<6, 3>(T, ptr, nc, ubase, A, eta, n, Cinv, p_W, b_store, sb_row, sb_col, S, ld, p_dst, p_r, p_flag, stream);
	else if(DC == 7 && DP == 3)
		tiles_enqueue_t<7, 3>(T, ptr, nc, ubase, A, eta, n, Cinv, p_W, b_store, sb_row, sb_col, S, ld, p_dst, p_r, p_flag, stream);
	else
		tiles_enqueue_t<3, 2>(T, ptr, nc, ubase, A, eta, n, Cinv, p_W, b_store, sb_row, sb_col, S, ld, p_dst, p_r, p_flag, stream);
}

// ---------------------------------------------------------------------------------------------
// host analysis
// ---------------------------------------------------------------------------------------------

namespace {

// the tile under construction: its blocks of S by key, in a small open-addressed table
struct TSlotTable {
	enum { SIZE = 256 }; // > 2 * SCHUR_TILE_SLOTS
	int64_t keys[SIZE];
	int16_t slots[SIZE];
	int used[SIZE];
	int n_used;
	TSlotTable() :n_used(0) { for(int i = 0; i < SIZE; ++ i) slots[i] = -1; }
	static size_t n_Hash(int64_t key) { return size_t((uint64_t(key) * 0x9E3779B97F4A7C15ull) >> 56) & (SIZE - 1); }
	int n_Find(int64_t key) const
	{
		for(size_t h = n_Hash(key);; h = (h + 1) & (SIZE - 1)) {
			if(slots[h] < 0)
				return -1;
			if(keys[h] == key)
				return slots[h];
		}
	}
	void Insert(int64_t key, int n_slot)
	{
		size_t h = n_Hash(key);
		while(slots[h] >= 0)
			h = (h + 1) & (SIZE - 1);
		keys[h] = key;
		slots[h] = int16_t(n_slot);
		used[n_used ++] = int(h);
	}
	void Clear()
	{
		for(int i = 0; i < n_used; ++ i)
			slots[used[i]] = -1;
		n_used = 0;
	}
};

struct TTileRun { // the tiles of one contiguous piece of the sorted landmark order
	std::vector<int32_t> tile_size, tile_lm;  // landmarks per tile; the landmarks
	std::vector<int32_t> tile_slots;          // blocks per tile
	std::vector<int64_t> slot_key;            // their keys, tile after tile
	std::vector<uint8_t> lm_slot;             // slots of the landmarks' pairs, landmark after landmark
	std::vector<int32_t> rejected;            // landmarks left to the lists
};

void build_run(TTileRun &R, const int32_t *p_order, int64_t n_first, int64_t n_last, int n_mode, int64_t nc, const int64_t *ptr,
	const int32_t *brow)
{
	const int n_cap = SCHUR_TILE_SLOTS, n_max_k = tile_max_k(SCHUR_TILE_SLOTS);
	const int n_max_points = std::max(1, dev_knob("SLAMPP_HIP_DEV_TILE_POINTS", int(SCHUR_TILE_MAX_POINTS)));
	TSlotTable table;
	std::vector<int32_t> cur_lm;
	std::vector<int64_t> cur_keys;
	std::vector<uint8_t> cur_slots;
	int64_t n_cur_pairs = 0;
	auto Close = [&]() {
		if(cur_lm.empty())
			return;
		const bool b_good = n_mode > 0 || n_cur_pairs >= 3 * int64_t(cur_keys.size());
		if(b_good) {
			R.tile_size.push_back(int32_t(cur_lm.size()));
			R.tile_lm.insert(R.tile_lm.end(), cur_lm.begin(), cur_lm.end());
			R.tile_slots.push_back(int32_t(cur_keys.size()));
			R.slot_key.insert(R.slot_key.end(), cur_keys.begin(), cur_keys.end());
			R.lm_slot.insert(R.lm_slot.end(), cur_slots.begin(), cur_slots.end());
		} else
			R.rejected.insert(R.rejected.end(), cur_lm.begin(), cur_lm.end());
		cur_lm.clear();
		cur_keys.clear();
		cur_slots.clear();
		n_cur_pairs = 0;
		table.Clear();
	};
	int64_t keys[SCHUR_TILE_SLOTS];
	for(int64_t i = n_first; i < n_last; ++ i) {
		const int32_t pt = p_order[i];
		const int64_t k0 = ptr[nc + pt], k = ptr[nc + pt + 1] - k0 - 1;
		if(k > n_max_k) {
			R.rejected.push_back(pt);
			continue;
		}
		const int n_pairs = int(k * (k + 1) / 2);
		int n_new = 0;
		for(int64_t b = 0, j = 0; b < k; ++ b) {
			for(int64_t a = 0; a <= b; ++ a, ++ j) {
				keys[j] = int64_t(brow[k0 + a]) * nc + brow[k0 + b]; // column = the smaller camera, row = the larger
				n_new += table.n_Find(keys[j]) < 0;
			}
		}
		if(int(cur_keys.size()) + n_new > n_cap || int(cur_lm.size()) >= n_max_points)
			Close();
		for(int j = 0; j < n_pairs; ++ j) {
			int n_slot = table.n_Find(keys[j]);
			if(n_slot < 0) {
				n_slot = int(cur_keys.size());
				table.Insert(keys[j], n_slot);
				cur_keys.push_back(keys[j]);
			}
			cur_slots.push_back(uint8_t(n_slot));
		}
		cur_lm.push_back(pt);
		n_cur_pairs += n_pairs;
	}
	Close();
}


// LSD radix sort of items by bits [n_first_bit, n_last_bit) of their 64-bit keys, 11 bits a pass (2 048 destinations a thread:
// the writes of a pass stay in the caches; with 16-bit digits every write was a miss), stable, on up to eight threads: a thread
// counts the digits of its range, the ranges' counts are laid out digit by digit and thread by thread -- where a serial pass
// would have put the elements --, and every thread moves its own range.  The items carry what later passes need: nothing is
// read through the permutation.
template <class TItem, class CKeyOf>
void radix_sort_items(raw_vector<TItem> &r_items, raw_vector<TItem> &r_tmp, int n_first_bit, int n_last_bit, CKeyOf key_of)
{
	enum { DIGIT_BITS = 11, DIGITS = 1 << DIGIT_BITS };
	const int64_t n = int64_t(r_items.size());
	r_tmp.resize(size_t(n));
	const int n_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), n / 32768)));
	std::vector<std::vector<int64_t> > cnt(n_workers, std::vector<int64_t>(DIGITS));
	auto For_Ranges = [&](const std::function<void(int, int64_t, int64_t)> &r_work) {
		std::vector<std::thread> threads;
		for(int t = 0; t < n_workers; ++ t) {
			const int64_t n_first = n * t / n_workers, n_last = n * (t + 1) / n_workers;
			if(t + 1 < n_workers)
				threads.emplace_back(r_work, t, n_first, n_last);
			else
				r_work(t, n_first, n_last);
		}
		for(size_t t = 0; t < threads.size(); ++ t)
			threads[t].join();
	};
	for(int n_shift = n_first_bit; n_shift < n_last_bit; n_shift += DIGIT_BITS) {
		const uint64_t n_mask = (n_last_bit - n_shift >= DIGIT_BITS)? uint64_t(DIGITS - 1) : (uint64_t(1) << (n_last_bit - n_shift)) - 1;
		For_Ranges([&](int t, int64_t n_first, int64_t n_last) {
			std::vector<int64_t> &r_cnt = cnt[t];
			std::fill(r_cnt.begin(), r_cnt.end(), 0);
			for(int64_t i = n_first; i < n_last; ++ i)
				++ r_cnt[(key_of(r_items[i]) >> n_shift) & n_mask];
		});
		int64_t n_sum = 0;
		for(int d = 0; d < DIGITS; ++ d) {
			for(int t = 0; t < n_workers; ++ t) {
				const int64_t n_here = cnt[t][d];
				cnt[t][d] = n_sum;
				n_sum += n_here;
			}
		}
		For_Ranges([&](int t, int64_t n_first, int64_t n_last) {
			std::vector<int64_t> &r_cnt = cnt[t];
			for(int64_t i = n_first; i < n_last; ++ i)
				r_tmp[r_cnt[(key_of(r_items[i]) >> n_shift) & n_mask] ++] = r_items[i];
		});
		r_items.swap(r_tmp);
	}
}

} // namespace

void schur_tiles_build(CSchurTiles &T, int n_mode, int DC, int DP, int64_t nc, int64_t np, const int64_t *ptr, const int32_t *brow,
	const std::vector<int32_t> &sb_row, const std::vector<int32_t> &sb_col, int64_t n_ablocks, hipStream_t stream)
{
	if(n_mode == 0 || !np || np > INT32_MAX)
		return;
	// n_mode: -1 = runs of at least four landmarks, tiles where landmarks share blocks, both only if together they take
	// half of the contributions; 1 = runs of two and every tile that can be formed; 2 = tiles only; 3 = runs only, of any length
	const bool b_use_runs = n_mode != 2, b_use_tiles = n_mode != 3;
	const int64_t n_min_run = (n_mode == 3)? 1 : (n_mode == 1)? 2 : 4;
	const int OB = 64 / DC;
	int64_t n_all_pairs = 0;
	for(int64_t pt = 0; pt < np; ++ pt) {
		const int64_t k = ptr[nc + pt + 1] - ptr[nc + pt] - 1;
		n_all_pairs += k * (k + 1) / 2;
	}
	T.n_all_pairs = n_all_pairs;
	const bool b_build_timing = getenv("SLAMPP_HIP_PLAN_TIMING") != 0;
	double t_build_phase = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
#define BUILD_PHASE(name) do { if(b_build_timing) { const double t_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); \
	fprintf(stderr, "[schur tiles] %-20s %8.2f ms\n", name, t_ - t_build_phase); t_build_phase = t_; } } while(0)
	raw_vector<uint8_t> handled(np, 0); // landmarks that do not go through the contribution lists
	raw_vector<int64_t> slot_key;       // the block of S of every partial block: runs first, then tiles
	CTrashList trash; // (schur_tiles.h: the big work arrays of the passes below, freed beside the uploads)

	// ---- runs: landmarks with identical camera lists, found by sorting hashes of the lists ----
	raw_vector<TRunJob> jobs;
	raw_vector<int32_t> run_lm, run_k;
	int64_t n_run_pairs = 0;
	if(b_use_runs) {
		// (the passes over the landmarks that do not depend on each other run on a few threads: at C5's two million landmarks
		// telling the classes apart was 98 ms of a 250 ms analysis on one core, a cache miss per list)
		auto For_Landmark_Ranges = [np](int64_t n_begin, const std::function<void(int64_t, int64_t)> &r_work) {
			const int64_t n = np - n_begin;
			const int n_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), n / 65536)));
			std::vector<std::thread> threads;
			for(int t = 0; t < n_workers; ++ t) {
				const int64_t n_first = n_begin + n * t / n_workers, n_last = n_begin + n * (t + 1) / n_workers;
				if(t + 1 < n_workers)
					threads.emplace_back(r_work, n_first, n_last);
				else
					r_work(n_first, n_last);
			}
			for(size_t t = 0; t < threads.size(); ++ t)
				threads[t].join();
		};
		raw_vector<uint64_t> hash(np), hash2(np); // (hash2: a second, independent hash of the same list -- see "same lists" below; raw_vector: not zero-filled, solver.h)
		raw_vector<int32_t> k_of(np); // observations per landmark (8 MB that stay in the caches better than two reads of ptr[] per use)
		For_Landmark_Ranges(0, [&](int64_t n_first, int64_t n_last) {
			for(int64_t pt = n_first; pt < n_last; ++ pt) {
				const int64_t k0 = ptr[nc + pt], k = ptr[nc + pt + 1] - k0 - 1;
				k_of[pt] = int32_t(k);
				uint64_t h = 0x9E3779B97F4A7C15ull * uint64_t(k + 1), h2 = 0xCBF29CE484222325ull ^ uint64_t(k);
				for(int64_t i = 0; i < k; ++ i) {
					h ^= uint64_t(brow[k0 + i]) + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
					h *= 0xFF51AFD7ED558CCDull;
					h2 = (h2 ^ (uint64_t(brow[k0 + i]) + 0x632BE59BD9B4E019ull * uint64_t(i + 1))) * 0x100000001B3ull;
					h2 ^= h2 >> 29;
				}
				hash[pt] = h;
				hash2[pt] = h2;
			}
		});
		BUILD_PHASE("hashes");
		// Radix sort by the hash (radix_sort_items): equal hashes stay in landmark order.  Round 6: what the passes and the
		// comparison of neighbours below need of a landmark -- both
		// hashes, the length of its list -- travels with it (24 bytes an element, read in order, instead of reads of hash[],
		// hash2[] and k_of[] in the order of the pass before: a cache miss each, 12 + 8 ms at C5's two million landmarks).
		struct TSortItem { uint64_t n_hash, n_hash2; int32_t n_k, n_landmark; };
		raw_vector<TSortItem> items(np), items_tmp(np);
		raw_vector<int32_t> order(np);
		For_Landmark_Ranges(0, [&](int64_t n_first, int64_t n_last) {
			for(int64_t pt = n_first; pt < n_last; ++ pt) {
				TSortItem &r_item = items[pt];
				r_item.n_hash = hash[pt];
				r_item.n_hash2 = hash2[pt];
				r_item.n_k = k_of[pt];
				r_item.n_landmark = int32_t(pt);
			}
		});
		// (by the upper 33 bits: three passes.  Lists whose hashes differ only below -- a few hundred pairs among two million --
		// may come out interleaved; the classes are then cut where "same lists" below says so, and the chains put classes of
		// one list back together: the same runs)
		radix_sort_items(items, items_tmp, 31, 64, [](const TSortItem &r_item) { return r_item.n_hash; });
		BUILD_PHASE("sort");
		auto Same = [&](int32_t p, int32_t q) -> bool {
			const int64_t kp0 = ptr[nc + p], kq0 = ptr[nc + q], k = ptr[nc + p + 1] - kp0 - 1;
			if(ptr[nc + q + 1] - kq0 - 1 != k)
				return false;
			for(int64_t i = 0; i < k; ++ i) {
				if(brow[kp0 + i] != brow[kq0 + i])
					return false;
			}
			return true;
		};
		const int64_t n_piece_max = std::max(1, std::min(64, dev_knob("SLAMPP_HIP_DEV_RUN_PIECE", 64)));
		std::vector<TRunJob> jobs_nt[5][2][2];
		// what a pass over runs leaves behind (round 6: the runs are emitted by a few threads, each into one of these, and the
		// results put behind each other in the runs' order -- the tables come out as one thread wrote them)
		struct TEmitOut {
			raw_vector<int32_t> run_lm, run_k;
			raw_vector<int64_t> slot_key;
			std::vector<TRunJob> jobs_nt[5][2][2];
			int64_t n_run_pairs = 0, n_prefix_points = 0;
			std::vector<int32_t> members, members_k; // (scratch)
		};
		// the jobs of one run: `members` (positions in `order`), longest camera list first -- every other member's list is that
		// list or a prefix of it; pieces of at most 64 landmarks, a job per pair of observation blocks, over the piece's
		// landmarks that reach into the row block (they are the first ones: sorted by length)
		// Longer pieces (a job takes its landmarks 64 at a time and keeps its sums across those sub-pieces) leave half or a
		// quarter of the partial blocks behind -- and were measured no faster (round 5: Venice-like C4 1.491 / 1.519 / 1.569 ms
		// at 1 / 2 / 4 times the base length: the longest jobs set the length of a launch; C5 1.197 / 1.223 / 1.183): the
		// kernel can, the analysis does not ask for it (development knob, tests/test_schur_gpu.py)
		const int64_t n_piece_mult = dev_knob_set("SLAMPP_HIP_DEV_RUN_PIECE_MULT")? std::max(1, std::min(4, dev_knob("SLAMPP_HIP_DEV_RUN_PIECE_MULT", 1))) : 1;
		auto Emit_Run = [&](TEmitOut &r_out, const int32_t *p_members, const int32_t *p_members_k /* null: every member has k_all observations */, int64_t n_members, int64_t k_all) {
			raw_vector<int32_t> &run_lm = r_out.run_lm, &run_k = r_out.run_k; // (this thread's: positions and offsets in the jobs are relative to them until the merge)
			raw_vector<int64_t> &slot_key = r_out.slot_key;
			std::vector<TRunJob> (&jobs_nt)[5][2][2] = r_out.jobs_nt;
			int64_t &n_run_pairs = r_out.n_run_pairs;
			// (pieces of 32 where a job keeps ten or sixteen accumulator tiles -- nine cameras and up at 6 x 6 --: those jobs are
			// the long ones, and more of them spread better: 164 + 123 -> 130 + 103 us for the two widest kernels of the
			// Venice-like C4, for 18 us more in the reduction of the partial blocks)
			const int64_t k_run = p_members_k? p_members_k[0] : k_all;
			const int64_t n_piece_len = ((std::min<int64_t>(k_run, OB) * DC > 48 && !dev_knob_set("SLAMPP_HIP_DEV_RUN_PIECE"))? 32 : n_piece_max) * n_piece_mult;
			for(int64_t f = 0; f < n_members; f += n_piece_len) {
				const int64_t n_piece = std::min<int64_t>(n_piece_len, n_members - f);
				const int32_t pt0 = p_members[f];
				const int64_t k0 = ptr[nc + pt0], k = ptr[nc + pt0 + 1] - k0 - 1; // the piece's longest list
				const int64_t n_blocks = (k + OB - 1) / OB;
				const int64_t n_lm_first = int64_t(run_lm.size());
				// (the landmarks' lengths come with the members, class by class, and `handled` is set in a pass of its own below:
				// a read of k_of[] and a write of handled[] per landmark, both in hash order, were most of the 20 ms this loop took
				// at C5's two million landmarks)
				run_lm.insert(run_lm.end(), p_members + f, p_members + f + n_piece);
				if(p_members_k) {
					run_k.insert(run_k.end(), p_members_k + f, p_members_k + f + n_piece);
					for(int64_t e = f; e < f + n_piece; ++ e)
						n_run_pairs += int64_t(p_members_k[e]) * (p_members_k[e] + 1) / 2;
				} else { // (one class: the same list for all of them -- round 6: no copy of the members, no array of equal lengths)
					run_k.insert(run_k.end(), size_t(n_piece), int32_t(k_all));
					n_run_pairs += n_piece * (k_all * (k_all + 1) / 2);
				}
				for(int64_t rb = 0; rb < n_blocks; ++ rb) {
					const int64_t n_kb_r = std::min<int64_t>(OB, k - rb * OB);
					int64_t n_reach = p_members_k? 0 : n_piece; // landmarks of the piece with observations in row block rb (one class: all of them)
					while(n_reach < n_piece && run_k[size_t(n_lm_first + n_reach)] > rb * OB)
						++ n_reach;
					for(int64_t cb = 0; cb <= rb; ++ cb) {
						const int64_t n_kb_c = std::min<int64_t>(OB, k - cb * OB);
						TRunJob job;
						job.n_first = int32_t(n_lm_first);
						job.n_points = int32_t(n_reach);
						job.n_k = int32_t(k);
						job.n_rb = int32_t(rb);
						job.n_cb = int32_t(cb);
						job.n_pad = 0;
						job.n_pbase = int64_t(slot_key.size());
						for(int64_t ob = 0; ob < n_kb_r; ++ ob) { // the order the kernel numbers its partial blocks in
							for(int64_t oa = 0; oa < ((rb == cb)? ob + 1 : n_kb_c); ++ oa)
								slot_key.push_back(int64_t(brow[k0 + cb * OB + oa]) * nc + brow[k0 + rb * OB + ob]);
						}
						const int n_lines = int(std::max(n_kb_r, n_kb_c)) * DC;
						const bool b_job_prefix = run_k[size_t(n_lm_first + n_reach - 1)] < k; // (sorted: the last one is the shortest)
						jobs_nt[(n_lines + 15) / 16][rb == cb][b_job_prefix].push_back(job);
					}
				}
			}
		};
		// Round 4: a run is a CHAIN of such classes -- landmarks seen by exactly the same cameras -- in which every class's camera
		// list continues the one before it (tracks born at the same camera and lost at different ones; identical lists are the
		// special case).  A landmark reads as zero beyond its own last observation (run_k), so the chain's partial blocks are
		// those of its longest list, written once per piece of 64 landmarks instead of once per class: at the Venice-like C4
		// the classes of the long tracks have 2 - 17 members and 66 - 465 blocks of S each.  Chains of short lists break
		// where the matrix-core tiles of a job would grow (16 lines a tile): a landmark of two cameras does not pay for ten.
		struct TClass { int64_t n_first, n_count; };
		std::vector<TClass> classes;
		raw_vector<uint8_t> same_as_previous(np); // order[i] is seen by the cameras of order[i - 1]
		if(np)
			same_as_previous[0] = 0;
		// Two neighbours of the sorted order have the same cameras if both 64-bit hashes and the lengths agree: 128 bits over at
		// most a few million lists (two different lists agreeing in both: below 1e-25 an analysis).  Reading the two lists
		// themselves -- four dependent cache misses per landmark, in hash order -- was 20 ms of C5's analysis on eight threads;
		// systems under 262 144 landmarks (where it costs nothing) still do (SLAMPP_HIP_DEV_RUN_HASH_ONLY, a development knob:
		// the hashes alone there too, for the tests).
		const bool b_compare_lists = np < (int64_t(1) << 18) && !dev_knob_set("SLAMPP_HIP_DEV_RUN_HASH_ONLY");
		if(np)
			order[0] = items[0].n_landmark;
		For_Landmark_Ranges(1, [&](int64_t n_first, int64_t n_last) {
			for(int64_t i = n_first; i < n_last; ++ i) {
				const TSortItem &r_p = items[i - 1], &r_q = items[i];
				order[i] = r_q.n_landmark;
				same_as_previous[i] = r_q.n_hash == r_p.n_hash && r_q.n_hash2 == r_p.n_hash2 && r_q.n_k == r_p.n_k &&
					(!b_compare_lists || Same(r_p.n_landmark, r_q.n_landmark));
			}
		});
		if(!b_compare_lists) {
			// the hashes alone said "same": every class of more than one landmark is held to its lists once, first member
			// against last (two lists per class instead of two per landmark; advisor, round 5: two unkeyed hashes of the same
			// 32-bit indices are not 128 independent bits).  A mismatch -- never seen -- sends every neighbour through Same().
			std::atomic<int> n_collisions(0);
			For_Landmark_Ranges(1, [&](int64_t n_first, int64_t n_last) {
				for(int64_t i = n_first; i < n_last; ++ i) {
					if(same_as_previous[i] && (i + 1 == np || !same_as_previous[i + 1])) { // the last member of a class
						int64_t f = i;
						while(f > 0 && same_as_previous[f])
							-- f; // (may leave this thread's range: read only)
						if(!Same(items[f].n_landmark, items[i].n_landmark))
							n_collisions.fetch_add(1, std::memory_order_relaxed);
					}
				}
			});
			if(n_collisions.load()) {
				For_Landmark_Ranges(1, [&](int64_t n_first, int64_t n_last) {
					for(int64_t i = n_first; i < n_last; ++ i)
						same_as_previous[i] = same_as_previous[i] && Same(items[i - 1].n_landmark, items[i].n_landmark);
				});
			}
		}
		BUILD_PHASE("  same lists");
		// Every list as long as every other: then no list continues another, a run is a class of landmarks with the same cameras,
		// and if the largest class is too small to be one there is no run anywhere -- the classes need not be formed, ordered and
		// walked to find that out (round 6: the uniform-visibility C4, three pairs of equal lists among half a million, spent 28 ms
		// of its 57 ms analysis ordering half a million classes of one landmark each).  The largest class = the longest stretch of
		// "same as previous" + 1: per range of the sorted order, then over the ranges' ends.
		bool b_runs_impossible = false;
		if(n_min_run >= 2 && np > 1) {
			struct TStretch { int64_t n_first, n_last, n_head, n_tail, n_longest; int32_t k_min, k_max; }; // head / tail: stretches at the range's ends
			std::vector<TStretch> parts;
			std::mutex t_mutex;
			For_Landmark_Ranges(0, [&](int64_t n_first, int64_t n_last) {
				TStretch t = {n_first, n_last, 0, 0, 0, INT32_MAX, 0};
				int64_t n_current = 0;
				bool b_head = true;
				for(int64_t i = n_first; i < n_last; ++ i) {
					if(same_as_previous[i])
						++ n_current;
					else {
						if(b_head)
							t.n_head = n_current;
						b_head = false;
						t.n_longest = std::max(t.n_longest, n_current);
						n_current = 0;
					}
					t.k_min = std::min(t.k_min, int32_t(items[i].n_k));
					t.k_max = std::max(t.k_max, int32_t(items[i].n_k));
				}
				if(b_head)
					t.n_head = n_current; // (the whole range is one stretch)
				t.n_tail = n_current;
				t.n_longest = std::max(t.n_longest, n_current);
				std::lock_guard<std::mutex> t_lock(t_mutex);
				parts.push_back(t);
			});
			std::sort(parts.begin(), parts.end(), [](const TStretch &a, const TStretch &b) { return a.n_first < b.n_first; });
			int64_t n_longest = 0, n_open = 0; // (n_open: the stretch that reaches the end of the ranges seen so far)
			int32_t k_min = INT32_MAX, k_max = 0;
			for(const TStretch &t : parts) {
				const bool b_whole = t.n_head == t.n_last - t.n_first; // every entry of the range continues a stretch
				n_longest = std::max(n_longest, std::max(t.n_longest, n_open + t.n_head));
				n_open = b_whole? n_open + t.n_head : t.n_tail;
				k_min = std::min(k_min, t.k_min);
				k_max = std::max(k_max, t.k_max);
			}
			const int64_t n_needed = (k_max > OB && n_min_run > 2)? 2 : n_min_run; // (what the emission below asks of a run)
			b_runs_impossible = k_min == k_max && n_longest + 1 < n_needed;
		}
		run_lm.reserve(b_runs_impossible? 0 : np);
		run_k.reserve(b_runs_impossible? 0 : np);
		for(int64_t i = 0; i < np && !b_runs_impossible;) {
			int64_t j = i + 1;
			while(j < np && same_as_previous[j])
				++ j;
			if(ptr[nc + order[i] + 1] - ptr[nc + order[i]] - 1 >= 1)
				classes.push_back(TClass{i, j - i});
			i = j;
		}
		auto List_Less = [&](int32_t p, int32_t q) -> bool { // lexicographic: a list sorts right before its continuations
			const int64_t kp0 = ptr[nc + p], kq0 = ptr[nc + q], kp = ptr[nc + p + 1] - kp0 - 1, kq = ptr[nc + q + 1] - kq0 - 1;
			for(int64_t i = 0; i < std::min(kp, kq); ++ i) {
				if(brow[kp0 + i] != brow[kq0 + i])
					return brow[kp0 + i] < brow[kq0 + i];
			}
			return kp < kq;
		};
		auto Is_Prefix = [&](int32_t p, int32_t q) -> bool { // the list of p starts the list of q
			const int64_t kp0 = ptr[nc + p], kq0 = ptr[nc + q], kp = ptr[nc + p + 1] - kp0 - 1, kq = ptr[nc + q + 1] - kq0 - 1;
			if(kp > kq)
				return false;
			for(int64_t i = 0; i < kp; ++ i) {
				if(brow[kp0 + i] != brow[kq0 + i])
					return false;
			}
			return true;
		};
		auto Tile_Class = [&](int32_t p) -> int { // 16-line tiles a side of the landmark's (first) job
			const int64_t k = ptr[nc + p + 1] - ptr[nc + p] - 1;
			return int((std::min<int64_t>(k, OB) * DC + 15) / 16);
		};
		BUILD_PHASE("  classes");
		const bool b_chains = !dev_knob_set("SLAMPP_HIP_DEV_NO_PREFIX_RUNS");
		if(b_chains) {
			// Round 6: by a 64-bit key of the first cameras of the list (a radix sort on a few threads), the lists themselves only
			// where keys tie -- a comparison sort that reads two lists per comparison was 96 ms at 500 000 landmarks with
			// 500 000 different lists (uniform visibility).  Camera + 1 in every field, zero behind the end of a short list:
			// the keys order the first fields' worth of cameras the way List_Less does, a list right before its continuations.
			const int64_t n_classes = int64_t(classes.size());
			int n_field_bits = 1;
			while((int64_t(1) << n_field_bits) <= nc)
				++ n_field_bits;
			const int n_fields = std::max(1, 64 / n_field_bits), n_key_bits = n_fields * n_field_bits;
			struct TClassItem { uint64_t n_key; TClass t_class; };
			raw_vector<TClassItem> class_items(n_classes), class_items_tmp;
			const int n_class_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), n_classes / 16384)));
			{
				std::vector<std::thread> threads;
				auto Keys = [&](int64_t n_first, int64_t n_last) {
					for(int64_t c = n_first; c < n_last; ++ c) {
						const int32_t p = order[classes[c].n_first];
						const int64_t k0 = ptr[nc + p], k = ptr[nc + p + 1] - k0 - 1;
						uint64_t n_key = 0;
						for(int f = 0; f < n_fields; ++ f)
							n_key = (n_key << n_field_bits) | ((f < k)? uint64_t(brow[k0 + f]) + 1 : 0);
						class_items[c].n_key = n_key;
						class_items[c].t_class = classes[c];
					}
				};
				for(int t = 0; t + 1 < n_class_workers; ++ t)
					threads.emplace_back(Keys, n_classes * t / n_class_workers, n_classes * (t + 1) / n_class_workers);
				Keys(n_classes * (n_class_workers - 1) / n_class_workers, n_classes);
				for(size_t t = 0; t < threads.size(); ++ t)
					threads[t].join();
			}
			radix_sort_items(class_items, class_items_tmp, 0, n_key_bits, [](const TClassItem &r_item) { return r_item.n_key; }); // (stable: classes of one key stay in the order they were found in)
			std::vector<TClass> sorted(classes.size());
			for(int64_t c = 0; c < n_classes; ++ c)
				sorted[c] = class_items[c].t_class;
			for(int64_t c0 = 0; c0 < n_classes;) { // keys that tie: lists longer than the key that agree in all of its fields
				int64_t c1 = c0 + 1;
				while(c1 < n_classes && class_items[c1].n_key == class_items[c0].n_key)
					++ c1;
				if(c1 - c0 > 1) {
					std::sort(sorted.begin() + c0, sorted.begin() + c1, [&](const TClass &a, const TClass &b) {
						const int32_t p = order[a.n_first], q = order[b.n_first];
						return List_Less(p, q) || (!List_Less(q, p) && a.n_first < b.n_first); });
				}
				c0 = c1;
			}
			classes.swap(sorted);
		}
		BUILD_PHASE("  class order");
		// what the chains ask of every class -- the tiles of its job, the length of its list, whether it continues the class
		// before it -- read from the lists on a few threads (round 6: 500 000 classes of one landmark each, uniform
		// visibility, were 15 ms of cache misses in the loop below)
		const int64_t n_classes_all = int64_t(classes.size());
		raw_vector<int32_t> class_k(n_classes_all);
		raw_vector<uint8_t> class_tiles(n_classes_all), class_continues(n_classes_all);
		{
			const int n_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), n_classes_all / 16384)));
			auto Look = [&](int64_t n_first, int64_t n_last) {
				for(int64_t c = n_first; c < n_last; ++ c) {
					const int32_t p = order[classes[c].n_first];
					class_k[c] = int32_t(ptr[nc + p + 1] - ptr[nc + p] - 1);
					class_tiles[c] = uint8_t(Tile_Class(p));
					class_continues[c] = b_chains && c > 0 && Is_Prefix(order[classes[c - 1].n_first], p);
				}
			};
			std::vector<std::thread> threads;
			for(int t = 0; t + 1 < n_workers; ++ t)
				threads.emplace_back(Look, n_classes_all * t / n_workers, n_classes_all * (t + 1) / n_workers);
			Look(n_classes_all * (n_workers - 1) / n_workers, n_classes_all);
			for(size_t t = 0; t < threads.size(); ++ t)
				threads[t].join();
		}
		BUILD_PHASE("  class lists");
		std::vector<int32_t> members, members_k; // (kept for the discard list below)
		struct TGroup { size_t c0, c1; int64_t n_members; };
		std::vector<TGroup> groups;
		int64_t n_emit_members = 0;
		for(size_t c0 = 0; c0 < classes.size();) {
			size_t c1 = c0 + 1;
			int64_t n_members = classes[c0].n_count;
			while(c1 < classes.size() && class_tiles[c1] == class_tiles[c0] && class_continues[c1]) {
				n_members += classes[c1].n_count;
				++ c1;
			}
			const int64_t k_longest = class_k[c1 - 1];
			// (a lone long track is no better off here than in the lists; two of them already share their partial blocks)
			if(n_members >= ((k_longest > OB && n_min_run > 2)? 2 : n_min_run)) {
				groups.push_back(TGroup{c0, c1, n_members});
				n_emit_members += n_members;
			}
			c0 = c1;
		}
		auto Emit_Groups = [&](TEmitOut &r_out, size_t g0, size_t g1) {
			for(size_t g = g0; g < g1; ++ g) {
				const size_t c0 = groups[g].c0, c1 = groups[g].c1;
				const int64_t n_members = groups[g].n_members, k_longest = class_k[c1 - 1];
				if(c1 - c0 == 1)
					Emit_Run(r_out, order.data() + classes[c0].n_first, 0, n_members, k_longest); // (the members as they lie in the sorted order)
				else {
					r_out.members.clear();
					r_out.members_k.clear();
					for(size_t c = c1; c > c0; -- c) { // longest list first
						const TClass &r_class = classes[c - 1];
						r_out.members.insert(r_out.members.end(), order.begin() + r_class.n_first, order.begin() + r_class.n_first + r_class.n_count);
						r_out.members_k.insert(r_out.members_k.end(), size_t(r_class.n_count), class_k[c - 1]);
					}
					Emit_Run(r_out, r_out.members.data(), r_out.members_k.data(), int64_t(r_out.members.size()), 0);
					r_out.n_prefix_points += n_members;
				}
			}
		};
		{
			const int n_emit_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(dev_knob("SLAMPP_HIP_DEV_EMIT_THREADS", 8), std::max(1u, std::thread::hardware_concurrency())), n_emit_members / 65536))); // (development aid, plan.h: 1 = one thread, for comparing the tables)
			std::vector<TEmitOut> outs;
			outs.resize(size_t(n_emit_workers));
			std::vector<size_t> g_begin(size_t(n_emit_workers) + 1, groups.size());
			{ // ranges of groups with about the same number of landmarks
				g_begin[0] = 0;
				int64_t n_seen = 0;
				int t = 1;
				for(size_t g = 0; g < groups.size() && t < n_emit_workers; ++ g) {
					n_seen += groups[g].n_members;
					while(t < n_emit_workers && n_seen >= n_emit_members * t / n_emit_workers)
						g_begin[size_t(t ++)] = g + 1;
				}
			}
			std::exception_ptr p_emit_error;
			std::mutex t_error_mutex;
			{
				std::vector<std::thread> threads;
				auto Work = [&](int t) {
					try {
						outs[size_t(t)].run_lm.reserve(size_t(n_emit_members / n_emit_workers + 65536));
						outs[size_t(t)].run_k.reserve(size_t(n_emit_members / n_emit_workers + 65536));
						Emit_Groups(outs[size_t(t)], g_begin[size_t(t)], g_begin[size_t(t) + 1]);
						for(int32_t n_landmark : outs[size_t(t)].run_lm)
							handled[size_t(n_landmark)] = 1; // (a landmark is in one run: no two threads write the same byte)
					} catch(...) {
						std::lock_guard<std::mutex> t_lock(t_error_mutex);
						p_emit_error = std::current_exception();
					}
				};
				for(int t = 1; t < n_emit_workers; ++ t)
					threads.emplace_back(Work, t);
				Work(0);
				for(size_t t = 0; t < threads.size(); ++ t)
					threads[t].join();
			}
			if(p_emit_error)
				std::rethrow_exception(p_emit_error);
			// behind each other, in the order of the runs: positions in the run list and offsets of the partial blocks become global
			if(n_emit_workers == 1) {
				run_lm.swap(outs[0].run_lm);
				run_k.swap(outs[0].run_k);
				slot_key.swap(outs[0].slot_key);
				for(int nt = 0; nt < 5; ++ nt)
					for(int d = 0; d < 2; ++ d)
						for(int x = 0; x < 2; ++ x)
							jobs_nt[nt][d][x].swap(outs[0].jobs_nt[nt][d][x]);
			} else {
				for(int t = 0; t < n_emit_workers; ++ t) {
					TEmitOut &r_out = outs[size_t(t)];
					const int64_t n_lm_base = int64_t(run_lm.size()), n_slot_base = int64_t(slot_key.size());
					for(int nt = 0; nt < 5; ++ nt) {
						for(int d = 0; d < 2; ++ d) {
							for(int x = 0; x < 2; ++ x) {
								for(TRunJob &r_job : r_out.jobs_nt[nt][d][x]) {
									r_job.n_first = int32_t(r_job.n_first + n_lm_base);
									r_job.n_pbase += n_slot_base;
								}
								jobs_nt[nt][d][x].insert(jobs_nt[nt][d][x].end(), r_out.jobs_nt[nt][d][x].begin(), r_out.jobs_nt[nt][d][x].end());
							}
						}
					}
					run_lm.insert(run_lm.end(), r_out.run_lm.begin(), r_out.run_lm.end());
					run_k.insert(run_k.end(), r_out.run_k.begin(), r_out.run_k.end());
					slot_key.insert(slot_key.end(), r_out.slot_key.begin(), r_out.slot_key.end());
				}
			}
			for(int t = 0; t < n_emit_workers; ++ t) {
				n_run_pairs += outs[size_t(t)].n_run_pairs;
				T.n_prefix_points += outs[size_t(t)].n_prefix_points;
			}
		}
		BUILD_PHASE("  emit (jobs)"); // (with the marks of the landmarks that are in runs: each emitting thread sets its own)
		for(int nt = 1; nt <= 4; ++ nt) {
			for(int d = 0; d < 2; ++ d) {
				// one launch per (tiles a side, diagonal): where some jobs of a class have landmarks that end early, all its jobs
				// run the masking instance (its masks do nothing for the others; ten launches of a few thousand waves each, one
				// after the other, cost more in ragged ends than the masks do)
				if(!jobs_nt[nt][d][1].empty()) {
					jobs_nt[nt][d][1].insert(jobs_nt[nt][d][1].end(), jobs_nt[nt][d][0].begin(), jobs_nt[nt][d][0].end());
					jobs_nt[nt][d][0].clear();
				}
				for(int x = 0; x < 2; ++ x) {
					T.n_run_job_first[nt][d][x] = int64_t(jobs.size());
					T.n_run_jobs[nt][d][x] = int64_t(jobs_nt[nt][d][x].size());
					jobs.insert(jobs.end(), jobs_nt[nt][d][x].begin(), jobs_nt[nt][d][x].end());
				}
			}
		}
		Discard_Later(trash, hash); Discard_Later(trash, hash2); Discard_Later(trash, k_of); Discard_Later(trash, items);
		Discard_Later(trash, items_tmp); Discard_Later(trash, order); Discard_Later(trash, same_as_previous);
		Discard_Later(trash, members); Discard_Later(trash, members_k);
	}
	BUILD_PHASE("runs -> jobs");
	if(b_build_timing) { // (what the tables hash to: one thread and several must agree -- SLAMPP_HIP_DEV_EMIT_THREADS)
		uint64_t n_hash = 1469598103934665603ull;
		auto Mix = [&n_hash](const void *p, size_t n_bytes) {
			const unsigned char *p_bytes = (const unsigned char*)p;
			for(size_t i = 0; i < n_bytes; ++ i)
				n_hash = (n_hash ^ p_bytes[i]) * 1099511628211ull;
		};
		Mix(run_lm.data(), run_lm.size() * sizeof(int32_t));
		Mix(run_k.data(), run_k.size() * sizeof(int32_t));
		Mix(slot_key.data(), slot_key.size() * sizeof(int64_t));
		for(const TRunJob &r_job : jobs) {
			const int64_t fields[7] = {r_job.n_first, r_job.n_points, r_job.n_k, r_job.n_rb, r_job.n_cb, r_job.n_pad, r_job.n_pbase};
			Mix(fields, sizeof(fields));
		}
		fprintf(stderr, "[schur tiles] run tables: %zu landmarks, %zu jobs, %zu partial blocks, hash %016llx\n", run_lm.size(), jobs.size(), slot_key.size(), (unsigned long long)n_hash);
	}
	const int64_t n_run_slots = int64_t(slot_key.size()), n_run_points = int64_t(run_lm.size());

	// ---- tiles: the other landmarks by (first camera, second camera) -- two counting sorts --, cut where a tile is full ----
	std::vector<TTileRun> runs;
	int64_t n_tiles = 0, n_tile_slots = 0, n_tile_points = 0, n_tile_pairs = 0;
	if(b_use_tiles && n_run_points < np) {
		// (round 6: the two cameras a landmark is sorted by travel with it -- (first camera, second camera, landmark) in one
		// 64-bit word, two counting passes over the words: the passes used to read ptr[] and brow[] of every landmark in the
		// order of the pass before, four cache misses a landmark and 23 ms at 500 000 landmarks)
		std::vector<int32_t> order;
		if(nc > 65536) { // (cameras that do not fit the 16-bit fields: the passes read the lists)
			std::vector<int32_t> tmp;
			for(int64_t pt = 0; pt < np; ++ pt) {
				if(!handled[pt])
					order.push_back(int32_t(pt));
			}
			tmp.resize(order.size());
			std::vector<int64_t> cnt(nc + 1);
			for(int n_pass = 0; n_pass < 2; ++ n_pass) {
				std::fill(cnt.begin(), cnt.end(), 0);
				auto Cam = [&](int64_t pt) -> int64_t {
					const int64_t k0 = ptr[nc + pt], k = ptr[nc + pt + 1] - k0 - 1;
					return (k == 0)? 0 : brow[k0 + ((n_pass == 0 && k > 1)? 1 : 0)]; // least significant first
				};
				for(size_t i = 0; i < order.size(); ++ i)
					++ cnt[Cam(order[i]) + 1];
				for(int64_t c = 0; c < nc; ++ c)
					cnt[c + 1] += cnt[c];
				for(size_t i = 0; i < order.size(); ++ i)
					tmp[cnt[Cam(order[i])] ++] = order[i];
				order.swap(tmp);
			}
		} else {
			raw_vector<uint64_t> items, items_tmp;
			items.reserve(size_t(np - n_run_points));
			for(int64_t pt = 0; pt < np; ++ pt) {
				if(handled[pt])
					continue;
				const int64_t k0 = ptr[nc + pt], k = ptr[nc + pt + 1] - k0 - 1;
				const uint64_t n_first_cam = (k == 0)? 0 : uint64_t(brow[k0]), n_second_cam = (k > 1)? uint64_t(brow[k0 + 1]) : n_first_cam;
				items.push_back((n_first_cam << 48) | (n_second_cam << 32) | uint64_t(pt));
			}
			const int64_t n_items = int64_t(items.size());
			items_tmp.resize(size_t(n_items));
			std::vector<int64_t> cnt(65537);
			for(int n_shift = 32; n_shift <= 48; n_shift += 16) { // least significant first
				std::fill(cnt.begin(), cnt.end(), 0);
				for(int64_t i = 0; i < n_items; ++ i)
					++ cnt[((items[i] >> n_shift) & 0xFFFF) + 1];
				for(int64_t c = 0; c < 65536; ++ c)
					cnt[c + 1] += cnt[c];
				for(int64_t i = 0; i < n_items; ++ i)
					items_tmp[cnt[(items[i] >> n_shift) & 0xFFFF] ++] = items[i];
				items.swap(items_tmp);
			}
			order.resize(size_t(n_items));
			for(int64_t i = 0; i < n_items; ++ i)
				order[i] = int32_t(uint32_t(items[i]));
		}
		const int64_t n_rest = int64_t(order.size());
		// built piecewise by a few threads (a piece boundary is a tile boundary)
		const int n_pieces = int(std::max<int64_t>(1, std::min<int64_t>(8, n_rest / 32768)));
		// Round 6: where the library decides by itself (n_mode < 0) and the runs alone do not carry half of the contributions, the
		// tiles are tried on a sample first -- the first 4 096 landmarks of every piece -- and left alone if runs and tiles
		// together would stay well under that half (under 30 %: the decision below asks for 50): landmarks that share no
		// cameras with their neighbours in this order (uniform visibility) end up on the lists whatever the tiles find, and
		// forming tiles of all 500 000 of them to learn that was 16 - 23 ms of the analysis.
		if(n_mode < 0 && n_rest >= 65536 && 2 * n_run_pairs < n_all_pairs) {
			const int64_t n_sample = 4096;
			std::vector<TTileRun> sample(n_pieces);
			std::vector<int64_t> sample_pairs(n_pieces, 0);
			std::vector<std::thread> threads;
			for(int t = 0; t < n_pieces; ++ t) {
				const int64_t n_first = n_rest * t / n_pieces, n_last = std::min(n_first + n_sample, n_rest * (t + 1) / n_pieces);
				auto Work = [&sample, &sample_pairs, &order, t, n_first, n_last, n_mode, nc, ptr, brow]() {
					build_run(sample[t], order.data(), n_first, n_last, n_mode, nc, ptr, brow);
					for(int64_t i = n_first; i < n_last; ++ i) {
						const int64_t k = ptr[nc + order[i] + 1] - ptr[nc + order[i]] - 1;
						sample_pairs[t] += k * (k + 1) / 2;
					}
				};
				if(t + 1 < n_pieces)
					threads.emplace_back(Work);
				else
					Work();
			}
			for(size_t t = 0; t < threads.size(); ++ t)
				threads[t].join();
			int64_t n_sample_all = 0, n_sample_tiled = 0;
			for(int t = 0; t < n_pieces; ++ t) {
				n_sample_all += sample_pairs[t];
				n_sample_tiled += int64_t(sample[t].lm_slot.size());
			}
			// pairs outside the runs, and the share of them the sample says tiles would take
			const int64_t n_rest_pairs = n_all_pairs - n_run_pairs;
			const double f_tiled = n_sample_all? double(n_sample_tiled) / double(n_sample_all) : 0.0;
			if(b_build_timing)
				fprintf(stderr, "[schur tiles] sample: tiles take %.1f %% of the contributions outside the runs\n", 100 * f_tiled);
			if(double(n_run_pairs) + f_tiled * double(n_rest_pairs) < 0.30 * double(n_all_pairs)) {
				BUILD_PHASE("tiles (sample)");
				memset(T.n_run_jobs, 0, sizeof(T.n_run_jobs));
				memset(T.n_run_job_first, 0, sizeof(T.n_run_job_first));
				return; // the lists keep everything
			}
		}
		runs.resize(n_pieces);
		std::vector<std::thread> threads;
		for(int t = 0; t < n_pieces; ++ t) {
			const int64_t n_first = n_rest * t / n_pieces, n_last = n_rest * (t + 1) / n_pieces;
			auto Work = [&runs, &order, t, n_first, n_last, n_mode, nc, ptr, brow]() {
				build_run(runs[t], order.data(), n_first, n_last, n_mode, nc, ptr, brow);
			};
			if(t + 1 < n_pieces)
				threads.emplace_back(Work);
			else
				Work();
		}
		for(size_t t = 0; t < threads.size(); ++ t)
			threads[t].join();
		for(int t = 0; t < n_pieces; ++ t) {
			n_tiles += int64_t(runs[t].tile_size.size());
			n_tile_slots += int64_t(runs[t].slot_key.size());
			n_tile_points += int64_t(runs[t].tile_lm.size());
			n_tile_pairs += int64_t(runs[t].lm_slot.size());
		}
	}
	BUILD_PHASE("tiles");
	const int64_t n_slots = n_run_slots + n_tile_slots;
	if((!n_tiles && jobs.empty()) || (n_mode < 0 && 2 * (n_tile_pairs + n_run_pairs) < n_all_pairs) || n_slots > INT32_MAX) {
		memset(T.n_run_jobs, 0, sizeof(T.n_run_jobs));
		memset(T.n_run_job_first, 0, sizeof(T.n_run_job_first));
		return; // the lists keep everything
	}
	// the run tables go to the device from here on, beside the rest of this analysis and of the caller's (round 5: 19 ms of C5's
	// cold call -- 50 MB out of pageable vectors --; schur_tiles_join waits for them).  The thread owns what it reads.
	{
		int n_device = 0;
		SLAMPP_HIP_CHECK(hipGetDevice(&n_device));
		T.p_run_upload = std::make_shared<TRunUpload>();
		TRunUpload *p_up = T.p_run_upload.get();
		p_up->jobs.swap(jobs);
		p_up->run_lm.swap(run_lm);
		p_up->run_k.swap(run_k);
		CSchurTiles *p_tiles = &T;
		auto Upload_Runs = [p_up, p_tiles, n_device, stream, ptr, nc, n_ablocks, DC, DP]() {
			try {
				SLAMPP_HIP_CHECK(hipSetDevice(n_device));
				p_tiles->d_run_jobs.Upload(p_up->jobs, stream);
				p_tiles->d_run_lm.Upload(p_up->run_lm, stream);
				p_tiles->d_run_k.Upload(p_up->run_k, stream);
				raw_vector<int64_t> run_rec(p_up->run_lm.size());
				{ // offset of every landmark's first U block in the values (a read of ptr[] per landmark, in hash order: a few threads)
					const int64_t n = int64_t(run_rec.size());
					const int n_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), n / 65536))); // (round 6: eight -- the analysis waits for this thread at its end)
					int64_t *p_rec = run_rec.data();
					const int32_t *p_lm = p_up->run_lm.data();
					auto Fill = [=](int64_t n_first, int64_t n_last) {
						for(int64_t i = n_first; i < n_last; ++ i) {
							const int64_t pt = p_lm[i], o0 = ptr[nc + pt] - ptr[nc] - pt;
							p_rec[i] = n_ablocks * DC * DC + o0 * DC * DP + pt * DP * DP;
						}
					};
					std::vector<std::thread> threads;
					for(int t = 0; t + 1 < n_workers; ++ t)
						threads.emplace_back(Fill, n * t / n_workers, n * (t + 1) / n_workers);
					Fill(n * (n_workers - 1) / n_workers, n);
					for(size_t t = 0; t < threads.size(); ++ t)
						threads[t].join();
				}
				p_tiles->d_run_rec.Upload(run_rec, stream);
				SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // (run_rec lives in this scope)
			} catch(...) {
				p_up->p_error = std::current_exception();
			}
		};
		if(p_up->run_lm.size() >= (size_t(1) << 18))
			p_up->t = std::thread(Upload_Runs);
		else
			Upload_Runs(); // (a small system: a thread's start-up is what it would save)
	}
	for(size_t i = 0; i < trash.size(); ++ i)
		T.trash.emplace_back(std::move(trash[i])); // (the caller frees them behind the analysis: schur.hip)
	trash.clear();
	const size_t n_run_jobs_all = T.p_run_upload->jobs.size();
	T.n_tiles = n_tiles;
	T.n_slots = n_slots;
	T.n_run_points = n_run_points;
	T.n_tile_points = n_tile_points;
	T.n_list_points = np - n_tile_points - n_run_points;
	T.n_tile_pairs = n_tile_pairs + n_run_pairs;
	for(size_t t = 0; t < runs.size(); ++ t) {
		for(size_t i = 0; i < runs[t].tile_slots.size(); ++ i)
			T.n_max_slots = std::max<int64_t>(T.n_max_slots, runs[t].tile_slots[i]);
		for(size_t i = 0; i < runs[t].tile_lm.size(); ++ i) {
			const int64_t pt = runs[t].tile_lm[i];
			T.n_max_k = std::max<int64_t>(T.n_max_k, ptr[nc + pt + 1] - ptr[nc + pt] - 1);
		}
	}
	if(getenv("SLAMPP_HIP_PLAN_TIMING")) {
		fprintf(stderr, "[schur] of %lld landmarks %lld in runs (%zu jobs), %lld in %lld tiles (largest %lld blocks); %lld partial blocks; "
			"%lld of %lld contributions\n", (long long)np, (long long)n_run_points, n_run_jobs_all, (long long)n_tile_points,
			(long long)n_tiles, (long long)T.n_max_slots, (long long)n_slots, (long long)T.n_tile_pairs, (long long)n_all_pairs);
	}

	std::vector<int32_t> tile_ptr(1, 0), tile_lm;
	std::vector<int64_t> tile_slot_ptr(1, n_run_slots), pair_ptr(np + 1, 0);
	std::vector<uint8_t> lm_slot;
	tile_lm.reserve(n_tile_points);
	slot_key.reserve(n_slots);
	lm_slot.reserve(n_tile_pairs);
	for(size_t t = 0; t < runs.size(); ++ t) {
		const TTileRun &R = runs[t];
		for(size_t i = 0; i < R.tile_size.size(); ++ i) {
			tile_ptr.push_back(tile_ptr.back() + R.tile_size[i]);
			tile_slot_ptr.push_back(tile_slot_ptr.back() + R.tile_slots[i]);
		}
		tile_lm.insert(tile_lm.end(), R.tile_lm.begin(), R.tile_lm.end());
		slot_key.insert(slot_key.end(), R.slot_key.begin(), R.slot_key.end());
		lm_slot.insert(lm_slot.end(), R.lm_slot.begin(), R.lm_slot.end());
	}
	{
		// where a tile landmark's slots start: they were appended in tile order (only the start of a range is ever read)
		int64_t n_off = 0;
		for(size_t i = 0; i < tile_lm.size(); ++ i) {
			const int64_t pt = tile_lm[i], k = ptr[nc + pt + 1] - ptr[nc + pt] - 1;
			pair_ptr[pt] = n_off;
			n_off += k * (k + 1) / 2;
			handled[pt] = 1;
		}
		pair_ptr[np] = n_off;
	}
	// the partial blocks of every block of S: key -> block by binary search on the sorted block list
	const int64_t n_sblocks = int64_t(sb_row.size());
	std::vector<int64_t> sb_keys(n_sblocks);
	for(int64_t i = 0; i < n_sblocks; ++ i)
		sb_keys[i] = int64_t(sb_col[i]) * nc + sb_row[i];
	// (a table over all camera pairs where that is small: a binary search per partial block and per contribution of the
	// list landmarks was a third of the analysis of the Venice-like leg)
	std::vector<int32_t> key_sb;
	if(nc * nc <= (int64_t(1) << 24)) {
		key_sb.assign(size_t(nc * nc), -1);
		for(int64_t i = 0; i < n_sblocks; ++ i)
			key_sb[sb_keys[i]] = int32_t(i);
	}
	auto Block_Of = [&](int64_t n_key) -> size_t {
		if(!key_sb.empty())
			return (key_sb[n_key] >= 0)? size_t(key_sb[n_key]) : sb_keys.size();
		const size_t k = size_t(std::lower_bound(sb_keys.begin(), sb_keys.end(), n_key) - sb_keys.begin());
		return (k < sb_keys.size() && sb_keys[k] == n_key)? k : sb_keys.size();
	};
	std::vector<int32_t> slot_sb(n_slots);
	std::vector<int64_t> sb_cnt(n_sblocks + 1, 0);
	for(int64_t g = 0; g < n_slots; ++ g) {
		const size_t k = Block_Of(slot_key[g]);
		if(k == sb_keys.size())
			throw std::logic_error("reduced camera system: a partial block is not in the block list");
		slot_sb[g] = int32_t(k);
		++ sb_cnt[k + 1];
	}
	std::vector<int64_t> rb_ptr(1, 0);
	std::vector<int32_t> rb_sb, rb_part(n_slots);
	{
		std::vector<int64_t> start(n_sblocks, -1);
		for(int64_t b = 0; b < n_sblocks; ++ b) {
			if(sb_cnt[b + 1]) {
				start[b] = rb_ptr.back();
				rb_sb.push_back(int32_t(b));
				rb_ptr.push_back(rb_ptr.back() + sb_cnt[b + 1]);
			}
		}
		for(int64_t g = 0; g < n_slots; ++ g) // ascending partial index inside every list
			rb_part[start[slot_sb[g]] ++] = int32_t(g);
	}
	T.n_rb = int64_t(rb_sb.size());
	BUILD_PHASE("partial block lists");

	if(T.n_tiles) { // (the tile kernel's tables: 16 bytes per landmark that nobody reads where every landmark is in a run -- 32 MB of C5's cold call)
		T.d_tile_ptr.Upload(tile_ptr, stream);
		T.d_tile_lm.Upload(tile_lm, stream);
		T.d_tile_slot_ptr.Upload(tile_slot_ptr, stream);
		T.d_pair_ptr.Upload(pair_ptr, stream);
		T.d_lm_slot.Upload(lm_slot, stream);
	}
	T.d_rb_ptr.Upload(rb_ptr, stream);
	T.d_rb_part.Upload(rb_part, stream);
	T.d_rb_sb.Upload(rb_sb, stream);
	T.d_P.Alloc(size_t(n_slots) * DC * DC);
	T.d_R.Alloc(size_t(n_slots) * DC);
	const raw_vector<uint8_t> &in_tile = handled;
	BUILD_PHASE("uploads (runs)");

	// the landmarks that stay with the contribution lists: lists of their own, for the blocks of S they touch
	std::vector<int64_t> xsb_ptr, xent_uoff, xcam_ptr;
	std::vector<int32_t> xsb_map, xent_a, xcam_obs;
	if(T.n_list_points > 0) {
		T.b_hybrid = true;
		const int64_t ubase = n_ablocks * DC * DC;
		// Two passes over the list landmarks (count, fill), each cut into landmark ranges for a few threads: thread t's
		// entries of a list follow those of thread t - 1, so every list comes out in (landmark, a, b) order, as a serial pass
		// would leave it (the Venice-like leg: 3.6 M entries, 40 ms as one pass through an intermediate array, a quarter of
		// the whole analysis)
		std::vector<int32_t> list_points;
		for(int64_t pt = 0; pt < np; ++ pt) {
			if(!in_tile[pt])
				list_points.push_back(int32_t(pt));
		}
		const int64_t n_list = int64_t(list_points.size());
		const int n_workers = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), n_list / 2048)));
		std::vector<std::vector<int64_t> > cnt_t(n_workers, std::vector<int64_t>(size_t(n_sblocks), 0)), cam_t(n_workers, std::vector<int64_t>(size_t(nc), 0));
		auto For_Ranges = [&](const std::function<void(int, int64_t, int64_t)> &r_work) {
			std::vector<std::thread> threads;
			for(int t = 0; t < n_workers; ++ t) {
				const int64_t n_first = n_list * t / n_workers, n_last = n_list * (t + 1) / n_workers;
				if(t + 1 < n_workers)
					threads.emplace_back(r_work, t, n_first, n_last);
				else
					r_work(t, n_first, n_last);
			}
			for(size_t t = 0; t < threads.size(); ++ t)
				threads[t].join();
		};
		std::atomic<int> n_missing(0);
		For_Ranges([&](int t, int64_t n_first, int64_t n_last) {
			for(int64_t i = n_first; i < n_last; ++ i) {
				const int64_t pt = list_points[i], k0 = ptr[nc + pt], k = ptr[nc + pt + 1] - k0 - 1;
				for(int64_t a = 0; a < k; ++ a) {
					++ cam_t[t][brow[k0 + a]];
					for(int64_t b = a; b < k; ++ b) {
						const size_t sb = Block_Of(int64_t(brow[k0 + a]) * nc + brow[k0 + b]);
						if(sb == sb_keys.size())
							++ n_missing;
						else
							++ cnt_t[t][sb];
					}
				}
			}
		});
		if(n_missing.load())
			throw std::logic_error("reduced camera system: a contribution's block is not in the block list");
		xsb_ptr.push_back(0);
		std::vector<int64_t> start(n_sblocks, -1);
		for(int64_t b = 0; b < n_sblocks; ++ b) {
			int64_t n_total = 0;
			for(int t = 0; t < n_workers; ++ t)
				n_total += cnt_t[t][b];
			if(n_total) {
				start[b] = xsb_ptr.back();
				xsb_map.push_back(int32_t(b));
				xsb_ptr.push_back(xsb_ptr.back() + n_total);
			}
			int64_t n_at = start[b];
			for(int t = 0; t < n_workers; ++ t) { // cnt_t[t][b] becomes where thread t's entries of list b start
				const int64_t n_mine = cnt_t[t][b];
				cnt_t[t][b] = n_at;
				n_at += n_mine;
			}
		}
		xcam_ptr.assign(nc + 1, 0);
		for(int64_t c = 0; c < nc; ++ c) {
			int64_t n_at = xcam_ptr[c];
			for(int t = 0; t < n_workers; ++ t) {
				const int64_t n_mine = cam_t[t][c];
				cam_t[t][c] = n_at;
				n_at += n_mine;
			}
			xcam_ptr[c + 1] = n_at;
		}
		xent_a.resize(size_t(xsb_ptr.back()));
		xent_uoff.resize(size_t(xsb_ptr.back()));
		xcam_obs.resize(size_t(xcam_ptr[nc]));
		For_Ranges([&](int t, int64_t n_first, int64_t n_last) {
			for(int64_t i = n_first; i < n_last; ++ i) {
				const int64_t pt = list_points[i], k0 = ptr[nc + pt], k = ptr[nc + pt + 1] - k0 - 1, o0 = k0 - ptr[nc] - pt;
				for(int64_t a = 0; a < k; ++ a) {
					xcam_obs[size_t(cam_t[t][brow[k0 + a]] ++)] = int32_t(o0 + a);
					for(int64_t b = a; b < k; ++ b) {
						const size_t sb = Block_Of(int64_t(brow[k0 + a]) * nc + brow[k0 + b]);
						const int64_t d = cnt_t[t][sb] ++;
						xent_a[size_t(d)] = int32_t(o0 + a);
						// the U block of observation b: the values hold [U .. U | C] per landmark
						xent_uoff[size_t(d)] = ubase + (o0 + b) * DC * DP + pt * DP * DP;
					}
				}
			}
		});
		const std::vector<int32_t> &xpoints = list_points;
		T.d_xpoints.Upload(xpoints, stream);
		T.n_xobs = int64_t(xcam_obs.size());
		T.n_xblocks = int64_t(xsb_map.size());
		T.n_xentries = int64_t(xent_a.size());
		T.d_xsb_ptr.Upload(xsb_ptr, stream);
		T.d_xsb_map.Upload(xsb_map, stream);
		T.d_xent_a.Upload(xent_a, stream);
		T.d_xent_uoff.Upload(xent_uoff, stream);
		T.d_xcam_ptr.Upload(xcam_ptr, stream);
		T.d_xcam_obs.Upload(xcam_obs, stream);
	}
	BUILD_PHASE("lists of the rest");
	SLAMPP_HIP_CHECK(hipStreamSynchronize(stream)); // the host vectors live on this stack frame
	BUILD_PHASE("sync");
#undef BUILD_PHASE
	T.b_enabled = true;
}

} // namespace slampp

#include "preload.h"
SLAMPP_PRELOAD_UNIT(schur_tiles) // (the handle's bring-up thread loads this unit's code object: capi.hip)
