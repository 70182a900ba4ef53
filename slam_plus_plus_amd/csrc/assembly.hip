// assembly.hip -- Lambda and eta assembled on the device from per-edge Jacobians.
//
// Stands where the reference computes the per-edge Hessian blocks on the host and runs its reduction plan
// (/root/reference/include/slam/BaseTypes_Binary.h:759-840 Calculate_Hessians_v2;
//  include/slam/NonlinearSolver_Lambda_Base.h:1634-1688 Refresh_Lambda, :1520-1580 unary factor), for one
// homogeneous set of binary edges (J0: rd x d0, J1: rd x d1, Sigma^-1: rd x rd, error: rd, robust weight w):
//   off-diagonal block (min id, max id)   = J0^T (w Sigma^-1) J1          (transposed if id0 > id1, :779-806)
//   diagonal block of vertex 0 / vertex 1 += J0^T (w Sigma^-1) J0  /  J1^T (w Sigma^-1) J1
//   eta of vertex 0                       += w^2 J0^T Sigma^-1 e   (sic: the reference multiplies by w twice, :813-815)
//   eta of vertex 1                       += w   J1^T Sigma^-1 e                                           (:836-838)
//   diagonal block / eta of the anchor    += U^T U / unary error   (NonlinearSolver_Lambda_Base.h:1551-1567)
// One wave per block of Lambda sums the contributions of its edges in a fixed order from host-built lists
// (no atomics, bit-reproducible) and writes the block where the solver's packed value array expects it, so
// that factor_solve_device can run on the result without Lambda ever visiting the host.  HBM-bound.
#include "solver.h"

#include <algorithm>
#include <cstring>

namespace slampp {

struct TAsmBlk { // 32 B: one block of Lambda and the edges that contribute to it
	int64_t dst;       // offset in the packed Lambda values
	int64_t e0;        // first entry of its edge list
	int64_t eta_off;   // diagonal blocks: scalar offset of the vertex in eta
	int32_t ne;        // number of entries
	int16_t rows, cols;
};

struct CAssemblyState {
	int64_t n_edges;
	int d0, d1, rd;
	int64_t n_offdiag, n_diag;
	CDevArray<TAsmBlk> d_offdiag, d_diag;
	CDevArray<int32_t> d_diag_vertex; // vertex of every record of d_diag (the lists below took some of them out)
	CDevArray<TAsmBlk> d_long;      // vertices with long edge lists (BA cameras, hubs): summed edge-parallel by their own kernel
	CDevArray<int32_t> d_long_vertex;
	int64_t n_long;
	CDevArray<TAsmBlk> d_small;     // low-dimensional vertices with short lists (BA landmarks): several share a wave
	CDevArray<int32_t> d_small_vertex;
	int64_t n_small;
	int n_small_elems;              // lanes per vertex in that kernel: d^2 + d of the largest of them
	int n_long_dim;                 // their common dimension (0: none qualify)
	int n_off_elems;                // largest off-diagonal block, in elements: 64 / that many blocks share a wave
	CDevArray<int32_t> d_entries;   // edge index * 2 + (off-diagonal: flipped; diagonal: side)
	CDevArray<double> d_unary;      // [64 + 8]: U^T U (column-major d x d), unary error
	double h_unary[72];             // what d_unary holds
	bool b_unary_valid;
	slampp_hip_solver *p_solver;
};

void assembly_destroy(CAssemblyState *p) { delete p; }

// (residual dimension, vertex dimension) pairs the edge-parallel kernel is instantiated for
static bool long_kernel_exists(int rd, int d)
{
	return (rd == 2 && (d == 6 || d == 7 || d == 3)) || (rd == 3 && d == 3) || (rd == 6 && d == 6) || (rd == 7 && d == 7);
}

CAssemblyState *assembly_setup(slampp_hip_solver &s, int64_t n_edges, const int64_t *v0, const int64_t *v1, int rd)
{
	if(!s.b_has_structure)
		throw std::invalid_argument("assembly_setup: set_structure was not called");
	if(n_edges <= 0 || !v0 || !v1 || rd <= 0 || rd > 8)
		throw std::invalid_argument("assembly_setup: bad edge set");
	const int64_t n = int64_t(s.cumsum.size()) - 1;
	const int64_t *cs = s.cumsum.data(), *ptr = s.bcol_ptr.data();
	const int32_t *brow = s.brow.data();
	for(int64_t e = 0; e < n_edges; ++ e) {
		if(v0[e] < 0 || v0[e] >= n || v1[e] < 0 || v1[e] >= n || v0[e] == v1[e])
			throw std::invalid_argument("assembly_setup: edge refers to a vertex outside Lambda, or to one vertex twice");
	}
	const int d0 = int(cs[v0[0] + 1] - cs[v0[0]]), d1 = int(cs[v1[0] + 1] - cs[v1[0]]);
	if(d0 > 7 || d1 > 7)
		throw std::domain_error("assembly: vertex dimensions above 7 are not supported");
	for(int64_t e = 0; e < n_edges; ++ e) {
		if(cs[v0[e] + 1] - cs[v0[e]] != d0 || cs[v1[e] + 1] - cs[v1[e]] != d1)
			throw std::domain_error("assembly: all edges of a set must join vertices of the same two dimensions");
	}
	// value offset of every block of Lambda
	std::vector<int64_t> voff(ptr[n] + 1, 0);
	for(int64_t c = 0; c < n; ++ c)
		for(int64_t k = ptr[c]; k < ptr[c + 1]; ++ k)
			voff[k + 1] = voff[k] + (cs[brow[k] + 1] - cs[brow[k]]) * (cs[c + 1] - cs[c]);
	// edge lists per block: count, then fill (stable: edges stay in their given order inside a list)
	std::vector<int64_t> blk_of_edge(n_edges);
	std::vector<int32_t> cnt_off(ptr[n], 0), cnt_diag(n, 0);
	for(int64_t e = 0; e < n_edges; ++ e) {
		const int64_t r = std::min(v0[e], v1[e]), c = std::max(v0[e], v1[e]);
		const int32_t *b = std::lower_bound(brow + ptr[c], brow + ptr[c + 1], int32_t(r));
		if(b == brow + ptr[c + 1] || *b != r)
			throw std::invalid_argument("assembly_setup: Lambda has no block for an edge");
		blk_of_edge[e] = b - brow;
		++ cnt_off[b - brow];
		++ cnt_diag[v0[e]];
		++ cnt_diag[v1[e]];
	}
	std::vector<TAsmBlk> offdiag, diag(n);
	std::vector<int64_t> off_slot(ptr[n], -1);
	int64_t n_entries = 0;
	for(int64_t c = 0; c < n; ++ c) {
		for(int64_t k = ptr[c]; k < ptr[c + 1]; ++ k) {
			if(brow[k] == c)
				continue;
			TAsmBlk b;
			b.dst = voff[k]; b.e0 = n_entries; b.ne = cnt_off[k]; b.eta_off = 0;
			b.rows = int16_t(cs[brow[k] + 1] - cs[brow[k]]); b.cols = int16_t(cs[c + 1] - cs[c]);
			n_entries += cnt_off[k];
			off_slot[k] = int64_t(offdiag.size());
			offdiag.push_back(b); // blocks without an edge are written as zeros
		}
	}
	for(int64_t v = 0; v < n; ++ v) {
		if(ptr[v + 1] == ptr[v] || brow[ptr[v + 1] - 1] != v)
			throw std::invalid_argument("assembly_setup: a diagonal block is missing");
		TAsmBlk &b = diag[v];
		b.dst = voff[ptr[v + 1] - 1]; b.e0 = n_entries; b.ne = cnt_diag[v]; b.eta_off = cs[v];
		b.rows = b.cols = int16_t(cs[v + 1] - cs[v]);
		if(b.rows > 7)
			throw std::domain_error("assembly: vertex dimensions above 7 are not supported");
		n_entries += cnt_diag[v];
	}
	if(n_edges >= (int64_t(1) << 30))
		throw std::domain_error("assembly: too many edges");
	std::vector<int32_t> entries(n_entries);
	{
		std::vector<int64_t> fill_off(offdiag.size()), fill_diag(n);
		for(size_t i = 0; i < offdiag.size(); ++ i) fill_off[i] = offdiag[i].e0;
		for(int64_t v = 0; v < n; ++ v) fill_diag[v] = diag[v].e0;
		for(int64_t e = 0; e < n_edges; ++ e) {
			entries[fill_off[off_slot[blk_of_edge[e]]] ++] = int32_t(e * 2 + (v0[e] > v1[e]));
			entries[fill_diag[v0[e]] ++] = int32_t(e * 2);
			entries[fill_diag[v1[e]] ++] = int32_t(e * 2 + 1);
		}
	}
	// Three kernels share the vertices: long lists go to the edge-parallel kernel (one lane per edge, tree reduction)
	// when it exists for their dimension; low-dimensional vertices with short lists (the landmarks of a BA system:
	// 12 of 64 lanes busy otherwise) are packed several to a wave when the residual is small enough to unroll; the
	// rest stay with one wave per vertex
	std::vector<TAsmBlk> long_blks, small_blks, plain_blks;
	std::vector<int32_t> long_vertex, small_vertex, plain_vertex;
	int n_long_dim = 0, n_small_elems = 0;
	{
		enum { LONG_LIST = 96 };
		const bool b_can_pack = rd == 2 || rd == 3;
		for(int64_t v = 0; v < n; ++ v) {
			const int d = diag[v].rows;
			if(diag[v].ne >= LONG_LIST && long_kernel_exists(rd, d) && (!n_long_dim || d == n_long_dim)) {
				n_long_dim = d;
				long_blks.push_back(diag[v]);
				long_vertex.push_back(int32_t(v));
			} else if(b_can_pack && 2 * (d * d + d) <= 64) {
				n_small_elems = std::max(n_small_elems, d * d + d);
				small_blks.push_back(diag[v]);
				small_vertex.push_back(int32_t(v));
			} else {
				plain_blks.push_back(diag[v]);
				plain_vertex.push_back(int32_t(v));
			}
		}
	}
	int n_off_elems = 1;
	for(size_t i = 0; i < offdiag.size(); ++ i) {
		n_off_elems = std::max(n_off_elems, int(offdiag[i].rows) * int(offdiag[i].cols));
		if(offdiag[i].ne == 1)
			offdiag[i].e0 = entries[offdiag[i].e0]; // a single entry rides in the record: one dependent load less
	}
	CAssemblyState *p = new CAssemblyState();
	try {
		p->n_long = int64_t(long_blks.size()); p->n_long_dim = n_long_dim; p->n_off_elems = n_off_elems;
		if(!long_blks.empty()) {
			p->d_long.Upload(long_blks, s.stream);
			p->d_long_vertex.Upload(long_vertex, s.stream);
		}
		p->n_small = int64_t(small_blks.size()); p->n_small_elems = n_small_elems;
		if(!small_blks.empty()) {
			p->d_small.Upload(small_blks, s.stream);
			p->d_small_vertex.Upload(small_vertex, s.stream);
		}
		p->n_edges = n_edges; p->d0 = d0; p->d1 = d1; p->rd = rd;
		p->n_offdiag = int64_t(offdiag.size()); p->n_diag = int64_t(plain_blks.size());
		p->d_offdiag.Upload(offdiag, s.stream);
		if(!plain_blks.empty()) {
			p->d_diag.Upload(plain_blks, s.stream);
			p->d_diag_vertex.Upload(plain_vertex, s.stream);
		}
		p->d_entries.Upload(entries, s.stream);
		p->d_unary.Alloc(72);
		p->b_unary_valid = false;
		p->p_solver = &s;
		SLAMPP_HIP_CHECK(hipStreamSynchronize(s.stream));
	} catch(...) {
		delete p;
		throw;
	}
	return p;
}

__device__ __forceinline__ void wave_sync_lds()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// t[b] = sum_a J[a + col * rd] * S[a + b * rd]  (row `col` of J^T Sigma^-1), b < rd
template <int RD>
__device__ __forceinline__ void jt_sigma_row(const double *__restrict__ J, const double *__restrict__ S, int col, int rd,
	double (&t)[RD? RD : 8])
{
	enum { N = RD? RD : 8 };
	double j[N];
	#pragma unroll
	for(int a = 0; a < N; ++ a)
		j[a] = (RD || a < rd)? J[a + col * rd] : 0.0;
	#pragma unroll
	for(int b = 0; b < N; ++ b) {
		double sum = 0;
		if(RD || b < rd) {
			#pragma unroll
			for(int a = 0; a < N; ++ a)
				if(RD || a < rd) sum += j[a] * S[a + b * rd];
		}
		t[b] = sum;
	}
}

template <int RD>
__device__ __forceinline__ double dot_rd(const double (&t)[RD? RD : 8], const double *__restrict__ v, int rd)
{
	enum { N = RD? RD : 8 };
	double sum = 0;
	#pragma unroll
	for(int b = 0; b < N; ++ b)
		if(RD || b < rd) sum += t[b] * v[b];
	return sum;
}

// off-diagonal blocks (row vertex < column vertex): sum over the block's edges of J_row^T (w S) J_col, one wave per
// block; its record and loop bounds stay in scalar registers.  An edge's J0, J1 and Sigma^-1 are fetched with one
// coalesced load each into LDS (a lane computing its element straight from global memory issued rd^2 + 2 rd loads per
// edge, 48 for an SE(3) edge), M = Sigma^-1 J1 is formed once per edge by the wave, and element (c0, c1) of
// J0^T M is rd LDS reads per operand.
template <int RD>
__global__ void __launch_bounds__(64)
assemble_offdiag_kernel(const TAsmBlk *__restrict__ blks, const int32_t *__restrict__ entries, int d0, int d1, int n_rd,
	const double *__restrict__ J0, const double *__restrict__ J1, const double *__restrict__ Si,
	const double *__restrict__ wgt, double *values, int b_accumulate)
{
	__shared__ double s_j0[64], s_j1[64], s_s[64], s_m[64];
	const int rd = RD? RD : n_rd;
	const TAsmBlk bd = blks[blockIdx.x];
	const int el = threadIdx.x;
	const bool b_el = el < bd.rows * bd.cols;
	const int r = b_el? el % bd.rows : 0, q = b_el? el / bd.rows : 0;
	const int ma = el % rd, mc = el / rd; // this lane's element of M = S J1 (rd x d1)
	double acc = (b_accumulate && b_el)? values[bd.dst + el] : 0.0;
	for(int i = 0; i < bd.ne; ++ i) {
		const int32_t ent = (bd.ne == 1)? int32_t(bd.e0) : entries[bd.e0 + i];
		const int64_t e = ent >> 1;
		const bool b_flip = ent & 1; // vertex 0 has the larger id: the stored block is (J0^T w S J1)^T
		const double v0 = (el < rd * d0)? J0[e * rd * d0 + el] : 0.0, v1 = (el < rd * d1)? J1[e * rd * d1 + el] : 0.0,
			vs = (el < rd * rd)? Si[e * rd * rd + el] : 0.0;
		const double w = wgt? wgt[e] : 1.0;
		wave_sync_lds(); // the previous edge's operands have been consumed
		s_j0[el] = v0;
		s_j1[el] = v1;
		s_s[el] = vs;
		wave_sync_lds();
		double m = 0;
		if(el < rd * d1) {
			#pragma unroll
			for(int b2 = 0; b2 < (RD? RD : 8); ++ b2)
				if(RD || b2 < rd) m += s_s[ma + b2 * rd] * s_j1[b2 + mc * rd];
		}
		s_m[el] = m;
		wave_sync_lds();
		// element (c0, c1) of J0^T (w S) J1 lands on (r, q) of the stored block
		const int c0 = b_flip? q : r, c1 = b_flip? r : q;
		double sum = 0;
		#pragma unroll
		for(int a2 = 0; a2 < (RD? RD : 8); ++ a2)
			if(RD || a2 < rd) sum += s_j0[a2 + c0 * rd] * s_m[a2 + c1 * rd];
		acc += sum * w;
	}
	if(b_el)
		values[bd.dst + el] = acc; // a structural block without an edge becomes zeros
}

// The same for small blocks with small residuals (a BA system: 6 x 3 blocks, one edge each, 2-d residuals): as many
// blocks share a wave as fit, lane = (slot, element) -- 2 M single-block waves with 18 busy lanes took 0.58 ms at C4,
// three to a wave 0.27 ms.  (Working on several such groups per wave with their loads batched was slower: 0.32 ms.)
template <int RD>
__global__ void __launch_bounds__(64)
assemble_offdiag_packed_kernel(const TAsmBlk *__restrict__ blks, int64_t n_blks, int n_per_wave, int n_slot_elems,
	const int32_t *__restrict__ entries, int d0, int d1,
	const double *__restrict__ J0, const double *__restrict__ J1, const double *__restrict__ Si,
	const double *__restrict__ wgt, double *values, int b_accumulate)
{
	const int sub = int(threadIdx.x) / n_slot_elems, el = int(threadIdx.x) - sub * n_slot_elems;
	const int64_t bi = int64_t(blockIdx.x) * n_per_wave + sub;
	if(sub >= n_per_wave || bi >= n_blks)
		return;
	const TAsmBlk bd = blks[bi];
	if(el >= bd.rows * bd.cols)
		return;
	const int r = el % bd.rows, q = el / bd.rows;
	double acc = b_accumulate? values[bd.dst + el] : 0.0;
	for(int i = 0; i < bd.ne; ++ i) {
		const int32_t ent = (bd.ne == 1)? int32_t(bd.e0) : entries[bd.e0 + i];
		const int64_t e = ent >> 1;
		const int c0 = (ent & 1)? q : r, c1 = (ent & 1)? r : q;
		double t[RD];
		jt_sigma_row<RD>(J0 + e * RD * d0, Si + e * RD * RD, c0, RD, t);
		acc += dot_rd<RD>(t, J1 + e * RD * d1 + c1 * RD, RD) * (wgt? wgt[e] : 1.0);
	}
	values[bd.dst + el] = acc;
}

// one wave per vertex: diagonal block (lanes r + q d) and eta (lanes 56 + q); an incident edge's Jacobian, Sigma^-1 and
// error are fetched with one coalesced load each into LDS, M = Sigma^-1 [J | e] is formed once per edge by the wave
// (lanes a + c rd the columns of J, lanes 56 + a the error), and a lane's element of J^T M is rd LDS reads per operand
template <int RD>
__global__ void __launch_bounds__(64)
assemble_diag_kernel(const TAsmBlk *__restrict__ blks, const int32_t *__restrict__ vertex_of,
	const int32_t *__restrict__ entries, int d0, int d1, int n_rd,
	const double *__restrict__ J0, const double *__restrict__ J1, const double *__restrict__ Si,
	const double *__restrict__ err, const double *__restrict__ wgt, int unary_vertex, const double *__restrict__ unary,
	double *values, double *eta, int b_accumulate)
{
	__shared__ double s_j[64], s_s[64], s_e[8], s_m[64];
	const int rd = RD? RD : n_rd;
	const TAsmBlk bd = blks[blockIdx.x];
	const int lane = threadIdx.x;
	const int d = bd.rows;
	const bool b_blk = lane < d * d, b_y = lane >= 56 && lane < 56 + d;
	const int r = b_blk? lane % d : (b_y? lane - 56 : 0), q = b_blk? lane / d : 0;
	const int ma = (lane < 56)? lane % rd : lane - 56, mc = lane / rd; // this lane's element of M
	double *p_dst = b_blk? values + bd.dst + lane : eta + bd.eta_off + r;
	double acc = (b_accumulate && (b_blk || b_y))? *p_dst : 0.0;
	for(int i = 0; i < bd.ne; ++ i) { // (requesting the records of four edges together changed nothing: 89 -> 91 us at C3)
		const int32_t ent = entries[bd.e0 + i];
		const int64_t e = ent >> 1;
		const int side = ent & 1;
		const double *J = side? J1 + e * rd * d1 : J0 + e * rd * d0;
		const double vj = (lane < rd * d)? J[lane] : 0.0, vs = (lane < rd * rd)? Si[e * rd * rd + lane] : 0.0,
			ve = (lane < rd)? err[e * rd + lane] : 0.0;
		const double w = wgt? wgt[e] : 1.0;
		wave_sync_lds(); // the previous edge's operands have been consumed
		s_j[lane] = vj;
		s_s[lane] = vs;
		if(lane < 8)
			s_e[lane] = ve;
		wave_sync_lds();
		double m = 0;
		if(lane < rd * d) { // (S J)(ma, mc)
			#pragma unroll
			for(int b2 = 0; b2 < (RD? RD : 8); ++ b2)
				if(RD || b2 < rd) m += s_s[ma + b2 * rd] * s_j[b2 + mc * rd];
		} else if(lane >= 56 && lane < 56 + rd) { // (S e)(ma)
			#pragma unroll
			for(int b2 = 0; b2 < (RD? RD : 8); ++ b2)
				if(RD || b2 < rd) m += s_s[ma + b2 * rd] * s_e[b2];
		}
		s_m[lane] = m; // lanes 56 .. 56 + rd - 1: S e; rd d <= 56, so the two ranges do not meet
		wave_sync_lds();
		double sum = 0;
		#pragma unroll
		for(int a2 = 0; a2 < (RD? RD : 8); ++ a2)
			if(RD || a2 < rd) sum += s_j[a2 + r * rd] * (b_blk? s_m[a2 + q * rd] : s_m[56 + a2]);
		// the reference weights vertex 0's right-hand side twice (BaseTypes_Binary.h:813-815 against :836-838)
		acc += sum * ((b_blk || side)? w : w * w);
	}
	if(!b_blk && !b_y)
		return;
	if(vertex_of[blockIdx.x] == unary_vertex)
		acc += b_blk? unary[lane] : unary[64 + r];
	*p_dst = acc;
}

// The same for low-dimensional vertices with short lists and small residuals (the landmarks of a BA system): several
// vertices share a wave (lane = (slot, element), elements 0 .. d^2-1 the block, d^2 .. d^2+d-1 the right-hand side), and
// the operands of UNR edges are requested before the first of them is used
template <int RD, int UNR>
__global__ void __launch_bounds__(64)
assemble_diag_packed_kernel(const TAsmBlk *__restrict__ blks, const int32_t *__restrict__ vertex_of, int64_t n_blks,
	int n_per_wave, int n_slot_elems, const int32_t *__restrict__ entries, int d0, int d1,
	const double *__restrict__ J0, const double *__restrict__ J1, const double *__restrict__ Si,
	const double *__restrict__ err, const double *__restrict__ wgt, int unary_vertex, const double *__restrict__ unary,
	double *values, double *eta, int b_accumulate)
{
	const int sub = int(threadIdx.x) / n_slot_elems, el = int(threadIdx.x) - sub * n_slot_elems;
	const int64_t bi = int64_t(blockIdx.x) * n_per_wave + sub;
	if(sub >= n_per_wave || bi >= n_blks)
		return;
	const TAsmBlk bd = blks[bi];
	const int d = bd.rows;
	const bool b_blk = el < d * d, b_y = !b_blk && el < d * d + d;
	if(!b_blk && !b_y)
		return;
	const int r = b_blk? el % d : el - d * d, q = b_blk? el / d : 0;
	double *p_dst = b_blk? values + bd.dst + el : eta + bd.eta_off + r;
	double acc = b_accumulate? *p_dst : 0.0;
	for(int i0 = 0; i0 < bd.ne; i0 += UNR) {
		int32_t ent[UNR];
		#pragma unroll
		for(int u = 0; u < UNR; ++ u)
			ent[u] = (i0 + u < bd.ne)? entries[bd.e0 + i0 + u] : -1;
		double jr[UNR][RD], sm[UNR][RD * RD], v[UNR][RD], w[UNR];
		#pragma unroll
		for(int u = 0; u < UNR; ++ u) {
			const int32_t en = (ent[u] >= 0)? ent[u] : 0;
			const int64_t e = en >> 1;
			const double *J = (en & 1)? J1 + e * RD * d1 : J0 + e * RD * d0;
			#pragma unroll
			for(int a = 0; a < RD; ++ a) {
				jr[u][a] = (ent[u] >= 0)? J[a + r * RD] : 0.0;
				v[u][a] = (ent[u] >= 0)? (b_blk? J[a + q * RD] : err[e * RD + a]) : 0.0;
			}
			#pragma unroll
			for(int a = 0; a < RD * RD; ++ a)
				sm[u][a] = (ent[u] >= 0)? Si[e * RD * RD + a] : 0.0;
			const double we = wgt? wgt[e] : 1.0;
			w[u] = (ent[u] < 0)? 0.0 : ((b_blk || (en & 1))? we : we * we); // vertex 0's right-hand side is weighted twice (see above)
		}
		#pragma unroll
		for(int u = 0; u < UNR; ++ u) {
			double sum = 0; // the arithmetic and its order are those of the one-wave kernel
			#pragma unroll
			for(int b = 0; b < RD; ++ b) {
				double t = 0;
				#pragma unroll
				for(int a = 0; a < RD; ++ a)
					t += jr[u][a] * sm[u][a + b * RD];
				sum += t * v[u][b];
			}
			if(ent[u] >= 0)
				acc += sum * w[u];
		}
	}
	if(vertex_of[bi] == unary_vertex)
		acc += b_blk? unary[el] : unary[64 + r];
	*p_dst = acc;
}

// N contiguous doubles with 16-byte loads where the address allows it
template <int N>
__device__ __forceinline__ void load_record(double (&r_dst)[N], const double *__restrict__ p_src)
{
	typedef double v2f64 __attribute__((ext_vector_type(2)));
	if(N > 1 && (reinterpret_cast<uintptr_t>(p_src) & 15) == 0) {
		#pragma unroll
		for(int i = 0; i + 1 < N; i += 2) {
			const v2f64 v = *reinterpret_cast<const v2f64*>(p_src + i);
			r_dst[i] = v.x;
			r_dst[i + 1] = v.y;
		}
		if(N & 1)
			r_dst[N - 1] = p_src[N - 1];
	} else {
		#pragma unroll
		for(int i = 0; i < N; ++ i)
			r_dst[i] = p_src[i];
	}
}

// Vertices with long edge lists (a BA camera sees thousands of points; one wave walking its list edge by edge took
// 1.1 ms at 2 000 edges per camera): one workgroup per vertex, one LANE per edge -- each lane reads its edge's Jacobian,
// Sigma^-1 and error (contiguous per edge) and keeps the lower triangle of J^T (w S) J and the D entries of the
// right-hand side in registers; the lanes are then summed by a butterfly, the waves through LDS, both in a fixed
// order (bit-reproducible).
template <int RD, int D>
__global__ void __launch_bounds__(256)
assemble_long_kernel(const TAsmBlk *__restrict__ blks, const int32_t *__restrict__ vertex_of, const int32_t *__restrict__ entries,
	int d0, int d1, const double *__restrict__ J0, const double *__restrict__ J1, const double *__restrict__ Si,
	const double *__restrict__ err, const double *__restrict__ wgt, int unary_vertex, const double *__restrict__ unary,
	double *values, double *eta, int b_accumulate)
{
	enum { NT = D * (D + 1) / 2, NV = NT + D, W = 4 };
	__shared__ double s_part[W][NV];
	const TAsmBlk bd = blks[blockIdx.x];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	double acc[NV];
	#pragma unroll
	for(int i = 0; i < NV; ++ i)
		acc[i] = 0;
	for(int i = tid; i < bd.ne; i += 64 * W) {
		const int32_t ent = entries[bd.e0 + i];
		const int64_t e = ent >> 1;
		const int side = ent & 1;
		const double *J = side? J1 + e * RD * d1 : J0 + e * RD * d0;
		const double *S = Si + e * RD * RD, *ev = err + e * RD;
		const double w = wgt? wgt[e] : 1.0;
		double j[RD][D], sm[RD][RD], se[RD];
		{
			// a lane's scattered loads are paid per instruction: 16 bytes each where the records start at multiples of 16 bytes
			double jf[RD * D], sf[RD * RD], ef[RD];
			load_record<RD * D>(jf, J);
			load_record<RD * RD>(sf, S);
			load_record<RD>(ef, ev);
			#pragma unroll
			for(int c = 0; c < D; ++ c)
				#pragma unroll
				for(int a = 0; a < RD; ++ a)
					j[a][c] = jf[a + c * RD];
			#pragma unroll
			for(int b = 0; b < RD; ++ b)
				#pragma unroll
				for(int a = 0; a < RD; ++ a)
					sm[a][b] = sf[a + b * RD];
			#pragma unroll
			for(int b = 0; b < RD; ++ b) { // (Sigma^-1 e)[b], by columns as the one-wave kernel sums it: sum_a .. S[a + b rd]
				double sum = 0;
				#pragma unroll
				for(int a = 0; a < RD; ++ a)
					sum += ef[a] * sm[a][b];
				se[b] = sum;
			}
		}
		const double wy = side? w : w * w; // the reference weights vertex 0's right-hand side twice (see above)
		int k = 0;
		#pragma unroll
		for(int q = 0; q < D; ++ q) {
			double t[RD]; // (w S J)[:, q]
			#pragma unroll
			for(int a = 0; a < RD; ++ a) {
				double sum = 0;
				#pragma unroll
				for(int b = 0; b < RD; ++ b)
					sum += sm[a][b] * j[b][q];
				t[a] = sum * w;
			}
			#pragma unroll
			for(int r = q; r < D; ++ r, ++ k) {
				double sum = 0;
				#pragma unroll
				for(int a = 0; a < RD; ++ a)
					sum += j[a][r] * t[a];
				acc[k] += sum;
			}
			double sum = 0;
			#pragma unroll
			for(int a = 0; a < RD; ++ a)
				sum += j[a][q] * se[a];
			acc[NT + q] += sum * wy;
		}
	}
	#pragma unroll
	for(int i = 0; i < NV; ++ i) {
		double v = acc[i];
		#pragma unroll
		for(int m = 32; m >= 1; m >>= 1)
			v += __shfl_xor(v, m);
		if(lane == 0)
			s_part[wave][i] = v;
	}
	__syncthreads();
	if(tid >= D * D + D)
		return;
	const bool b_blk = tid < D * D;
	const int r = b_blk? tid % D : tid - D * D, q = b_blk? tid / D : 0;
	int idx;
	if(b_blk) {
		const int hi = (r > q)? r : q, lo = (r > q)? q : r; // lower triangle, column by column: column lo starts after lo (2D - lo + 1) / 2
		idx = lo * (2 * D - lo + 1) / 2 + (hi - lo);
	} else
		idx = NT + r;
	double sum = 0;
	#pragma unroll
	for(int ww = 0; ww < W; ++ ww)
		sum += s_part[ww][idx];
	double *p_dst = b_blk? values + bd.dst + tid : eta + bd.eta_off + r;
	if(b_accumulate)
		sum += *p_dst;
	if(vertex_of[blockIdx.x] == unary_vertex)
		sum += b_blk? unary[tid] : unary[64 + r];
	*p_dst = sum;
}

void assembly_enqueue(CAssemblyState &a, const double *J0, const double *J1, const double *Si, const double *err,
	const double *wgt, int64_t n_unary_vertex, const double *p_unary_factor, const double *p_unary_error,
	double *values_out, double *eta_out, int b_accumulate)
{
	slampp_hip_solver &s = *a.p_solver;
	hipStream_t st = s.stream;
	if(!J0 || !J1 || !Si || !err || !values_out || !eta_out)
		throw std::invalid_argument("assemble: null device pointer");
	int n_unary = -1;
	if(p_unary_factor) {
		if(n_unary_vertex < 0 || n_unary_vertex >= int64_t(s.cumsum.size()) - 1)
			throw std::invalid_argument("assemble: the unary factor's vertex is outside Lambda");
		const int d = int(s.cumsum[n_unary_vertex + 1] - s.cumsum[n_unary_vertex]);
		double h[72];
		memset(h, 0, sizeof(h));
		for(int q = 0; q < d; ++ q) {
			for(int r = 0; r < d; ++ r) {
				double sum = 0; // U^T U, U column-major d x d (NonlinearSolver_Lambda_Base.h:1551-1553)
				for(int k = 0; k < d; ++ k)
					sum += p_unary_factor[k + r * d] * p_unary_factor[k + q * d];
				h[r + q * d] = sum;
			}
		}
		for(int q = 0; q < d && p_unary_error; ++ q)
			h[64 + q] = p_unary_error[q];
		if(!a.b_unary_valid || memcmp(h, a.h_unary, sizeof(h))) {
			SLAMPP_HIP_CHECK(hipStreamSynchronize(st)); // an earlier launch may still read d_unary
			memcpy(a.h_unary, h, sizeof(h));
			SLAMPP_HIP_CHECK(hipMemcpyAsync(a.d_unary.p(), a.h_unary, sizeof(h), hipMemcpyHostToDevice, st));
			a.b_unary_valid = true;
		}
		n_unary = int(n_unary_vertex);
	}
	s.Phase_Begin("assemble");
	const bool b_pack_off = (a.rd == 2 || a.rd == 3) && 2 * a.n_off_elems <= 64;
	enum { OFF_UNR = 1, DIAG_UNR = 4 }; // landmarks at C4: 297 us one wave each, 186 us packed with the operands of 4 edges in flight
	const int n_off_per_wave = std::max(1, 64 / a.n_off_elems);
	const unsigned n_off_grid = unsigned((a.n_offdiag + int64_t(n_off_per_wave) * OFF_UNR - 1) / (int64_t(n_off_per_wave) * OFF_UNR));
	const int n_small_per_wave = a.n_small? std::max(1, 64 / a.n_small_elems) : 1;
	const unsigned n_small_grid = unsigned((a.n_small + n_small_per_wave - 1) / n_small_per_wave);
#define LAUNCH_PACKED(RD) do { \
		if(a.n_offdiag > 0 && b_pack_off) \
			hipLaunchKernelGGL((assemble_offdiag_packed_kernel<RD>), dim3(n_off_grid), dim3(64), 0, st, a.d_offdiag.p(), \
				a.n_offdiag, n_off_per_wave, a.n_off_elems, a.d_entries.p(), a.d0, a.d1, J0, J1, Si, wgt, values_out, b_accumulate); \
		if(a.n_small > 0) \
			hipLaunchKernelGGL((assemble_diag_packed_kernel<RD, DIAG_UNR>), dim3(n_small_grid), dim3(64), 0, st, a.d_small.p(), \
				a.d_small_vertex.p(), a.n_small, n_small_per_wave, a.n_small_elems, a.d_entries.p(), a.d0, a.d1, J0, J1, Si, err, wgt, \
				n_unary, a.d_unary.p(), values_out, eta_out, b_accumulate); } while(0)
	if(a.rd == 2)
		LAUNCH_PACKED(2);
	else if(a.rd == 3)
		LAUNCH_PACKED(3);
#undef LAUNCH_PACKED
#define LAUNCH_ASM(RD) do { \
		if(a.n_offdiag > 0 && !b_pack_off) \
			hipLaunchKernelGGL(assemble_offdiag_kernel<RD>, dim3(unsigned(a.n_offdiag)), dim3(64), 0, st, a.d_offdiag.p(), \
				a.d_entries.p(), a.d0, a.d1, a.rd, J0, J1, Si, wgt, values_out, b_accumulate); \
		if(a.n_diag > 0) \
			hipLaunchKernelGGL(assemble_diag_kernel<RD>, dim3(unsigned(a.n_diag)), dim3(64), 0, st, a.d_diag.p(), a.d_diag_vertex.p(), \
				a.d_entries.p(), a.d0, a.d1, a.rd, J0, J1, Si, err, wgt, n_unary, a.d_unary.p(), values_out, eta_out, \
				b_accumulate); } while(0)
#define LAUNCH_LONG(RD, D) hipLaunchKernelGGL((assemble_long_kernel<RD, D>), dim3(unsigned(a.n_long)), dim3(256), 0, st, \
		a.d_long.p(), a.d_long_vertex.p(), a.d_entries.p(), a.d0, a.d1, J0, J1, Si, err, wgt, n_unary, a.d_unary.p(), \
		values_out, eta_out, b_accumulate)
	switch(a.rd) {
	case 2: LAUNCH_ASM(2); break;
	case 3: LAUNCH_ASM(3); break;
	case 6: LAUNCH_ASM(6); break;
	case 7: LAUNCH_ASM(7); break;
	default: LAUNCH_ASM(0); break;
	}
	if(a.n_long > 0) {
		const int key = a.rd * 10 + a.n_long_dim;
		switch(key) {
		case 26: LAUNCH_LONG(2, 6); break;
		case 27: LAUNCH_LONG(2, 7); break;
		case 23: LAUNCH_LONG(2, 3); break;
		case 33: LAUNCH_LONG(3, 3); break;
		case 66: LAUNCH_LONG(6, 6); break;
		case 77: LAUNCH_LONG(7, 7); break;
		default: throw std::logic_error("assembly: no edge-parallel kernel for a vertex that was set aside for it");
		}
	}
#undef LAUNCH_LONG
#undef LAUNCH_ASM
	s.Phase_End();
	SLAMPP_HIP_CHECK(hipGetLastError());
}

// Levenberg-Marquardt damping on device-resident values: alpha onto the diagonal of the diagonal blocks of block
// columns [n_first, n_last) -- the reference's ApplyDamping (include/slam/NonlinearSolver_Lambda_LM.h:228-239)
__global__ void damping_kernel(const int64_t *__restrict__ p_off_dim, int64_t n_first, int64_t n_last, double f_alpha, double *values)
{
	const int64_t v = n_first + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if(v >= n_last)
		return;
	const int64_t off = p_off_dim[2 * v], d = p_off_dim[2 * v + 1];
	for(int64_t r = 0; r < d; ++ r)
		values[off + r * (d + 1)] += f_alpha;
}

void damping_enqueue(const int64_t *p_off_dim_dev, int64_t n_first, int64_t n_last, double f_alpha, double *p_values_dev,
	hipStream_t stream)
{
	if(n_last > n_first)
		hipLaunchKernelGGL(damping_kernel, dim3(unsigned((n_last - n_first + 255) / 256)), dim3(256), 0, stream, p_off_dim_dev,
			n_first, n_last, f_alpha, p_values_dev);
}

size_t assembly_device_bytes(const CAssemblyState *p)
{
	return p->d_offdiag.n_Bytes() + p->d_diag.n_Bytes() + p->d_diag_vertex.n_Bytes() + p->d_long.n_Bytes() + p->d_long_vertex.n_Bytes() +
		p->d_small.n_Bytes() + p->d_small_vertex.n_Bytes() +
		p->d_entries.n_Bytes() + p->d_unary.n_Bytes();
}

} // namespace slampp
