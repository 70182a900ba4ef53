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
	// groups of consecutive vertices whose edges fit in LDS together: every edge record is read once for the off-diagonal
	// block and both diagonal contributions (assemble_group_kernel)
	int64_t n_groups;
	CDevArray<int32_t> d_grp_words; // per group, n_grp_pkg_stride words: n_edges, n_blocks, n_entries, 0 | edges (n_grp_cap_edges, padded) | blocks (6 words each) | entries (slot * 2 + flag)
	int n_grp_pkg_stride, n_grp_cap_edges;
	int n_grp_resident[2];          // workgroups of the kernel a CU holds (without / with accumulation; 0: not asked yet)
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

enum { GRP_LDS_BYTES = 160 * 1024 / 6 - 576 - 1536, GRP_PACKAGE_WORDS = 1024, GRP_PKG_PER_THREAD = 4, GRP_MAX_INSTR = 4 }; // LDS of a workgroup (six to a CU); its package; copy instructions per wave // LDS of a group: its edges' records; its package

// the group kernel's shape for an edge set: bytes of an edge in LDS (J0 | J1 | Sigma^-1 | error as they lie in memory,
// M = Sigma^-1 [J0 J1 e], the weight), edges per wave, copy instructions per wave
struct TAsmGroupShape {
	int n_edge_bytes, n_ew, n_instr;
};

static inline TAsmGroupShape asm_group_shape(int rd, int d0, int d1)
{
	const int n_sz = (rd % 2 == 0 && (rd == 2 || rd == 6))? 16 : 4; // (the instantiations with 16-byte copies: assemble_group_kernel's SZ)
	const int n_raw = rd * (d0 + d1 + rd + 1), n_mw = (rd * (d0 + d1 + 1) + 1) & ~1, n_units = n_raw * 8 / n_sz;
	TAsmGroupShape t;
	t.n_edge_bytes = (n_raw + n_mw + 1) * 8;
	t.n_ew = std::min(std::min(int(GRP_LDS_BYTES) / t.n_edge_bytes / 4, 64 * int(GRP_MAX_INSTR) / n_units), 32);
	t.n_instr = (t.n_ew * n_units + 63) / 64;
	return t;
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
	// Groups of consecutive vertices (a pose graph's odometry chain makes neighbours of them): the workgroup of a group
	// brings the records of all edges that touch its vertices into LDS once and writes every block of their columns --
	// an edge inside a group is read once instead of three times (off-diagonal block, two diagonal blocks), one that
	// leaves it twice.  A group closes at GRP_VERTICES vertices or when its edges would not fit; vertices with more edges
	// than fit on their own stay with the kernels below.
	std::vector<int64_t> grp_ptr;
	std::vector<int32_t> grp_words;
	std::vector<uint8_t> b_grouped(n, 0);
	int n_grp_pkg_cap = 0, n_grp_cap_edges = 0;
	{
		const int n_grp_vertices = s.n_assembly_groups; // option "assembly_groups": as many as fit by default, 0 = the one-wave kernels for everything
		const int n_mode = n_grp_vertices > 0;
		const TAsmGroupShape t_shape = asm_group_shape(rd, d0, d1);
		const int n_cap_edges = 4 * t_shape.n_ew; // as many edges as the workgroup copies in one go: n_ew per wave
		n_grp_cap_edges = n_cap_edges;
		if(n_mode > 0 && n_cap_edges >= 8) {
			std::vector<int64_t> stamp(n_edges, -1);
			std::vector<int32_t> slot_of(n_edges, 0);
			std::vector<int32_t> g_edges, g_blocks, g_entries;
			std::vector<uint16_t> g_items;
			std::vector<int64_t> g_vertices;
			auto Close = [&]() {
				if(g_vertices.empty())
					return;
				g_blocks.clear(); g_entries.clear(); g_items.clear();
				for(size_t i = 0; i < g_vertices.size(); ++ i) {
					const int64_t v = g_vertices[i];
					for(int64_t k = ptr[v]; k < ptr[v + 1]; ++ k) {
						const bool b_diag = brow[k] == v;
						const TAsmBlk &b = b_diag? diag[v] : offdiag[off_slot[k]];
						const int32_t n_begin = int32_t(g_entries.size());
						for(int64_t i2 = 0; i2 < b.ne; ++ i2) {
							const int32_t ent = entries[b.e0 + i2];
							g_entries.push_back(slot_of[ent >> 1] * 2 + (ent & 1));
						}
						g_blocks.push_back(int32_t(uint64_t(b.dst) & 0xffffffffu));
						g_blocks.push_back(int32_t(uint64_t(b.dst) >> 32));
						g_blocks.push_back(b_diag? int32_t(b.eta_off) : -1);
						g_blocks.push_back(int32_t(v));
						g_blocks.push_back(n_begin | (int32_t(b.ne) << 16));
						g_blocks.push_back(int32_t(b.rows) | (int32_t(b.cols) << 8));
						for(int q = 0; q < int(b.cols) + (b_diag? 1 : 0); ++ q)
							g_items.push_back(uint16_t((g_blocks.size() / 6 - 1) * 8 + q)); // (block, column; column `cols` of a diagonal block: eta)
					}
					b_grouped[v] = 1;
				}
				// the lanes of a wave walk their blocks' edge lists in step: blocks with lists of the same length next to each other
				std::stable_sort(g_items.begin(), g_items.end(), [&](uint16_t a, uint16_t b) {
					return (uint32_t(g_blocks[6 * (a >> 3) + 4]) >> 16) < (uint32_t(g_blocks[6 * (b >> 3) + 4]) >> 16); });
				grp_ptr.push_back(int64_t(grp_words.size()));
				grp_words.push_back(int32_t(g_edges.size()));
				grp_words.push_back(int32_t(g_blocks.size() / 6));
				grp_words.push_back(int32_t(g_entries.size()));
				grp_words.push_back(int32_t(g_items.size()));
				grp_words.insert(grp_words.end(), g_edges.begin(), g_edges.end());
				grp_words.insert(grp_words.end(), size_t(n_cap_edges) - g_edges.size(), g_edges.empty()? 0 : g_edges[0]); // (valid addresses for the requests)
				grp_words.insert(grp_words.end(), g_blocks.begin(), g_blocks.end());
				grp_words.insert(grp_words.end(), g_entries.begin(), g_entries.end());
				for(size_t i = 0; i < g_items.size(); i += 2)
					grp_words.push_back(int32_t(uint32_t(g_items[i]) | (uint32_t((i + 1 < g_items.size())? g_items[i + 1] : 0) << 16)));
				n_grp_pkg_cap = std::max(n_grp_pkg_cap, int(4 + n_cap_edges + g_blocks.size() + g_entries.size() + (g_items.size() + 1) / 2));
				g_edges.clear(); g_vertices.clear();
			};
			int64_t n_group = 0, n_group_words = 0; // number of the open group; words of its blocks and entries so far
			for(int64_t v = 0; v < n; ++ v) {
				const int64_t n_v_words = (6 + 4) * (ptr[v + 1] - ptr[v]) + 3 * int64_t(cnt_diag[v]) + 1; // blocks and their items, entries
				if(cnt_diag[v] > n_cap_edges || n_v_words + 4 + n_cap_edges > GRP_PACKAGE_WORDS || int(diag[v].rows) * int(diag[v].rows) + int(diag[v].rows) > 64) {
					Close(); // (consecutive vertices only)
					++ n_group; n_group_words = 0;
					continue;
				}
				int n_new = 0;
				for(int64_t i = 0; i < diag[v].ne; ++ i)
					n_new += stamp[entries[diag[v].e0 + i] >> 1] != n_group;
				if(int(g_vertices.size()) == n_grp_vertices || int(g_edges.size()) + n_new > n_cap_edges ||
				   n_group_words + n_v_words + 4 + n_cap_edges > GRP_PACKAGE_WORDS) {
					Close();
					++ n_group; n_group_words = 0;
				}
				for(int64_t i = 0; i < diag[v].ne; ++ i) {
					const int64_t e = entries[diag[v].e0 + i] >> 1;
					if(stamp[e] != n_group) {
						stamp[e] = n_group;
						slot_of[e] = int32_t(g_edges.size());
						g_edges.push_back(int32_t(e));
					}
				}
				g_vertices.push_back(v);
				n_group_words += n_v_words;
			}
			Close();
			grp_ptr.push_back(int64_t(grp_words.size()));
			if(grp_ptr.size() == 1)
				grp_ptr.clear();
			else { // one stride for all packages: a workgroup finds its next one without a lookup
				n_grp_pkg_cap = (n_grp_pkg_cap + 3) & ~3; // (the records behind it start at a multiple of 16 bytes)
				std::vector<int32_t> t_fixed((grp_ptr.size() - 1) * size_t(n_grp_pkg_cap), 0);
				for(size_t i = 0; i + 1 < grp_ptr.size(); ++ i)
					std::copy(grp_words.begin() + grp_ptr[i], grp_words.begin() + grp_ptr[i + 1], t_fixed.begin() + i * size_t(n_grp_pkg_cap));
				grp_words.swap(t_fixed);
			}
			// the kernels below keep what the groups did not take
			std::vector<TAsmBlk> offdiag_rest;
			for(int64_t c = 0; c < n; ++ c) {
				if(b_grouped[c])
					continue;
				for(int64_t k = ptr[c]; k < ptr[c + 1]; ++ k)
					if(brow[k] != c) offdiag_rest.push_back(offdiag[off_slot[k]]);
			}
			offdiag.swap(offdiag_rest);
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
			if(b_grouped[v])
				continue;
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
		if(!offdiag.empty())
			p->d_offdiag.Upload(offdiag, s.stream);
		p->n_groups = grp_ptr.empty()? 0 : int64_t(grp_ptr.size()) - 1;
		p->n_grp_pkg_stride = n_grp_pkg_cap; p->n_grp_cap_edges = n_grp_cap_edges;
		p->n_grp_resident[0] = p->n_grp_resident[1] = 0;
		if(p->n_groups)
			p->d_grp_words.Upload(grp_words, s.stream);
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

// rd consecutive doubles out of LDS, two per instruction where rd is even (every vector of a record then starts at a multiple of 16 bytes)
template <int RD>
__device__ __forceinline__ void lds_read_vector(double (&r_dst)[RD? RD : 8], const double *p_src, int rd)
{
	typedef double v2f64 __attribute__((ext_vector_type(2)));
	if(RD && RD % 2 == 0) {
		#pragma unroll
		for(int i = 0; i < RD; i += 2) {
			const v2f64 v = *reinterpret_cast<const v2f64*>(p_src + i);
			r_dst[i] = v.x;
			r_dst[i + 1] = v.y;
		}
	} else {
		#pragma unroll
		for(int i = 0; i < (RD? RD : 8); ++ i)
			r_dst[i] = (RD || i < rd)? p_src[i] : 0.0;
	}
}

// a load that writes LDS directly: SZ bytes per lane from p_src (per lane) to p_lds_base (wave-uniform) + lane * SZ
template <int SZ>
__device__ __forceinline__ void load_to_lds(const void *p_src, void *p_lds_base)
{
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass has no such builtin, and drops the kernel's stub without a word if it sees it)
	if(SZ == 16) {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p_src,
			(__attribute__((address_space(3))) void*)p_lds_base, 16, 0, 0);
	} else {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p_src,
			(__attribute__((address_space(3))) void*)p_lds_base, 4, 0, 0);
	}
#endif
}

// Groups of consecutive vertices (assembly_setup), a few per workgroup, one after another.  The records of all edges
// that touch a group's vertices are copied into LDS as they lie in memory (J0 | J1 | Sigma^-1 | error of an edge next to
// each other) by loads that write LDS directly -- no registers, no LDS store instructions, one instruction per wave and
// kilobyte; M = Sigma^-1 [J0 J1 e] is formed once per edge; then every block of the group's columns -- off-diagonal
// blocks, diagonal blocks, right-hand sides -- is summed out of LDS, a lane per (block, column).  The arithmetic and
// its order are those of the one-wave kernels above.
// What the numbers said on the way here (C3-shaped graph, 157 MB, the two one-wave kernels: 90 us): records requested
// when a group's turn came, through registers, a lane per element of M and of the blocks: 96 us; requested one group
// ahead: 109-125 us (not the loads: 1 250 wave-instructions per group); a lane per column, 16-byte LDS reads: 65 us;
// no load behind a branch inside the loop: 60 us, by the kernel's own clock 1.4 k cycles per group to store the
// records, 1.7 k to request the next ones (64-bit address arithmetic for 28 loads), 1.2 k for M, 4.2 k for the blocks
// -- dependent LDS reads and six-term sums with three waves per SIMD (150 registers, 40 of them records in flight).
// Hence this form: nothing in flight in registers, few registers, as many workgroups per CU as LDS allows, and the
// other workgroups of the CU to cover a workgroup's wait for its next group.
template <int RD, bool b_accumulate>
__global__ void __launch_bounds__(256)
assemble_group_kernel(const int32_t *__restrict__ grp_words, int n_groups, int n_pkg_stride, int d0, int d1,
	int n_rd, int n_ew, int n_instr, const double *__restrict__ J0, const double *__restrict__ J1,
	const double *__restrict__ Si, const double *__restrict__ err, const double *__restrict__ wgt, int unary_vertex,
	const double *__restrict__ unary, double *values, double *eta, long long *p_timing)
{
	__shared__ double s_unary[72];
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];
	enum { PKW = GRP_PKG_PER_THREAD, NJ = GRP_MAX_INSTR, SZ = (RD && RD % 2 == 0)? 16 : 4 };
	const int rd = RD? RD : n_rd;
	const int L0 = rd * d0, L1 = rd * d1, LS = rd * rd, NM = rd * (d0 + d1 + 1);
	const int o_j1 = L0, o_s = L0 + L1, o_e = o_s + LS, RAW = o_e + rd, MW = (NM + 1) & ~1; // (even where rd is even)
	const int n_cap = 4 * n_ew, NU = RAW * 8 / SZ; // edge slots; copy units of an edge
	int32_t *s_pkg = reinterpret_cast<int32_t*>(s_dyn);
	double *s_raw = s_dyn + n_pkg_stride / 2, *s_m = s_raw + n_cap * RAW, *s_w = s_m + n_cap * MW;
	const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
	int g = blockIdx.x;
	if(g >= n_groups)
		return;
	const long long n_t_begin = p_timing? __builtin_readcyclecounter() : 0;
	if(t < 72)
		s_unary[t] = unary[t];
	if(t < n_cap)
		s_w[t] = 1.0; // (stays if there are no weights)
	// this lane's part in the copy of the wave's n_ew edges: instruction j moves units 64 j .. 64 j + 63 of them
	int n_unit[NJ];             // doubles per edge of the source array | (edge slot in the package's list + 1) << 8; 0: nothing to copy
	const char *p_unit_src[NJ]; // the source array, at this lane's bytes of edge 0
	#pragma unroll
	for(int j = 0; j < NJ; ++ j) {
		const int u = 64 * j + lane, n_rel = u / NU, n_byte = (u - n_rel * NU) * SZ;
		const bool b_valid = j < n_instr && u < n_ew * NU;
		const int n_seg = (n_byte < o_j1 * 8)? 0 : ((n_byte < o_s * 8)? 1 : ((n_byte < o_e * 8)? 2 : 3));
		n_unit[j] = b_valid? (((n_seg == 0)? L0 : ((n_seg == 1)? L1 : ((n_seg == 2)? LS : rd))) | ((wave * n_ew + n_rel + 1) << 8)) : 0;
		p_unit_src[j] = reinterpret_cast<const char*>((n_seg == 0)? J0 : ((n_seg == 1)? J1 : ((n_seg == 2)? Si : err))) +
			(n_byte - 8 * ((n_seg == 0)? 0 : ((n_seg == 1)? o_j1 : ((n_seg == 2)? o_s : o_e))));
	}
	const int n_w_slot = (wgt && lane < 2 * n_ew)? wave * n_ew + (lane >> 1) : -1; // the weights: 4-byte units
	// (every address is formed -- and pinned, or the compiler moves the arithmetic back to where it is used -- before the
	// first copy is issued: with a copy in flight, the compiler waits for everything in flight wherever the value of an
	// ordinary load, the edge numbers, is used: 900 cycles between two copies)
	auto Request = [&](const int32_t (&eid)[NJ], int32_t n_w_eid) {
		const char *p_src[NJ];
		#pragma unroll
		for(int j = 0; j < NJ; ++ j) {
			p_src[j] = p_unit_src[j] + int64_t(eid[j]) * ((n_unit[j] & 0xff) * 8);
			asm volatile("" : "+v"(p_src[j]));
		}
		const char *p_w_src = reinterpret_cast<const char*>(wgt + n_w_eid) + 4 * (lane & 1);
		asm volatile("" : "+v"(p_w_src));
		#pragma unroll
		for(int j = 0; j < NJ; ++ j) {
			if(n_unit[j])
				load_to_lds<SZ>(p_src[j], reinterpret_cast<char*>(s_raw + wave * n_ew * RAW) + 64 * j * SZ);
		}
		if(n_w_slot >= 0)
			load_to_lds<4>(p_w_src, s_w + wave * n_ew);
	};
	int32_t eid[NJ], n_w_eid, pk[PKW];
	{
		const int32_t *p_pkg = grp_words + int64_t(g) * n_pkg_stride;
		#pragma unroll
		for(int j = 0; j < NJ; ++ j)
			eid[j] = p_pkg[4 + max((n_unit[j] >> 8) - 1, 0)]; // (the list is padded with edges of the group)
		n_w_eid = p_pkg[4 + max(n_w_slot, 0)];
		#pragma unroll
		for(int j = 0; j < PKW; ++ j)
			pk[j] = p_pkg[min(t + 256 * j, n_pkg_stride - 1)];
		__syncthreads(); // (s_w)
		Request(eid, n_w_eid);
	}
	for(;;) {
		const int g_next = g + int(gridDim.x);
		long long n_t0 = p_timing? __builtin_readcyclecounter() : 0, n_t1 = 0, n_t2 = 0, n_t3 = 0;
		// the group's package is in registers, its records on their way into LDS
		#pragma unroll
		for(int j = 0; j < PKW; ++ j)
			if(t + 256 * j < n_pkg_stride) s_pkg[t + 256 * j] = pk[j];
		__syncthreads(); // (waits for the records as well)
		if(p_timing) n_t1 = __builtin_readcyclecounter();
		{
			// the package of the next group and the edge numbers this lane will ask for
			const int32_t *p_pkg = grp_words + int64_t((g_next < n_groups)? g_next : g) * n_pkg_stride;
			#pragma unroll
			for(int j = 0; j < NJ; ++ j)
				eid[j] = p_pkg[4 + max((n_unit[j] >> 8) - 1, 0)];
			n_w_eid = p_pkg[4 + max(n_w_slot, 0)];
			#pragma unroll
			for(int j = 0; j < PKW; ++ j)
				pk[j] = p_pkg[min(t + 256 * j, n_pkg_stride - 1)];
		}
		if(p_timing) n_t2 = __builtin_readcyclecounter();
		const int n_edges = s_pkg[0], n_blocks = s_pkg[1], n_items = s_pkg[3];
		const int32_t *s_blk = s_pkg + 4 + n_cap, *s_ent = s_blk + 6 * n_blocks;
		const uint16_t *s_item = reinterpret_cast<const uint16_t*>(s_ent + s_pkg[2]);
		// M = Sigma^-1 [J0 J1 e]: a lane per (edge, column) -- a lane per element issued 2 rd LDS reads for every product
		for(int i = t; i < n_edges * 16; i += 256) {
			const int mc = i & 15;
			if(mc < d0 + d1 + 1) {
				const double *p_rec = s_raw + (i >> 4) * RAW;
				double x[RD? RD : 8], m[RD? RD : 8];
				lds_read_vector<RD>(x, (mc < d0 + d1)? p_rec + mc * rd : p_rec + o_e, rd); // (J0 and J1 follow one another)
				#pragma unroll
				for(int a2 = 0; a2 < (RD? RD : 8); ++ a2)
					m[a2] = 0;
				#pragma unroll
				for(int b2 = 0; b2 < (RD? RD : 8); ++ b2) {
					if(RD || b2 < rd) {
						double sc[RD? RD : 8];
						lds_read_vector<RD>(sc, p_rec + o_s + b2 * rd, rd);
						#pragma unroll
						for(int a2 = 0; a2 < (RD? RD : 8); ++ a2)
							m[a2] += sc[a2] * x[b2];
					}
					__builtin_amdgcn_sched_barrier(0); // (one column of Sigma^-1 at a time: few registers, many workgroups)
				}
				double *p_m = s_m + (i >> 4) * MW + mc * rd;
				#pragma unroll
				for(int a2 = 0; a2 < (RD? RD : 8); ++ a2)
					if(RD || a2 < rd) p_m[a2] = m[a2];
			}
		}
		__syncthreads();
		if(p_timing) n_t3 = __builtin_readcyclecounter();
		// a lane per (block, column): column q of the stored block (or the right-hand side of a diagonal block, column
		// `cols`) is sum over the block's edges of A^T b, A one of J0 / J1 / their M, b a column of the other
		for(int i = t; i < n_items; i += 256) {
			const int n_item = s_item[i];
			const int32_t *p_blk = s_blk + 6 * (n_item >> 3);
			const int q = n_item & 7;
			const int64_t dst = int64_t(uint64_t(uint32_t(p_blk[0])) | (uint64_t(uint32_t(p_blk[1])) << 32));
			const int eta_off = p_blk[2], n_ent0 = p_blk[4] & 0xffff, n_ents = int(uint32_t(p_blk[4]) >> 16);
			const int rows = p_blk[5] & 0xff, cols = (p_blk[5] >> 8) & 0xff;
			const bool b_diag = eta_off >= 0, b_blk = q < cols;
			double *p_dst = b_blk? values + dst + q * rows : eta + eta_off;
			double acc[8];
			#pragma unroll
			for(int r = 0; r < 8; ++ r)
				acc[r] = (b_accumulate && r < rows)? p_dst[r] : 0.0;
			for(int j = 0; j < n_ents; ++ j) {
				const int32_t ent = s_ent[n_ent0 + j];
				const double *p_rec = s_raw + (ent >> 1) * RAW, *p_M = s_m + (ent >> 1) * MW;
				const double *p_A, *p_b;
				double w = s_w[ent >> 1];
				if(!b_diag) { // element (c0, c1) of J0^T (w S) J1 lands on (r, q) of the stored block, transposed if vertex 0 has the larger id
					p_A = (ent & 1)? p_M + d0 * rd : p_rec;
					p_b = (ent & 1)? p_rec + q * rd : p_M + (d0 + q) * rd;
				} else {
					const int side = ent & 1;
					p_A = p_rec + (side? o_j1 : 0);
					p_b = p_M + (b_blk? ((side? d0 : 0) + q) * rd : (d0 + d1) * rd);
					// the reference weights vertex 0's right-hand side twice (BaseTypes_Binary.h:813-815 against :836-838)
					w = (b_blk || side)? w : w * w;
				}
				double bv[RD? RD : 8];
				lds_read_vector<RD>(bv, p_b, rd);
				#pragma unroll
				for(int r = 0; r < 8; ++ r) {
					if(r < rows) {
						double av[RD? RD : 8];
						lds_read_vector<RD>(av, p_A + r * rd, rd);
						double sum = 0;
						#pragma unroll
						for(int a2 = 0; a2 < (RD? RD : 8); ++ a2)
							if(RD || a2 < rd) sum += av[a2] * bv[a2];
						acc[r] += sum * w;
					}
					__builtin_amdgcn_sched_barrier(0); // (one row at a time: see above)
				}
			}
			if(b_diag && p_blk[3] == unary_vertex) {
				#pragma unroll
				for(int r = 0; r < 8; ++ r)
					if(r < rows) acc[r] += b_blk? s_unary[q * rows + r] : s_unary[64 + r];
			}
			#pragma unroll
			for(int r = 0; r < 8; ++ r)
				if(r < rows) p_dst[r] = acc[r];
		}
		long long n_t4 = 0;
		if(p_timing && blockIdx.x < 64 && lane == 0) { // development aid: where a group's time goes (wait, requests, M, blocks), summed over the workgroup's groups, wave by wave
			n_t4 = __builtin_readcyclecounter();
			long long *p_t = p_timing + blockIdx.x * 32 + wave * 8;
			p_t[0] += n_t1 - n_t0; p_t[1] += n_t2 - n_t1; p_t[2] += n_t3 - n_t2; p_t[3] += n_t4 - n_t3; p_t[4] += 1;
		}
		if(g_next >= n_groups) {
			if(p_timing && t == 0) {
				p_timing[64 * 32 + 2 * blockIdx.x] = n_t_begin;
				p_timing[64 * 32 + 2 * blockIdx.x + 1] = __builtin_readcyclecounter();
			}
			break;
		}
		g = g_next;
		__syncthreads(); // the group's records and package are no longer needed
		long long n_t5 = p_timing? __builtin_readcyclecounter() : 0;
		Request(eid, n_w_eid);
		if(p_timing && blockIdx.x < 64 && lane == 0) {
			long long *p_t = p_timing + blockIdx.x * 32 + wave * 8;
			p_t[5] += n_t5 - n_t4; p_t[6] += __builtin_readcyclecounter() - n_t5;
		}
	}
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
	if(a.n_groups > 0) {
		const TAsmGroupShape t_shape = asm_group_shape(a.rd, a.d0, a.d1);
		const size_t n_lds = size_t(a.n_grp_pkg_stride) * 4 + size_t(a.n_grp_cap_edges) * t_shape.n_edge_bytes;
		static const int n_cus = [] { hipDeviceProp_t t_prop; int n_dev = 0; return (hipGetDevice(&n_dev) == hipSuccess &&
			hipGetDeviceProperties(&t_prop, n_dev) == hipSuccess)? t_prop.multiProcessorCount : 256; }();
		// a persistent grid: as many workgroups as the chip holds at once (registers and LDS decide)
		int &n_resident = a.n_grp_resident[b_accumulate != 0];
#define GROUPS_RESIDENT(RD, b_acc) do { if(!n_resident) { \
			if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n_resident, assemble_group_kernel<RD, b_acc>, 256, n_lds) != hipSuccess || n_resident < 1) \
				n_resident = 1; } } while(0)
#define GROUPS_RESIDENT_RD(RD) do { if(b_accumulate) GROUPS_RESIDENT(RD, true); else GROUPS_RESIDENT(RD, false); } while(0)
		switch(a.rd) {
		case 2: GROUPS_RESIDENT_RD(2); break;
		case 3: GROUPS_RESIDENT_RD(3); break;
		case 6: GROUPS_RESIDENT_RD(6); break;
		case 7: GROUPS_RESIDENT_RD(7); break;
		default: GROUPS_RESIDENT_RD(0); break;
		}
#undef GROUPS_RESIDENT_RD
#undef GROUPS_RESIDENT
		const unsigned n_grid = unsigned(std::min(a.n_groups, int64_t(n_cus) * n_resident));
		long long *p_timing = 0;
		const bool b_timing = getenv("SLAMPP_HIP_ASM_TIMING") != 0; // development aid (prints only)
		if(b_timing) {
			SLAMPP_HIP_CHECK(hipMalloc(&p_timing, (64 * 32 + 2 * n_grid) * sizeof(long long)));
			SLAMPP_HIP_CHECK(hipMemsetAsync(p_timing, 0, (64 * 32 + 2 * n_grid) * sizeof(long long), st));
		}
#define LAUNCH_GROUPS_ACC(RD, b_acc) hipLaunchKernelGGL((assemble_group_kernel<RD, b_acc>), dim3(n_grid), dim3(256), n_lds, st, \
			a.d_grp_words.p(), int(a.n_groups), a.n_grp_pkg_stride, a.d0, a.d1, a.rd, t_shape.n_ew, t_shape.n_instr, \
			J0, J1, Si, err, wgt, n_unary, a.d_unary.p(), values_out, eta_out, p_timing)
#define LAUNCH_GROUPS(RD) do { if(b_accumulate) LAUNCH_GROUPS_ACC(RD, true); else LAUNCH_GROUPS_ACC(RD, false); } while(0)
		switch(a.rd) {
		case 2: LAUNCH_GROUPS(2); break;
		case 3: LAUNCH_GROUPS(3); break;
		case 6: LAUNCH_GROUPS(6); break;
		case 7: LAUNCH_GROUPS(7); break;
		default: LAUNCH_GROUPS(0); break;
		}
#undef LAUNCH_GROUPS
#undef LAUNCH_GROUPS_ACC
		if(p_timing) {
			std::vector<long long> t_h(64 * 32 + 2 * n_grid);
			long long *h = t_h.data();
			SLAMPP_HIP_CHECK(hipMemcpyAsync(h, p_timing, t_h.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
			SLAMPP_HIP_CHECK(hipStreamSynchronize(st));
			{
				long long n_min = h[64 * 32], n_max = 0;
				for(unsigned i = 0; i < n_grid; ++ i) {
					n_min = std::min(n_min, h[64 * 32 + 2 * i]);
					n_max = std::max(n_max, h[64 * 32 + 2 * i + 1]);
				}
				std::vector<long long> t_begin, t_len;
				for(unsigned i = 0; i < n_grid; ++ i) {
					t_begin.push_back(h[64 * 32 + 2 * i] - n_min);
					t_len.push_back(h[64 * 32 + 2 * i + 1] - h[64 * 32 + 2 * i]);
				}
				std::sort(t_begin.begin(), t_begin.end());
				std::sort(t_len.begin(), t_len.end());
				fprintf(stderr, "[assemble] first begin to last end %lld ticks; begins: median %lld, 90 %% %lld, last %lld; lifetimes: min %lld median %lld max %lld\n",
					n_max - n_min, t_begin[n_grid / 2], t_begin[n_grid * 9 / 10], t_begin.back(), t_len[0], t_len[n_grid / 2], t_len.back());
			}
			SLAMPP_HIP_CHECK(hipFree(p_timing));
			for(int w = 0; w < 4; ++ w) {
				double f[7] = {0, 0, 0, 0, 0, 0, 0};
				for(int i = 0; i < 64; ++ i)
					for(int j = 0; j < 7; ++ j) f[j] += double(h[i * 32 + w * 8 + j]);
				fprintf(stderr, "[assemble] wave %d: per group (clock ticks): wait %.0f, requests %.0f, M %.0f, blocks %.0f, barrier %.0f, copies %.0f (%.1f groups per workgroup; grid %u, LDS %zu B)\n",
					w, f[0] / f[4], f[1] / f[4], f[2] / f[4], f[3] / f[4], f[5] / f[4], f[6] / f[4], f[4] / 64, n_grid, n_lds);
			}
		}
	}
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
		p->d_entries.n_Bytes() + p->d_unary.n_Bytes() + p->d_grp_words.n_Bytes();
}

} // namespace slampp
