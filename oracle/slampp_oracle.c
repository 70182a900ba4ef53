/*
 * oracle/slampp_oracle.c -- TEST INFRASTRUCTURE.  A plain-C, single-threaded CPU restatement of
 * the reference's Lambda-solve path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (libslampp_hip.so) never does.
 *
 * Pinned against the compiled reference (oracle/_ref/ref_harness, built from /root/reference by
 * oracle/Makefile.ref) through the golden vectors in tests/golden/ -- see tests/test_oracle.py.
 *
 * What is restated (file:line relative to /root/reference):
 *   - symmetric block permutation keeping the upper triangle, blocks that land below the
 *     diagonal are transposed                         src/slam/BlockMatrix.cpp:8183-8349
 *   - block elimination tree                           src/slam/BlockMatrix.cpp:9403-9451
 *   - ereach (pattern of a column of R)                src/slam/BlockMatrix.cpp:9453-9545
 *   - up-looking block Cholesky  R^T R = Lambda        src/slam/BlockMatrix.cpp:9547-9785
 *       per column j, for k in ereach(j):  R_kj = R_kk^-T (A_kj - sum_i R_ik^T R_ij),
 *       then R_jj = chol(A_jj - sum_i R_ij^T R_ij); returns "not positive definite" exactly
 *       where Eigen's LLT does (pivot <= 0)            :9752-9771
 *   - x = R^-1 R^-T b by block substitution            src/slam/BlockMatrix.cpp:8637-8719, 8898-
 *     orchestrated as in CLinearSolver_UberBlock::Solve_PosDef_Blocky
 *                                                      include/slam/LinearSolver_UberBlock.h:312-426
 *   - the Schur-complement solve of CLinearSolver_Schur::Solve_PosDef_Blocky, steps 2-13
 *                                                      include/slam/LinearSolver_Schur.h:1699-1886
 *     with the dense reduced solve of CLinearSolver_DenseEigen (Eigen LLT)
 *                                                      src/slam/LinearSolver_Schur.cpp:2314-2331
 *   - block-diagonal inverse (Eigen .inverse(), closed form up to 4x4; here Gauss-Jordan on
 *     the SPD block, same result to rounding)          include/slam/BlockMatrixBase.h:1257-1270
 *
 * The fill-reducing ordering (the reference calls AMD) is an input: the solution does not
 * depend on it beyond rounding, and the tests pass several orderings to show that.
 *
 * Also here: oracle_exec_plan(), a CPU replay of the product's elimination plan
 * (slampp_hip_plan_view) so that ordering / symbolic analysis / schedule can be tested without a GPU.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct {
	int64_t n;          /* block columns */
	int64_t *cs;        /* [n+1] scalar offsets */
	int64_t *ptr;       /* [n+1] */
	int32_t *row;       /* block rows, sorted per column, upper triangle (row <= col) */
	int64_t *off;       /* value offset per block */
	double *val;
} TBlockMat;

static void bm_free(TBlockMat *m)
{
	free(m->cs); free(m->ptr); free(m->row); free(m->off); free(m->val);
	memset(m, 0, sizeof(*m));
}

/* ---- symmetric permutation, upper triangle kept (BlockMatrix.cpp:8183-8349) ----
 * perm[new] = old.  Block (r,c), r <= c, goes to (min,max) of the new indices and is
 * transposed if the order of its two indices flips. */
static int bm_permute_upper(const int64_t n, const int64_t *cs, const int64_t *ptr, const int32_t *brow,
	const double *val, const int32_t *perm, TBlockMat *out)
{
	int64_t i, c, k, nb = ptr[n];
	int32_t *pinv = (int32_t*)malloc(sizeof(int32_t) * n);
	int64_t *aoff = (int64_t*)malloc(sizeof(int64_t) * (nb + 1));
	int64_t *cnt = (int64_t*)calloc(n + 1, sizeof(int64_t));
	int64_t *fill;
	memset(out, 0, sizeof(*out));
	out->n = n;
	out->cs = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
	out->ptr = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
	out->row = (int32_t*)malloc(sizeof(int32_t) * (nb? nb : 1));
	out->off = (int64_t*)malloc(sizeof(int64_t) * (nb + 1));
	for(i = 0; i < n; ++ i)
		pinv[perm? perm[i] : i] = (int32_t)i;
	out->cs[0] = 0;
	for(i = 0; i < n; ++ i) {
		int64_t o = perm? perm[i] : i;
		out->cs[i + 1] = out->cs[i] + (cs[o + 1] - cs[o]);
	}
	aoff[0] = 0;
	for(c = 0; c < n; ++ c) {
		for(k = ptr[c]; k < ptr[c + 1]; ++ k) {
			int64_t r = brow[k];
			int64_t nr = pinv[r], nc = pinv[c];
			aoff[k + 1] = aoff[k] + (cs[r + 1] - cs[r]) * (cs[c + 1] - cs[c]);
			++ cnt[(nr > nc? nr : nc) + 1];
		}
	}
	for(i = 0; i < n; ++ i)
		cnt[i + 1] += cnt[i];
	memcpy(out->ptr, cnt, sizeof(int64_t) * (n + 1));
	fill = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
	memcpy(fill, cnt, sizeof(int64_t) * (n + 1));
	out->val = (double*)malloc(sizeof(double) * (aoff[nb]? aoff[nb] : 1));
	{
		/* two passes: place (row, source) pairs, sort rows per column, then copy values */
		int64_t *src = (int64_t*)malloc(sizeof(int64_t) * (nb? nb : 1));
		for(c = 0; c < n; ++ c) {
			for(k = ptr[c]; k < ptr[c + 1]; ++ k) {
				int64_t nr = pinv[brow[k]], nc = pinv[c];
				int64_t col = nr > nc? nr : nc, rw = nr > nc? nc : nr; /* upper triangle: row <= col */
				int64_t d = fill[col] ++;
				out->row[d] = (int32_t)rw;
				src[d] = k;
			}
		}
		for(c = 0; c < n; ++ c) { /* insertion sort by row (columns are short) */
			int64_t a, b;
			for(a = out->ptr[c] + 1; a < out->ptr[c + 1]; ++ a) {
				int32_t rr = out->row[a];
				int64_t ss = src[a];
				for(b = a; b > out->ptr[c] && out->row[b - 1] > rr; -- b) {
					out->row[b] = out->row[b - 1];
					src[b] = src[b - 1];
				}
				out->row[b] = rr;
				src[b] = ss;
			}
		}
		out->off[0] = 0;
		for(c = 0; c < n; ++ c) {
			int64_t w = out->cs[c + 1] - out->cs[c];
			for(k = out->ptr[c]; k < out->ptr[c + 1]; ++ k) {
				int64_t rw = out->row[k], h = out->cs[rw + 1] - out->cs[rw];
				int64_t s = src[k], a, b;
				int64_t oc = 0, orow = brow[s];
				const double *sv = val + aoff[s];
				double *dv;
				/* find the source column of block s */
				{ int64_t lo = 0, hi = n; while(hi - lo > 1) { int64_t mid = (lo + hi) / 2; if(ptr[mid] <= s) lo = mid; else hi = mid; } oc = lo; }
				out->off[k + 1] = out->off[k] + h * w;
				dv = out->val + out->off[k];
				if(pinv[orow] <= pinv[oc]) { /* same orientation: h x w as stored */
					memcpy(dv, sv, sizeof(double) * h * w);
				} else { /* transposed: source is w x h */
					for(a = 0; a < h; ++ a)
						for(b = 0; b < w; ++ b)
							dv[a + b * h] = sv[b + a * w];
				}
			}
		}
		free(src);
	}
	free(pinv); free(aoff); free(cnt); free(fill);
	return 0;
}

/* ---- elimination tree on the block structure (BlockMatrix.cpp:9403-9451) ---- */
static void bm_etree(const TBlockMat *A, int64_t *parent, int64_t *ancestor)
{
	int64_t j, k;
	for(j = 0; j < A->n; ++ j) {
		parent[j] = -1;
		ancestor[j] = -1;
		for(k = A->ptr[j]; k < A->ptr[j + 1]; ++ k) {
			int64_t i = A->row[k];
			while(i != -1 && i < j) {
				int64_t next = ancestor[i];
				ancestor[i] = j;
				if(next == -1)
					parent[i] = j;
				i = next;
			}
		}
	}
}

/* ---- ereach of column j: pattern of R(0:j-1, j), topologically ordered in s[top..n-1]
 * (BlockMatrix.cpp:9453-9545, the block twin of cs_ereach) ---- */
static int64_t bm_ereach(const TBlockMat *A, int64_t j, const int64_t *parent, int64_t *s, int64_t *mark)
{
	int64_t top = A->n, k, len, i;
	mark[j] = j;
	for(k = A->ptr[j]; k < A->ptr[j + 1]; ++ k) {
		i = A->row[k];
		if(i >= j)
			continue;
		for(len = 0; mark[i] != j; i = parent[i]) {
			s[len ++] = i;
			mark[i] = j;
		}
		while(len > 0)
			s[-- top] = s[-- len];
	}
	return top;
}

static int cmp_i64(const void *a, const void *b)
{
	int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
	return (x > y) - (x < y);
}

/* dense upper Cholesky of a d x d column-major block in place (A = R^T R, R upper), what
 * Eigen::LLT<MatrixXd, Upper> computes (BlockMatrix.cpp:9765-9771). returns 0 if not PD */
static int dense_chol_upper(double *a, int64_t d)
{
	int64_t i, j, k;
	for(j = 0; j < d; ++ j) {
		double s = a[j + j * d];
		for(k = 0; k < j; ++ k)
			s -= a[k + j * d] * a[k + j * d];
		if(!(s > 0))
			return 0;
		s = sqrt(s);
		a[j + j * d] = s;
		for(i = j + 1; i < d; ++ i) {
			double t = a[j + i * d];
			for(k = 0; k < j; ++ k)
				t -= a[k + j * d] * a[k + i * d];
			a[j + i * d] = t / s;
		}
		for(i = j + 1; i < d; ++ i)
			a[i + j * d] = 0; /* strictly lower part is not part of R */
	}
	return 1;
}

/* solves R_kk^T X = B in place, R_kk upper dk x dk, B dk x w (BlockMatrix.cpp:9724-9726) */
static void upper_transpose_solve(const double *R, int64_t dk, double *B, int64_t w)
{
	int64_t c, i, t;
	for(c = 0; c < w; ++ c) {
		double *b = B + c * dk;
		for(i = 0; i < dk; ++ i) {
			double s = b[i];
			for(t = 0; t < i; ++ t)
				s -= R[t + i * dk] * b[t];
			b[i] = s / R[i + i * dk];
		}
	}
}

typedef struct {
	int64_t n;
	const int64_t *cs;
	int64_t *ptr;      /* [n+1] */
	int64_t *row;      /* sorted, last of each column = diagonal */
	int64_t *off;
	double *val;
	int64_t nblocks, nvals;
} TFactor;

static void factor_free(TFactor *R)
{
	free(R->ptr); free(R->row); free(R->off); free(R->val);
	memset(R, 0, sizeof(*R));
}

/* up-looking block Cholesky (BlockMatrix.cpp:9547-9785). returns 0 ok, 1 not positive definite */
static int bm_cholesky(const TBlockMat *A, TFactor *R)
{
	const int64_t n = A->n;
	int64_t *parent = (int64_t*)malloc(sizeof(int64_t) * n), *anc = (int64_t*)malloc(sizeof(int64_t) * n);
	int64_t *stack = (int64_t*)malloc(sizeof(int64_t) * n), *mark = (int64_t*)malloc(sizeof(int64_t) * n);
	int64_t j, u, k, nb = 0, nv = 0, result = 0;
	memset(R, 0, sizeof(*R));
	R->n = n;
	R->cs = A->cs;
	bm_etree(A, parent, anc);
	/* symbolic pass: count blocks / values of every column */
	R->ptr = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
	for(j = 0; j < n; ++ j)
		mark[j] = -1;
	R->ptr[0] = 0;
	for(j = 0; j < n; ++ j) {
		int64_t top = bm_ereach(A, j, parent, stack, mark);
		int64_t w = A->cs[j + 1] - A->cs[j];
		for(u = top; u < n; ++ u)
			nv += (A->cs[stack[u] + 1] - A->cs[stack[u]]) * w;
		nv += w * w;
		nb += n - top + 1;
		R->ptr[j + 1] = nb;
	}
	R->row = (int64_t*)malloc(sizeof(int64_t) * nb);
	R->off = (int64_t*)malloc(sizeof(int64_t) * (nb + 1));
	R->val = (double*)calloc(nv? nv : 1, sizeof(double));
	R->nblocks = nb;
	R->nvals = nv;
	for(j = 0; j < n; ++ j)
		mark[j] = -1;
	R->off[0] = 0;
	for(j = 0; j < n && !result; ++ j) {
		const int64_t w = A->cs[j + 1] - A->cs[j];
		const int64_t top = bm_ereach(A, j, parent, stack, mark);
		const int64_t cnt = n - top;
		int64_t base = R->ptr[j], a;
		/* the reference keeps the column sorted by row (insertion at :9618-9627); sort up front */
		qsort(stack + top, (size_t)cnt, sizeof(int64_t), cmp_i64);
		for(u = 0; u < cnt; ++ u) {
			R->row[base + u] = stack[top + u];
			R->off[base + u + 1] = R->off[base + u] + (A->cs[stack[top + u] + 1] - A->cs[stack[top + u]]) * w;
		}
		R->row[base + cnt] = j;
		R->off[base + cnt + 1] = R->off[base + cnt] + w * w;
		/* copy A(k,j) into R(k,j) (zero where A has no block, :9650) */
		for(a = A->ptr[j]; a < A->ptr[j + 1]; ++ a) {
			int64_t r = A->row[a], lo = base, hi = base + cnt + 1;
			while(hi - lo > 1) { int64_t mid = (lo + hi) / 2; if(R->row[mid] <= r) lo = mid; else hi = mid; }
			memcpy(R->val + R->off[lo], A->val + A->off[a], sizeof(double) * (A->cs[r + 1] - A->cs[r]) * w);
		}
		/* cmod + late division, rows ascending (a valid topological order of ereach) */
		for(u = 0; u < cnt; ++ u) {
			const int64_t kk = R->row[base + u];
			const int64_t dk = A->cs[kk + 1] - A->cs[kk];
			double *Rkj = R->val + R->off[base + u];
			int64_t pj = base, pk = R->ptr[kk];
			const int64_t pk_diag = R->ptr[kk + 1] - 1;
			/* merge the block lists of columns j (rows < k) and k (:9688-9715) */
			for(; pj < base + u; ++ pj) {
				int64_t i = R->row[pj];
				while(pk < pk_diag && R->row[pk] < i)
					++ pk;
				if(pk < pk_diag && R->row[pk] == i) {
					const int64_t di = A->cs[i + 1] - A->cs[i];
					const double *Rik = R->val + R->off[pk], *Rij = R->val + R->off[pj];
					int64_t r, c, t;
					for(c = 0; c < w; ++ c)
						for(r = 0; r < dk; ++ r) {
							double s = 0;
							for(t = 0; t < di; ++ t)
								s += Rik[t + r * di] * Rij[t + c * di];
							Rkj[r + c * dk] -= s;
						}
				}
			}
			upper_transpose_solve(R->val + R->off[pk_diag], dk, Rkj, w);
		}
		/* cdiv: diagonal block (:9728-9771) */
		{
			double *Rjj = R->val + R->off[base + cnt];
			for(k = base; k < base + cnt; ++ k) {
				const int64_t i = R->row[k], di = A->cs[i + 1] - A->cs[i];
				const double *Rij = R->val + R->off[k];
				int64_t r, c, t;
				for(c = 0; c < w; ++ c)
					for(r = 0; r <= c; ++ r) { /* upper triangle only, as the reference does */
						double s = 0;
						for(t = 0; t < di; ++ t)
							s += Rij[t + r * di] * Rij[t + c * di];
						Rjj[r + c * w] -= s;
					}
			}
			if(!dense_chol_upper(Rjj, w))
				result = 1;
		}
	}
	free(parent); free(anc); free(stack); free(mark);
	return (int)result;
}

/* x = R^-1 R^-T b (BlockMatrix.cpp:8637-8719 then 8898-) */
static void factor_solve(const TFactor *R, double *x)
{
	const int64_t n = R->n;
	int64_t j, k, r, t;
	for(j = 0; j < n; ++ j) { /* R^T y = b: column j of R gives row j of R^T */
		const int64_t w = R->cs[j + 1] - R->cs[j];
		double *xj = x + R->cs[j];
		const int64_t kd = R->ptr[j + 1] - 1;
		for(k = R->ptr[j]; k < kd; ++ k) {
			const int64_t i = R->row[k], di = R->cs[i + 1] - R->cs[i];
			const double *B = R->val + R->off[k], *xi = x + R->cs[i];
			for(r = 0; r < w; ++ r) {
				double s = 0;
				for(t = 0; t < di; ++ t)
					s += B[t + r * di] * xi[t];
				xj[r] -= s;
			}
		}
		{
			const double *D = R->val + R->off[kd];
			for(r = 0; r < w; ++ r) {
				double s = xj[r];
				for(t = 0; t < r; ++ t)
					s -= D[t + r * w] * xj[t];
				xj[r] = s / D[r + r * w];
			}
		}
	}
	for(j = n; j > 0; -- j) { /* R x = y */
		const int64_t c = j - 1, w = R->cs[c + 1] - R->cs[c];
		double *xj = x + R->cs[c];
		const int64_t kd = R->ptr[c + 1] - 1;
		const double *D = R->val + R->off[kd];
		for(r = w; r > 0; -- r) {
			double s = xj[r - 1];
			for(t = r; t < w; ++ t)
				s -= D[(r - 1) + t * w] * xj[t];
			xj[r - 1] = s / D[(r - 1) + (r - 1) * w];
		}
		for(k = R->ptr[c]; k < kd; ++ k) {
			const int64_t i = R->row[k], di = R->cs[i + 1] - R->cs[i];
			const double *B = R->val + R->off[k];
			double *xi = x + R->cs[i];
			for(r = 0; r < w; ++ r)
				for(t = 0; t < di; ++ t)
					xi[t] -= B[t + r * di] * xj[r];
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * public: sparse solve.  perm[new] = old (NULL = natural order).  rhs overwritten with x.
 * stats_out (may be NULL): [0] factor blocks, [1] factor values incl. diagonal blocks in full.
 * returns 0 ok, 1 not positive definite, -1 bad input
 * ------------------------------------------------------------------------------------------ */
int oracle_solve_sparse(int64_t n, const int64_t *cs, const int64_t *ptr, const int32_t *brow,
	const double *val, double *rhs_inout, const int32_t *perm, double *stats_out)
{
	TBlockMat P;
	TFactor R;
	int result;
	int64_t i, t;
	double *b;
	if(n <= 0 || !cs || !ptr || !brow || !val || !rhs_inout)
		return -1;
	bm_permute_upper(n, cs, ptr, brow, val, perm, &P);
	b = (double*)malloc(sizeof(double) * cs[n]);
	for(i = 0; i < n; ++ i) { /* Permute_RightHandSide_Vector (BlockMatrix.cpp:9291-9401) */
		int64_t o = perm? perm[i] : i, d = cs[o + 1] - cs[o];
		for(t = 0; t < d; ++ t)
			b[P.cs[i] + t] = rhs_inout[cs[o] + t];
	}
	result = bm_cholesky(&P, &R);
	if(!result) {
		factor_solve(&R, b);
		for(i = 0; i < n; ++ i) {
			int64_t o = perm? perm[i] : i, d = cs[o + 1] - cs[o];
			for(t = 0; t < d; ++ t)
				rhs_inout[cs[o] + t] = b[P.cs[i] + t];
		}
	}
	if(stats_out) {
		stats_out[0] = (double)R.nblocks;
		stats_out[1] = (double)R.nvals;
		stats_out[2] = stats_out[3] = 0; /* smallest / largest diagonal entry of R: (max / min)^2 is a lower bound of cond_2(Lambda) */
		if(!result) {
			double f_min = 1e300, f_max = 0;
			for(i = 0; i < n; ++ i) {
				const int64_t d = P.cs[i + 1] - P.cs[i];
				const double *p_diag = R.val + R.off[R.ptr[i + 1] - 1];
				for(t = 0; t < d; ++ t) {
					const double f = p_diag[t + t * d];
					f_min = (f < f_min)? f : f_min;
					f_max = (f > f_max)? f : f_max;
				}
			}
			stats_out[2] = f_min;
			stats_out[3] = f_max;
		}
	}
	factor_free(&R);
	bm_free(&P);
	free(b);
	return result;
}

/* ------------------------------------------------------------------------------------------
 * Schur-complement solve (LinearSolver_Schur.h:1699-1886), Lambda ordered cameras first:
 *   Lambda = | A U |   S = A - U C^-1 U^T,  S dx = x - U C^-1 l,  dl = C^-1 (l - U^T dx)
 *            | V C |
 * C must be block diagonal (LinearSolver_Schur.h:1721).  S_out (N x N col-major, upper triangle
 * filled as the reference's schur_compl, rest zero) and rhs_reduced_out may be NULL.
 * returns 0 ok, 1 not positive definite, -1 bad structure
 * ------------------------------------------------------------------------------------------ */
static int spd_inverse(const double *a, double *inv, int64_t d)
{
	/* Gauss-Jordan without pivoting on an SPD block; fails on a non-positive pivot */
	double m[64 * 2];
	int64_t i, j, k;
	if(d > 8)
		return 0;
	for(i = 0; i < d; ++ i)
		for(j = 0; j < d; ++ j) {
			m[i * 2 * d + j] = a[i + j * d];
			m[i * 2 * d + d + j] = (i == j);
		}
	for(k = 0; k < d; ++ k) {
		double p = m[k * 2 * d + k];
		if(!(p > 0))
			return 0;
		for(j = 0; j < 2 * d; ++ j)
			m[k * 2 * d + j] /= p;
		for(i = 0; i < d; ++ i) {
			double f;
			if(i == k)
				continue;
			f = m[i * 2 * d + k];
			for(j = 0; j < 2 * d; ++ j)
				m[i * 2 * d + j] -= f * m[k * 2 * d + j];
		}
	}
	for(i = 0; i < d; ++ i)
		for(j = 0; j < d; ++ j)
			inv[i + j * d] = m[i * 2 * d + d + j];
	return 1;
}

int oracle_solve_schur(int64_t n, const int64_t *cs, const int64_t *ptr, const int32_t *brow,
	const double *val, double *rhs_inout, int64_t n_cut, double *S_out, double *rhs_reduced_out)
{
	int64_t c, k, r, t, q, N, nl, nb;
	int64_t *aoff;
	double *S, *x, *l, *Cinv, *W; /* W = U C^-1, stored like U */
	int result = 0;
	if(n <= 1 || n_cut <= 0 || n_cut >= n)
		return -1;
	nb = ptr[n];
	N = cs[n_cut];
	nl = cs[n] - N;
	aoff = (int64_t*)malloc(sizeof(int64_t) * (nb + 1));
	aoff[0] = 0;
	for(c = 0; c < n; ++ c)
		for(k = ptr[c]; k < ptr[c + 1]; ++ k)
			aoff[k + 1] = aoff[k] + (cs[brow[k] + 1] - cs[brow[k]]) * (cs[c + 1] - cs[c]);
	for(c = n_cut; c < n; ++ c) { /* C block diagonal: only camera rows and the diagonal */
		for(k = ptr[c]; k < ptr[c + 1]; ++ k) {
			if(brow[k] >= n_cut && brow[k] != c) { free(aoff); return -1; }
		}
		if(ptr[c + 1] == ptr[c] || brow[ptr[c + 1] - 1] != c) { free(aoff); return -1; }
	}
	S = (double*)calloc((size_t)(N * N), sizeof(double));
	x = (double*)malloc(sizeof(double) * N);
	l = (double*)malloc(sizeof(double) * (nl? nl : 1));
	Cinv = (double*)malloc(sizeof(double) * 64 * (n - n_cut));
	W = (double*)malloc(sizeof(double) * (aoff[nb]? aoff[nb] : 1));
	memcpy(x, rhs_inout, sizeof(double) * N);
	memcpy(l, rhs_inout + N, sizeof(double) * nl);
	/* S := A (upper blocks; step 8 adds A to the product, :1767) */
	for(c = 0; c < n_cut; ++ c) {
		int64_t w = cs[c + 1] - cs[c];
		for(k = ptr[c]; k < ptr[c + 1]; ++ k) {
			int64_t rr = brow[k], h = cs[rr + 1] - cs[rr];
			for(q = 0; q < w; ++ q)
				for(r = 0; r < h; ++ r)
					S[(cs[rr] + r) + (cs[c] + q) * N] += val[aoff[k] + r + q * h];
		}
	}
	for(c = n_cut; c < n && !result; ++ c) {
		const int64_t dp = cs[c + 1] - cs[c], kd = ptr[c + 1] - 1;
		double *ci = Cinv + 64 * (c - n_cut);
		const double *lp = l + (cs[c] - N);
		int64_t ka, kb;
		/* step 4: C^-1 (InverseOf_BlockDiag_FBS_Parallel, :1723-1726) */
		if(!spd_inverse(val + aoff[kd], ci, dp)) {
			result = 1;
			break;
		}
		/* step 6: W = U C^-1 (the reference computes -U C^-1; signs are folded below) */
		for(k = ptr[c]; k < kd; ++ k) {
			const int64_t h = cs[brow[k] + 1] - cs[brow[k]];
			const double *U = val + aoff[k];
			double *Wk = W + aoff[k];
			for(q = 0; q < dp; ++ q)
				for(r = 0; r < h; ++ r) {
					double s = 0;
					for(t = 0; t < dp; ++ t)
						s += U[r + t * h] * ci[t + q * dp];
					Wk[r + q * h] = s;
				}
		}
		/* step 7: S -= W U^T, upper triangle only (:1757-1759); step 9: x -= W l (:1829-1830) */
		for(ka = ptr[c]; ka < kd; ++ ka) {
			const int64_t ra = brow[ka], ha = cs[ra + 1] - cs[ra];
			const double *Wa = W + aoff[ka];
			for(r = 0; r < ha; ++ r) {
				double s = 0;
				for(t = 0; t < dp; ++ t)
					s += Wa[r + t * ha] * lp[t];
				x[cs[ra] + r] -= s;
			}
			for(kb = ka; kb < kd; ++ kb) {
				const int64_t rb = brow[kb], hb = cs[rb + 1] - cs[rb];
				const double *Ub = val + aoff[kb];
				for(q = 0; q < hb; ++ q)
					for(r = 0; r < ha; ++ r) {
						double s = 0;
						for(t = 0; t < dp; ++ t)
							s += Wa[r + t * ha] * Ub[q + t * hb];
						S[(cs[ra] + r) + (cs[rb] + q) * N] -= s;
					}
			}
		}
	}
	if(!result) {
		/* keep only the upper triangle (the diagonal blocks of A are stored in full) */
		for(q = 0; q < N; ++ q)
			for(r = q + 1; r < N; ++ r)
				S[r + q * N] = 0;
		if(S_out)
			memcpy(S_out, S, sizeof(double) * N * N);
		if(rhs_reduced_out)
			memcpy(rhs_reduced_out, x, sizeof(double) * N);
		/* step 10: dense LLT on S (LinearSolver_Schur.cpp:2314-2331), then the two substitutions */
		for(q = 0; q < N && !result; ++ q) { /* R^T R = S, R upper, column by column */
			double s = S[q + q * N];
			for(t = 0; t < q; ++ t)
				s -= S[t + q * N] * S[t + q * N];
			if(!(s > 0)) { result = 1; break; }
			s = sqrt(s);
			S[q + q * N] = s;
			for(c = q + 1; c < N; ++ c) {
				double v = S[q + c * N];
				for(t = 0; t < q; ++ t)
					v -= S[t + q * N] * S[t + c * N];
				S[q + c * N] = v / s;
			}
		}
	}
	if(!result) {
		for(q = 0; q < N; ++ q) {
			double s = x[q];
			for(t = 0; t < q; ++ t)
				s -= S[t + q * N] * x[t];
			x[q] = s / S[q + q * N];
		}
		for(q = N; q > 0; -- q) {
			double s = x[q - 1];
			for(t = q; t < N; ++ t)
				s -= S[(q - 1) + t * N] * x[t];
			x[q - 1] = s / S[(q - 1) + (q - 1) * N];
		}
		memcpy(rhs_inout, x, sizeof(double) * N);
		/* steps 11-12: dl = C^-1 (l - U^T dx) (:1869-1881) */
		for(c = n_cut; c < n; ++ c) {
			const int64_t dp = cs[c + 1] - cs[c], kd = ptr[c + 1] - 1;
			const double *ci = Cinv + 64 * (c - n_cut);
			double v[8], *out = rhs_inout + cs[c];
			for(t = 0; t < dp; ++ t)
				v[t] = l[cs[c] - N + t];
			for(k = ptr[c]; k < kd; ++ k) {
				const int64_t rr = brow[k], h = cs[rr + 1] - cs[rr];
				const double *U = val + aoff[k];
				for(t = 0; t < dp; ++ t) {
					double s = 0;
					for(r = 0; r < h; ++ r)
						s += U[r + t * h] * x[cs[rr] + r];
					v[t] -= s;
				}
			}
			for(r = 0; r < dp; ++ r) {
				double s = 0;
				for(t = 0; t < dp; ++ t)
					s += ci[r + t * dp] * v[t];
				out[r] = s;
			}
		}
	}
	free(aoff); free(S); free(x); free(l); free(Cinv); free(W);
	return result;
}

/* ------------------------------------------------------------------------------------------
 * CPU replay of the product's elimination plan (see slampp_hip_plan_view in include/slampp_hip.h):
 * the same left-looking arithmetic the HIP kernels perform, stage by stage, task by task.
 * Checks the schedule as it goes: every operand block must have been produced earlier.
 * returns 0 ok, 1 not positive definite, 2 schedule violation
 * ------------------------------------------------------------------------------------------ */
int oracle_exec_plan(int64_t n, const int32_t *perm, const int32_t *dim, const int64_t *lptr,
	const int32_t *lrow, const int64_t *loff, const int64_t *asrc, const int32_t *atrans,
	const int64_t *pptr, const int32_t *pa, const int32_t *pb, const int64_t *rptr, const int32_t *rblk,
	int64_t n_stages, const int32_t *stage_ptr, const int64_t *task_ptr, const int32_t *task_cols,
	const int64_t *cs_old, int64_t l_values, const double *val, double *rhs_inout,
	const int32_t *dense_pos, int64_t dense_dim)
{
	const int64_t nbl = lptr[n];
	double *L = (double*)calloc(l_values? l_values : 1, sizeof(double));
	double *Linv = (double*)calloc(64 * n, sizeof(double));
	int64_t *cs_new = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
	int32_t *blk_col = (int32_t*)malloc(sizeof(int32_t) * (nbl? nbl : 1));
	int32_t *done_stage = (int32_t*)malloc(sizeof(int32_t) * n); /* stage in which a column was finished */
	int32_t *done_task = (int32_t*)malloc(sizeof(int32_t) * n);
	double *w;
	int64_t s, t, c, j, k, e;
	int result = 0;
	cs_new[0] = 0;
	for(j = 0; j < n; ++ j) {
		cs_new[j + 1] = cs_new[j] + dim[j];
		done_stage[j] = -1;
		done_task[j] = -1;
		for(k = lptr[j]; k < lptr[j + 1]; ++ k)
			blk_col[k] = (int32_t)j;
	}
	w = (double*)calloc(cs_new[n], sizeof(double));
#define OPERAND_READY(col, stage, task) (done_stage[col] >= 0 && (done_stage[col] < (stage) || done_task[col] == (task)))
	for(s = 0; s < n_stages && !result; ++ s) {
		for(t = stage_ptr[s]; t < stage_ptr[s + 1] && !result; ++ t) {
			for(c = task_ptr[t]; c < task_ptr[t + 1] && !result; ++ c) {
				const int64_t dj = dim[j = task_cols[c]];
				const int64_t k0 = lptr[j];
				double *Ljj = L + loff[k0], *Li = Linv + 64 * j;
				int64_t r, q, u;
				for(k = k0; k < lptr[j + 1]; ++ k) {
					const int64_t i = lrow[k], di = dim[i];
					double *B = L + loff[k];
					if(asrc[k] >= 0) {
						const double *src = val + asrc[k];
						for(q = 0; q < dj; ++ q)
							for(r = 0; r < di; ++ r)
								B[r + q * di] = (atrans[k] || k == k0)? src[q + r * dj] : src[r + q * di];
					}
					for(e = pptr[k]; e < pptr[k + 1]; ++ e) {
						const int64_t cc = blk_col[pa[e]], dc = dim[cc];
						const double *Pa = L + loff[pa[e]], *Pb = L + loff[pb[e]];
						if(blk_col[pb[e]] != cc || lrow[pa[e]] != i || lrow[pb[e]] != j ||
						   !OPERAND_READY(cc, s, t)) {
							result = 2;
							break;
						}
						for(q = 0; q < dj; ++ q)
							for(r = 0; r < di; ++ r) {
								double sum = 0;
								for(u = 0; u < dc; ++ u)
									sum += Pa[r + u * di] * Pb[q + u * dj];
								B[r + q * di] -= sum;
							}
					}
					if(result)
						break;
					if(k == k0) { /* diagonal: lower Cholesky + inverse */
						for(q = 0; q < dj; ++ q) {
							double p = Ljj[q + q * dj];
							for(u = 0; u < q; ++ u)
								p -= Ljj[q + u * dj] * Ljj[q + u * dj];
							if(!(p > 0)) { result = 1; p = 1; }
							p = sqrt(p);
							Ljj[q + q * dj] = p;
							for(r = q + 1; r < dj; ++ r) {
								double v = Ljj[r + q * dj];
								for(u = 0; u < q; ++ u)
									v -= Ljj[r + u * dj] * Ljj[q + u * dj];
								Ljj[r + q * dj] = v / p;
							}
							for(r = 0; r < q; ++ r)
								Ljj[r + q * dj] = 0;
						}
						for(q = 0; q < dj; ++ q)
							for(r = 0; r < dj; ++ r) {
								double v;
								if(r < q)
									v = 0;
								else if(r == q)
									v = 1 / Ljj[r + r * dj];
								else {
									double sum = 0;
									for(u = q; u < r; ++ u)
										sum += Ljj[r + u * dj] * Li[u + q * dj];
									v = -sum / Ljj[r + r * dj];
								}
								Li[r + q * dj] = v;
							}
					} else { /* L(i,j) = acc * Linv^T */
						double tmp[64];
						for(q = 0; q < dj; ++ q)
							for(r = 0; r < di; ++ r) {
								double sum = 0;
								for(u = 0; u <= q; ++ u)
									sum += B[r + u * di] * Li[q + u * dj];
								tmp[r + q * di] = sum;
							}
						memcpy(B, tmp, sizeof(double) * di * dj);
					}
				}
				done_stage[j] = (int32_t)s;
				done_task[j] = (int32_t)t;
			}
		}
	}
	for(j = 0; j < n && !result; ++ j) {
		if(done_stage[j] < 0 && !(dense_dim && dense_pos[j] >= 0))
			result = 2; /* a column was never scheduled */
	}
	/* forward substitution of the block-eliminated columns (needed before the dense top's right-hand side) */
	if(!result) {
		/* forward substitution in schedule order */
		for(s = 0; s < n_stages; ++ s)
			for(t = stage_ptr[s]; t < stage_ptr[s + 1]; ++ t)
				for(c = task_ptr[t]; c < task_ptr[t + 1]; ++ c) {
					const int64_t dj = dim[j = task_cols[c]];
					double v[8];
					int64_t r, u;
					for(r = 0; r < dj; ++ r)
						v[r] = rhs_inout[cs_old[perm[j]] + r];
					for(e = rptr[j]; e < rptr[j + 1]; ++ e) {
						const int64_t kb = rblk[e], cc = blk_col[kb], dc = dim[cc];
						const double *B = L + loff[kb];
						if(lrow[kb] != j) result = 2;
						for(r = 0; r < dj; ++ r)
							for(u = 0; u < dc; ++ u)
								v[r] -= B[r + u * dj] * w[cs_new[cc] + u];
					}
					for(r = 0; r < dj; ++ r) {
						double sum = 0;
						for(u = 0; u <= r; ++ u)
							sum += Linv[64 * j + r + u * dj] * v[u];
						w[cs_new[j] + r] = sum;
					}
				}
	}
	if(!result && dense_dim) {
		/* dense top: assemble the Schur complement onto the dense-top columns from the kept update
		 * lists (sources outside the dense top only), factor it densely, solve for x of those columns */
		const int64_t N = dense_dim;
		double *Dm = (double*)calloc((size_t)(N * N), sizeof(double)), *rd = (double*)calloc((size_t)N, sizeof(double));
		for(j = 0; j < n && !result; ++ j) {
			int64_t dj, r, q, u;
			if(dense_pos[j] < 0)
				continue;
			dj = dim[j];
			for(k = lptr[j]; k < lptr[j + 1] && !result; ++ k) {
				const int64_t i = lrow[k], di = dim[i];
				if(dense_pos[i] < 0) { result = 2; break; }
				for(q = 0; q < dj; ++ q)
					for(r = 0; r < di; ++ r) {
						double v = 0;
						if(asrc[k] >= 0)
							v = (atrans[k] || k == lptr[j])? val[asrc[k] + q + r * dj] : val[asrc[k] + r + q * di];
						Dm[(dense_pos[i] + r) + (dense_pos[j] + q) * N] = v;
					}
				for(e = pptr[k]; e < pptr[k + 1]; ++ e) {
					const int64_t cc = blk_col[pa[e]], dc = dim[cc];
					const double *Pa = L + loff[pa[e]], *Pb = L + loff[pb[e]];
					if(dense_pos[cc] >= 0 || done_stage[cc] < 0 || lrow[pa[e]] != i || lrow[pb[e]] != j) { result = 2; break; }
					for(q = 0; q < dj; ++ q)
						for(r = 0; r < di; ++ r) {
							double sum = 0;
							for(u = 0; u < dc; ++ u)
								sum += Pa[r + u * di] * Pb[q + u * dj];
							Dm[(dense_pos[i] + r) + (dense_pos[j] + q) * N] -= sum;
						}
				}
			}
			for(r = 0; r < dj; ++ r)
				rd[dense_pos[j] + r] = rhs_inout[cs_old[perm[j]] + r];
			for(e = rptr[j]; e < rptr[j + 1]; ++ e) {
				const int64_t kb = rblk[e], cc = blk_col[kb], dc = dim[cc];
				const double *B = L + loff[kb];
				if(dense_pos[cc] >= 0) { result = 2; break; }
				for(r = 0; r < dj; ++ r)
					for(u = 0; u < dc; ++ u)
						rd[dense_pos[j] + r] -= B[r + u * dj] * w[cs_new[cc] + u];
			}
		}
		if(!result) { /* positions no column maps to (the plan aligns independent chains to tile boundaries): identity */
			char *covered = (char*)calloc((size_t)N, 1);
			int64_t q;
			for(j = 0; j < n; ++ j) {
				if(dense_pos[j] >= 0)
					memset(covered + dense_pos[j], 1, (size_t)dim[j]);
			}
			for(q = 0; q < N; ++ q) {
				if(!covered[q])
					Dm[q + q * N] = 1.0;
			}
			free(covered);
		}
		if(!result) { /* dense lower Cholesky + two substitutions */
			int64_t r, q, u;
			for(q = 0; q < N && !result; ++ q) { /* right-looking, column-oriented: every inner loop is contiguous */
				double p = Dm[q + q * N], *cq = Dm + q * N;
				if(!(p > 0)) { result = 1; break; }
				p = sqrt(p);
				cq[q] = p;
				for(r = q + 1; r < N; ++ r)
					cq[r] /= p;
				for(u = q + 1; u < N; ++ u) {
					const double f = cq[u];
					double *cu = Dm + u * N;
					if(f != 0)
						for(r = u; r < N; ++ r)
							cu[r] -= cq[r] * f;
				}
			}
			for(q = 0; q < N && !result; ++ q) {
				double v = rd[q];
				for(u = 0; u < q; ++ u)
					v -= Dm[q + u * N] * rd[u];
				rd[q] = v / Dm[q + q * N];
			}
			for(q = N; q > 0 && !result; -- q) {
				double v = rd[q - 1];
				for(u = q; u < N; ++ u)
					v -= Dm[u + (q - 1) * N] * rd[u];
				rd[q - 1] = v / Dm[(q - 1) + (q - 1) * N];
			}
			for(j = 0; j < n && !result; ++ j) {
				if(dense_pos[j] < 0)
					continue;
				for(r = 0; r < dim[j]; ++ r) {
					w[cs_new[j] + r] = rd[dense_pos[j] + r];
					rhs_inout[cs_old[perm[j]] + r] = rd[dense_pos[j] + r];
				}
			}
		}
		free(Dm); free(rd);
	}
	if(!result) {
		/* backward substitution of the block-eliminated columns, schedule reversed */
		for(s = n_stages; s > 0; -- s)
			for(t = stage_ptr[s]; t > stage_ptr[s - 1]; -- t)
				for(c = task_ptr[t]; c > task_ptr[t - 1]; -- c) {
					const int64_t dj = dim[j = task_cols[c - 1]];
					double v[8];
					int64_t r, u;
					for(r = 0; r < dj; ++ r)
						v[r] = w[cs_new[j] + r];
					for(k = lptr[j] + 1; k < lptr[j + 1]; ++ k) {
						const int64_t i = lrow[k], di = dim[i];
						const double *B = L + loff[k];
						for(r = 0; r < dj; ++ r)
							for(u = 0; u < di; ++ u)
								v[r] -= B[u + r * di] * w[cs_new[i] + u];
					}
					for(r = 0; r < dj; ++ r) {
						double sum = 0;
						for(u = r; u < dj; ++ u)
							sum += Linv[64 * j + u + r * dj] * v[u];
						w[cs_new[j] + r] = sum;
					}
					/* note: w[j] must not be read as y_j by anyone after this point; later columns
					 * (smaller j) only read x_i, i > j */
					for(r = 0; r < dj; ++ r)
						rhs_inout[cs_old[perm[j]] + r] = w[cs_new[j] + r];
				}
	}
#undef OPERAND_READY
	free(L); free(Linv); free(cs_new); free(blk_col); free(done_stage); free(done_task); free(w);
	return result;
}

/* ---------------------------------------------------------------------------------------------
 * Lambda / eta assembly from per-edge Jacobians (the row after the solve in SURVEY.md section 8f).
 * Restates, for one homogeneous set of binary edges:
 *   per-edge Hessian blocks and right-hand sides       include/slam/BaseTypes_Binary.h:759-840
 *       H0S = J0^T Sigma^-1 [* w]                       :771-777
 *       off-diagonal (min id, max id) = H0S J1, or J1^T H0S^T when id0 > id1   :779-806
 *       vertex 0: H0S J0,  rhs H0S e [* w]  (w enters twice for a robust edge)  :809-823
 *       vertex 1: J1^T Sigma^-1 J1 [* w],  rhs J1^T (Sigma^-1 e) [* w]          :824-840
 *   the sums per block / per vertex (the reduction plan) include/slam/NonlinearSolver_Lambda_Base.h:1634-1688
 *   the unary factor on the anchor vertex: += U^T U, eta += unary error        :1520-1580
 * Pinned against Lambda / eta recorded from the reference's CNonlinearSolver_Lambda (ref_harness lambda_dump,
 * tests/golden/assembly_*.npz).  weight may be NULL (not a robust edge).  Returns 0, or -1 on a bad edge.
 * ------------------------------------------------------------------------------------------- */
int oracle_assemble_lambda(int64_t n, const int64_t *cs, const int64_t *ptr, const int32_t *brow,
	int64_t n_edges, const int64_t *v0, const int64_t *v1, int64_t rd,
	const double *J0, const double *J1, const double *sigma_inv, const double *err, const double *weight,
	int64_t unary_vertex, const double *unary_factor, const double *unary_error,
	double *values, double *eta)
{
	int64_t *voff = (int64_t*)malloc((size_t)(ptr[n] + 1) * sizeof(int64_t));
	int64_t c, k, e, i, j, a, b;
	if(!voff)
		return -1;
	voff[0] = 0;
	for(c = 0; c < n; ++ c)
		for(k = ptr[c]; k < ptr[c + 1]; ++ k)
			voff[k + 1] = voff[k] + (cs[brow[k] + 1] - cs[brow[k]]) * (cs[c + 1] - cs[c]);
	memset(values, 0, (size_t)voff[ptr[n]] * sizeof(double));
	memset(eta, 0, (size_t)cs[n] * sizeof(double));
	for(e = 0; e < n_edges; ++ e) {
		const int64_t id0 = v0[e], id1 = v1[e];
		const int64_t d0 = cs[id0 + 1] - cs[id0], d1 = cs[id1 + 1] - cs[id1];
		const double *j0 = J0 + e * rd * d0, *j1 = J1 + e * rd * d1, *S = sigma_inv + e * rd * rd, *er = err + e * rd;
		const double w = weight? weight[e] : 1.0;
		double H0S[8 * 8], H1S[8 * 8], Se[8]; /* H0S: d0 x rd column-major */
		const int64_t r = (id0 < id1)? id0 : id1, cc = (id0 < id1)? id1 : id0;
		double *dst = 0, *d00, *d11;
		if(id0 == id1 || d0 > 8 || d1 > 8 || rd > 8) { free(voff); return -1; }
		for(k = ptr[cc]; k < ptr[cc + 1]; ++ k)
			if(brow[k] == r) dst = values + voff[k];
		if(!dst || brow[ptr[id0 + 1] - 1] != id0 || brow[ptr[id1 + 1] - 1] != id1) { free(voff); return -1; }
		d00 = values + voff[ptr[id0 + 1] - 1];
		d11 = values + voff[ptr[id1 + 1] - 1];
		for(i = 0; i < d0; ++ i)
			for(b = 0; b < rd; ++ b) {
				double s = 0;
				for(a = 0; a < rd; ++ a) s += j0[a + i * rd] * S[a + b * rd];
				H0S[i + b * d0] = s * w;
			}
		for(i = 0; i < d1; ++ i)
			for(b = 0; b < rd; ++ b) {
				double s = 0;
				for(a = 0; a < rd; ++ a) s += j1[a + i * rd] * S[a + b * rd];
				H1S[i + b * d1] = s;
			}
		for(a = 0; a < rd; ++ a) {
			double s = 0;
			for(b = 0; b < rd; ++ b) s += S[a + b * rd] * er[b];
			Se[a] = s;
		}
		/* off-diagonal: (i, j) of H0S J1 goes to (i, j) of a d0 x d1 block, or to (j, i) of a d1 x d0 block */
		for(i = 0; i < d0; ++ i)
			for(j = 0; j < d1; ++ j) {
				double s = 0;
				for(b = 0; b < rd; ++ b) s += H0S[i + b * d0] * j1[b + j * rd];
				if(id0 < id1) dst[i + j * d0] += s;
				else dst[j + i * d1] += s;
			}
		for(i = 0; i < d0; ++ i) {
			double s = 0;
			for(j = 0; j < d0; ++ j) {
				double h = 0;
				for(b = 0; b < rd; ++ b) h += H0S[i + b * d0] * j0[b + j * rd];
				d00[i + j * d0] += h;
			}
			for(b = 0; b < rd; ++ b) s += H0S[i + b * d0] * er[b];
			eta[cs[id0] + i] += s * w; /* sic: w is already in H0S (:813-815) */
		}
		for(i = 0; i < d1; ++ i) {
			double s = 0;
			for(j = 0; j < d1; ++ j) {
				double h = 0;
				for(b = 0; b < rd; ++ b) h += H1S[i + b * d1] * j1[b + j * rd];
				d11[i + j * d1] += h * w;
			}
			for(a = 0; a < rd; ++ a) s += j1[a + i * rd] * Se[a];
			eta[cs[id1] + i] += s * w;
		}
	}
	if(unary_factor && unary_vertex >= 0 && unary_vertex < n) {
		const int64_t d = cs[unary_vertex + 1] - cs[unary_vertex];
		double *dd = values + voff[ptr[unary_vertex + 1] - 1];
		for(i = 0; i < d; ++ i)
			for(j = 0; j < d; ++ j) {
				double s = 0;
				for(k = 0; k < d; ++ k) s += unary_factor[k + i * d] * unary_factor[k + j * d];
				dd[i + j * d] += s;
			}
		for(i = 0; i < d && unary_error; ++ i)
			eta[cs[unary_vertex] + i] += unary_error[i];
	}
	free(voff);
	return 0;
}
