/* gmg_oracle.c -- CPU ORACLE (test infrastructure only; never linked into the product).
 *
 * Plain-C restatement of the hot path of the reference application
 * (/root/reference/src/step-50.cc:938-1017, LaplaceProblem::solve) and of the third-party
 * library behaviour it delegates to.  deal.II (>= 9.0.0, CMakeLists.txt:29), Trilinos
 * Epetra/Ifpack and p4est are NOT vendored in the reference and are not installed, so the
 * reference binary is unbuildable here; this restatement is pinned against the reference's
 * own golden logs (tests/golden/, see tests/test_oracle_golden.py).
 *
 * What is restated, and from where:
 *   oracle_spmv ............ TrilinosWrappers::SparseMatrix::vmult   (call sites :991, V-cycle)
 *   cg_solve ............... deal.II SolverCG<vector_t>::solve       (:943, :963, :991-992)
 *                            operation order: g=Ax-b | -b ; res=|g| ; h=M^-1 g ; d=-h ;
 *                            gh=g.h ; loop { h=Ad ; alpha=gh/(d.h) ; x+=alpha d ;
 *                            g+=alpha h ; res=|g| ; check ; h=M^-1 g ; beta=(g.h)/gh ;
 *                            d=beta d-h }.  Identity branch: d=-g, gh=res*res.
 *   SolverControl .......... success if res<=tol, else failure if step>=max   (:942, :962)
 *   oracle_vcycle .......... PreconditionMG::vmult + Multigrid::level_v_step (:980-989):
 *                            copy_to_mg ; pre apply ; t=A u (+I u) ; t=defect-t ;
 *                            defect[l-1]+=P^T t ; recurse ; u+=P u[l-1] ;
 *                            defect-=I^T u ; post smooth ; copy_from_mg (dst zeroed first)
 *   smoother ............... MGSmootherPrecondition<.., Smoother, ..>, set_steps(2) (:969-973)
 *       JACOBI  Ifpack point relaxation: y = omega * r * (1/a_ii)
 *       SSOR    Ifpack symmetric Gauss-Seidel, 1 sweep, zero start, damping omega, in local
 *               row order: y_i += omega (r_i - sum_j a_ij y_j) (1/a_ii), i up then down
 *       CHEBYSHEV  not in the reference: defined by this build as the
 *               Ifpack_Chebyshev recurrence on D^-1 A over [lmax/ratio, lmax], lmax = the
 *               Gershgorin bound max_i sum_j |a_ij|/|a_ii| unless given.  PARITY UNPINNED
 *               by the reference; pinned GPU-vs-this-file only.
 *   coarse solver .......... MGCoarseGridIterativeSolver<SolverCG, PreconditionIdentity>,
 *                            SolverControl(1000, 1e-10 absolute)       (:960-967)
 *
 * Floating point: compile with -ffp-contract=off so that products and sums round
 * separately, in CSR order; the HIP SpMV keeps the same order and is compared bit-exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_SMOOTHER_JACOBI 0
#define ORACLE_SMOOTHER_SSOR 1
#define ORACLE_SMOOTHER_CHEBYSHEV 2

#define ORACLE_PRECOND_GMG 0
#define ORACLE_PRECOND_JACOBI 1
#define ORACLE_PRECOND_IDENTITY 2

#define ORACLE_OK 0
#define ORACLE_ERR_OUTER_NOCONV 2
#define ORACLE_ERR_COARSE_NOCONV 3
#define ORACLE_ERR_ARG 4

typedef struct {
  int64_t n_rows, n_cols;
  int64_t *rowptr;
  int32_t *col;
  double *val;
} csr_t;

typedef struct {
  csr_t A, I, P; /* P: this level -> next finer level (n_{l+1} x n_l) */
  int has_I, has_P;
  double *invdiag;
  double cheb_lmax;
  int64_t n_copy;
  int32_t *copy_global, *copy_level;
  double *sol, *def, *t, *w1, *w2;
} level_t;

typedef struct oracle_mg {
  int n_levels;
  level_t *lv;
  csr_t S; /* system matrix */
  int has_S;
  double *S_invdiag;
  int smoother, steps, cheb_degree, ssor_blocks;
  double omega, cheb_ratio, cheb_lmax_user;
  double coarse_tol;
  int coarse_maxit;
  int threads;
  int64_t coarse_iters_total;
  int last_coarse_iters;
  int error;
} oracle_mg;

static int g_threads = 1;

static void csr_free(csr_t *m) {
  free(m->rowptr); free(m->col); free(m->val);
  memset(m, 0, sizeof *m);
}

static void csr_copy(csr_t *m, int64_t nr, int64_t nc, const int64_t *rp, const int32_t *col, const double *val) {
  csr_free(m);
  m->n_rows = nr; m->n_cols = nc;
  int64_t nnz = rp[nr];
  m->rowptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nr + 1));
  m->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  m->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  memcpy(m->rowptr, rp, sizeof(int64_t) * (size_t)(nr + 1));
  memcpy(m->col, col, sizeof(int32_t) * (size_t)nnz);
  memcpy(m->val, val, sizeof(double) * (size_t)nnz);
}

/* ------------------------------------------------------------------ vector_t ops (A4) */

static void v_zero(double *x, int64_t n) { memset(x, 0, sizeof(double) * (size_t)n); }

static double v_dot(const double *x, const double *y, int64_t n) {
  double s = 0.0;
  if (g_threads > 1) {
#pragma omp parallel for reduction(+ : s) num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += x[i] * y[i];
  } else {
    for (int64_t i = 0; i < n; ++i) s += x[i] * y[i];
  }
  return s;
}

static void v_axpy(double *y, double a, const double *x, int64_t n) { /* y += a x */
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
}

static void v_sadd(double *y, double s, double a, const double *x, int64_t n) { /* y = s y + a x */
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int64_t i = 0; i < n; ++i) y[i] = s * y[i] + a * x[i];
}

static void v_equ(double *y, double a, const double *x, int64_t n) { /* y = a x */
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int64_t i = 0; i < n; ++i) y[i] = a * x[i];
}

static int v_all_zero(const double *x, int64_t n) {
  for (int64_t i = 0; i < n; ++i)
    if (x[i] != 0.0) return 0;
  return 1;
}

/* ------------------------------------------------------------------ SpMV family (A3, A10) */

static void spmv(const csr_t *A, const double *x, double *y, int add) {
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int64_t i = 0; i < A->n_rows; ++i) {
    double s = add ? y[i] : 0.0;
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) s += A->val[k] * x[A->col[k]];
    y[i] = s;
  }
}

/* y (+)= A^T x, sequential scatter in ascending row order (== stable CSR transpose order) */
static void spmv_t(const csr_t *A, const double *x, double *y, int add) {
  if (!add) v_zero(y, A->n_cols);
  for (int64_t i = 0; i < A->n_rows; ++i) {
    const double xi = x[i];
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) y[A->col[k]] += A->val[k] * xi;
  }
}

int oracle_spmv(int64_t n_rows, const int64_t *rowptr, const int32_t *col, const double *val,
                const double *x, double *y) {
  csr_t A = {n_rows, 0, (int64_t *)rowptr, (int32_t *)col, (double *)val};
  spmv(&A, x, y, 0);
  return ORACLE_OK;
}

int oracle_spmv_transpose(int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int32_t *col,
                          const double *val, const double *x, double *y, int add) {
  csr_t A = {n_rows, n_cols, (int64_t *)rowptr, (int32_t *)col, (double *)val};
  spmv_t(&A, x, y, add);
  return ORACLE_OK;
}

void oracle_set_threads(int n) {
  g_threads = n > 0 ? n : 1;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ object graph */

oracle_mg *oracle_mg_create(int n_levels) {
  oracle_mg *mg = (oracle_mg *)calloc(1, sizeof *mg);
  mg->n_levels = n_levels;
  mg->lv = (level_t *)calloc((size_t)n_levels, sizeof(level_t));
  mg->smoother = ORACLE_SMOOTHER_SSOR; /* src/step-50.cc:970 */
  mg->omega = 0.5;                     /* :972 */
  mg->steps = 2;                       /* :973 */
  mg->cheb_degree = 2;
  mg->ssor_blocks = 1; /* 1 = the reference on one rank; B = block Jacobi of rank-local SGS on B ranks */
  mg->cheb_ratio = 30.0;
  mg->cheb_lmax_user = 0.0;
  mg->coarse_tol = 1e-10; /* :962 */
  mg->coarse_maxit = 1000;
  return mg;
}

void oracle_mg_destroy(oracle_mg *mg) {
  if (!mg) return;
  for (int l = 0; l < mg->n_levels; ++l) {
    level_t *L = &mg->lv[l];
    csr_free(&L->A); csr_free(&L->I); csr_free(&L->P);
    free(L->invdiag); free(L->copy_global); free(L->copy_level);
    free(L->sol); free(L->def); free(L->t); free(L->w1); free(L->w2);
  }
  csr_free(&mg->S);
  free(mg->S_invdiag);
  free(mg->lv);
  free(mg);
}

static double *inverse_diagonal(const csr_t *A, double *gersh) {
  double *d = (double *)calloc((size_t)A->n_rows, sizeof(double));
  double lmax = 0.0;
  for (int64_t i = 0; i < A->n_rows; ++i) {
    double aii = 0.0, rs = 0.0;
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) {
      if (A->col[k] == i) aii = A->val[k];
      rs += fabs(A->val[k]);
    }
    d[i] = 1.0 / aii;
    double r = rs / fabs(aii);
    if (r > lmax) lmax = r;
  }
  if (gersh) *gersh = lmax;
  return d;
}

int oracle_mg_set_level_matrix(oracle_mg *mg, int level, int64_t n, const int64_t *rp, const int32_t *col,
                               const double *val) {
  if (level < 0 || level >= mg->n_levels) return ORACLE_ERR_ARG;
  level_t *L = &mg->lv[level];
  csr_copy(&L->A, n, n, rp, col, val);
  free(L->invdiag);
  L->invdiag = inverse_diagonal(&L->A, &L->cheb_lmax);
  free(L->sol); free(L->def); free(L->t); free(L->w1); free(L->w2);
  L->sol = (double *)calloc((size_t)n, sizeof(double));
  L->def = (double *)calloc((size_t)n, sizeof(double));
  L->t = (double *)calloc((size_t)n, sizeof(double));
  L->w1 = (double *)calloc((size_t)n, sizeof(double));
  L->w2 = (double *)calloc((size_t)n, sizeof(double));
  return ORACLE_OK;
}

int oracle_mg_set_edge_matrix(oracle_mg *mg, int level, int64_t n, const int64_t *rp, const int32_t *col,
                              const double *val) {
  if (level < 0 || level >= mg->n_levels) return ORACLE_ERR_ARG;
  level_t *L = &mg->lv[level];
  csr_copy(&L->I, n, n, rp, col, val);
  L->has_I = rp[n] > 0;
  return ORACLE_OK;
}

/* P maps level -> level+1 */
int oracle_mg_set_prolongation(oracle_mg *mg, int level, int64_t n_fine, int64_t n_coarse, const int64_t *rp,
                               const int32_t *col, const double *val) {
  if (level < 0 || level >= mg->n_levels - 1) return ORACLE_ERR_ARG;
  level_t *L = &mg->lv[level];
  csr_copy(&L->P, n_fine, n_coarse, rp, col, val);
  L->has_P = 1;
  return ORACLE_OK;
}

int oracle_mg_set_copy_indices(oracle_mg *mg, int level, int64_t n, const int32_t *global_idx,
                               const int32_t *level_idx) {
  if (level < 0 || level >= mg->n_levels) return ORACLE_ERR_ARG;
  level_t *L = &mg->lv[level];
  free(L->copy_global); free(L->copy_level);
  L->n_copy = n;
  L->copy_global = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  L->copy_level = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  memcpy(L->copy_global, global_idx, sizeof(int32_t) * (size_t)n);
  memcpy(L->copy_level, level_idx, sizeof(int32_t) * (size_t)n);
  return ORACLE_OK;
}

int oracle_mg_set_system_matrix(oracle_mg *mg, int64_t n, const int64_t *rp, const int32_t *col, const double *val) {
  csr_copy(&mg->S, n, n, rp, col, val);
  mg->has_S = 1;
  free(mg->S_invdiag);
  mg->S_invdiag = inverse_diagonal(&mg->S, NULL);
  return ORACLE_OK;
}

int oracle_mg_set_smoother(oracle_mg *mg, int kind, double omega, int steps, int cheb_degree, double cheb_ratio,
                           double cheb_lmax) {
  if (kind < 0 || kind > 2 || steps < 0) return ORACLE_ERR_ARG;
  mg->smoother = kind; mg->omega = omega; mg->steps = steps;
  if (cheb_degree > 0) mg->cheb_degree = cheb_degree;
  if (cheb_ratio > 0) mg->cheb_ratio = cheb_ratio;
  mg->cheb_lmax_user = cheb_lmax;
  return ORACLE_OK;
}

int oracle_mg_set_ssor_blocks(oracle_mg *mg, int n_blocks) {
  mg->ssor_blocks = n_blocks < 1 ? 1 : n_blocks;
  return ORACLE_OK;
}

int oracle_mg_set_coarse(oracle_mg *mg, double abs_tol, int max_it) {
  mg->coarse_tol = abs_tol; mg->coarse_maxit = max_it;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------ smoother "preconditioner" S (A7) */

static void smoother_apply_inverse(oracle_mg *mg, int l, double *y, const double *r) {
  level_t *L = &mg->lv[l];
  const int64_t n = L->A.n_rows;
  const double om = mg->omega;
  if (mg->smoother == ORACLE_SMOOTHER_JACOBI) {
    for (int64_t i = 0; i < n; ++i) y[i] = (om * r[i]) * L->invdiag[i];
  } else if (mg->smoother == ORACLE_SMOOTHER_SSOR) {
    v_zero(y, n);
    const csr_t *A = &L->A;
    /* Ifpack's local matrix on each rank is the diagonal block: off-rank columns are dropped.
     * Blocks = equal runs of consecutive rows, as in gmg_coulomb.hip:setup_sgs. */
    int nb = mg->ssor_blocks < 1 ? 1 : mg->ssor_blocks;
    if (nb > (n + 63) / 64) nb = (int)((n + 63) / 64);
    if (nb < 1) nb = 1;
    for (int b = 0; b < nb; ++b) {
      const int64_t rb = n * b / nb, re = n * (b + 1) / nb;
      for (int64_t i = rb; i < re; ++i) {
        double s = 0.0;
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
          if (A->col[k] >= rb && A->col[k] < re) s += A->val[k] * y[A->col[k]];
        y[i] += om * (r[i] - s) * L->invdiag[i];
      }
      for (int64_t i = re; i-- > rb;) {
        double s = 0.0;
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
          if (A->col[k] >= rb && A->col[k] < re) s += A->val[k] * y[A->col[k]];
        y[i] += om * (r[i] - s) * L->invdiag[i];
      }
    }
  } else { /* Chebyshev on D^-1 A, Ifpack_Chebyshev recurrence, zero start */
    const double lmax = mg->cheb_lmax_user > 0 ? mg->cheb_lmax_user : L->cheb_lmax;
    const double alpha = lmax / mg->cheb_ratio, beta = lmax;
    const double delta = 2.0 / (beta - alpha), theta = 0.5 * (beta + alpha), s1 = theta * delta;
    double *w = L->w2, *v = L->w1;
    double rhok = 1.0 / s1;
    for (int64_t i = 0; i < n; ++i) {
      w[i] = (r[i] * L->invdiag[i]) / theta;
      y[i] = w[i];
    }
    for (int deg = 1; deg < mg->cheb_degree; ++deg) {
      spmv(&L->A, y, v, 0);
      const double rhokp1 = 1.0 / (2.0 * s1 - rhok);
      const double d1 = rhokp1 * rhok, d2 = 2.0 * rhokp1 * delta;
      rhok = rhokp1;
      for (int64_t i = 0; i < n; ++i) {
        w[i] = d1 * w[i] + d2 * ((r[i] - v[i]) * L->invdiag[i]);
        y[i] += w[i];
      }
    }
  }
}

/* MGSmootherPrecondition::apply (first step from zero) / ::smooth */
static void smooth(oracle_mg *mg, int l, double *u, const double *rhs, int from_zero) {
  level_t *L = &mg->lv[l];
  const int64_t n = L->A.n_rows;
  double *res = L->t; /* t is free during smoothing */
  double *d = (double *)malloc(sizeof(double) * (size_t)n);
  int first = 0;
  if (from_zero && mg->steps > 0) {
    smoother_apply_inverse(mg, l, u, rhs);
    first = 1;
  }
  for (int s = first; s < mg->steps; ++s) {
    spmv(&L->A, u, res, 0);
    for (int64_t i = 0; i < n; ++i) res[i] = rhs[i] - res[i]; /* r.sadd(-1, 1, rhs) */
    smoother_apply_inverse(mg, l, d, res);
    for (int64_t i = 0; i < n; ++i) u[i] += d[i];
  }
  free(d);
}

/* ------------------------------------------------------------------ CG (A2, A9) */

typedef void (*apply_fn)(void *ctx, double *dst, const double *src);

typedef struct {
  int iters;
  double res0, res;
  int status;
} cg_result;

static void cg_solve(const csr_t *A, double *x, const double *b, double tol, int maxit, apply_fn precond, void *pctx,
                     cg_result *out) {
  const int64_t n = A->n_rows;
  double *g = (double *)malloc(sizeof(double) * (size_t)n);
  double *d = (double *)malloc(sizeof(double) * (size_t)n);
  double *h = (double *)malloc(sizeof(double) * (size_t)n);
  int it = 0;
  double res, gh, alpha, beta;
  if (!v_all_zero(x, n)) {
    spmv(A, x, g, 0);
    v_axpy(g, -1.0, b, n);
  } else {
    v_equ(g, -1.0, b, n);
  }
  res = sqrt(v_dot(g, g, n));
  out->res0 = res;
  out->status = ORACLE_OK;
  if (res <= tol) { /* SolverControl::check at step 0 */
    out->iters = 0; out->res = res;
    free(g); free(d); free(h);
    return;
  }
  if (precond) {
    precond(pctx, h, g);
    v_equ(d, -1.0, h, n);
    gh = v_dot(g, h, n);
  } else {
    v_equ(d, -1.0, g, n);
    gh = res * res;
  }
  for (;;) {
    ++it;
    spmv(A, d, h, 0);
    alpha = v_dot(d, h, n);
    alpha = gh / alpha;
    v_axpy(x, alpha, d, n);
    v_axpy(g, alpha, h, n);
    res = sqrt(v_dot(g, g, n));
    if (res <= tol) break;
    if (it >= maxit || res != res) { out->status = 1; break; }
    if (precond) {
      precond(pctx, h, g);
      beta = gh;
      gh = v_dot(g, h, n);
      beta = gh / beta;
      v_sadd(d, beta, -1.0, h, n);
    } else {
      beta = gh;
      gh = res * res;
      beta = gh / beta;
      v_sadd(d, beta, -1.0, g, n);
    }
  }
  out->iters = it; out->res = res;
  free(g); free(d); free(h);
}

/* ------------------------------------------------------------------ V-cycle (A5, A6, A8) */

static void level_v_step(oracle_mg *mg, int l) {
  level_t *L = &mg->lv[l];
  const int64_t n = L->A.n_rows;
  if (l == 0) {
    cg_result r;
    v_zero(L->sol, n);
    cg_solve(&L->A, L->sol, L->def, mg->coarse_tol, mg->coarse_maxit, NULL, NULL, &r);
    mg->coarse_iters_total += r.iters;
    mg->last_coarse_iters = r.iters;
    if (r.status != ORACLE_OK) mg->error = ORACLE_ERR_COARSE_NOCONV;
    return;
  }
  level_t *C = &mg->lv[l - 1];
  smooth(mg, l, L->sol, L->def, 1);           /* pre_smooth->apply */
  spmv(&L->A, L->sol, L->t, 0);               /* t = A u */
  if (L->has_I) spmv(&L->I, L->sol, L->t, 1); /* edge_out->vmult_add */
  for (int64_t i = 0; i < n; ++i) L->t[i] = L->def[i] - L->t[i]; /* t.sadd(-1,1,defect) */
  spmv_t(&C->P, L->t, C->def, 1);             /* restrict_and_add */
  level_v_step(mg, l - 1);
  spmv(&C->P, C->sol, L->t, 0);               /* prolongate */
  for (int64_t i = 0; i < n; ++i) L->sol[i] += L->t[i];
  if (L->has_I) {                             /* edge_in->Tvmult ; defect -= t */
    spmv_t(&L->I, L->sol, L->t, 0);
    for (int64_t i = 0; i < n; ++i) L->def[i] -= L->t[i];
  }
  smooth(mg, l, L->sol, L->def, 0);           /* post_smooth->smooth */
}

int oracle_vcycle(oracle_mg *mg, double *dst, const double *src) {
  mg->error = ORACLE_OK;
  for (int l = 0; l < mg->n_levels; ++l) { /* copy_to_mg */
    level_t *L = &mg->lv[l];
    v_zero(L->def, L->A.n_rows);
    v_zero(L->sol, L->A.n_rows);
    for (int64_t k = 0; k < L->n_copy; ++k) L->def[L->copy_level[k]] = src[L->copy_global[k]];
  }
  level_v_step(mg, mg->n_levels - 1);
  const int64_t n = mg->has_S ? mg->S.n_rows : 0;
  if (n) v_zero(dst, n); /* copy_from_mg: dst = 0 first, coarse to fine */
  for (int l = 0; l < mg->n_levels; ++l) {
    level_t *L = &mg->lv[l];
    for (int64_t k = 0; k < L->n_copy; ++k) dst[L->copy_global[k]] = L->sol[L->copy_level[k]];
  }
  return mg->error;
}

static void precond_gmg(void *ctx, double *dst, const double *src) { oracle_vcycle((oracle_mg *)ctx, dst, src); }

static void precond_jacobi(void *ctx, double *dst, const double *src) { /* :996-1004, omega = 0.6 */
  oracle_mg *mg = (oracle_mg *)ctx;
  for (int64_t i = 0; i < mg->S.n_rows; ++i) dst[i] = (0.6 * src[i]) * mg->S_invdiag[i];
}

/* The outer solve of LaplaceProblem::solve(): tol = rel_tol * |b|_2 (:942), max 500. */
int oracle_solve(oracle_mg *mg, double *x, const double *b, double rel_tol, int max_it, int precond_kind, int *iters,
                 double *res0, double *res, int64_t *coarse_iters) {
  if (!mg->has_S) return ORACLE_ERR_ARG;
  const int64_t n = mg->S.n_rows;
  const double tol = rel_tol * sqrt(v_dot(b, b, n));
  cg_result r;
  mg->coarse_iters_total = 0;
  mg->error = ORACLE_OK;
  apply_fn p = precond_kind == ORACLE_PRECOND_GMG ? precond_gmg : precond_kind == ORACLE_PRECOND_JACOBI ? precond_jacobi : NULL;
  cg_solve(&mg->S, x, b, tol, max_it, p, mg, &r);
  if (iters) *iters = r.iters;
  if (res0) *res0 = r.res0;
  if (res) *res = r.res;
  if (coarse_iters) *coarse_iters = mg->coarse_iters_total;
  if (mg->error) return mg->error;
  return r.status == ORACLE_OK ? ORACLE_OK : ORACLE_ERR_OUTER_NOCONV;
}

/* Stand-alone coarse solve (level 0), for the coarse-CG parity tests and the CPU baseline. */
int oracle_coarse_solve(oracle_mg *mg, double *x, const double *b, int *iters, double *res) {
  level_t *L = &mg->lv[0];
  cg_result r;
  v_zero(x, L->A.n_rows);
  cg_solve(&L->A, x, b, mg->coarse_tol, mg->coarse_maxit, NULL, NULL, &r);
  if (iters) *iters = r.iters;
  if (res) *res = r.res;
  return r.status == ORACLE_OK ? ORACLE_OK : ORACLE_ERR_COARSE_NOCONV;
}

/* one smoother call on a level, for unit parity tests: from_zero=1 -> apply, 0 -> smooth */
int oracle_smooth(oracle_mg *mg, int level, double *u, const double *rhs, int from_zero) {
  if (level < 0 || level >= mg->n_levels) return ORACLE_ERR_ARG;
  smooth(mg, level, u, rhs, from_zero);
  return ORACLE_OK;
}

double oracle_cheb_lmax(oracle_mg *mg, int level) { return mg->lv[level].cheb_lmax; }
