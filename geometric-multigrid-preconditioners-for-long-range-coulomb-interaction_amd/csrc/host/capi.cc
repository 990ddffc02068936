// capi.cc -- flat C entry points over LaplaceProblem<dim> for tests/ and bench.py (ctypes).
// Not part of the drop-in boundary (that is include/gmg_coulomb.h); this is how Python drives
// the host-side C++ the same way src/main.cc of the reference drives LaplaceProblem.
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "laplace_problem.h"
#include "partition.h"
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace step50;

struct step50_problem {
  int dim = 3;
  std::unique_ptr<LaplaceProblem<2>> p2;
  std::unique_ptr<LaplaceProblem<3>> p3;
  std::string err;
};

#define DISPATCH(h, expr) ((h)->dim == 2 ? (h)->p2->expr : (h)->p3->expr)

namespace {
template <class F>
int guarded(step50_problem *h, F f) {
  try {
    return f();
  } catch (const std::exception &e) {
    h->err = e.what();
    return -1;
  }
}
const CSRMatrix *pick_matrix(step50_problem *h, int kind, int level) {
  // kind 0 system, 1 level, 2 edge, 3 prolongation
  if (kind == 0) return &DISPATCH(h, system_matrix);
  if (kind == 1) DISPATCH(h, ensure_level_matrix(level));  // (level 0 may have been left to the device: assemble it now)
  if (kind == 3) DISPATCH(h, ensure_prolongation(level));   // (the transfers likewise)
  auto &v = kind == 1 ? DISPATCH(h, mg_matrices) : kind == 2 ? DISPATCH(h, mg_interface_matrices) : DISPATCH(h, mg_prolongation);
  if (level < 0 || level >= (int)v.size()) return nullptr;
  return &v[(size_t)level];
}
}  // namespace

extern "C" {

step50_problem *step50_create(const char *prm_text, char *err, int err_len) {
  auto *h = new step50_problem();
  try {
    ParameterReader prm;
    prm.declare_parameters();
    prm.parse_input_from_string(prm_text);
    Parameters par = Parameters::from(prm);
    h->dim = par.dim;
    if (par.dim == 2) h->p2.reset(new LaplaceProblem<2>(par));
    else if (par.dim == 3) h->p3.reset(new LaplaceProblem<3>(par));
    else throw std::runtime_error("Only 2d and 3d dimensions are supported.");  // src/main.cc:92-95
    return h;
  } catch (const std::exception &e) {
    if (err && err_len > 0) { std::strncpy(err, e.what(), (size_t)err_len - 1); err[err_len - 1] = 0; }
    delete h;
    return nullptr;
  }
}
void step50_destroy(step50_problem *h) { delete h; }
const char *step50_last_error(step50_problem *h) {
  const std::string &le = DISPATCH(h, last_error);
  return h->err.empty() ? le.c_str() : h->err.c_str();
}
const char *step50_log(step50_problem *h) { return DISPATCH(h, log).c_str(); }
void step50_set_echo(step50_problem *h, int on) { DISPATCH(h, echo) = on != 0; }

int step50_read_lammps(step50_problem *h, const char *path) {
  return guarded(h, [&] { DISPATCH(h, read_lammps_input_file(path)); return DISPATCH(h, lammpsinput) ? 0 : 1; });
}
int step50_set_atoms(step50_problem *h, int64_t n, const double *q, const double *xyz) {
  return guarded(h, [&] {
    std::vector<double> qq(q, q + n), xx(xyz, xyz + 3 * n);
    DISPATCH(h, set_atoms(qq, xx));
    return 0;
  });
}
int step50_set_nacl_atoms(step50_problem *h, int n_cells) {
  return guarded(h, [&] {
    std::vector<double> q;
    std::vector<double> x = nacl_lattice(n_cells, q);
    DISPATCH(h, set_atoms(q, x));
    return 0;
  });
}
int64_t step50_n_atoms(step50_problem *h) { return (int64_t)DISPATCH(h, number_of_atoms); }
int step50_get_atoms(step50_problem *h, double *q, double *xyz) {
  const auto &qq = DISPATCH(h, charges);
  const auto &xx = DISPATCH(h, atom_positions);
  std::memcpy(q, qq.data(), sizeof(double) * qq.size());
  std::memcpy(xyz, xx.data(), sizeof(double) * xx.size());
  return 0;
}

// one pass of the loop body of LaplaceProblem::run (src/step-50.cc:1484-1561)
int step50_run_cycle(step50_problem *h, int cycle, int on_device) {
  return guarded(h, [&] { return DISPATCH(h, run_cycle((unsigned)cycle, on_device != 0)); });
}
// the timed bench step: repeat solve() of the current cycle from the same initial guess
int step50_solve_again(step50_problem *h) {
  return guarded(h, [&] { return DISPATCH(h, solve_again()); });
}
// marking-rule study (tools/marking_rule_scan.py): per active cell of the cycle just estimated, the Kelly face sum
// eta_K^2, the residual term h_K^2 int_K (4 pi rho)^2, the cell's level, and its centre
int64_t step50_n_active_cells(step50_problem *h) { return (int64_t)DISPATCH(h, estimator_kelly_sq).size(); }
int step50_estimator_components(step50_problem *h, double *kelly_sq, double *residual_sq, int32_t *level, double *centre) {
  return guarded(h, [&] {
    const auto &k = DISPATCH(h, estimator_kelly_sq);
    const auto &r = DISPATCH(h, estimator_residual_sq);
    std::memcpy(kelly_sq, k.data(), sizeof(double) * k.size());
    std::memcpy(residual_sq, r.data(), sizeof(double) * r.size());
    auto fill = [&](auto &P) {
      for (size_t a = 0; a < P.active_cells.size(); ++a) {
        const auto &ac = P.active_cells[a];
        level[a] = ac.level;
        double x0[3] = {0, 0, 0};
        const double hh = P.triangulation.cell_size(ac.level);
        P.triangulation.cell_origin(ac.level, P.triangulation.levels[(size_t)ac.level][(size_t)ac.index], x0);
        for (int d = 0; d < 3; ++d) centre[3 * a + d] = x0[d] + 0.5 * hh;
      }
    };
    if (h->dim == 2) fill(*h->p2); else fill(*h->p3);
    return 0;
  });
}
// bench: re-upload the current cycle's operators with another smoother
int step50_set_smoother(step50_problem *h, const char *smoother, int ssor_blocks) {
  return guarded(h, [&] { return DISPATCH(h, set_smoother(std::string(smoother), ssor_blocks)); });
}
// CPU-only continuation of a cycle for tests: inject a solution, then estimator + energy
int step50_finish_cycle_with(step50_problem *h, const double *x, int64_t n) {
  return guarded(h, [&] {
    std::vector<double> v(x, x + n);
    DISPATCH(h, set_solution(v));
    DISPATCH(h, finish_cycle());
    return 0;
  });
}
int step50_n_reports(step50_problem *h) { return (int)DISPATCH(h, reports).size(); }

struct step50_report {
  int32_t cycle, cg_iterations, status, has_energy, n_levels, pad;
  int64_t active_cells, dofs, coarse_iterations;
  int64_t dofs_by_level[16];
  double rhs_l1, rhs_l2, rhs_linf, matrix_l1, matrix_linf, matrix_frobenius;
  double starting_value, convergence_value, sol_l1, sol_l2, sol_linf, refine_threshold;
  double energy_analytical, energy_short, energy_fe_long, energy_self, energy_total, energy_abs_error;
  double solve_seconds, energy_norm_error;
  double build_matrices_ms;  // device time of MGTransferPrebuilt::build_matrices (gmg_build_transfer) for this cycle; 0: built on the host
};
int step50_get_report(step50_problem *h, int i, step50_report *out) {
  const auto &reps = DISPATCH(h, reports);
  if (i < 0) i += (int)reps.size();
  if (i < 0 || i >= (int)reps.size()) return 1;
  const CycleReport &r = reps[(size_t)i];
  std::memset(out, 0, sizeof *out);
  out->cycle = r.cycle; out->cg_iterations = r.cg_iterations; out->status = r.status; out->has_energy = r.has_energy;
  out->n_levels = (int32_t)r.dofs_by_level.size();
  out->active_cells = r.active_cells; out->dofs = r.dofs; out->coarse_iterations = r.coarse_iterations;
  for (size_t l = 0; l < r.dofs_by_level.size() && l < 16; ++l) out->dofs_by_level[l] = r.dofs_by_level[l];
  out->rhs_l1 = r.rhs_l1; out->rhs_l2 = r.rhs_l2; out->rhs_linf = r.rhs_linf;
  out->matrix_l1 = r.matrix_l1; out->matrix_linf = r.matrix_linf; out->matrix_frobenius = r.matrix_frobenius;
  out->starting_value = r.starting_value; out->convergence_value = r.convergence_value;
  out->sol_l1 = r.sol_l1; out->sol_l2 = r.sol_l2; out->sol_linf = r.sol_linf; out->refine_threshold = r.refine_threshold;
  out->energy_analytical = r.energy_analytical; out->energy_short = r.energy_short; out->energy_fe_long = r.energy_fe_long;
  out->energy_self = r.energy_self; out->energy_total = r.energy_total; out->energy_abs_error = r.energy_abs_error;
  out->solve_seconds = r.solve_seconds;
  out->energy_norm_error = r.energy_norm_error;
  out->build_matrices_ms = r.build_matrices_ms;
  return 0;
}

// ---- access to what solve() consumes, so tests can hand the same inputs to the oracle
int step50_n_levels(step50_problem *h) { return h->dim == 2 ? h->p2->triangulation.n_levels() : h->p3->triangulation.n_levels(); }
int step50_matrix_shape(step50_problem *h, int kind, int level, int64_t *n_rows, int64_t *n_cols, int64_t *nnz) {
  const CSRMatrix *m = pick_matrix(h, kind, level);
  if (!m) return 1;
  *n_rows = m->n_rows; *n_cols = m->n_cols; *nnz = m->nnz();
  return 0;
}
int step50_matrix_copy(step50_problem *h, int kind, int level, int64_t *rowptr, int32_t *col, double *val) {
  const CSRMatrix *m = pick_matrix(h, kind, level);
  if (!m) return 1;
  std::memcpy(rowptr, m->rowptr.data(), sizeof(int64_t) * m->rowptr.size());
  std::memcpy(col, m->col.data(), sizeof(int32_t) * m->col.size());
  std::memcpy(val, m->val.data(), sizeof(double) * m->val.size());
  return 0;
}
int64_t step50_copy_indices_size(step50_problem *h, int level) { return (int64_t)DISPATCH(h, copy_global)[(size_t)level].size(); }
int step50_copy_indices(step50_problem *h, int level, int32_t *global_idx, int32_t *level_idx) {
  const auto &g = DISPATCH(h, copy_global)[(size_t)level];
  const auto &l = DISPATCH(h, copy_level)[(size_t)level];
  std::memcpy(global_idx, g.data(), sizeof(int32_t) * g.size());
  std::memcpy(level_idx, l.data(), sizeof(int32_t) * l.size());
  return 0;
}
int64_t step50_n_dofs(step50_problem *h) { return (int64_t)DISPATCH(h, vertex_of_dof).size(); }
int step50_get_vector(step50_problem *h, int which, double *out) {
  // 0 system_rhs, 1 solution (after constraints.distribute), 2 initial guess
  const auto &v = which == 0 ? DISPATCH(h, system_rhs) : which == 1 ? DISPATCH(h, solution) : DISPATCH(h, initial_guess);
  std::memcpy(out, v.data(), sizeof(double) * v.size());
  return 0;
}
// check vector of the reference's rhs test: per DoF, the integrated charge density of its cells
int step50_total_charge_density(step50_problem *h, double *out) {
  const std::vector<double> v = DISPATCH(h, total_charge_density_vector());
  std::memcpy(out, v.data(), sizeof(double) * v.size());
  return 0;
}
int step50_dof_coordinates(step50_problem *h, double *xyz) {
  const auto &keys = DISPATCH(h, vertex_of_dof);
  for (size_t i = 0; i < keys.size(); ++i) {
    if (h->dim == 2) h->p2->triangulation.vertex_coords(keys[i], xyz + 3 * i);
    else h->p3->triangulation.vertex_coords(keys[i], xyz + 3 * i);
  }
  return 0;
}
int step50_constrained_mask(step50_problem *h, int8_t *out) {
  const auto &c = DISPATCH(h, constraint_of_dof);
  for (size_t i = 0; i < c.size(); ++i) out[i] = c[i] >= 0;
  return 0;
}
void *step50_gmg_context(step50_problem *h) { return DISPATCH(h, gmg); }

// host threads of the replicated setup (one process per GPU: cores / world size)
void step50_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

// one process per GPU: rank, world size and the 128-byte id of gmg_comm_unique_id (rank 0)
int step50_set_communicator(step50_problem *h, int rank, int n_ranks, const void *id128) {
  return guarded(h, [&] {
    DISPATCH(h, set_communicator(rank, n_ranks, std::string((const char *)id128, 128)));
    return 0;
  });
}

// partition.h on one operator, for the CPU tests of the distributed layout: kind/level as in
// step50_matrix_shape.  Two calls: sizes first, then the arrays.
struct step50_local_info { int64_t n_rows, n_cols, nnz, row_begin, n_neighbors, n_send; };
static LocalOperator g_last_local;
int step50_localize(step50_problem *h, int kind, int level, int rank, int n_ranks, step50_local_info *out) {
  return guarded(h, [&] {
    const CSRMatrix *m = pick_matrix(h, kind, level);
    if (!m) return 1;
    g_last_local = localize(*m, rank, n_ranks);
    out->n_rows = g_last_local.A.n_rows; out->n_cols = g_last_local.A.n_cols; out->nnz = g_last_local.A.nnz();
    out->row_begin = g_last_local.row_begin; out->n_neighbors = (int64_t)g_last_local.halo.neighbor_rank.size();
    out->n_send = (int64_t)g_last_local.halo.send_idx.size();
    return 0;
  });
}
int step50_localize_copy(int64_t *rowptr, int32_t *col, double *val, int32_t *neighbor_rank, int32_t *send_count,
                         int32_t *send_idx, int32_t *recv_count, int64_t *ghost_global) {
  const LocalOperator &L = g_last_local;
  std::memcpy(rowptr, L.A.rowptr.data(), sizeof(int64_t) * L.A.rowptr.size());
  std::memcpy(col, L.A.col.data(), sizeof(int32_t) * L.A.col.size());
  std::memcpy(val, L.A.val.data(), sizeof(double) * L.A.val.size());
  std::memcpy(neighbor_rank, L.halo.neighbor_rank.data(), sizeof(int32_t) * L.halo.neighbor_rank.size());
  std::memcpy(send_count, L.halo.send_count.data(), sizeof(int32_t) * L.halo.send_count.size());
  std::memcpy(send_idx, L.halo.send_idx.data(), sizeof(int32_t) * L.halo.send_idx.size());
  std::memcpy(recv_count, L.halo.recv_count.data(), sizeof(int32_t) * L.halo.recv_count.size());
  std::memcpy(ghost_global, L.ghost_global.data(), sizeof(int64_t) * L.ghost_global.size());
  return 0;
}

}  // extern "C"
