// laplace_problem.h -- host-side C++ mirror of the reference's Step50::LaplaceProblem<dim>
// (/root/reference/include/step_50.h:111-202) for the GMG-CG hot path on MI355X.
//
// Same member names and meaning as the reference class; what the reference delegates to
// deal.II / Trilinos / p4est is done here by forest.h (mesh), plain CSR containers (matrices)
// and -- for everything on the hot path -- by the C-ABI of include/gmg_coulomb.h.  The outer
// CG loop of solve() stays in this C++ (SolverCG below) and calls through that ABI.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>

#include "flat_map.h"
#include <vector>

#include "forest.h"
#include "gmg_coulomb.h"

namespace step50 {

struct CSRMatrix {
  int64_t n_rows = 0, n_cols = 0;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> col;
  std::vector<double> val;
  int64_t nnz() const { return rowptr.empty() ? 0 : rowptr.back(); }
  void add(int32_t r, int32_t c, double v);  // entry must be in the pattern
  double l1_norm() const, linfty_norm() const, frobenius_norm() const;  // src/step-50.cc:950-952
};

// flat view of a deal.II .prm file (ParameterHandler keys of src/step-50.cc:13-95)
class ParameterReader {
 public:
  void declare_parameters();                      // defaults, src/step-50.cc:13-95
  void read_parameters(const std::string &file);  // src/step-50.cc:98-101
  void parse_input_from_string(const std::string &text);
  std::string get(const std::string &key) const;
  double get_double(const std::string &key) const;
  long get_integer(const std::string &key) const;
  bool get_bool(const std::string &key) const;
  void set(const std::string &key, const std::string &value) { values[key] = value; }

 private:
  std::map<std::string, std::string> values;
};

// the reference's 20 constructor arguments (include/step_50.h:115-118) + this build's additions
struct Parameters {
  unsigned int degree = 1;
  std::string Problemtype = "Step16", PreconditionerType = "GMG", LammpsInputFile = "atom_8.data",
              Boundary_conditions = "Inhomogeneous";
  double domain_size_left = -1, domain_size_right = 1, mesh_size_h = 0.25;
  unsigned int repetitions_for_vacuum = 1, number_of_global_refinement = 2, number_of_adaptive_refinement_cycles = 2;
  double r_c = 0.5, nonzero_density_radius_parameter = 3;
  bool flag_rhs_assembly = false, flag_analytical_solution = false, flag_rhs_field = false, flag_atoms_support = false,
       flag_output_time = true;
  unsigned int quadrature_degree_rhs = 1;
  int dim = 2;
  // additions (the smoother is a source edit in the reference, src/step-50.cc:969-970)
  std::string smoother = "SSOR";  // Jacobi | SSOR | Chebyshev
  double smoother_omega = 0.5;
  int smoother_steps = 2, chebyshev_degree = 2;
  bool densities_on_device = true;  // compute_charge_densities() through gmg_charge_density when a device is in use
  int ssor_blocks = 1;  // 1: exact sequential SGS (mpirun=1); B: rank-local SGS on B blocks (mpirun=B)
  bool device_resident_outer_cg = false;  // true: gmg_cg_solve instead of the host SolverCG
  std::string partition_level0 = "auto";  // one process per GPU: auto | always | never
  std::string refinement_estimator = "Kelly + residual";  // HEAD (:1040-1089) | "Kelly": the indicator of the older cluster runs
  double short_range_cutoff = 0.0;        // in smoothing lengths; 0: all pairs (the reference)
  bool energy_for_large_systems = false;  // evaluate the energy also for >= 300 atoms (needs the cutoff; the reference skips it, :1554)
  bool rhs_on_device = true;            // gmg_rhs_assemble: F integrated on the device from densities that stay there
  bool transfer_on_device = true;       // gmg_build_transfer instead of building P_l here and uploading it
  bool level0_matrix_on_device = true;  // gmg_set_level_matrix_lattice instead of assembling + uploading level 0 (3D, constant coefficient, lexicographic, unpartitioned)
  std::string level0_numbering = "lexicographic";  // lexicographic | cell-wise (deal.II's first-touch order): level 0 carries no smoother
  static Parameters from(const ParameterReader &prm);
};

struct CycleReport {  // the values the reference prints per cycle (src/step-50.cc:946-952, 1009-1014, ...)
  int cycle = 0;
  int64_t active_cells = 0, dofs = 0;
  std::vector<int64_t> dofs_by_level;
  double rhs_l1 = 0, rhs_l2 = 0, rhs_linf = 0, matrix_l1 = 0, matrix_linf = 0, matrix_frobenius = 0;
  double starting_value = 0, convergence_value = 0, sol_l1 = 0, sol_l2 = 0, sol_linf = 0;
  int cg_iterations = 0;
  int64_t coarse_iterations = 0;
  double refine_threshold = 0;
  bool has_energy = false;
  double energy_analytical = 0, energy_short = 0, energy_fe_long = 0, energy_self = 0, energy_total = 0, energy_abs_error = 0;
  double energy_norm_error = 0;
  double solve_seconds = 0;  // first residual to convergence, excluding upload / build_matrices
  double build_matrices_ms = 0;  // device time of mg_transfer.build_matrices (:957-958) when the device builds the transfers
  int status = 0;
};

template <int dim>
class LaplaceProblem {
 public:
  explicit LaplaceProblem(const Parameters &p);
  ~LaplaceProblem();
  void run();  // src/step-50.cc:1463-1573

  // ---- protected in the reference (tests subclass it); public here for the C binding
  void read_lammps_input_file(const std::string &filename);              // :181-258
  void set_atoms(const std::vector<double> &q, const std::vector<double> &xyz);  // synthetic input of the same shape
  void make_initial_grid();                                              // :1490-1528
  void setup_system(unsigned int cycle);                                 // :646-732
  void rhs_assembly_optimization();                                      // :260-306
  void compute_charge_densities();                                       // :509-575
  void ensure_host_densities();                                          // copy device-resident densities out when the host needs them
  void compute_moments();                                                // :577-644
  void assemble_system();                                                // :735-833
  void assemble_multigrid();                                             // :835-933
  void assemble_level(int l);                                            // one level's matrix + interface matrix (:869-931)
  void ensure_level_matrix(int l);                                       // assemble a level that was left to the device, on demand
  bool decide_level0_on_device() const;                                  // level 0 formed on the device (gmg_set_level_matrix_lattice)?
  void level0_cell_matrix(double *Ke) const;
  void build_transfer();                                                 // mg_transfer.build_matrices, :957-958
  void build_prolongation(int l);                                        // P_l on the host (the device builds it otherwise)
  void ensure_prolongation(int l);
  int ensure_context();                                                  // gmg_create (+ communicator) on first use
  int upload();                                                          // hand the operators over the C-ABI
  int solve();                                                           // :938-1017
  void estimate_error_and_mark_cells();                                  // :1020-1090
  void refine_grid(unsigned int cycle);                                  // :1095-1121
  void postprocess_electrostatic_energy();                               // :1310-1420
  void postprocess_error_in_energy_norm();                               // :1423-1461
  std::vector<double> total_charge_density_vector() const;               // tests_rhs_rc_variation/rc_variation.cc:110-215
  int run_cycle(unsigned int cycle, bool on_device = true);              // one iteration of the loop in run()
  void finish_cycle();                                   // estimator + energy, the tail of the loop body
  void set_solution(const std::vector<double> &x);       // test hook, see laplace_problem.cc
  int solve_again();  // repeat the solve of the current cycle from the same initial guess (bench step)
  int set_smoother(const std::string &smoother, int ssor_blocks);  // other smoother on the current cycle's operators (re-uploads them)

  // ---- data, named as in the reference where it exists (include/step_50.h:146-200)
  Parameters par;
  Forest<dim> triangulation;
  std::vector<double> charges, atom_positions;  // positions: 3 * n
  unsigned int number_of_atoms = 0;
  bool lammpsinput = false;
  CSRMatrix system_matrix;
  std::vector<double> solution, system_rhs, initial_guess;
  std::vector<CSRMatrix> mg_matrices, mg_interface_matrices, mg_prolongation;  // P_l: level l -> l+1
  std::vector<std::vector<int32_t>> copy_global, copy_level;
  double dipole_moment[3] = {0, 0, 0};
  std::vector<CycleReport> reports;
  std::string log;  // everything pcout would have printed
  bool echo = false;
  gmg_context *gmg = nullptr;
  bool operators_uploaded = false, densities_on_device = false;
  bool solve_on_device_requested = false, level0_on_device = false, transfer_on_device = false;
  bool densities_device_resident = false;  // compute_charge_densities left them in HBM for gmg_rhs_assemble
  double build_matrices_ms = 0.0;  // device time of gmg_build_transfer for the current cycle's operators
  std::string last_error;
  // one process per GPU (the reference: one MPI rank per subdomain, src/main.cc:8); the host
  // setup is replicated, the operators are cut by partition.h at upload()
  int rank = 0, n_ranks = 1;
  bool distributed = false;
  bool level0_partitioned = false;  // decided per cycle in upload() ("Partition level 0")
  // rows a rank must get rid of for a partitioned level 0 to pay (24 ps per row and coarse iteration on one MI355X):
  // over RCCL an iteration gains three collectives + their launches (~60 us), over the peer transport three
  // one-workgroup kernels and three flag latencies (~17 us)
  static constexpr int64_t kPartitionMinRowsSaved = 2500000, kPartitionMinRowsSavedPeer = 700000;
  std::string comm_id;  // gmg_comm_unique_id of rank 0, broadcast by the launcher
  void set_communicator(int rank_, int n_ranks_, const std::string &id) { rank = rank_; n_ranks = n_ranks_; comm_id = id; distributed = true; }

  // DoF bookkeeping (Q1: DoFs = vertices)
  struct ActiveCell { int32_t level, index; };
  std::vector<ActiveCell> active_cells;
  std::vector<std::vector<int32_t>> active_index_of_cell;  // [level][cell] -> position in active_cells or -1
  VertexMap dof_of_vertex;
  std::vector<uint64_t> vertex_of_dof;
  std::vector<VertexMap> level_dof_of_vertex;
  std::vector<std::vector<uint64_t>> level_vertex_of_dof;
  std::vector<int32_t> active_cell_dof_table;                 // [active cell][vertex]: the DoFs of every active cell (cell_dofs)
  std::vector<std::vector<int32_t>> level_cell_dof_table;     // [level][cell][vertex] (level_cell_dofs)
  // constraints (hanging nodes + Dirichlet), resolved: masters are unconstrained DoFs
  struct ConstraintLine { std::vector<std::pair<int32_t, double>> entries; double inhomogeneity = 0; bool hanging = false; };
  std::vector<int32_t> constraint_of_dof;  // -1 = unconstrained
  std::vector<ConstraintLine> constraint_lines;
  std::vector<std::vector<char>> level_boundary, level_refinement_edge;  // MGConstrainedDoFs
  std::vector<std::vector<double>> density_values_for_each_cell;         // [active cell][q]
  std::vector<float> error_per_cell;
  std::vector<double> estimator_kelly_sq, estimator_residual_sq;  // per active cell: face-jump sum / h_K^2 int (4 pi rho)^2 (kept for the marking-rule study)
  std::vector<std::vector<char>> refine_flags;

  void pcout(const std::string &s);
  void distribute_dofs();
  void make_constraints();
  void cell_dofs(const ActiveCell &c, int32_t *out) const;
  void level_cell_dofs(int level, int32_t cell, int32_t *out) const;
  double boundary_value(const double x[3]) const;
  double rhs_function(const double x[3]) const;
  double coefficient(const double x[3]) const;
  void atoms_of_root_cell(const int rc[3], std::vector<int32_t> &out) const;
  double fe_value_at(const std::vector<double> &u, const double x[3]) const;
  void distribute_constraints(std::vector<double> &u) const;  // constraints.distribute, :1016
  void set_zero_constraints(std::vector<double> &u) const;    // constraints.set_zero, :1119

 private:
  struct AtomBins;
  std::unique_ptr<AtomBins> bins;
  double *d_solution = nullptr, *d_rhs = nullptr, *d_full = nullptr;
  int64_t d_n = 0, d_nvec = 0, d_begin = 0;
  int solve_on_device(CycleReport &rep);
};

// deal.II SolverCG<vector_t> restated on top of the C-ABI: every vector is a device pointer,
// every operation one ABI call (reference: src/step-50.cc:943, 991-992; operation order as in
// oracle/gmg_oracle.c:cg_solve).
struct SolverControl {
  int max_steps;
  double tolerance;
  double initial_value = 0, last_value = 0;
  int last_step = 0;
};
int SolverCG_solve(gmg_context *ctx, SolverControl &control, int64_t n, int64_t n_vec, double *x, const double *b,
                   const std::string &preconditioner);

std::vector<double> nacl_lattice(int n_cells, std::vector<double> &charges);  // the reference's atom/*.data generator

}  // namespace step50
