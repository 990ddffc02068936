// laplace_problem.cc -- see laplace_problem.h.  Reference line numbers are those of
// /root/reference/src/step-50.cc unless another file is named.
#include "laplace_problem.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include "partition.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace step50 {

// STEP50_TIMING=2: wall time of the pieces of a host phase on stderr (since the previous call; nullptr restarts the clock)
void sublap(const char *what) {
  static const bool on = std::getenv("STEP50_TIMING") != nullptr && std::atoi(std::getenv("STEP50_TIMING")) >= 2;
  static auto t_last = std::chrono::steady_clock::now();
  if (!on) return;
  const auto now = std::chrono::steady_clock::now();
  if (what) std::fprintf(stderr, "[step50]     . %-32s %8.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
  t_last = now;
}

// ======================================================================== CSR helpers

void CSRMatrix::add(int32_t r, int32_t c, double v) {
  const int32_t *b = col.data() + rowptr[(size_t)r], *e = col.data() + rowptr[(size_t)r + 1];
  const int32_t *p = std::lower_bound(b, e, c);
  if (p == e || *p != c) throw std::logic_error("CSRMatrix::add: entry outside the sparsity pattern");
  val[(size_t)(p - col.data())] += v;
}
double CSRMatrix::l1_norm() const {
  std::vector<double> s((size_t)n_cols, 0.0);
  for (int64_t k = 0; k < nnz(); ++k) s[(size_t)col[(size_t)k]] += std::fabs(val[(size_t)k]);
  double m = 0;
  for (double v : s) m = std::max(m, v);
  return m;
}
double CSRMatrix::linfty_norm() const {
  double m = 0;
  for (int64_t i = 0; i < n_rows; ++i) {
    double s = 0;
    for (int64_t k = rowptr[(size_t)i]; k < rowptr[(size_t)i + 1]; ++k) s += std::fabs(val[(size_t)k]);
    m = std::max(m, s);
  }
  return m;
}
double CSRMatrix::frobenius_norm() const {
  double s = 0;
  for (double v : val) s += v * v;
  return std::sqrt(s);
}

namespace {

// Pattern from per-cell coupling lists: all pairs inside a list are stored entries (value 0).
CSRMatrix pattern_from_cells(int64_t n, const std::vector<int64_t> &cptr, const std::vector<int32_t> &citems) {
  const int64_t n_cells = (int64_t)cptr.size() - 1;
  std::vector<int64_t> dptr((size_t)n + 1, 0);
  for (int32_t d : citems) dptr[(size_t)d + 1]++;
  for (int64_t i = 0; i < n; ++i) dptr[(size_t)i + 1] += dptr[(size_t)i];
  std::vector<int32_t> dcells((size_t)dptr[(size_t)n]);
  {
    std::vector<int64_t> pos(dptr.begin(), dptr.end() - 1);
    for (int64_t c = 0; c < n_cells; ++c)
      for (int64_t k = cptr[(size_t)c]; k < cptr[(size_t)c + 1]; ++k) dcells[(size_t)pos[(size_t)citems[(size_t)k]]++] = (int32_t)c;
  }
  CSRMatrix A;
  A.n_rows = A.n_cols = n;
  A.rowptr.assign((size_t)n + 1, 0);
  // rows in chunks, one thread per chunk: the sorted union of the row's cell lists goes to the chunk's buffer, the buffers
  // are copied to their place once the row pointers are known (each row is sorted once)
  const int64_t chunk = 1 << 13;
  const int64_t n_chunks = (n + chunk - 1) / chunk;
  std::vector<std::vector<int32_t>> ccols((size_t)n_chunks);
#pragma omp parallel
  {
    std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 1)
    for (int64_t ch = 0; ch < n_chunks; ++ch) {
      std::vector<int32_t> &out = ccols[(size_t)ch];
      const int64_t i1 = std::min(n, (ch + 1) * chunk);
      out.reserve((size_t)(i1 - ch * chunk) * 27);
      for (int64_t i = ch * chunk; i < i1; ++i) {
        tmp.clear();
        for (int64_t q = dptr[(size_t)i]; q < dptr[(size_t)i + 1]; ++q) {
          const int32_t c = dcells[(size_t)q];
          tmp.insert(tmp.end(), citems.begin() + cptr[(size_t)c], citems.begin() + cptr[(size_t)c + 1]);
        }
        std::sort(tmp.begin(), tmp.end());
        const auto e = std::unique(tmp.begin(), tmp.end());
        A.rowptr[(size_t)i + 1] = e - tmp.begin();
        out.insert(out.end(), tmp.begin(), e);
      }
    }
  }
  for (int64_t i = 0; i < n; ++i) A.rowptr[(size_t)i + 1] += A.rowptr[(size_t)i];
  A.col.resize((size_t)A.rowptr[(size_t)n]);
  A.val.assign((size_t)A.rowptr[(size_t)n], 0.0);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t ch = 0; ch < n_chunks; ++ch) {
    std::copy(ccols[(size_t)ch].begin(), ccols[(size_t)ch].end(), A.col.begin() + A.rowptr[(size_t)(ch * chunk)]);
    std::vector<int32_t>().swap(ccols[(size_t)ch]);
  }
  return A;
}

CSRMatrix csr_from_triplets(int64_t n_rows, int64_t n_cols, std::vector<std::array<int64_t, 2>> &rc, std::vector<double> &v,
                            bool sum_duplicates) {
  std::vector<size_t> order(rc.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rc[a] < rc[b]; });
  CSRMatrix A;
  A.n_rows = n_rows; A.n_cols = n_cols;
  A.rowptr.assign((size_t)n_rows + 1, 0);
  for (size_t q = 0; q < order.size(); ++q) {
    const size_t i = order[q];
    if (q > 0 && rc[i] == rc[order[q - 1]]) {
      if (sum_duplicates) A.val.back() += v[i];
      continue;  // "set" semantics: first writer wins, later ones are identical
    }
    A.col.push_back((int32_t)rc[i][1]);
    A.val.push_back(v[i]);
    A.rowptr[(size_t)rc[i][0] + 1]++;
  }
  for (int64_t i = 0; i < n_rows; ++i) A.rowptr[(size_t)i + 1] += A.rowptr[(size_t)i];
  return A;
}

// Gauss-Legendre on [0,1]
void gauss01(int n, std::vector<double> &x, std::vector<double> &w) {
  x.resize((size_t)n); w.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1, p2 = 0;
      for (int j = 1; j <= n; ++j) { const double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1) * z * p2 - (j - 1.0) * p3) / j; }
      pp = n * (z * p1 - p2) / (z * z - 1);
      const double dz = p1 / pp;
      z -= dz;
      if (std::fabs(dz) < 1e-16) break;
    }
    x[(size_t)(n - 1 - i)] = 0.5 * (z + 1);
    w[(size_t)(n - 1 - i)] = 1.0 / ((1 - z * z) * pp * pp);
  }
}

template <int dim>
struct Quadrature {  // QGauss<dim>(n), tensor product with x fastest
  std::vector<std::array<double, 3>> p;
  std::vector<double> w;
  std::vector<std::array<double, 1 << dim>> shape;                       // [q][i]
  std::vector<std::array<std::array<double, 3>, 1 << dim>> grad;        // [q][i][d], unit cell
  explicit Quadrature(int n) {
    std::vector<double> x1, w1;
    gauss01(n, x1, w1);
    const int nz = dim == 3 ? n : 1;
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          p.push_back({x1[(size_t)i], x1[(size_t)j], dim == 3 ? x1[(size_t)k] : 0.0});
          w.push_back(w1[(size_t)i] * w1[(size_t)j] * (dim == 3 ? w1[(size_t)k] : 1.0));
        }
    shape.resize(p.size()); grad.resize(p.size());
    for (size_t q = 0; q < p.size(); ++q)
      for (int a = 0; a < (1 << dim); ++a) {
        double f[3], df[3];
        for (int d = 0; d < dim; ++d) {
          const int b = (a >> d) & 1;
          f[d] = b ? p[q][(size_t)d] : 1.0 - p[q][(size_t)d];
          df[d] = b ? 1.0 : -1.0;
        }
        double v = 1;
        for (int d = 0; d < dim; ++d) v *= f[d];
        shape[q][(size_t)a] = v;
        for (int e = 0; e < 3; ++e) {
          double g = 0;
          if (e < dim) { g = 1; for (int d = 0; d < dim; ++d) g *= (d == e) ? df[d] : f[d]; }
          grad[q][(size_t)a][(size_t)e] = g;
        }
      }
  }
};

std::string fmt(const char *f, double v) {
  char buf[64];
  std::snprintf(buf, sizeof buf, f, v);
  return buf;
}

}  // namespace

// ======================================================================== parameters

void ParameterReader::declare_parameters() {
  values = {{"Number of global refinement", "2"}, {"Domain limit left", "-1"}, {"Domain limit right", "1"},
            {"Mesh size", "0.25"}, {"Vacuum repetitions", "1"}, {"Problem", "Step16"}, {"Dimension", "2"},
            {"Boundary conditions selection", "Inhomogeneous"}, {"Number of Adaptive Refinement", "2"},
            {"smoothing length", "0.5"}, {"Nonzero Density radius parameter around each charge", "3"},
            {"Output and calculation of Analytical solution", "false"}, {"Output of RHS field", "false"},
            {"Output of support of each atom", "false"}, {"Flag for RHS evaluation optimization", "false"},
            {"Quadrature points for RHS function", "1"}, {"Output time summary table", "true"},
            {"Polynomial degree", "1"}, {"Preconditioner", "GMG"}, {"Lammps input file", "atom_8.data"},
            // additions of this build (the reference selects the smoother by editing :969-970)
            {"Smoother", "SSOR"}, {"Smoother damping", "0.5"}, {"Smoother steps", "2"}, {"Chebyshev degree", "2"},
            {"Device resident outer CG", "false"}, {"SSOR blocks", "1"}, {"Charge densities on device", "true"},
            {"Partition level 0", "auto"},
            // HEAD marks with Kelly + the cell residual (:1040-1089); the cluster logs (January 2018) are reproduced by
            // the Kelly indicator alone (tools/marking_rule_scan.py)
            {"Refinement estimator", "Kelly + residual"},
            // level 0 is the undivided lattice: numbered lexicographically (x fastest) its operator is a pure 27-point
            // stencil for the device's plane-by-plane kernel; "cell-wise" = deal.II's first-touch order on level 0 as well
            {"Level 0 numbering", "lexicographic"},
            // the level-0 matrix formed on the device from (lattice size, cell matrix) instead of assembled here and uploaded
            // as CSR (gmg_set_level_matrix_lattice; 3D constant-coefficient problems with a lexicographic, unpartitioned level 0)
            {"Level 0 matrix on device", "true"},
            // MGTransferPrebuilt::build_matrices on the device (gmg_build_transfer) instead of here + upload
            {"Transfer matrices on device", "true"},
            // the right-hand side integrated on the device from densities that stay there (gmg_rhs_assemble)
            {"RHS on device", "true"},
            // the short-ranged pair sum over the pairs closer than this many smoothing lengths, found through
            // cell bins (erfc(6) = 2e-17: beyond 6 r_c a pair contributes nothing in double precision); 0 = all pairs as the
            // reference (:1325-1332).  With it the energy is also evaluated for the large systems the reference skips (:1554).
            {"Short-range cutoff in smoothing lengths", "0"}, {"Energy for large systems", "false"}};
}
void ParameterReader::parse_input_from_string(const std::string &text) {
  std::istringstream in(text);
  std::string line;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    const size_t s = line.find("set ");
    if (s == std::string::npos) continue;
    const size_t eq = line.find('=', s);
    if (eq == std::string::npos) continue;
    auto trim = [](std::string t) {
      const size_t a = t.find_first_not_of(" \t\r"), b = t.find_last_not_of(" \t\r");
      return a == std::string::npos ? std::string() : t.substr(a, b - a + 1);
    };
    const std::string key = trim(line.substr(s + 4, eq - s - 4));
    if (!values.count(key)) throw std::runtime_error("ParameterHandler: undeclared entry <" + key + ">");
    values[key] = trim(line.substr(eq + 1));
  }
}
void ParameterReader::read_parameters(const std::string &file) {
  std::ifstream in(file);
  if (!in) throw std::runtime_error("cannot open parameter file " + file);
  std::stringstream ss;
  ss << in.rdbuf();
  parse_input_from_string(ss.str());
}
std::string ParameterReader::get(const std::string &k) const { return values.at(k); }
double ParameterReader::get_double(const std::string &k) const { return std::stod(values.at(k)); }
long ParameterReader::get_integer(const std::string &k) const { return std::stol(values.at(k)); }
bool ParameterReader::get_bool(const std::string &k) const { return values.at(k) == "true"; }

Parameters Parameters::from(const ParameterReader &prm) {  // src/main.cc:25-68
  Parameters p;
  p.number_of_global_refinement = (unsigned)prm.get_integer("Number of global refinement");
  p.domain_size_left = prm.get_double("Domain limit left");
  p.domain_size_right = prm.get_double("Domain limit right");
  p.mesh_size_h = prm.get_double("Mesh size");
  p.repetitions_for_vacuum = (unsigned)prm.get_integer("Vacuum repetitions");
  p.number_of_adaptive_refinement_cycles = (unsigned)prm.get_integer("Number of Adaptive Refinement");
  p.r_c = prm.get_double("smoothing length");
  p.nonzero_density_radius_parameter = prm.get_double("Nonzero Density radius parameter around each charge");
  p.flag_analytical_solution = prm.get_bool("Output and calculation of Analytical solution");
  p.flag_rhs_field = prm.get_bool("Output of RHS field");
  p.flag_atoms_support = prm.get_bool("Output of support of each atom");
  p.flag_rhs_assembly = prm.get_bool("Flag for RHS evaluation optimization");
  p.quadrature_degree_rhs = (unsigned)prm.get_integer("Quadrature points for RHS function");
  p.flag_output_time = prm.get_bool("Output time summary table");
  p.degree = (unsigned)prm.get_integer("Polynomial degree");
  p.PreconditionerType = prm.get("Preconditioner");
  p.Problemtype = prm.get("Problem");
  p.dim = (int)prm.get_integer("Dimension");
  p.Boundary_conditions = prm.get("Boundary conditions selection");
  p.LammpsInputFile = prm.get("Lammps input file");
  p.smoother = prm.get("Smoother");
  p.smoother_omega = prm.get_double("Smoother damping");
  p.smoother_steps = (int)prm.get_integer("Smoother steps");
  p.chebyshev_degree = (int)prm.get_integer("Chebyshev degree");
  p.device_resident_outer_cg = prm.get_bool("Device resident outer CG");
  p.ssor_blocks = (int)prm.get_integer("SSOR blocks");
  p.densities_on_device = prm.get_bool("Charge densities on device");
  p.partition_level0 = prm.get("Partition level 0");
  p.refinement_estimator = prm.get("Refinement estimator");
  if (p.refinement_estimator != "Kelly + residual" && p.refinement_estimator != "Kelly")
    throw std::runtime_error("Refinement estimator must be <Kelly + residual> or <Kelly>");
  p.level0_matrix_on_device = prm.get_bool("Level 0 matrix on device");
  p.transfer_on_device = prm.get_bool("Transfer matrices on device");
  p.rhs_on_device = prm.get_bool("RHS on device");
  p.short_range_cutoff = prm.get_double("Short-range cutoff in smoothing lengths");
  p.energy_for_large_systems = prm.get_bool("Energy for large systems");
  p.level0_numbering = prm.get("Level 0 numbering");
  if (p.level0_numbering != "lexicographic" && p.level0_numbering != "cell-wise")
    throw std::runtime_error("Level 0 numbering must be <lexicographic> or <cell-wise>");
  return p;
}

std::vector<double> nacl_lattice(int n, std::vector<double> &q) {
  // atom/atom_n{1,3,5,7,10,20}_*.data of the reference: n^3 unit cells of 8 ions, spacing 0.5,
  // cells ordered x outermost, z innermost (verified against the files, tests/test_host.py)
  static const double base[8][3] = {{0, 0, 0}, {.5, 0, 0}, {.5, .5, 0}, {0, .5, 0}, {.5, 0, .5}, {0, 0, .5}, {0, .5, .5}, {.5, .5, .5}};
  std::vector<double> x;
  q.clear();
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b)
      for (int c = 0; c < n; ++c)
        for (int k = 0; k < 8; ++k) {
          x.push_back(base[k][0] + a); x.push_back(base[k][1] + b); x.push_back(base[k][2] + c);
          q.push_back(k % 2 == 0 ? 1.0 : -1.0);
        }
  return x;
}

// ======================================================================== atoms

template <int dim>
struct LaplaceProblem<dim>::AtomBins {
  double lo[3] = {0, 0, 0}, size = 1;
  int n[3] = {1, 1, 1};
  std::vector<int64_t> ptr;
  std::vector<int32_t> items;
};

template <int dim>
LaplaceProblem<dim>::LaplaceProblem(const Parameters &p) : par(p) {
  if (p.degree != 1) throw std::runtime_error("only Q1 elements (Polynomial degree = 1) are supported");
  pcout("Problem type is:   " + par.Problemtype);
  pcout("Preconditioner :    " + par.PreconditionerType);
  pcout(par.flag_rhs_assembly ? "Rhs assembly optimization ENABLED" : "Without rhs assembly optimization");
}

template <int dim>
LaplaceProblem<dim>::~LaplaceProblem() {
  if (gmg) {
    for (double *p : {d_solution, d_rhs, d_full})
      if (p) gmg_vec_free(gmg, p);
    gmg_destroy(gmg);
  }
}

template <int dim>
void LaplaceProblem<dim>::pcout(const std::string &s) {
  log += s;
  log += '\n';
  if (echo) std::cout << s << std::endl;
}

template <int dim>
void LaplaceProblem<dim>::read_lammps_input_file(const std::string &filename) {
  // token-counting reader of :181-258: token #2 = number of atoms, token #35 starts the records
  if (dim != 3) {
    lammpsinput = false;
    pcout("\nReading of Lammps input file implemented for 3D only\n");
    return;
  }
  std::ifstream file(filename);
  if (!file.is_open()) {
    lammpsinput = false;
    pcout("Unable to open the file.");
    return;
  }
  lammpsinput = true;
  std::string tok;
  unsigned int count = 0;
  while (!file.eof()) {
    if (count == 2) {
      file >> number_of_atoms;
      pcout("Number of atoms: " + std::to_string(number_of_atoms));
      charges.resize(number_of_atoms);
      atom_positions.resize(3 * (size_t)number_of_atoms);
    } else if (count == 35) {
      for (unsigned int i = 0; i < number_of_atoms; ++i) {
        double a, b, type;
        file >> a >> b >> type >> charges[i] >> atom_positions[3 * i] >> atom_positions[3 * i + 1] >> atom_positions[3 * i + 2];
      }
    } else {
      file >> tok;
    }
    ++count;
  }
}

template <int dim>
void LaplaceProblem<dim>::set_atoms(const std::vector<double> &q, const std::vector<double> &xyz) {
  charges = q;
  atom_positions = xyz;
  number_of_atoms = (unsigned)q.size();
  lammpsinput = true;
  pcout("Number of atoms: " + std::to_string(number_of_atoms));
}

template <int dim>
void LaplaceProblem<dim>::rhs_assembly_optimization() {
  // :260-306 keeps, per cell, the atoms with ANY cell vertex closer than cutoff * r_c; children
  // inherit the list unchanged (:441-450), so the list is a function of the root cell alone.
  // Instead of the O(cells x atoms) scan the atoms are binned once; atoms_of_root_cell()
  // evaluates the same predicate on the candidates of the neighbouring bins.
  bins.reset(new AtomBins());
  AtomBins &B = *bins;
  const double cut = par.nonzero_density_radius_parameter * par.r_c;
  double hi[3];
  for (int d = 0; d < 3; ++d) { B.lo[d] = 1e300; hi[d] = -1e300; }
  for (unsigned i = 0; i < number_of_atoms; ++i)
    for (int d = 0; d < 3; ++d) {
      B.lo[d] = std::min(B.lo[d], atom_positions[3 * i + (size_t)d]);
      hi[d] = std::max(hi[d], atom_positions[3 * i + (size_t)d]);
    }
  B.size = std::max(cut, 1e-12);
  int64_t total = 1;
  for (int d = 0; d < 3; ++d) { B.n[d] = std::max(1, (int)std::floor((hi[d] - B.lo[d]) / B.size) + 1); total *= B.n[d]; }
  B.ptr.assign((size_t)total + 1, 0);
  auto bin_of = [&](unsigned i) {
    int b[3];
    for (int d = 0; d < 3; ++d) b[d] = std::min(B.n[d] - 1, std::max(0, (int)std::floor((atom_positions[3 * i + (size_t)d] - B.lo[d]) / B.size)));
    return (int64_t)b[0] + B.n[0] * ((int64_t)b[1] + (int64_t)B.n[1] * b[2]);
  };
  for (unsigned i = 0; i < number_of_atoms; ++i) B.ptr[(size_t)bin_of(i) + 1]++;
  for (int64_t b = 0; b < total; ++b) B.ptr[(size_t)b + 1] += B.ptr[(size_t)b];
  B.items.resize(number_of_atoms);
  std::vector<int64_t> pos(B.ptr.begin(), B.ptr.end() - 1);
  for (unsigned i = 0; i < number_of_atoms; ++i) B.items[(size_t)pos[(size_t)bin_of(i)]++] = (int32_t)i;
}

template <int dim>
void LaplaceProblem<dim>::atoms_of_root_cell(const int rc[3], std::vector<int32_t> &out) const {
  out.clear();
  const AtomBins &B = *bins;
  const double cut = par.nonzero_density_radius_parameter * par.r_c;
  const double h = triangulation.h0;
  double lo[3], hi[3];
  int b0[3], b1[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = triangulation.origin + h * rc[d];
    hi[d] = triangulation.origin + h * (rc[d] + 1);
    b0[d] = (int)std::floor((lo[d] - cut - B.lo[d]) / B.size);
    b1[d] = (int)std::floor((hi[d] + cut - B.lo[d]) / B.size);
    b0[d] = std::max(b0[d], 0);
    b1[d] = std::min(b1[d], B.n[d] - 1);
  }
  for (int z = b0[2]; z <= b1[2]; ++z)
    for (int y = b0[1]; y <= b1[1]; ++y)
      for (int x = b0[0]; x <= b1[0]; ++x) {
        const int64_t b = (int64_t)x + B.n[0] * ((int64_t)y + (int64_t)B.n[1] * z);
        for (int64_t k = B.ptr[(size_t)b]; k < B.ptr[(size_t)b + 1]; ++k) {
          const int32_t i = B.items[(size_t)k];
          double d2 = 0;
          for (int d = 0; d < 3; ++d) {  // nearest vertex, direction by direction
            const double a = atom_positions[3 * (size_t)i + (size_t)d];
            const double dl = a - lo[d], dh = a - hi[d];
            const double m = std::fabs(dl) <= std::fabs(dh) ? dl : dh;
            d2 += m * m;
          }
          if (std::sqrt(d2) < cut) out.push_back(i);
        }
      }
  std::sort(out.begin(), out.end());  // std::set iteration order of the reference (:559-560)
}

// ======================================================================== problem functions

template <int dim>
double LaplaceProblem<dim>::coefficient(const double x[3]) const {
  if (par.Problemtype == "Step16") {  // include/step_50.h:246-254
    double s = 0;
    for (int d = 0; d < dim; ++d) s += x[d] * x[d];
    return s < 0.25 ? 5.0 : 1.0;
  }
  return 1.0;
}

template <int dim>
double LaplaceProblem<dim>::rhs_function(const double x[3]) const {
  if (par.Problemtype == "Step16") return 10.0;  // include/step_50.h:240-244
  double s = 0;                                   // include/step_50.h:321-329
  for (int d = 0; d < dim; ++d) s += x[d] * x[d];
  const double c = s / (par.r_c * par.r_c);
  return (8.0 * std::exp(-4.0 * c) - std::exp(-c)) / (std::pow(par.r_c, 3) * std::pow(M_PI, 1.5));
}

template <int dim>
double LaplaceProblem<dim>::boundary_value(const double x[3]) const {
  if (par.Boundary_conditions == "Homogeneous") return 0.0;
  if (par.Boundary_conditions == "Exact") {  // Analytical_Solution::value, include/step_50.h:338-353
    if (par.Problemtype != "GaussianCharges") return 0.0;
    double v = 0;
    const double inv = 1.0 / (std::sqrt(M_PI) * par.r_c);
    for (unsigned i = 0; i < number_of_atoms; ++i) {
      double r2 = 0;
      for (int d = 0; d < dim; ++d) { const double t = x[d] - atom_positions[3 * i + (size_t)d]; r2 += t * t; }
      const double r = std::sqrt(r2);
      v += r < 1e-10 ? charges[i] * 2.0 * inv : charges[i] * (std::erf(r / par.r_c) / r);
    }
    return v;
  }
  // Inhomogeneous: NonZeroDBC with x0 = 0, dipole p0, quadrupole forced to 0 (:623-624, step_50.h:378-385)
  double r2 = 0, px = 0;
  for (int d = 0; d < dim; ++d) { r2 += x[d] * x[d]; px += dipole_moment[d] * x[d]; }
  const double r = std::sqrt(r2);
  return px / std::pow(r, 3) + (0.5 * 0.0) / std::pow(r, 5);
}

// ======================================================================== mesh + DoFs

template <int dim>
void LaplaceProblem<dim>::make_initial_grid() {
  if (par.Problemtype == "Step16") {  // :1496-1497
    triangulation.create_lattice(1, par.domain_size_left, par.domain_size_right - par.domain_size_left);
    triangulation.refine_global((int)par.number_of_global_refinement);
  } else {  // :1504-1526
    const double a = 2 * par.mesh_size_h;
    const double N = (par.domain_size_right - par.domain_size_left) / a;
    const double M = par.repetitions_for_vacuum;
    const unsigned int reps = (unsigned int)(2 * (N + 2 * M));
    const double lo = par.domain_size_left - M * a, hi = par.domain_size_right + M * a;
    if (std::pow((double)reps, dim) > 2.0e8) throw std::runtime_error("initial lattice too large for this host");
    // "Need to set #Global_ref = 0" (:1503): the reference applies no global refinement here
    triangulation.create_lattice((int)reps, lo, (hi - lo) / reps);
  }
}

template <int dim>
void LaplaceProblem<dim>::cell_dofs(const ActiveCell &c, int32_t *out) const {
  // (tables filled by distribute_dofs: the assembly, the estimator and the transfer loops ask for the same cell again and again,
  // and a hash look-up per vertex was a third of assemble_system)
  const int32_t *t = &active_cell_dof_table[(size_t)active_index_of_cell[(size_t)c.level][(size_t)c.index] * (1 << dim)];
  for (int a = 0; a < (1 << dim); ++a) out[a] = t[a];
}
template <int dim>
void LaplaceProblem<dim>::level_cell_dofs(int level, int32_t ci, int32_t *out) const {
  const int32_t *t = &level_cell_dof_table[(size_t)level][(size_t)ci * (1 << dim)];
  for (int a = 0; a < (1 << dim); ++a) out[a] = t[a];
}

template <int dim>
void LaplaceProblem<dim>::distribute_dofs() {
  // DoFs are numbered in the order cells meet their vertices: active cells by (level, index) for
  // the active mesh, all cells of a level by index for the level DoFs.
  const int L = triangulation.n_levels();
  active_cells.clear();
  sublap(nullptr);
  active_index_of_cell.assign((size_t)L, {});
  for (int l = 0; l < L; ++l) {
    active_index_of_cell[(size_t)l].assign(triangulation.levels[(size_t)l].size(), -1);
    for (size_t c = 0; c < triangulation.levels[(size_t)l].size(); ++c)
      if (triangulation.levels[(size_t)l][c].first_child < 0) {
        active_index_of_cell[(size_t)l][c] = (int32_t)active_cells.size();
        active_cells.push_back({l, (int32_t)c});
      }
  }
  // (level-0 vertices index a dense array by lattice position, the others a hash table: flat_map.h)
  const int lattice[3] = {triangulation.n0 + 1, triangulation.n0 + 1, dim == 3 ? triangulation.n0 + 1 : 1};
  dof_of_vertex.reset(kMaxLevelShift, lattice);
  vertex_of_dof.clear();
  sublap("dofs: active cell list");
  dof_of_vertex.reserve_sparse(active_cells.size() - (size_t)std::min<int64_t>((int64_t)active_cells.size(), (int64_t)triangulation.levels[0].size()));
  active_cell_dof_table.resize(active_cells.size() * (size_t)(1 << dim));
  size_t slot = 0;
  for (const ActiveCell &ac : active_cells) {
    const Cell &cell = triangulation.levels[(size_t)ac.level][(size_t)ac.index];
    for (int a = 0; a < (1 << dim); ++a) {
      const uint64_t key = triangulation.vertex_key(ac.level, cell, a);
      const auto ins = dof_of_vertex.emplace(key, (int32_t)vertex_of_dof.size());
      if (ins.second) vertex_of_dof.push_back(key);
      active_cell_dof_table[slot++] = *ins.first;
    }
  }
  level_dof_of_vertex.assign((size_t)L, {});
  sublap("dofs: active numbering");
  level_vertex_of_dof.assign((size_t)L, {});
  level_cell_dof_table.assign((size_t)L, {});
  for (int l = 0; l < L; ++l) {
    auto &map = level_dof_of_vertex[(size_t)l];
    auto &vec = level_vertex_of_dof[(size_t)l];
    auto &tab = level_cell_dof_table[(size_t)l];
    const int none[3] = {0, 0, 0};
    map.reset(kMaxLevelShift, l == 0 ? lattice : none);
    if (l > 0) map.reserve_sparse(triangulation.levels[(size_t)l].size() * 2);
    tab.resize(triangulation.levels[(size_t)l].size() * (size_t)(1 << dim));
    if (l == 0 && par.level0_numbering == "lexicographic") {
      // Level 0 is the undivided lattice (subdivided_hyper_rectangle, src/step-50.cc:1526) and carries no smoother: its
      // numbering only permutes rows / columns of A_0, P_0 and the copy indices (results change in the last bits of a few
      // sums).  Vertex keys order by (z, y, x): ascending keys = lexicographic DoFs, and A_0 becomes a pure 27-point stencil
      // with strides 1, nx, nx ny -- what the device's plane-by-plane kernel (csrc/gmg_lattice.hpp) wants.  deal.II's real
      // level numbering on the reference's p4est partition is not reproducible here either way.
      // DoF of a vertex = its lattice position: no table is searched, no key is sorted.
      const int64_t nx = lattice[0], ny = lattice[1], nz = lattice[2];
      vec.resize((size_t)(nx * ny * nz));
#pragma omp parallel for schedule(static)
      for (int64_t k = 0; k < nz; ++k)
        for (int64_t j = 0; j < ny; ++j)
          for (int64_t i = 0; i < nx; ++i) {
            const uint64_t key = pack3((uint64_t)i << kMaxLevelShift, (uint64_t)j << kMaxLevelShift, (uint64_t)k << kMaxLevelShift);
            vec[(size_t)(i + nx * (j + ny * k))] = key;
          }
      for (size_t i = 0; i < vec.size(); ++i) map.emplace(vec[i], (int32_t)i);
      const auto &cells = triangulation.levels[0];
#pragma omp parallel for schedule(static)
      for (int64_t c = 0; c < (int64_t)cells.size(); ++c)
        for (int a = 0; a < (1 << dim); ++a) {
          const Cell &cell = cells[(size_t)c];
          tab[(size_t)c * (size_t)(1 << dim) + (size_t)a] =
              (int32_t)((cell.c[0] + (a & 1)) + nx * ((cell.c[1] + ((a >> 1) & 1)) + ny * (dim == 3 ? cell.c[2] + ((a >> 2) & 1) : 0)));
        }
      continue;
    }
    size_t ls = 0;
    for (const Cell &cell : triangulation.levels[(size_t)l])
      for (int a = 0; a < (1 << dim); ++a) {
        const uint64_t key = triangulation.vertex_key(l, cell, a);
        const auto ins = map.emplace(key, (int32_t)vec.size());
        if (ins.second) vec.push_back(key);
        tab[ls++] = *ins.first;
      }
  }
  sublap("dofs: level numbering");
}

template <int dim>
void LaplaceProblem<dim>::make_constraints() {
  // DoFTools::make_hanging_node_constraints + VectorTools::interpolate_boundary_values +
  // constraints.close() (:661-696), and MGConstrainedDoFs (:704-706).
  const int64_t n = (int64_t)vertex_of_dof.size();
  constraint_of_dof.assign((size_t)n, -1);
  constraint_lines.clear();
  const int shift0 = kMaxLevelShift;
  for (const ActiveCell &ac : active_cells) {
    const int l = ac.level;
    const Cell &c = triangulation.levels[(size_t)l][(size_t)ac.index];
    for (int d = 0; d < dim; ++d)
      for (int side = 0; side < 2; ++side) {
        int nb[3] = {c.c[0], c.c[1], c.c[2]};
        nb[d] += side ? 1 : -1;
        const int32_t N = triangulation.find(l, nb[0], nb[1], nb[2]);
        if (N < 0 || triangulation.active(l, N)) continue;
        // the face of this coarse cell is refined on the other side: its centre and edge
        // mid-points are hanging nodes
        uint64_t corner[4][3];
        int nc = 0;
        for (int a = 0; a < (1 << dim); ++a)
          if (((a >> d) & 1) == side) {
            uint64_t v[3];
            Forest<dim>::unpack(triangulation.vertex_key(l, c, a), v);
            for (int e = 0; e < 3; ++e) corner[nc][e] = v[e];
            ++nc;
          }
        auto add_line = [&](const int *ids, int m) {
          uint64_t s[3] = {0, 0, 0};
          for (int q = 0; q < m; ++q)
            for (int e = 0; e < 3; ++e) s[e] += corner[ids[q]][e];
          const uint64_t key = pack3(s[0] / (uint64_t)m, s[1] / (uint64_t)m, s[2] / (uint64_t)m);
          const int32_t *it = dof_of_vertex.find(key);
          if (!it) throw std::logic_error("hanging node without a DoF: mesh is not 2:1 balanced");
          if (constraint_of_dof[(size_t)*it] >= 0) return;
          ConstraintLine line;
          line.hanging = true;
          for (int q = 0; q < m; ++q)
            line.entries.push_back({dof_of_vertex.at(pack3(corner[ids[q]][0], corner[ids[q]][1], corner[ids[q]][2])), 1.0 / m});
          constraint_of_dof[(size_t)*it] = (int32_t)constraint_lines.size();
          constraint_lines.push_back(line);
        };
        if (dim == 2) {
          const int e[2] = {0, 1};
          add_line(e, 2);
        } else {
          const int all[4] = {0, 1, 2, 3};
          add_line(all, 4);
          const int edges[4][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}};
          for (auto &e : edges) add_line(e, 2);
        }
      }
  }
  (void)shift0;
  // Dirichlet lines for boundary DoFs that are not already (hanging-node) constrained
  if (lammpsinput) compute_moments();
  for (int64_t i = 0; i < n; ++i) {
    if (!triangulation.vertex_on_boundary(vertex_of_dof[(size_t)i]) || constraint_of_dof[(size_t)i] >= 0) continue;
    double x[3];
    triangulation.vertex_coords(vertex_of_dof[(size_t)i], x);
    ConstraintLine line;
    line.inhomogeneity = boundary_value(x);
    constraint_of_dof[(size_t)i] = (int32_t)constraint_lines.size();
    constraint_lines.push_back(line);
  }
  // close(): masters that are themselves (Dirichlet) constrained fold into the inhomogeneity
  for (ConstraintLine &line : constraint_lines) {
    if (!line.hanging) continue;
    std::vector<std::pair<int32_t, double>> kept;
    for (auto &e : line.entries) {
      const int32_t cm = constraint_of_dof[(size_t)e.first];
      if (cm < 0) { kept.push_back(e); continue; }
      const ConstraintLine &m = constraint_lines[(size_t)cm];
      if (m.hanging) throw std::logic_error("hanging node constrained to a hanging node");
      line.inhomogeneity += e.second * m.inhomogeneity;
    }
    line.entries.swap(kept);
  }
  // MGConstrainedDoFs: boundary indices and refinement-edge indices per level
  const int L = triangulation.n_levels();
  level_boundary.assign((size_t)L, {});
  level_refinement_edge.assign((size_t)L, {});
  for (int l = 0; l < L; ++l) {
    const auto &vec = level_vertex_of_dof[(size_t)l];
    level_boundary[(size_t)l].assign(vec.size(), 0);
    level_refinement_edge[(size_t)l].assign(vec.size(), 0);
    for (size_t i = 0; i < vec.size(); ++i) level_boundary[(size_t)l][i] = triangulation.vertex_on_boundary(vec[i]);
    if (l == 0) continue;
    const int nl = triangulation.n0 << l;
    for (size_t ci = 0; ci < triangulation.levels[(size_t)l].size(); ++ci) {
      const Cell &c = triangulation.levels[(size_t)l][ci];
      for (int d = 0; d < dim; ++d)
        for (int side = 0; side < 2; ++side) {
          int nb[3] = {c.c[0], c.c[1], c.c[2]};
          nb[d] += side ? 1 : -1;
          if (nb[d] < 0 || nb[d] >= nl) continue;  // domain boundary
          if (triangulation.find(l, nb[0], nb[1], nb[2]) >= 0) continue;
          for (int a = 0; a < (1 << dim); ++a)
            if (((a >> d) & 1) == side)
              level_refinement_edge[(size_t)l][(size_t)level_dof_of_vertex[(size_t)l].at(triangulation.vertex_key(l, c, a))] = 1;
        }
    }
  }
}

template <int dim>
void LaplaceProblem<dim>::setup_system(unsigned int cycle) {
  distribute_dofs();
  const int64_t n = (int64_t)vertex_of_dof.size();
  solution.assign((size_t)n, 0.0);
  system_rhs.assign((size_t)n, 0.0);
  if (cycle == 0 && par.flag_rhs_assembly && lammpsinput) rhs_assembly_optimization();
  sublap("rhs_assembly_optimization");
  if (lammpsinput) compute_charge_densities();
  sublap("compute_charge_densities");
  make_constraints();  // calls compute_moments() first when atoms are present (:675-679)
  sublap("make_constraints");
}

template <int dim>
void LaplaceProblem<dim>::compute_charge_densities() {
  // :509-575, rho(x_q) = 4 pi / (r_c^3 pi^1.5) sum_k q_k exp(-|x_q - x_k|^2 / r_c^2)
  const Quadrature<dim> quad((int)(par.degree + par.quadrature_degree_rhs));
  const size_t nq = quad.p.size();
  density_values_for_each_cell.assign(active_cells.size(), {});
  densities_device_resident = false;
  if (densities_on_device && dim == 3) {
    // the same sums on the MI355X (gmg_charge_density), cell geometry in, rho out
    if (ensure_context() != GMG_OK) throw std::runtime_error("charge densities on the device: " + last_error);
    const size_t nc = active_cells.size();
    std::vector<double> lo(3 * nc), hh(nc), rlo(3 * nc), qp(3 * nq), dens(nc * nq);
    for (size_t ci = 0; ci < nc; ++ci) {
      const ActiveCell &ac = active_cells[ci];
      const Cell &cell = triangulation.levels[(size_t)ac.level][(size_t)ac.index];
      double x0[3];
      triangulation.cell_origin(ac.level, cell, x0);
      hh[ci] = triangulation.cell_size(ac.level);
      for (int d = 0; d < 3; ++d) {
        lo[3 * ci + (size_t)d] = x0[d];
        rlo[3 * ci + (size_t)d] = triangulation.origin + triangulation.h0 * (cell.c[d] >> ac.level);
      }
    }
    for (size_t q = 0; q < nq; ++q)
      for (int d = 0; d < 3; ++d) qp[3 * q + (size_t)d] = quad.p[q][(size_t)d];
    // "RHS on device": the densities stay in HBM (dens = NULL) and gmg_rhs_assemble integrates them there; the host gets a
    // copy only if something asks for it (HEAD's residual estimator, the rc_variation check vector): ensure_host_densities()
    densities_device_resident = par.rhs_on_device;
    const int rc = gmg_charge_density(gmg, (int64_t)nc, lo.data(), hh.data(), rlo.data(), triangulation.h0, (int64_t)number_of_atoms,
                                      atom_positions.data(), charges.data(), par.r_c, par.nonzero_density_radius_parameter * par.r_c,
                                      par.flag_rhs_assembly ? 1 : 0, (int)nq, qp.data(), densities_device_resident ? nullptr : dens.data());
    if (rc != GMG_OK) throw std::runtime_error(std::string("gmg_charge_density: ") + gmg_last_error(gmg));
    if (densities_device_resident) { density_values_for_each_cell.clear(); return; }
    for (size_t ci = 0; ci < nc; ++ci) density_values_for_each_cell[ci].assign(dens.begin() + (std::ptrdiff_t)(ci * nq), dens.begin() + (std::ptrdiff_t)((ci + 1) * nq));
    return;
  }
  const double constant_value = 4.0 * M_PI / (std::pow(par.r_c, 3) * std::pow(M_PI, 1.5));
  const double r_c_squared_inverse = 1.0 / (par.r_c * par.r_c);
#pragma omp parallel
  {
    std::vector<int32_t> atoms;
#pragma omp for schedule(dynamic, 256)
    for (int64_t ci = 0; ci < (int64_t)active_cells.size(); ++ci) {
      const ActiveCell &ac = active_cells[(size_t)ci];
      const Cell &cell = triangulation.levels[(size_t)ac.level][(size_t)ac.index];
      double x0[3];
      triangulation.cell_origin(ac.level, cell, x0);
      const double h = triangulation.cell_size(ac.level);
      const bool use_list = par.flag_rhs_assembly;
      if (use_list) {
        const int rc[3] = {cell.c[0] >> ac.level, cell.c[1] >> ac.level, cell.c[2] >> ac.level};
        atoms_of_root_cell(rc, atoms);
      }
      std::vector<double> &dens = density_values_for_each_cell[(size_t)ci];
      dens.assign(nq, 0.0);
      const size_t na = use_list ? atoms.size() : (size_t)number_of_atoms;
      for (size_t q = 0; q < nq; ++q) {
        double xq[3] = {0, 0, 0};
        for (int d = 0; d < dim; ++d) xq[d] = x0[d] + h * quad.p[q][(size_t)d];
        double s = 0.0;
        for (size_t t = 0; t < na; ++t) {
          const size_t k = use_list ? (size_t)atoms[t] : t;
          double r2 = 0;
          for (int d = 0; d < 3; ++d) { const double dd = atom_positions[3 * k + (size_t)d] - xq[d]; r2 += dd * dd; }
          const double r = std::sqrt(r2);  // the reference squares the distance again (:547-548)
          s += constant_value * std::exp(-(r * r) * r_c_squared_inverse) * charges[k];
        }
        dens[q] = s;
      }
    }
  }
}

template <int dim>
void LaplaceProblem<dim>::ensure_host_densities() {
  if (!lammpsinput || !densities_device_resident || density_values_for_each_cell.size() == active_cells.size()) return;
  const Quadrature<dim> quad((int)(par.degree + par.quadrature_degree_rhs));
  const size_t nq = quad.p.size(), nc = active_cells.size();
  std::vector<double> dens(nc * nq);
  if (gmg_get_charge_density(gmg, (int64_t)nc, (int)nq, dens.data()) != GMG_OK) throw std::runtime_error(std::string("gmg_get_charge_density: ") + gmg_last_error(gmg));
  density_values_for_each_cell.assign(nc, {});
  for (size_t ci = 0; ci < nc; ++ci) density_values_for_each_cell[ci].assign(dens.begin() + (std::ptrdiff_t)(ci * nq), dens.begin() + (std::ptrdiff_t)((ci + 1) * nq));
}

template <int dim>
void LaplaceProblem<dim>::compute_moments() {
  // :577-644 -- the quadrupole is integrated and then overwritten with 0 (:623-624): only the
  // dipole survives.
  for (int d = 0; d < 3; ++d) dipole_moment[d] = 0;
  for (unsigned k = 0; k < number_of_atoms; ++k)
    for (int d = 0; d < 3; ++d) dipole_moment[d] += charges[k] * atom_positions[3 * k + (size_t)d];
}

// ======================================================================== assembly

namespace {
template <int dim, class Coef>
void cell_matrix(const Quadrature<dim> &q2, double h, const double x0[3], Coef coef, bool constant, double K[1 << dim][1 << dim]) {
  constexpr int nv = 1 << dim;
  for (int i = 0; i < nv; ++i)
    for (int j = 0; j < nv; ++j) K[i][j] = 0;
  const double scale = std::pow(h, dim - 2);  // JxW / h^2 from the two gradients
  for (size_t q = 0; q < q2.p.size(); ++q) {
    double c = 1.0;
    if (!constant) {
      double x[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) x[d] = x0[d] + h * q2.p[q][(size_t)d];
      c = coef(x);
    }
    for (int i = 0; i < nv; ++i)
      for (int j = 0; j < nv; ++j) {
        double g = 0;
        for (int d = 0; d < dim; ++d) g += q2.grad[q][(size_t)i][(size_t)d] * q2.grad[q][(size_t)j][(size_t)d];
        K[i][j] += c * g * q2.w[q] * scale;
      }
  }
}
}  // namespace

template <int dim>
void LaplaceProblem<dim>::assemble_system() {
  constexpr int nv = 1 << dim;
  const int64_t n = (int64_t)vertex_of_dof.size();
  const Quadrature<dim> q_laplace((int)par.degree + 1), q_rhs((int)(par.degree + par.quadrature_degree_rhs));
  const bool constant_coef = par.Problemtype != "Step16";
  // coupling lists: the cell's DoFs plus the masters of its constrained DoFs
  sublap(nullptr);
  std::vector<int64_t> cptr(active_cells.size() + 1, 0);
  std::vector<int32_t> citems;
  citems.reserve(active_cells.size() * nv);
  for (size_t ci = 0; ci < active_cells.size(); ++ci) {
    int32_t dofs[nv];
    cell_dofs(active_cells[ci], dofs);
    const size_t begin = citems.size();
    for (int a = 0; a < nv; ++a) {
      citems.push_back(dofs[a]);
      const int32_t cl = constraint_of_dof[(size_t)dofs[a]];
      if (cl >= 0)
        for (auto &e : constraint_lines[(size_t)cl].entries) citems.push_back(e.first);
    }
    std::sort(citems.begin() + (std::ptrdiff_t)begin, citems.end());
    citems.erase(std::unique(citems.begin() + (std::ptrdiff_t)begin, citems.end()), citems.end());
    cptr[ci + 1] = (int64_t)citems.size();
  }
  sublap("assemble: coupling lists");
  system_matrix = pattern_from_cells(n, cptr, citems);
  sublap("assemble: sparsity pattern");
  system_rhs.assign((size_t)n, 0.0);
  // "RHS on device": the cell loop records WHERE every F_i goes (DoF, slot = cell * 2^dim + i, weight) and which Dirichlet
  // terms it loses instead of forming F from densities the host does not have; gmg_rhs_assemble does the arithmetic
  const bool rhs_dev = lammpsinput && densities_device_resident;
  std::vector<int32_t> plan_dof, plan_slot, term_slot;
  std::vector<uint8_t> plan_code;
  std::vector<double> term_value, coef_table(256, 0.0);
  int n_codes = 1;  // code 0: the slot's value as it is
  auto code_of = [&](double w) -> int {
    for (int c = 1; c < n_codes; ++c)
      if (coef_table[(size_t)c] == w) return c;
    if (n_codes >= 256) throw std::runtime_error("RHS on device: more than 255 distinct constraint weights");
    coef_table[(size_t)n_codes] = w;
    return n_codes++;
  };
  if (rhs_dev) { plan_dof.reserve(active_cells.size() * nv); plan_slot.reserve(active_cells.size() * nv); plan_code.reserve(active_cells.size() * nv); }

  double Kc[nv][nv];
  if (constant_coef) {
    double x0[3] = {0, 0, 0};
    cell_matrix<dim>(q_laplace, 1.0, x0, [&](const double *) { return 1.0; }, true, Kc);
  }
  // One pass over the cells per THREAD for the matrix, restricted to the rows the thread owns (SURVEY 8 / VERDICT r02 #8:
  // threaded host setup): a cell is visited by the threads whose row range meets its coupling list, every entry still
  // receives its cells' contributions in cell order -- the same bits as the sequential loop, whatever the thread count.
  // The right-hand side (or the plan for gmg_rhs_assemble) follows in a sequential pass of its own.
  auto cell_K = [&](size_t ci, double (&K)[nv][nv], double (&x0)[3], double &h) {
    const ActiveCell &ac = active_cells[ci];
    const Cell &cell = triangulation.levels[(size_t)ac.level][(size_t)ac.index];
    h = triangulation.cell_size(ac.level);
    triangulation.cell_origin(ac.level, cell, x0);
    if (constant_coef) {
      const double s = std::pow(h, dim - 2);
      for (int i = 0; i < nv; ++i)
        for (int j = 0; j < nv; ++j) K[i][j] = Kc[i][j] * s;
    } else {
      cell_matrix<dim>(q_laplace, h, x0, [&](const double *x) { return coefficient(x); }, false, K);
    }
  };
  auto cell_lines = [&](size_t ci, int32_t (&dofs)[nv], const ConstraintLine *(&line)[nv]) {
    cell_dofs(active_cells[ci], dofs);
    // ConstraintMatrix::distribute_local_to_global (:793-795, 825-828)
    for (int a = 0; a < nv; ++a) {
      const int32_t cl = constraint_of_dof[(size_t)dofs[a]];
      line[a] = cl >= 0 ? &constraint_lines[(size_t)cl] : nullptr;
    }
  };
#pragma omp parallel
  {
    int nt = 1, tid = 0;
#ifdef _OPENMP
    nt = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
    const int64_t r0 = n * tid / nt, r1 = n * (tid + 1) / nt;
    auto add = [&](int32_t r, int32_t c, double v) { if (r >= r0 && r < r1) system_matrix.add(r, c, v); };
    for (size_t ci = 0; ci < active_cells.size(); ++ci) {
      // (the coupling list of a cell is sorted: its first and last entries bound the rows the cell can touch)
      if (citems[(size_t)cptr[ci + 1] - 1] < r0 || citems[(size_t)cptr[ci]] >= r1) continue;
      double K[nv][nv], x0[3], h;
      cell_K(ci, K, x0, h);
      int32_t dofs[nv];
      const ConstraintLine *line[nv];
      cell_lines(ci, dofs, line);
      for (int i = 0; i < nv; ++i) {
        if (line[i]) add(dofs[i], dofs[i], std::fabs(K[i][i]));
        for (int j = 0; j < nv; ++j) {
          if (!line[i] && !line[j]) { add(dofs[i], dofs[j], K[i][j]); continue; }
          if (line[i] && line[i]->entries.empty()) continue;
          if (line[j] && line[j]->entries.empty()) continue;
          if (line[i] && line[j]) {
            for (auto &ri : line[i]->entries)
              for (auto &rj : line[j]->entries) add(ri.first, rj.first, ri.second * rj.second * K[i][j]);
          } else if (line[i]) {
            for (auto &ri : line[i]->entries) add(ri.first, dofs[j], ri.second * K[i][j]);
          } else {
            for (auto &rj : line[j]->entries) add(dofs[i], rj.first, rj.second * K[i][j]);
          }
        }
      }
    }
  }
  sublap("assemble: matrix pass");
  for (size_t ci = 0; ci < active_cells.size(); ++ci) {
    double K[nv][nv], x0[3], h;
    int32_t dofs[nv];
    const ConstraintLine *line[nv];
    cell_lines(ci, dofs, line);
    bool any_inhom = false;
    for (int a = 0; a < nv; ++a) any_inhom = any_inhom || (line[a] && line[a]->inhomogeneity != 0.0);
    if (!rhs_dev || any_inhom) cell_K(ci, K, x0, h);
    else { h = triangulation.cell_size(active_cells[ci].level); }
    double F[nv];
    for (int i = 0; i < nv; ++i) F[i] = 0;
    const double jxw = std::pow(h, dim);
    if (!rhs_dev && !lammpsinput) triangulation.cell_origin(active_cells[ci].level, triangulation.levels[(size_t)active_cells[ci].level][(size_t)active_cells[ci].index], x0);
    for (size_t q = 0; q < q_rhs.p.size() && !rhs_dev; ++q) {
      double dens;
      if (lammpsinput) dens = density_values_for_each_cell[ci][q];
      else {
        double x[3] = {0, 0, 0};
        for (int d = 0; d < dim; ++d) x[d] = x0[d] + h * q_rhs.p[q][(size_t)d];
        dens = rhs_function(x);
      }
      for (int i = 0; i < nv; ++i) F[i] += q_rhs.shape[q][(size_t)i] * dens * q_rhs.w[q] * jxw;
    }
    for (int i = 0; i < nv; ++i) {
      double Fi = F[i];
      const int32_t slot = (int32_t)(ci * nv + (size_t)i);
      for (int j = 0; j < nv && any_inhom; ++j)
        if (line[j] && line[j]->inhomogeneity != 0.0) {
          if (rhs_dev) { term_slot.push_back(slot); term_value.push_back(K[i][j] * line[j]->inhomogeneity); }
          else Fi -= K[i][j] * line[j]->inhomogeneity;
        }
      if (line[i]) {
        for (auto &ri : line[i]->entries) {
          if (rhs_dev) { plan_dof.push_back(ri.first); plan_slot.push_back(slot); plan_code.push_back((uint8_t)code_of(ri.second)); }
          else system_rhs[(size_t)ri.first] += ri.second * Fi;
        }
      } else if (rhs_dev) {
        plan_dof.push_back(dofs[i]); plan_slot.push_back(slot); plan_code.push_back(0);
      } else {
        system_rhs[(size_t)dofs[i]] += Fi;
      }
    }
  }
  sublap("assemble: rhs pass");
  if (rhs_dev) {
    // per-DoF gather lists in the order the loop above would have added (a stable counting sort by DoF)
    std::vector<int64_t> dof_ptr((size_t)n + 1, 0);
    for (int32_t d : plan_dof) dof_ptr[(size_t)d + 1]++;
    for (int64_t i = 0; i < n; ++i) dof_ptr[(size_t)i + 1] += dof_ptr[(size_t)i];
    std::vector<int32_t> entry_slot(plan_dof.size());
    std::vector<uint8_t> entry_code(plan_dof.size());
    {
      std::vector<int64_t> pos(dof_ptr.begin(), dof_ptr.end() - 1);
      for (size_t e = 0; e < plan_dof.size(); ++e) {
        const int64_t q = pos[(size_t)plan_dof[e]]++;
        entry_slot[(size_t)q] = plan_slot[e];
        entry_code[(size_t)q] = plan_code[e];
      }
    }
    std::vector<uint8_t> cell_level(active_cells.size());
    for (size_t ci = 0; ci < active_cells.size(); ++ci) cell_level[ci] = (uint8_t)active_cells[ci].level;
    double jxw_of_level[16];
    for (int l = 0; l < 16; ++l) jxw_of_level[l] = std::pow(triangulation.cell_size(l), dim);
    std::vector<double> shape(q_rhs.p.size() * nv), weight(q_rhs.p.size());
    for (size_t q = 0; q < q_rhs.p.size(); ++q) {
      weight[q] = q_rhs.w[q];
      for (int i = 0; i < nv; ++i) shape[q * nv + (size_t)i] = q_rhs.shape[q][(size_t)i];
    }
    double *d_out = nullptr;
    auto chk = [&](int rc, const char *what) { if (rc != GMG_OK) throw std::runtime_error(std::string(what) + ": " + gmg_last_error(gmg)); };
    chk(gmg_vec_alloc(gmg, n, &d_out), "gmg_vec_alloc");
    chk(gmg_rhs_assemble(gmg, (int64_t)active_cells.size(), (int)q_rhs.p.size(), dim, shape.data(), weight.data(), cell_level.data(), jxw_of_level,
                         (int64_t)term_slot.size(), term_slot.data(), term_value.data(), n, dof_ptr.data(), entry_slot.data(), entry_code.data(),
                         coef_table.data(), d_out), "gmg_rhs_assemble");
    chk(gmg_vec_download(gmg, system_rhs.data(), d_out, n), "gmg_vec_download");
    gmg_vec_free(gmg, d_out);
    sublap("assemble: rhs on device");
  }
}

template <int dim>
void LaplaceProblem<dim>::assemble_multigrid() {
  constexpr int nv = 1 << dim;
  const int L = triangulation.n_levels();
  const Quadrature<dim> q_laplace((int)par.degree + 1);
  const bool constant_coef = par.Problemtype != "Step16";
  mg_matrices.assign((size_t)L, {});
  mg_interface_matrices.assign((size_t)L, {});
  level0_on_device = decide_level0_on_device();
  for (int l = 0; l < L; ++l) {
    if (l == 0 && level0_on_device) continue;  // formed on the device at upload; assembled here only when somebody asks for it
    assemble_level(l);
  }
}

// Is level 0 the kind of operator gmg_set_level_matrix_lattice forms (and is it going to stay whole on this rank)?
template <int dim>
bool LaplaceProblem<dim>::decide_level0_on_device() const {
  if (!par.level0_matrix_on_device || !solve_on_device_requested || dim != 3 || par.Problemtype == "Step16" || par.level0_numbering != "lexicographic") return false;
  if (triangulation.n0 + 1 < 5) return false;
  if (distributed) {  // the same decision upload() takes about partitioning level 0
    const int64_t n0 = (int64_t)level_vertex_of_dof[0].size();
    const bool peer_transport = comm_id.compare(0, 8, "GMGPEER:") == 0;
    const bool part = par.partition_level0 == "always" ||
                      (par.partition_level0 != "never" && n0 - n0 / n_ranks >= (peer_transport ? kPartitionMinRowsSavedPeer : kPartitionMinRowsSaved));
    if (part) return false;
  }
  return true;
}

// the 8 x 8 cell matrix every level-0 cell adds (constant coefficient), as assemble_level forms it
template <int dim>
void LaplaceProblem<dim>::level0_cell_matrix(double *Ke) const {
  constexpr int nv = 1 << dim;
  const Quadrature<dim> q_laplace((int)par.degree + 1);
  double Kc[nv][nv];
  double x0[3] = {0, 0, 0};
  cell_matrix<dim>(q_laplace, 1.0, x0, [&](const double *) { return 1.0; }, true, Kc);
  const double s = std::pow(triangulation.cell_size(0), dim - 2);
  for (int i = 0; i < nv; ++i)
    for (int j = 0; j < nv; ++j) Ke[i * nv + j] = Kc[i][j] * s;
}

template <int dim>
void LaplaceProblem<dim>::ensure_level_matrix(int l) {
  if (l >= 0 && l < (int)mg_matrices.size() && mg_matrices[(size_t)l].n_rows == 0 && !level_vertex_of_dof[(size_t)l].empty()) assemble_level(l);
}

template <int dim>
void LaplaceProblem<dim>::assemble_level(int l) {
  constexpr int nv = 1 << dim;
  const Quadrature<dim> q_laplace((int)par.degree + 1);
  const bool constant_coef = par.Problemtype != "Step16";
  double Kc[nv][nv];
  if (constant_coef) {
    double x0[3] = {0, 0, 0};
    cell_matrix<dim>(q_laplace, 1.0, x0, [&](const double *) { return 1.0; }, true, Kc);
  }
  {
    const auto &cells = triangulation.levels[(size_t)l];
    const int64_t n = (int64_t)level_vertex_of_dof[(size_t)l].size();
    std::vector<int64_t> cptr(cells.size() + 1, 0);
    std::vector<int32_t> citems(cells.size() * nv);
    for (size_t ci = 0; ci < cells.size(); ++ci) {
      level_cell_dofs(l, (int32_t)ci, &citems[ci * nv]);
      cptr[ci + 1] = (int64_t)(ci + 1) * nv;
    }
    CSRMatrix &A = mg_matrices[(size_t)l];
    A = pattern_from_cells(n, cptr, citems);
    const auto &bnd = level_boundary[(size_t)l];
    const auto &edge = level_refinement_edge[(size_t)l];
    std::vector<std::array<int64_t, 2>> irc;
    std::vector<double> iv;
    const double h = triangulation.cell_size(l);
    for (size_t ci = 0; ci < cells.size(); ++ci) {
      double K[nv][nv];
      if (constant_coef) {
        const double s = std::pow(h, dim - 2);
        for (int i = 0; i < nv; ++i)
          for (int j = 0; j < nv; ++j) K[i][j] = Kc[i][j] * s;
      } else {
        double x0[3];
        triangulation.cell_origin(l, cells[ci], x0);
        cell_matrix<dim>(q_laplace, h, x0, [&](const double *x) { return coefficient(x); }, false, K);
      }
      const int32_t *dofs = &citems[ci * nv];
      for (int i = 0; i < nv; ++i) {
        const bool ci_ = bnd[(size_t)dofs[i]] || edge[(size_t)dofs[i]];
        if (ci_) { A.add(dofs[i], dofs[i], std::fabs(K[i][i])); continue; }
        for (int j = 0; j < nv; ++j) {
          const bool cj = bnd[(size_t)dofs[j]] || edge[(size_t)dofs[j]];
          if (!cj) A.add(dofs[i], dofs[j], K[i][j]);
        }
      }
      // interface ("edge") matrix, :892-925: i on the refinement edge, j not, neither on the boundary
      for (int i = 0; i < nv; ++i) {
        if (!edge[(size_t)dofs[i]] || bnd[(size_t)dofs[i]]) continue;
        for (int j = 0; j < nv; ++j)
          if (!edge[(size_t)dofs[j]] && !bnd[(size_t)dofs[j]]) {
            irc.push_back({dofs[i], dofs[j]});
            iv.push_back(K[i][j]);
          }
      }
    }
    mg_interface_matrices[(size_t)l] = csr_from_triplets(n, n, irc, iv, true);
  }
}

template <int dim>
void LaplaceProblem<dim>::build_prolongation(int l) {
  // MGTransferPrebuilt::build_matrices (:957-958) for one level pair: Q1 embedding per child, columns of coarse boundary DoFs
  // zeroed.  With "Transfer matrices on device" the device forms the same operator from the two levels' vertex tables
  // (gmg_build_transfer) and this host version only runs when somebody asks for the matrix (tests, the CPU oracle).
  constexpr int nv = 1 << dim;
  {
    std::vector<std::array<int64_t, 2>> rc;
    std::vector<double> v;
    const auto &cells = triangulation.levels[(size_t)l];
    for (size_t ci = 0; ci < cells.size(); ++ci) {
      if (cells[ci].first_child < 0) continue;
      int32_t pd[nv];
      level_cell_dofs(l, (int32_t)ci, pd);
      for (int a = 0; a < nv; ++a) {
        int32_t cd[nv];
        level_cell_dofs(l + 1, cells[ci].first_child + a, cd);
        for (int b = 0; b < nv; ++b)
          for (int p = 0; p < nv; ++p) {
            double w = 1;
            for (int d = 0; d < dim; ++d) {
              const double pos = 0.5 * (((a >> d) & 1) + ((b >> d) & 1));
              w *= ((p >> d) & 1) ? pos : 1.0 - pos;
            }
            if (w == 0.0 || level_boundary[(size_t)l][(size_t)pd[p]]) continue;
            rc.push_back({cd[b], pd[p]});
            v.push_back(w);
          }
      }
    }
    mg_prolongation[(size_t)l] =
        csr_from_triplets((int64_t)level_vertex_of_dof[(size_t)l + 1].size(), (int64_t)level_vertex_of_dof[(size_t)l].size(), rc, v, false);
  }
}

template <int dim>
void LaplaceProblem<dim>::ensure_prolongation(int l) {
  if (l >= 0 && l < (int)mg_prolongation.size() && mg_prolongation[(size_t)l].n_rows == 0) build_prolongation(l);
}

template <int dim>
void LaplaceProblem<dim>::build_transfer() {
  // mg_transfer.build_matrices (:957-958) + the copy_indices of PreconditionMG (active cells, not on the refinement edge)
  constexpr int nv = 1 << dim;
  const int L = triangulation.n_levels();
  mg_prolongation.assign((size_t)std::max(0, L - 1), {});
  transfer_on_device = par.transfer_on_device && solve_on_device_requested;
  for (int l = 0; l + 1 < L && !transfer_on_device; ++l) build_prolongation(l);
  copy_global.assign((size_t)L, {});
  copy_level.assign((size_t)L, {});
  // (active cells are listed by level, then index: one pass; DoFs from the cell tables, no vertex is looked up)
  std::vector<std::vector<char>> seen((size_t)L);
  for (int l = 0; l < L; ++l) seen[(size_t)l].assign(level_vertex_of_dof[(size_t)l].size(), 0);
  for (size_t ai = 0; ai < active_cells.size(); ++ai) {
    const ActiveCell &ac = active_cells[ai];
    const size_t l = (size_t)ac.level;
    for (int a = 0; a < nv; ++a) {
      const int32_t ld = level_cell_dof_table[l][(size_t)ac.index * nv + (size_t)a];
      if (seen[l][(size_t)ld] || level_refinement_edge[l][(size_t)ld]) continue;
      seen[l][(size_t)ld] = 1;
      copy_global[l].push_back(active_cell_dof_table[ai * nv + (size_t)a]);
      copy_level[l].push_back(ld);
    }
  }
}

// ======================================================================== device hand-over + solve

#define GMGC(call)                                                    \
  do {                                                                \
    const int rc_ = (call);                                           \
    if (rc_ != GMG_OK) {                                              \
      last_error = std::string(#call) + ": " + (gmg ? gmg_last_error(gmg) : "no context"); \
      return rc_;                                                     \
    }                                                                 \
  } while (0)

template <int dim>
int LaplaceProblem<dim>::ensure_context() {
  if (gmg) return GMG_OK;
  const char *dev_env = std::getenv("STEP50_DEVICE");  // one process per GPU: LOCAL_RANK
  const int rc = gmg_create(&gmg, dev_env ? std::atoi(dev_env) : 0, 1);
  if (rc != GMG_OK) { last_error = "gmg_create failed: no usable MI355X / HIP runtime"; gmg = nullptr; return rc; }
  if (distributed) GMGC(gmg_comm_init(gmg, rank, n_ranks, comm_id.data()));  // once: the id is single-use
  return GMG_OK;
}

template <int dim>
int LaplaceProblem<dim>::upload() {
  const int L = triangulation.n_levels();
  const bool had_context = gmg != nullptr && operators_uploaded;
  int rc = ensure_context();
  if (rc != GMG_OK) return rc;
  if (had_context) {  // next adaptive cycle: same context (stream, RCCL communicator), new operators
    for (double *p : {d_solution, d_rhs, d_full})
      if (p) gmg_vec_free(gmg, p);
    d_solution = d_rhs = d_full = nullptr;
    GMGC(gmg_reset(gmg, L));
  } else {
    GMGC(gmg_reset(gmg, L));  // the context may have been created early (charge densities) with 1 level
  }
  operators_uploaded = true;
  build_matrices_ms = 0.0;
  // before the level matrices: sizes the SGS schedule.  0 = one block per rank: the reference's smoother on that many ranks
  GMGC(gmg_set_ssor_blocks(gmg, par.ssor_blocks > 0 ? par.ssor_blocks : std::max(1, distributed ? n_ranks : 1)));
  const CSRMatrix &S = system_matrix;
  if (distributed) {
    // system matrix + outer-CG vectors and level 0 are row-partitioned (canonical equal chunks),
    // levels >= 1, transfers and copy indices are replicated.  A level 0 whose
    // coarse-CG iteration is shorter than the three collectives it would need stays replicated.
    const int64_t n0 = (int64_t)level_vertex_of_dof[0].size();
    // rows taken off every rank's coarse iteration against what the exchange costs on the transport in use
    const bool peer_transport = comm_id.compare(0, 8, "GMGPEER:") == 0;
    level0_partitioned = par.partition_level0 == "always" ||
                         (par.partition_level0 != "never" && n0 - n0 / n_ranks >= (peer_transport ? kPartitionMinRowsSavedPeer : kPartitionMinRowsSaved));
    GMGC(gmg_set_global_sizes(gmg, S.n_rows, level0_partitioned ? n0 : 0));
    const LocalOperator Sl = localize(S, rank, n_ranks);
    GMGC(gmg_set_system_matrix(gmg, Sl.A.n_rows, Sl.A.n_cols, Sl.A.rowptr.data(), Sl.A.col.data(), Sl.A.val.data()));
    GMGC(gmg_set_halo_plan(gmg, GMG_SYSTEM, (int)Sl.halo.neighbor_rank.size(), Sl.halo.neighbor_rank.data(),
                           Sl.halo.send_count.data(), Sl.halo.send_idx.data(), Sl.halo.recv_count.data()));
    d_begin = Sl.row_begin; d_n = Sl.A.n_rows; d_nvec = Sl.A.n_cols;
  } else {
    GMGC(gmg_set_system_matrix(gmg, S.n_rows, S.n_cols, S.rowptr.data(), S.col.data(), S.val.data()));
    d_begin = 0; d_n = d_nvec = S.n_rows;
  }
  for (int l = 0; l < L; ++l) {
    if (l == 0 && level0_on_device && !(distributed && level0_partitioned)) {
      // the lattice operator is formed on the device from its size and the cell matrix
      double Ke[64];
      level0_cell_matrix(Ke);
      const int32_t nvv[3] = {triangulation.n0 + 1, triangulation.n0 + 1, triangulation.n0 + 1};
      GMGC(gmg_set_level_matrix_lattice(gmg, 0, nvv, Ke));
    } else {
      ensure_level_matrix(l);
    }
    const CSRMatrix &A = mg_matrices[(size_t)l];
    if (l == 0 && level0_on_device && !(distributed && level0_partitioned)) {
    } else if (distributed && level0_partitioned && l == 0) {
      const LocalOperator Al = localize(A, rank, n_ranks);
      GMGC(gmg_set_level_matrix(gmg, 0, Al.A.n_rows, Al.A.n_cols, Al.A.rowptr.data(), Al.A.col.data(), Al.A.val.data()));
      GMGC(gmg_set_halo_plan(gmg, 0, (int)Al.halo.neighbor_rank.size(), Al.halo.neighbor_rank.data(), Al.halo.send_count.data(),
                             Al.halo.send_idx.data(), Al.halo.recv_count.data()));
    } else {
      GMGC(gmg_set_level_matrix(gmg, l, A.n_rows, A.n_cols, A.rowptr.data(), A.col.data(), A.val.data()));
    }
    const CSRMatrix &I = mg_interface_matrices[(size_t)l];
    if (I.nnz() > 0) GMGC(gmg_set_edge_matrix(gmg, l, I.n_rows, I.n_cols, I.rowptr.data(), I.col.data(), I.val.data()));
    GMGC(gmg_set_copy_indices(gmg, l, (int64_t)copy_global[(size_t)l].size(), copy_global[(size_t)l].data(), copy_level[(size_t)l].data()));
    if (l + 1 < L && transfer_on_device) {
      // P_l and its transpose are formed on the device from the vertex tables of the two levels
      std::vector<uint8_t> bnd(level_boundary[(size_t)l].begin(), level_boundary[(size_t)l].end());
      double ms = 0.0;
      GMGC(gmg_build_transfer(gmg, l, dim, (int64_t)level_vertex_of_dof[(size_t)l].size(), level_vertex_of_dof[(size_t)l].data(), bnd.data(),
                              (int64_t)level_vertex_of_dof[(size_t)l + 1].size(), level_vertex_of_dof[(size_t)l + 1].data(),
                              (uint64_t)1 << (kMaxLevelShift - (l + 1)), &ms));
      build_matrices_ms += ms;
    } else if (l + 1 < L) {
      const CSRMatrix &P = mg_prolongation[(size_t)l];
      GMGC(gmg_set_prolongation(gmg, l, P.n_rows, P.n_cols, P.rowptr.data(), P.col.data(), P.val.data()));
    }
  }
  const int kind = par.smoother == "Jacobi" ? GMG_SMOOTHER_JACOBI : par.smoother == "Chebyshev" ? GMG_SMOOTHER_CHEBYSHEV : GMG_SMOOTHER_SSOR;
  GMGC(gmg_set_smoother(gmg, kind, par.smoother_omega, par.smoother_steps, par.chebyshev_degree, 0.0, 0.0));
  GMGC(gmg_set_coarse(gmg, 1e-10, 1000));  // :962
  GMGC(gmg_vec_alloc(gmg, d_nvec, &d_solution));
  GMGC(gmg_vec_alloc(gmg, d_nvec, &d_rhs));
  const int64_t chunk = (S.n_rows + n_ranks - 1) / n_ranks;
  GMGC(gmg_vec_alloc(gmg, chunk * n_ranks, &d_full));
  GMGC(gmg_vec_upload(gmg, d_rhs, system_rhs.data() + d_begin, d_n));
  GMGC(gmg_comm_barrier(gmg));  // the ranks' assembly times differ by seconds; inside the solve they wait for each other in kernels
  return GMG_OK;
}

int SolverCG_solve(gmg_context *ctx, SolverControl &control, int64_t n, int64_t n_vec, double *x, const double *b,
                   const std::string &preconditioner) {
  // deal.II SolverCG<vector_t>::solve; "GMG" -> PreconditionMG::vmult, "Jacobi" -> omega 0.6 (:996-1004)
  double *g = nullptr, *d = nullptr, *h = nullptr;
  int rc = GMG_OK;
  if ((rc = gmg_vec_alloc(ctx, n_vec, &g)) || (rc = gmg_vec_alloc(ctx, n_vec, &d)) || (rc = gmg_vec_alloc(ctx, n_vec, &h))) return rc;
  auto precondition = [&](double *dst, const double *src) {
    return preconditioner == "GMG" ? gmg_precondition(ctx, dst, src) : gmg_precondition_jacobi(ctx, 0.6, dst, src);
  };
  int it = 0, zero = 0;
  double res = 0, gh = 0, alpha = 0, beta = 0, tmp = 0;
  do {
    if ((rc = gmg_vec_all_zero(ctx, x, n, &zero))) break;
    if (!zero) {
      if ((rc = gmg_spmv(ctx, GMG_SYSTEM, g, x))) break;
      gmg_vec_add(ctx, g, -1.0, b, n);
    } else {
      gmg_vec_equ(ctx, g, -1.0, b, n);
    }
    if ((rc = gmg_vec_dot(ctx, g, g, n, &tmp))) break;
    res = std::sqrt(tmp);
    control.initial_value = res;
    if (res <= control.tolerance) break;
    if ((rc = precondition(h, g))) break;
    gmg_vec_equ(ctx, d, -1.0, h, n);
    if ((rc = gmg_vec_dot(ctx, g, h, n, &gh))) break;
    for (;;) {
      ++it;
      if ((rc = gmg_spmv(ctx, GMG_SYSTEM, h, d))) break;
      if ((rc = gmg_vec_dot(ctx, d, h, n, &alpha))) break;
      alpha = gh / alpha;
      gmg_vec_add(ctx, x, alpha, d, n);
      gmg_vec_add(ctx, g, alpha, h, n);
      if ((rc = gmg_vec_dot(ctx, g, g, n, &tmp))) break;
      res = std::sqrt(tmp);
      if (res <= control.tolerance) break;
      if (it >= control.max_steps || res != res) { rc = GMG_ERR_OUTER_NOCONV; break; }
      if ((rc = precondition(h, g))) break;
      beta = gh;
      if ((rc = gmg_vec_dot(ctx, g, h, n, &gh))) break;
      beta = gh / beta;
      gmg_vec_sadd(ctx, d, beta, -1.0, h, n);
    }
  } while (0);
  control.last_step = it;
  control.last_value = res;
  gmg_synchronize(ctx);
  gmg_vec_free(ctx, g); gmg_vec_free(ctx, d); gmg_vec_free(ctx, h);
  return rc;
}

template <int dim>
int LaplaceProblem<dim>::solve_on_device(CycleReport &rep) {
  GMGC(gmg_vec_upload(gmg, d_solution, initial_guess.data() + d_begin, d_n));
  gmg_stats st0;
  gmg_stats_get(gmg, &st0);
  double l1, l2, li;
  GMGC(gmg_vec_norms(gmg, d_rhs, d_n, &l1, &l2, &li));  // system_rhs.l2_norm(), :942
  SolverControl control{500, 1e-8 * l2};
  GMGC(gmg_synchronize(gmg));
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  if (par.device_resident_outer_cg) {
    rc = gmg_cg_solve(gmg, d_solution, d_rhs, 1e-8, 500, par.PreconditionerType == "GMG" ? GMG_PRECOND_GMG : GMG_PRECOND_JACOBI,
                      &control.last_step, &control.initial_value, &control.last_value);
  } else {
    rc = SolverCG_solve(gmg, control, d_n, d_nvec, d_solution, d_rhs, par.PreconditionerType);
  }
  gmg_synchronize(gmg);
  rep.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  rep.build_matrices_ms = build_matrices_ms;
  rep.starting_value = control.initial_value;
  rep.cg_iterations = control.last_step;
  rep.convergence_value = control.last_value;
  rep.status = rc;
  gmg_stats st;
  gmg_stats_get(gmg, &st);
  rep.coarse_iterations = st.coarse_iterations - st0.coarse_iterations;
  if (rc != GMG_OK) { last_error = std::string("solve: ") + gmg_last_error(gmg); return rc; }
  GMGC(gmg_vec_norms(gmg, d_solution, d_n, &rep.sol_l1, &rep.sol_l2, &rep.sol_linf));  // :1012-1014
  GMGC(gmg_vec_allgather(gmg, (int64_t)solution.size(), d_full, d_solution));  // every rank keeps the whole solution
  GMGC(gmg_vec_download(gmg, solution.data(), d_full, (int64_t)solution.size()));
  return GMG_OK;
}

template <int dim>
int LaplaceProblem<dim>::solve() {
  // :938-1017.  The operators were handed over by upload(); this is the timed hot path.
  CycleReport &rep = reports.back();
  rep.rhs_l1 = rep.rhs_l2 = rep.rhs_linf = 0;
  for (double v : system_rhs) { rep.rhs_l1 += std::fabs(v); rep.rhs_l2 += v * v; rep.rhs_linf = std::max(rep.rhs_linf, std::fabs(v)); }
  rep.rhs_l2 = std::sqrt(rep.rhs_l2);
  rep.matrix_l1 = system_matrix.l1_norm();
  rep.matrix_linf = system_matrix.linfty_norm();
  rep.matrix_frobenius = system_matrix.frobenius_norm();
  pcout("   L1 rhs norm " + fmt("%.10e", rep.rhs_l1));
  pcout("   L2 rhs norm " + fmt("%.10e", rep.rhs_l2));
  pcout("   LInfinity rhs norm " + fmt("%.10e", rep.rhs_linf));
  pcout("   L1 Matrix norm " + fmt("%.10e", rep.matrix_l1));
  pcout("   LInfinity Matrix norm " + fmt("%.10e", rep.matrix_linf));
  pcout("   Frobenius Matrix norm " + fmt("%.10e", rep.matrix_frobenius));
  const int rc = solve_on_device(rep);
  if (rc != GMG_OK) return rc;
  pcout("   Starting value " + fmt("%.10f", rep.starting_value));
  pcout("   CG converged in " + std::to_string(rep.cg_iterations) + " iterations.");
  pcout("   Convergence value " + fmt("%.10e", rep.convergence_value));
  pcout("   L1 solution norm " + fmt("%.10e", rep.sol_l1));
  pcout("   L2 solution norm " + fmt("%.10e", rep.sol_l2));
  pcout("   LInfinity solution norm " + fmt("%.10e", rep.sol_linf));
  distribute_constraints(solution);  // :1016
  return GMG_OK;
}

template <int dim>
int LaplaceProblem<dim>::solve_again() {
  CycleReport rep = reports.back();
  const int rc = solve_on_device(rep);
  reports.back().solve_seconds = rep.solve_seconds;
  reports.back().cg_iterations = rep.cg_iterations;
  reports.back().coarse_iterations = rep.coarse_iterations;
  return rc;
}

// bench: the operators of the current cycle with another smoother ("Jacobi" | "SSOR" | "Chebyshev"; SSOR in
// `ssor_blocks` blocks = the reference's smoother on that many ranks).  The SGS schedule is built at upload.
template <int dim>
int LaplaceProblem<dim>::set_smoother(const std::string &smoother, int ssor_blocks) {
  if (smoother != "Jacobi" && smoother != "SSOR" && smoother != "Chebyshev") { last_error = "unknown smoother " + smoother; return GMG_ERR_INVALID; }
  if (!operators_uploaded) { last_error = "set_smoother: no cycle has been run"; return GMG_ERR_INVALID; }
  par.smoother = smoother;
  par.ssor_blocks = std::max(0, ssor_blocks);
  return upload();
}

template <int dim>
void LaplaceProblem<dim>::distribute_constraints(std::vector<double> &u) const {
  for (size_t i = 0; i < constraint_of_dof.size(); ++i) {
    const int32_t cl = constraint_of_dof[i];
    if (cl < 0) continue;
    const ConstraintLine &line = constraint_lines[(size_t)cl];
    double v = line.inhomogeneity;
    for (auto &e : line.entries) v += e.second * u[(size_t)e.first];
    u[i] = v;
  }
}
template <int dim>
void LaplaceProblem<dim>::set_zero_constraints(std::vector<double> &u) const {
  for (size_t i = 0; i < constraint_of_dof.size(); ++i)
    if (constraint_of_dof[i] >= 0) u[i] = 0.0;
}

// ======================================================================== post-processing

template <int dim>
double LaplaceProblem<dim>::fe_value_at(const std::vector<double> &u, const double x[3]) const {
  // GridTools::find_active_cell_around_point + FEValues::get_function_values (:1354-1363)
  int c[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) {
    c[d] = (int)std::floor((x[d] - triangulation.origin) / triangulation.h0);
    c[d] = std::min(std::max(c[d], 0), triangulation.n0 - 1);
  }
  int level = 0;
  int32_t ci = triangulation.find(0, c[0], c[1], c[2]);
  while (!triangulation.active(level, ci)) {
    const Cell &cell = triangulation.levels[(size_t)level][(size_t)ci];
    const double h = triangulation.cell_size(level);
    double x0[3];
    triangulation.cell_origin(level, cell, x0);
    int a = 0;
    for (int d = 0; d < dim; ++d)
      if (x[d] >= x0[d] + 0.5 * h) a |= 1 << d;
    ci = cell.first_child + a;
    ++level;
  }
  const Cell &cell = triangulation.levels[(size_t)level][(size_t)ci];
  const double h = triangulation.cell_size(level);
  double x0[3];
  triangulation.cell_origin(level, cell, x0);
  double t[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) t[d] = (x[d] - x0[d]) / h;
  double v = 0;
  for (int a = 0; a < (1 << dim); ++a) {
    double w = 1;
    for (int d = 0; d < dim; ++d) w *= ((a >> d) & 1) ? t[d] : 1.0 - t[d];
    v += w * u[(size_t)dof_of_vertex.at(triangulation.vertex_key(level, cell, a))];
  }
  return v;
}

template <int dim>
void LaplaceProblem<dim>::postprocess_electrostatic_energy() {
  // :1310-1420
  CycleReport &rep = reports.back();
  double analytical = 0, shortr = 0, fe = 0, self = 0;
  const bool large = number_of_atoms >= 300;  // the reference's gate (:1554): the all-pairs sums are O(N^2)
  const double rcut = par.short_range_cutoff * par.r_c;
  if (!large)
    for (unsigned i = 0; i < number_of_atoms; ++i)
      for (unsigned j = i + 1; j < number_of_atoms; ++j) {
        double r2 = 0;
        for (int d = 0; d < 3; ++d) { const double t = atom_positions[3 * i + (size_t)d] - atom_positions[3 * j + (size_t)d]; r2 += t * t; }
        analytical += charges[i] * charges[j] / std::sqrt(r2);
      }
  if (rcut <= 0.0) {  // the reference's loop (:1325-1332)
    for (unsigned i = 0; i < number_of_atoms; ++i)
      for (unsigned j = i + 1; j < number_of_atoms; ++j) {
        double r2 = 0;
        for (int d = 0; d < 3; ++d) { const double t = atom_positions[3 * i + (size_t)d] - atom_positions[3 * j + (size_t)d]; r2 += t * t; }
        const double r = std::sqrt(r2);
        shortr += charges[i] * (charges[j] * (std::erfc(r / par.r_c) / r));
      }
  } else {
    // pairs closer than rcut through bins of edge rcut: per atom i the atoms j > i of the 27 neighbouring bins, j ascending
    // -- the reference's i < j order restricted to the pairs that contribute
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (unsigned i = 0; i < number_of_atoms; ++i)
      for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], atom_positions[3 * i + (size_t)d]); hi[d] = std::max(hi[d], atom_positions[3 * i + (size_t)d]); }
    int bn[3];
    for (int d = 0; d < 3; ++d) bn[d] = std::max(1, std::min(512, (int)std::floor((hi[d] - lo[d]) / rcut) + 1));
    auto bin_of = [&](unsigned i, int b[3]) { for (int d = 0; d < 3; ++d) b[d] = std::min(bn[d] - 1, std::max(0, (int)std::floor((atom_positions[3 * i + (size_t)d] - lo[d]) / rcut))); };
    std::vector<std::vector<unsigned>> bins((size_t)bn[0] * bn[1] * bn[2]);
    for (unsigned i = 0; i < number_of_atoms; ++i) { int b[3]; bin_of(i, b); bins[(size_t)b[0] + (size_t)bn[0] * ((size_t)b[1] + (size_t)bn[1] * b[2])].push_back(i); }
    std::vector<double> part(number_of_atoms, 0.0);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t ii = 0; ii < (int64_t)number_of_atoms; ++ii) {
      const unsigned i = (unsigned)ii;
      int b[3];
      bin_of(i, b);
      std::vector<unsigned> cand;
      for (int z = std::max(0, b[2] - 1); z <= std::min(bn[2] - 1, b[2] + 1); ++z)
        for (int y = std::max(0, b[1] - 1); y <= std::min(bn[1] - 1, b[1] + 1); ++y)
          for (int x = std::max(0, b[0] - 1); x <= std::min(bn[0] - 1, b[0] + 1); ++x)
            for (unsigned j : bins[(size_t)x + (size_t)bn[0] * ((size_t)y + (size_t)bn[1] * z)])
              if (j > i) cand.push_back(j);
      std::sort(cand.begin(), cand.end());
      double acc = 0.0;
      for (unsigned j : cand) {
        double r2 = 0;
        for (int d = 0; d < 3; ++d) { const double t = atom_positions[3 * i + (size_t)d] - atom_positions[3 * j + (size_t)d]; r2 += t * t; }
        const double r = std::sqrt(r2);
        if (r < rcut) acc += charges[i] * (charges[j] * (std::erfc(r / par.r_c) / r));
      }
      part[i] = acc;
    }
    for (unsigned i = 0; i < number_of_atoms; ++i) shortr += part[i];  // fixed order: deterministic
  }
  for (unsigned i = 0; i < number_of_atoms; ++i) {
    fe += 0.5 * charges[i] * fe_value_at(solution, &atom_positions[3 * i]);
    self += charges[i] * charges[i] / (std::sqrt(M_PI) * par.r_c);
  }
  rep.has_energy = true;
  rep.energy_analytical = analytical; rep.energy_short = shortr; rep.energy_fe_long = fe; rep.energy_self = self;
  rep.energy_total = shortr + fe - self;
  rep.energy_abs_error = std::fabs(std::fabs(analytical) - std::fabs(rep.energy_total));
  if (large) pcout("\nTotal analytical electrostatic energy :   (not evaluated: all pairs of " + std::to_string(number_of_atoms) + " atoms)");
  else pcout("\nTotal analytical electrostatic energy :   " + fmt("%.10e", analytical));
  pcout("Short-ranged energy contribution :  " + fmt("%.10e", shortr));
  pcout("FE solution long-ranged energy contribution :    " + fmt("%.10e", fe));
  pcout("Self energy contribution : " + fmt("%.10e", self));
  pcout("Total electrostatic energy with split in short- and long-ranged : " + fmt("%.10e", rep.energy_total));
  pcout("Absolute Error between both energies :\t" + fmt("%.10e", rep.energy_abs_error) + "\n");
  pcout("Relative Error in total electrostatic energy :\t" + fmt("%.10e", std::fabs((std::fabs(analytical) - std::fabs(rep.energy_total)) / analytical)));
}

template <int dim>
void LaplaceProblem<dim>::postprocess_error_in_energy_norm() {
  // :1423-1461 -- || grad phi_h - grad phi_exact ||_L2 with QGauss(degree+1) per cell and the
  // analytical gradient of include/step_50.h:355-369 (GaussianCharges only; the reference
  // dereferences a null exact_solution for Step16).
  if (par.Problemtype != "GaussianCharges") return;
  constexpr int nv = 1 << dim;
  const Quadrature<dim> quad((int)par.degree + 1);
  const double inv_constant = 1.0 / (std::sqrt(M_PI) * par.r_c);
  double Error = 0.0;
#pragma omp parallel for reduction(+ : Error) schedule(dynamic, 512)
  for (int64_t ci = 0; ci < (int64_t)active_cells.size(); ++ci) {
    const ActiveCell &ac = active_cells[(size_t)ci];
    const Cell &cell = triangulation.levels[(size_t)ac.level][(size_t)ac.index];
    const double h = triangulation.cell_size(ac.level);
    double x0[3];
    triangulation.cell_origin(ac.level, cell, x0);
    int32_t dofs[nv];
    cell_dofs(ac, dofs);
    for (size_t q = 0; q < quad.p.size(); ++q) {
      double xq[3] = {0, 0, 0}, gh[3] = {0, 0, 0}, ga[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) xq[d] = x0[d] + h * quad.p[q][(size_t)d];
      for (int a = 0; a < nv; ++a)
        for (int d = 0; d < dim; ++d) gh[d] += solution[(size_t)dofs[a]] * quad.grad[q][(size_t)a][(size_t)d] / h;
      for (unsigned i = 0; i < number_of_atoms; ++i) {
        double r2 = 0, dir[3] = {0, 0, 0};
        for (int d = 0; d < dim; ++d) { dir[d] = xq[d] - atom_positions[3 * i + (size_t)d]; r2 += dir[d] * dir[d]; }
        const double r = std::sqrt(r2);
        const double f = charges[i] * (((2.0 * r * std::exp(-std::pow(r / par.r_c, 2)) * inv_constant) - std::erf(r / par.r_c)) / std::pow(r, 2));
        for (int d = 0; d < dim; ++d) ga[d] += f * dir[d] / r;
      }
      double n2 = 0;
      for (int d = 0; d < dim; ++d) n2 += (gh[d] - ga[d]) * (gh[d] - ga[d]);
      Error += n2 * quad.w[q] * std::pow(h, dim);
    }
  }
  reports.back().energy_norm_error = std::sqrt(Error);
  pcout("Error in FE solution in energy norm:  " + fmt("%.10e", std::sqrt(Error)));
}

// ======================================================================== adaptive loop

template <int dim>
int LaplaceProblem<dim>::run_cycle(unsigned int cycle, bool on_device) {
  pcout("Cycle " + std::to_string(cycle) + ":");
  // STEP50_TIMING=1: wall time of the host phases on stderr (the reference's TimerOutput sections, :1463-1560)
  static const bool timing = std::getenv("STEP50_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[step50] cycle %u %-22s %8.3f s\n", cycle, what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  densities_on_device = on_device && par.densities_on_device;
  solve_on_device_requested = on_device;
  if (cycle == 0) make_initial_grid();
  else refine_grid(cycle);
  lap("mesh");
  reports.emplace_back();
  CycleReport &rep = reports.back();
  rep.cycle = (int)cycle;
  rep.active_cells = triangulation.n_active_cells();
  pcout("   Number of active cells:       " + std::to_string(rep.active_cells));
  if (cycle == 0) {
    setup_system(cycle);
    initial_guess.assign(vertex_of_dof.size(), 0.0);
  }
  rep.dofs = (int64_t)vertex_of_dof.size();
  std::string s = "   Number of degrees of freedom: " + std::to_string(rep.dofs) + " (by level: ";
  for (int l = 0; l < triangulation.n_levels(); ++l) {
    rep.dofs_by_level.push_back((int64_t)level_vertex_of_dof[(size_t)l].size());
    s += std::to_string(rep.dofs_by_level.back()) + (l == triangulation.n_levels() - 1 ? ")" : ", ");
  }
  pcout(s);
  lap("setup_system");
  assemble_system();
  lap("assemble_system");
  if (par.PreconditionerType == "GMG") assemble_multigrid();
  lap("assemble_multigrid");
  build_transfer();
  lap("build_transfer");
  if (!on_device) return GMG_OK;
  int rc = upload();
  lap("upload");
  if (rc != GMG_OK) return rc;
  rc = solve();
  lap("solve");
  if (rc != GMG_OK) return rc;
  finish_cycle();
  lap("estimate + mark");
  return GMG_OK;
}

// The check vector of the reference's rhs test (tests_rhs_rc_variation/rc_variation.cc:110-215,
// charge_density_test): every DoF of a cell receives sum_q rho(x_q) JxW_q of that cell (no shape
// function weights), constrained DoFs receive nothing (homogeneous lines: distribute_local_to_global
// drops them).  Uses the densities of the last assembled right-hand side.
template <int dim>
std::vector<double> LaplaceProblem<dim>::total_charge_density_vector() const {
  constexpr int nv = 1 << dim;
  std::vector<double> t(vertex_of_dof.size(), 0.0);
  const_cast<LaplaceProblem<dim> *>(this)->ensure_host_densities();
  if (!lammpsinput || density_values_for_each_cell.size() != active_cells.size()) return t;
  const Quadrature<dim> q_rhs((int)(par.degree + par.quadrature_degree_rhs));
  for (size_t ci = 0; ci < active_cells.size(); ++ci) {
    const ActiveCell &ac = active_cells[ci];
    const double jxw = std::pow(triangulation.cell_size(ac.level), dim);
    double cell_sum = 0.0;
    for (size_t q = 0; q < q_rhs.p.size(); ++q) cell_sum += density_values_for_each_cell[ci][q] * (q_rhs.w[q] * jxw);
    int32_t dofs[nv];
    cell_dofs(ac, dofs);
    for (int a = 0; a < nv; ++a) {
      const int32_t cl = constraint_of_dof[(size_t)dofs[a]];
      if (cl < 0) t[(size_t)dofs[a]] += cell_sum;
      else
        for (auto &e : constraint_lines[(size_t)cl].entries) t[(size_t)e.first] += e.second * cell_sum;
    }
  }
  return t;
}

template <int dim>
void LaplaceProblem<dim>::finish_cycle() {
  estimate_error_and_mark_cells();                                               // :1552
  if (lammpsinput && (number_of_atoms < 300 || (par.energy_for_large_systems && par.short_range_cutoff > 0.0))) {
    postprocess_electrostatic_energy();     // :1554-1555
    if (number_of_atoms < 300) postprocess_error_in_energy_norm();  // :1556 (O(cells x atoms): under the reference's small-system gate only)
  }
}

// Test hook: a solution computed elsewhere (the CPU oracle in tests/) takes the place of solve();
// x is the pre-distribute vector as the solver returns it.
template <int dim>
void LaplaceProblem<dim>::set_solution(const std::vector<double> &x) {
  solution = x;
  distribute_constraints(solution);
}

template <int dim>
void LaplaceProblem<dim>::run() {
  pcout("Running with the MI355X C-ABI backend on " + std::to_string(n_ranks) + " rank(s)...");  // :1466-1474
  pcout("Dimension:\t" + std::to_string(dim));
  if (!lammpsinput && number_of_atoms == 0) read_lammps_input_file(par.LammpsInputFile);
  for (unsigned int cycle = 0; cycle < par.number_of_adaptive_refinement_cycles; ++cycle) {
    const int rc = run_cycle(cycle, true);
    if (rc != GMG_OK) throw std::runtime_error("cycle " + std::to_string(cycle) + " failed: " + last_error);
  }
}

}  // namespace step50

#include "adaptive.inc"

namespace step50 {
template class LaplaceProblem<2>;
template class LaplaceProblem<3>;
}  // namespace step50
