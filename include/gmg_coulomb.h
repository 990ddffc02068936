/* gmg_coulomb.h -- C-ABI of libgmgcoulomb.so: the MI355X (gfx950) implementation of the
 * GMG-preconditioned CG hot path of the Step50 Poisson/Coulomb solver.
 *
 * The reference (vinayak-gholap1993/Geometric-Multigrid-preconditioners-for-long-range-
 * Coulomb-interaction) has no FFI for this path: LaplaceProblem::solve()
 * (src/step-50.cc:938-1017) wires deal.II class templates together inline.  The seams are
 * deal.II's duck-typed concepts; every entry point below names the concept
 * member / call site it stands in for.  A maintainer binds them with the 20-line adapter
 * classes shown in INTEGRATION.md and passes those to SolverCG::solve at :991.
 *
 * Conventions
 *   - every function returns int: GMG_OK or a GMG_ERR_* code; no C++ exception crosses the ABI;
 *     gmg_last_error() gives the text of the last failure on that context.
 *   - matrices are handed over as host CSR arrays (int64 rowptr, int32 col, fp64 val) and
 *     copied; they are immutable until replaced (the reference rebuilds them once per
 *     adaptive cycle, src/step-50.cc:1545-1548).  Explicitly stored zeros are kept.
 *   - vectors are device pointers (fp64) obtained from gmg_vec_alloc(); the caller owns them.
 *   - one host thread per context, one HIP stream per context; calls on one context are
 *     serialised by the caller (the reference runs single-threaded ranks, src/main.cc:8).
 *   - functions that return a scalar to the host (dot, norms, solves) block until it is
 *     available; everything else is asynchronous on the context's stream.
 *   - distributed runs: one process per GPU, rows of every operator are the locally owned
 *     range, columns index [owned | ghost] entries; see gmg_set_halo_plan().
 */
#ifndef GMG_COULOMB_H
#define GMG_COULOMB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMG_OK 0
#define GMG_ERR_INVALID 1        /* bad argument / wrong call order                         */
#define GMG_ERR_OUTER_NOCONV 2   /* outer CG hit max_it: deal.II SolverControl::NoConvergence, :942 */
#define GMG_ERR_COARSE_NOCONV 3  /* coarse CG hit 1000 its: NoConvergence from :962-967     */
#define GMG_ERR_HIP 4            /* a HIP runtime call failed                               */
#define GMG_ERR_COMM 5           /* RCCL failure / communicator not initialised             */
#define GMG_ERR_UNSUPPORTED 6

/* `which` argument of gmg_spmv & co.: a level number >= 0, or the active-mesh matrix */
#define GMG_SYSTEM (-1)

/* smoother kinds: src/step-50.cc:969 (Jacobi, commented), :970 (SSOR); Chebyshev is this
 * build's addition for BASELINE config 3 (not in the reference, parity unpinned)        */
#define GMG_SMOOTHER_JACOBI 0
#define GMG_SMOOTHER_SSOR 1
#define GMG_SMOOTHER_CHEBYSHEV 2

/* preconditioner of gmg_cg_solve: prm "Preconditioner = GMG | Jacobi", src/step-50.cc:954, :996 */
#define GMG_PRECOND_GMG 0
#define GMG_PRECOND_JACOBI 1
#define GMG_PRECOND_IDENTITY 2

typedef struct gmg_context gmg_context;

/* ---- lifetime ----------------------------------------------------------------------- */
/* Replaces the construction of the object graph in solve(), src/step-50.cc:954-989.
 * n_levels = triangulation.n_global_levels() (:709).                                      */
int gmg_create(gmg_context **ctx, int device_id, int n_levels);
int gmg_destroy(gmg_context *ctx);
/* New adaptive cycle (src/step-50.cc:1484): drops every operator, keeps stream + communicator. */
int gmg_reset(gmg_context *ctx, int n_levels);
const char *gmg_last_error(const gmg_context *ctx);
int gmg_synchronize(gmg_context *ctx);

/* ---- operators ---------------------------------------------------------------------- */
/* system_matrix (include/step_50.h:156; filled src/step-50.cc:793-795, 831).              */
int gmg_set_system_matrix(gmg_context *ctx, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                          const int32_t *col, const double *val);
/* mg_matrices[level] (include/step_50.h:168; filled src/step-50.cc:869-889, 930).         */
int gmg_set_level_matrix(gmg_context *ctx, int level, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                         const int32_t *col, const double *val);
/* mg_matrices[0] of the undivided lattice (Triangulation::subdivided_hyper_rectangle, src/step-50.cc:1526; assembled
 * :869-889) FORMED ON THE DEVICE instead of handed over as CSR: nv[0..2] vertices per direction
 * (>= 5), level DoFs numbered lexicographically (x fastest), Ke = the 8 x 8 cell matrix every cell adds (row-major, vertex a
 * = bx + 2 by + 4 bz, as deal.II orders them), all faces Dirichlet (MGConstrainedDoFs boundary indices, :704-706: the row of
 * a boundary DoF keeps sum |Ke[a][a]| on the diagonal, its couplings become stored zeros).  Values are the same bits
 * gmg_set_level_matrix would receive (a vertex's cells added in cell order).  Level 0 only; not with a row-partitioned
 * level 0.  The operator then serves y = A x (gmg_spmv) and the coarse CG.                                                */
int gmg_set_level_matrix_lattice(gmg_context *ctx, int level, const int32_t nv[3], const double Ke[64]);
/* mg_interface_matrices[level] (include/step_50.h:169; src/step-50.cc:892-925, 931); used as
 * both edge_out and edge_in, mg.set_edge_matrices(down, up) at :986.  Zeros may be pruned. */
int gmg_set_edge_matrix(gmg_context *ctx, int level, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                        const int32_t *col, const double *val);
/* MGTransferPrebuilt::build_matrices product (src/step-50.cc:957-958): P maps `level` to
 * `level+1`, n_fine x n_coarse; the library builds the transpose for restrict_and_add.     */
int gmg_set_prolongation(gmg_context *ctx, int level, int64_t n_fine, int64_t n_coarse, const int64_t *rowptr,
                         const int32_t *col, const double *val);
/* MGTransferPrebuilt copy_indices[level] (used by PreconditionMG::vmult, :988-989).        */
int gmg_set_copy_indices(gmg_context *ctx, int level, int64_t n, const int32_t *global_idx, const int32_t *level_idx);
/* mg_smoother.initialize(mg_matrices, AdditionalData(omega)); set_steps(steps), :971-973.
 * cheb_*: Chebyshev degree, eigenvalue ratio and lambda_max of D^-1 A (0 = Gershgorin).    */
int gmg_set_smoother(gmg_context *ctx, int kind, double omega, int steps, int cheb_degree, double cheb_ratio,
                     double cheb_lmax);
/* SolverControl coarse_solver_control(1000, 1e-10, false, false), :962.                    */
int gmg_set_coarse(gmg_context *ctx, double abs_tol, int max_it);

/* ---- vector_t (LA::MPI::Vector, include/step_50.h:154) ----------------------------- */
int gmg_vec_alloc(gmg_context *ctx, int64_t n, double **dptr);
int gmg_vec_free(gmg_context *ctx, double *dptr);
int gmg_vec_upload(gmg_context *ctx, double *dst_dev, const double *src_host, int64_t n);
int gmg_vec_download(gmg_context *ctx, double *dst_host, const double *src_dev, int64_t n);
int gmg_vec_set_zero(gmg_context *ctx, double *x, int64_t n);                               /* v = 0        */
int gmg_vec_equ(gmg_context *ctx, double *y, double a, const double *x, int64_t n);         /* y.equ(a,x)   */
int gmg_vec_add(gmg_context *ctx, double *y, double a, const double *x, int64_t n);         /* y.add(a,x)   */
int gmg_vec_sadd(gmg_context *ctx, double *y, double s, double a, const double *x, int64_t n); /* y.sadd(s,a,x) */
int gmg_vec_dot(gmg_context *ctx, const double *x, const double *y, int64_t n, double *out); /* x*y (all-reduced) */
int gmg_vec_norms(gmg_context *ctx, const double *x, int64_t n, double *l1, double *l2, double *linf); /* :946-948, :1012-1014 */
int gmg_vec_all_zero(gmg_context *ctx, const double *x, int64_t n, int *out);               /* x.all_zero() */

/* ---- the concepts SolverCG / Multigrid consume ------------------------------------- */
/* matrix.vmult(dst, src): SolverCG matrix concept (:991 system_matrix, :965 coarse_matrix);
 * also mg::Matrix::vmult(level, ...) (:975).  Includes the ghost import.                  */
int gmg_spmv(gmg_context *ctx, int which, double *dst, const double *src);
/* PreconditionMG::vmult(dst, src): copy_to_mg, one V-cycle (Multigrid::cycle), copy_from_mg
 * (:980-989).  Fails with GMG_ERR_COARSE_NOCONV like the reference's exception.           */
int gmg_precondition(gmg_context *ctx, double *dst, const double *src);
/* PreconditionJacobi(omega).vmult on the system matrix (:999-1004, omega = 0.6).           */
int gmg_precondition_jacobi(gmg_context *ctx, double omega, double *dst, const double *src);
/* MGCoarseGridIterativeSolver::operator()(0, dst, src) (:965-967): unpreconditioned CG from
 * zero on mg_matrices[0], device resident; returns iteration count and final residual.    */
int gmg_coarse_solve(gmg_context *ctx, double *dst, const double *src, int *iterations, double *residual);
/* MGSmootherBase::apply (from_zero != 0) / ::smooth (from_zero == 0) on one level (:983-984). */
int gmg_smoother_step(gmg_context *ctx, int level, double *u, const double *rhs, int from_zero);
/* MGTransferBase::prolongate(level+1, dst, src) / restrict_and_add(level+1, dst, src).     */
int gmg_prolongate(gmg_context *ctx, int level, double *dst_fine, const double *src_coarse);
int gmg_restrict_and_add(gmg_context *ctx, int level, double *dst_coarse, const double *src_fine);

/* ---- optional: the whole outer solve on the device side of the ABI ------------------ */
/* solver.solve(system_matrix, solution, system_rhs, preconditioner) (:991-992 / :1003-1004)
 * with tol = rel_tol * |b|_2 (:942).  The north-star layout keeps this loop in the host C++
 * (csrc/host/laplace_problem.cc) on top of the calls above; this entry runs the same
 * operation order inside the library to avoid the per-call launch latency.               */
int gmg_cg_solve(gmg_context *ctx, double *x, const double *b, double rel_tol, int max_it, int precond,
                 int *iterations, double *starting_value, double *convergence_value);

/* ---- next row N1: Gaussian charge density at the quadrature points -------- */
/* compute_charge_densities() (src/step-50.cc:509-575) with the atom lists of
 * rhs_assembly_optimization() (:260-306) evaluated on the fly: for every cell, rho at its nq
 * quadrature points, summed over the atoms closer than `cutoff` to any vertex of the cell's
 * ROOT cell (use_lists != 0; children inherit the parent's list, :441-450) or over all atoms.
 * Host arrays in, host array out (dens[n_cells * nq], incl. the factor 4 pi of :522).       */
/* dens == NULL keeps the densities on the device for gmg_rhs_assemble (gmg_get_charge_density copies them out on demand).   */
int gmg_charge_density(gmg_context *ctx, int64_t n_cells, const double *cell_lo, const double *cell_h,
                       const double *root_lo, double root_h, int64_t n_atoms, const double *atom_xyz,
                       const double *atom_q, double r_c, double cutoff, int use_lists, int nq,
                       const double *quadrature_points, double *dens);
/* the densities gmg_charge_density kept on the device (dens == NULL), copied out: [n_cells * nq]                        */
int gmg_get_charge_density(gmg_context *ctx, int64_t n_cells, int nq, double *dens);
/* The right-hand side of assemble_system (src/step-50.cc:813-828) from the densities the preceding
 * gmg_charge_density(..., dens = NULL) left on the device -- they never cross PCIe:
 *   per cell   F_i = sum_q shape[q][i] * rho_q * weight[q] * jxw_of_level[level]      (:813-820, the reference's operand order)
 *   Dirichlet  F[term_slot[t]] -= term_value[t], t ascending   (term_value = K_ij g_j, :825-828; slots ascending in the list)
 *   per DoF d  rhs[d] = sum over e in [dof_ptr[d], dof_ptr[d+1]) of (entry_coef[e] == 0 ? F[slot] : coef_table[code] * F[slot])
 * with slot = cell * 2^dim + vertex, the entries of a DoF in the order the reference's cell loop adds them
 * (distribute_local_to_global: hanging-node rows contribute to their masters with the constraint weight).  Every output
 * value is one sequential sum: deterministic, no atomics.  shape: [nq][2^dim]; cell_level: [n_cells]; rhs: device vector.  */
int gmg_rhs_assemble(gmg_context *ctx, int64_t n_cells, int nq, int dim, const double *shape, const double *weight,
                     const uint8_t *cell_level, const double *jxw_of_level /* [16] */, int64_t n_terms, const int32_t *term_slot,
                     const double *term_value, int64_t n_dofs, const int64_t *dof_ptr, const int32_t *entry_slot,
                     const uint8_t *entry_coef, const double *coef_table /* [256] */, double *rhs);

/* ---- distributed (one process per GPU, RCCL over xGMI) ------------------------------ */
#define GMG_UNIQUE_ID_BYTES 128
int gmg_comm_unique_id(void *out_id);                       /* rank 0, then broadcast by the host.  Default: an RCCL id.  With
                                                             * GMG_COMM_TRANSPORT=peer in the environment the id names the
                                                             * peer-to-peer transport for the GPUs of one node (mailboxes mapped
                                                             * with hipIpc, messages written by kernels straight into the peer's
                                                             * HBM, flags instead of collectives); it also works between
                                                             * processes sharing one GPU (tests)                              */
int gmg_comm_init(gmg_context *ctx, int rank, int n_ranks, const void *id);
/* The ranks meet on the host (call it when the operators are set, before the solve: the ranks' host-side setup times
 * differ by seconds, the kernels of the peer transport wait for each other by polling).  No-op without a communicator. */
int gmg_comm_barrier(gmg_context *ctx);
/* What the communicator turned out to be (bench.py records it): out[0] ranks, out[1] transport (0 none, 1 RCCL,
 * 2 peer-to-peer stores), out[2] peer mailbox in fine-grained memory (0 / 1), out[3] shared direction ring of the
 * coarse CG (-1 not allocated, 0 plain hipMalloc, 1 fine-grained), out[4] distinct GPUs under the ranks, out[5]
 * level 0 row-partitioned (0 / 1), out[6..7] reserved (0).  Stands where the reference would print
 * Utilities::MPI::n_mpi_processes (src/step-50.cc:120-122).                                                      */
int gmg_comm_info(gmg_context *ctx, int64_t out[8]);
/* Distributed layout: the system matrix / outer-CG vectors and level 0 (matrix,
 * coarse CG) are row-partitioned in equal chunks -- gmg_partition_range gives the canonical
 * owned range, mirroring locally_owned_dofs() of the reference (:656-657) -- while levels >= 1,
 * the transfers and the copy-index lists are passed whole (global numbering) on every rank.
 * n_level0_global = 0 keeps level 0 replicated as well (matrix passed whole, every rank runs
 * the single-GPU coarse CG): for a level 0 too small to pay three collectives per iteration.
 * Call order: gmg_comm_init, gmg_set_global_sizes, then the gmg_set_* of the operators.   */
int gmg_set_global_sizes(gmg_context *ctx, int64_t n_system_global, int64_t n_level0_global);
int gmg_partition_range(int64_t n_global, int rank, int n_ranks, int64_t *begin, int64_t *end);
/* dst_full (n_global entries, padded to n_ranks * ceil(n_global / n_ranks)) <- every rank's
 * owned slice; the reference does this with a ghosted vector assignment (:1026-1028).     */
int gmg_vec_allgather(gmg_context *ctx, int64_t n_global, double *dst_full, const double *src_local);
/* Epetra_Import plan of one operator: for each neighbour the owned local rows to send and
 * the number of ghost values received; ghosts are stored behind the owned entries in
 * neighbour order.  `which` as in gmg_spmv.                                              */
int gmg_set_halo_plan(gmg_context *ctx, int which, int n_neighbors, const int32_t *neighbor_rank,
                      const int32_t *send_count, const int32_t *send_idx, const int32_t *recv_count);

/* mg_transfer.build_matrices(mg_dof_handler) (src/step-50.cc:957-958, inside the Solve timer) ON THE DEVICE: P_level (level
 * -> level + 1, Q1 embedding, columns of coarse boundary DoFs dropped) and its transpose from the two levels' DoF tables
 * instead of a host-built CSR (gmg_set_prolongation).  coarse_vertex[i] / fine_vertex[i]: the vertex of level DoF i as
 * x | y << 21 | z << 42 in units of a lattice on which the fine level's vertices are `fine_spacing` apart (a power of two)
 * and the coarse level's 2 * fine_spacing; coarse_boundary[i] = 1 for DoFs on the domain boundary (MGConstrainedDoFs,
 * :704-706).  Rows come out in ascending column order, the transpose in ascending source-row order (the order of the
 * reference's sequential Tvmult): identical to what gmg_set_prolongation receives / derives.  build_ms (may be NULL)
 * returns the device time of the build.                                                                              */
int gmg_build_transfer(gmg_context *ctx, int level, int dim, int64_t n_coarse, const uint64_t *coarse_vertex,
                       const uint8_t *coarse_boundary, int64_t n_fine, const uint64_t *fine_vertex, uint64_t fine_spacing,
                       double *build_ms);
/* The CSR of P_level (transposed = 0) or of its transpose as the device holds it (for tests of gmg_build_transfer against
 * the host-built operator): with rowptr == NULL only the sizes are returned.                                          */
int gmg_get_transfer(gmg_context *ctx, int level, int transposed, int64_t *n_rows, int64_t *n_cols, int64_t *nnz,
                     int64_t *rowptr, int32_t *col, double *val);

/* ---- measurement -------------------------------------------------------------------- */
typedef struct gmg_stats {
  int64_t coarse_solves;        /* calls of the coarse solver since the last reset           */
  int64_t coarse_iterations;    /* inner CG iterations summed over those calls               */
  int64_t vcycles;
  int64_t spmv0_samples;        /* live level-0 SpMV launches bracketed by HIP events        */
  double spmv0_ms_total;        /* summed event time of those launches                       */
  int64_t spmv0_rows, spmv0_nnz; /* shape of the level-0 operator (for algorithmic bytes)    */
  int64_t cgupd_samples;
  double cgupd_ms_total;
  int64_t coarse_variant;       /* 1 = fused (SpMV + direction update), 2 = unfused, of the last solve */
  int64_t spmv0_layout;         /* 0 = CSR row windows; else 1 (SELL-64) + 2 (8-bit value codes) + 4 (16-bit column offsets) + 8 (pattern-run kernel) + 16 (row classes) + 32 (plane-by-plane lattice kernel) + 64 (formed on the device: no CSR behind it) */
  int64_t spmv0_matrix_bytes;   /* bytes of the level-0 operator one SpMV streams in its device layout */
  int64_t spmv0_pattern_slices, spmv0_slices; /* slices served by a column pattern / all slices */
  int64_t coarse_enqueued;      /* coarse iterations enqueued, incl. those that returned at once after convergence */
  int64_t spmv0_noop_samples;   /* sampled level-0 launches that returned at once (after convergence): ...    */
  double spmv0_noop_ms_total;   /* ... their summed event time                                                */
  int64_t sgs_samples;          /* SSOR sweep launches (levels >= 1) bracketed by HIP events while profiling is on   */
  double sgs_ms_total;          /* their summed event time                                                           */
  int64_t sgs_substeps;         /* dependent steps those launches walked (the sweep is latency bound)                */
  int64_t sgs_stream_bytes;     /* record bytes they streamed                                                        */
  int64_t sgs_launches;         /* all SSOR sweep launches while profiling was on (every sample_every-th one is timed)  */
  double build_matrices_ms;     /* device time of gmg_build_transfer calls since the last reset (the reference counts build_matrices in its Solve timer, :941-958) */
} gmg_stats;
int gmg_stats_reset(gmg_context *ctx);
int gmg_stats_get(gmg_context *ctx, gmg_stats *out);
/* attach HIP start / stop events to every `sample_every`-th level-0 SpMV launch and every `sample_every`-th SSOR
 * sweep launch (0 = off); when the event pool is full the sampling stops, nothing ever synchronises.        */
int gmg_set_profiling(gmg_context *ctx, int sample_every);
/* streaming-read and copy bandwidth of this device (GB/s) on n_bytes per array: the measured
 * ceiling bench.py prints beside the 8 TB/s spec peak.                                    */
int gmg_calibrate_hbm(gmg_context *ctx, int64_t n_bytes, int reps, double *read_gbps, double *copy_gbps);
/* tuning knobs: coarse_chunk = iterations enqueued between host convergence checks (0 = predicted
 * from the previous solve); cg_variant = 0 auto by size, 1 fused two-kernel iteration (SpMV also
 * forms d = beta d - g), 2 three-kernel iteration.                                          */
int gmg_set_tuning(gmg_context *ctx, int coarse_chunk, int cg_variant);
/* Diagnostic / measurement options by name (defaults are the production paths).  Keys: host_threads,
 * debug_upload, disable_sell, disable_patterns, disable_compression, disable_sellp, disable_rowclass, sell_grid, sellp_cost,
 * cg_variant, coarse_chunk, sgs_y_slots (doubles of LDS the SSOR sweep may use for y: small values force
 * several LDS ranges), sgs_disable_wave (SSOR through the generic CSR sweep), sgs_disable_phase (the one-wave sweep),
 * sgs_phase_profile (cycle counters of the four-wave sweep: same results, one rank only), sgs_profile (instrumented
 * one-wave sweep; its wrong-result timing modes exist only in a -DGMG_EXPERIMENTS build, tools/build_experiments.sh),
 * sgs_lds_bytes_override (tests: a value over the CU's 160 KB makes the sweep's launch fail -> GMG_ERR_HIP).  Options that shape a device
 * layout take effect at the next gmg_set_*_matrix.  The same keys are read once from the environment
 * variable GMG_OPTIONS="key=value,..." at gmg_create (for profiling scripts around bench.py).        */
int gmg_set_option(gmg_context *ctx, const char *key, double value);
/* SSOR blocks B: 1 = exact sequential sweep (the reference on one rank); B > 1 = what the
 * reference's smoother does on B MPI ranks: symmetric Gauss-Seidel inside each block of
 * consecutive rows, couplings between blocks dropped (Ifpack's rank-local matrix).  Call before
 * the level matrices are set.                                                              */
int gmg_set_ssor_blocks(gmg_context *ctx, int n_blocks);

#ifdef __cplusplus
}
#endif
#endif /* GMG_COULOMB_H */
