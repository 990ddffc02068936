// gmg_coulomb.hip -- implementation of the C-ABI in include/gmg_coulomb.h for MI355X (gfx950).
//
// Host-side orchestration of the kernels in gmg_device.hpp: operator upload (tiling for the
// LDS row-window SpMV, explicit transposes, inverse diagonals, SGS level schedule), the
// V-cycle of deal.II's Multigrid::level_v_step, the device-resident coarse CG, and the
// vector_t primitives the host-side outer CG calls.  Reference call sites are cited in the
// header next to each entry point; operation order follows /root/reference/src/step-50.cc
// :938-1017 as restated in SURVEY.md 3.2.
#include "../../include/gmg_coulomb.h"
#include "gmg_device.hpp"
#include "gmg_sgs.hpp"
#include "gmg_sgs_phase.hpp"
#include "gmg_sgs_dep.hpp"
#include "gmg_sgs_reg.hpp"
#ifdef GMG_EXPERIMENTS
#include "gmg_sgs_chain.hpp"  // hand-over through an LDS word instead of s_barrier: measured slower, kept as an experiment
#endif
#include "gmg_lattice.hpp"
#include "gmg_transfer.hpp"
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gmg_comm.hpp"

using namespace gmg;

namespace {

struct DevCSR {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int32_t *rowptr = nullptr, *col = nullptr;
  double *val = nullptr;
  int32_t *tile_row = nullptr;
  int n_tiles = 0, tiles_per_xcd = 0, grid = 0;
  bool valid = false;
  HaloPlan halo;
  // SELL-64 copy (regular-width operators only), see gmg_device.hpp
  bool sell = false;
  int32_t *slice_ptr = nullptr;  // in quads
  int32_t *slice_base = nullptr; // COL16: smallest column per slice
  void *sell_vals = nullptr;     // uchar4 codes (VAL8) or double2 pairs
  void *sell_cols = nullptr;     // ushort4 offsets (COL16) or int4 columns
  double *sell_dict = nullptr;   // VAL8: 256 doubles
  int32_t *sell_spat = nullptr, *sell_pat = nullptr;  // column-pattern ids per slice / pattern table
  int sellp_pid = -1, sellp_centre[9] = {};  // the nine-runs-of-three pattern served by spmv_sellp_kernel
  int32_t *sellp_wave_ptr = nullptr;          // slice range of every wave of spmv_sellp_kernel
  int4 *sellp_wave_rr = nullptr;              // strided fast waves: {first slice, stride, pairs, one more slice or -1}
  bool use_sellp = false;
  bool rowclass = false;  // run-pattern slices take their coefficients from a per-row class table (N4)
  uint8_t *sellp_rowcls = nullptr;
  double *sellp_ctab = nullptr;
  int n_classes = 0;
  int n_patterns = 0, n_pattern_slices = 0;
  bool val8 = false, col16 = false;
  int n_slices = 0, sell_grid = 0;
  int64_t sell_quads = 0;
  // lattice interior walked plane by plane (gmg_lattice.hpp): rows [lat_R0, lat_R1) by class table, the other slices from the SELL streams
  bool lattice = false;
  uint8_t *lat_rowcls = nullptr;
  double *lat_ctab = nullptr;
  int32_t *lat_gen = nullptr;
  bool lat_only = false;       // made by gmg_set_level_matrix_lattice: class table + row classes, no CSR / SELL copy
  int lat_W = 0;               // rows of the window that repeats with the plane stride (== lat_nxy: one contiguous interior)
  int64_t lat_fast_rows = 0;
  int lat_classes = 0, lat_nx = 0, lat_nxy = 0, lat_R0 = 0, lat_R1 = 0, lat_C = 0, lat_K = 0, lat_S = 0, lat_fast_blocks = 0, lat_n_gen = 0, lat_grid = 0;
  int64_t lat_gen_bytes = 0;  // stream bytes of the slices outside the interior
};

struct SgsPlan {
  // generic level-scheduled sweep straight from the CSR copy (fallback: rows with unsorted columns)
  int32_t *stage_ptr = nullptr, *stage_rows = nullptr, *block_row = nullptr, *block_stage = nullptr;
  int n_blocks = 0;
  int n_stages_max = 0;
  // wavefront sweep with y in LDS (gmg_sgs.hpp)
  bool wave = false;
  SwRange *w_ranges = nullptr;
  int32_t *w_block_rng = nullptr, *w_ws_ci = nullptr, *w_ci_row = nullptr, *w_row_ci = nullptr, *w_rpos_f = nullptr, *w_rpos_b = nullptr;
  char *w_stream = nullptr;
  double *w_ycur = nullptr, *w_iso_diag = nullptr, *w_iso_invd = nullptr;
  int w_y_slots = 0, w_lds_bytes = 0, w_n_ranges = 0;
  int64_t w_n_coupled = 0, w_stream_bytes = 0, w_steps = 0, w_stages = 0;
  std::vector<int32_t> host_block_row;  // n_blocks + 1 (several ranks: who sweeps which rows)
  double *w_stage = nullptr;            // staging of the all-gather of the swept pieces
  int64_t w_stage_len = 0;
  // four-wave variant (gmg_sgs_phase.hpp): same lists, its own ranges and record stream
  bool phased = false;
  bool dep = false;  // records laid out for gmg_sgs_dep.hpp (field-major, late = updated within the last three steps)
  bool reg = false;  // records laid out for gmg_sgs_reg.hpp (field-major, loaded straight into registers; no staging regions in LDS)
  PhRange *p_ranges = nullptr;
  uint4 *p_blk_tab = nullptr;
  std::vector<PhRange> host_pranges;
};

struct Level {
  DevCSR A, I, It, P, Pt;  // P: this level -> next finer one; Pt its transpose
  bool has_I = false, has_P = false;
  int64_t n = 0;       // owned rows
  int64_t n_vec = 0;   // owned + ghost entries of a level vector
  int64_t n_copy = 0;
  int32_t *copy_g = nullptr, *copy_l = nullptr;
  double *sol = nullptr, *def = nullptr, *t = nullptr, *w1 = nullptr, *w2 = nullptr, *w3 = nullptr;
  // distributed runs keep levels >= 1 replicated and level 0 row-partitioned: the V-cycle sees
  // level 0 through these full-length replicas (aliases of sol/def on a single GPU)
  double *sol_full = nullptr, *def_full = nullptr;
  double *invd = nullptr;
  double cheb_lmax = 0.0;
  SgsPlan sgs;
};

}  // namespace

struct gmg_context {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_levels = 0;
  std::vector<Level> lv;
  DevCSR S;
  double *S_invd = nullptr;
  double *S_tmp = nullptr;  // system-sized scratch with ghost tail (halo import of src)
  // smoother / coarse parameters (src/step-50.cc:962, 970-973)
  int smoother = GMG_SMOOTHER_SSOR;
  double omega = 0.5;
  int steps = 2;
  int cheb_degree = 2;
  double cheb_ratio = 30.0, cheb_lmax_user = 0.0;
  double coarse_tol = 1e-10;
  int coarse_maxit = 1000;
  // coarse CG work space (level-0 sized)
  int64_t cg_n = 0;
  double *cg_g = nullptr, *cg_d0 = nullptr, *cg_d1 = nullptr, *cg_h = nullptr;
  double *cg_ring[kXRing] = {};  // direction vectors of the last kXRing iterations (three-kernel coarse CG), allocated on first use
  int64_t cg_ring_len = 0;
  // peer transport, partitioned level 0: the ring is one shared allocation the neighbours write their halo entries into
  char *ring_shared[kPeerMaxRanks] = {};
  int64_t ring_stride = 0;                      // doubles between my ring slots
  int64_t peer_meta[kPeerMaxRanks][4 + kPeerMaxRanks] = {};  // per rank: ring stride, owned rows, ghost offset of every source
  unsigned long long peer_tag0 = 1;             // tags of the next coarse solve start here
  unsigned int *peer_push_cnt = nullptr;
  CGState *st = nullptr;       // device
  CGState *st_host = nullptr;  // pinned, 2 slots (the chunk being checked / the speculative one)
  CGState st_final{};
  hipEvent_t ev_chunk[2] = {nullptr, nullptr};
  double *part_a = nullptr, *part_b = nullptr;  // reduction partials (4 * kMaxPartials each)
  double *scal_dev = nullptr;                   // 8 doubles
  double *scal_host = nullptr;                  // pinned, 8 doubles
  int *sgs_abort = nullptr;                     // pinned, device-visible: set by the SSOR sweep if its wave protocol broke
  // tuning / measurement
  int coarse_chunk = 0;
  // diagnostic options (gmg_set_option / GMG_OPTIONS); the defaults are the fast paths
  int sgs_y_slots = 0;      // 0 = kSwYSlots; tests shrink it to force several LDS ranges
  bool sgs_disable_wave = false, sgs_disable_phase = false, sgs_chain = false, sgs_dep = false, sgs_reg = false, debug_upload = false, sgs_profile = false;
  int sgs_profile_mode = 0;
  int sgs_phase_chunk = 0;         // steps per chunk of one shape (0: default)
  int sgs_pf_lead_kb = 0;          // diagnostics: minimum lead of the prefetch wave in KB (0: default; < 0: no prefetch)
  bool sgs_phase_nocascade = false;  // every step gathers all T1 slots of its shape (comparison)
  bool sgs_phase_nosplit = false;  // the whole tail is gathered in the dependent phase (comparison / tests)
  int sgs_phase_profile = 0;  // > 0: print cycles per step of every range of the four-wave sweep
  int sgs_groups = 0;  // 0: chosen per sweep direction; 1..4: forced (experiments)
  int sgs_lds_bytes_override = 0;  // tests: request this much dynamic LDS for the SSOR sweep (over the limit: the launch is rejected)
  bool disable_sell = false, disable_patterns = false, disable_compression = false, disable_sellp = false, disable_rowclass = false, disable_lattice = false;
  int lattice_segments = 0;  // segments per XCD slab of the lattice kernel (0: by size)
  int lattice_max_blocks = 0;  // cap on the marching workgroups (0: only the reduction partials cap them)
  int sell_grid = 0;        // workgroups of the SELL kernels (0 = by size)
  int sellp_rr = 0;         // round-robin slice order of the fast waves: 0 by size, 1 on, 2 off
  double sellp_cost = 4.0;  // cost of a streamed slice in pattern slices (wave balancing of spmv_sellp_kernel)
  int ssor_blocks = 1;  // 1 = exact sequential SGS; B > 1 = block Jacobi of SGS (the reference on B ranks)
  int cg_variant = 0;  // 0 auto, 1 fused 2-kernel iteration, 2 unfused 3-kernel iteration
  int last_coarse_iters = 0;
  int prof_every = 0;
  std::vector<hipEvent_t> ev_a, ev_b;  // sampled level-0 SpMV launches
  std::vector<hipEvent_t> ev_c, ev_d;  // sampled update-kernel launches
  std::vector<hipEvent_t> ev_e, ev_f;  // SSOR sweep launches
  bool launch_refused = false;  // launch_op met an operator / mode combination it has no kernel for (reported by launch_status)
  int ev_used = 0, ev2_used = 0, ev3_used = 0;
  long long sgs_launch_no = 0;
  hipEvent_t timed_start = nullptr, timed_stop = nullptr;  // next launch carries these as its dispatch start / stop events
  gmg_stats stats{};
  double *dens_dev = nullptr;  // charge densities kept on the device (gmg_charge_density with dens == NULL): [cells][nq]
  int64_t dens_cells = 0;
  int dens_nq = 0;
  Comm comm;
  bool dist = false;             // communicator initialised: level 0 + system rows are partitioned
  int64_t sys_global = 0, l0_global = 0;  // l0_global == 0 on a communicator: level 0 is replicated, only the outer CG is partitioned
  double *sys_full_a = nullptr, *sys_full_b = nullptr;  // replicated src / dst of the V-cycle
  std::string err;
};

namespace {

#define HIPC(call)                                                                                   \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                  \
      return GMG_ERR_HIP;                                                                            \
    }                                                                                                \
  } while (0)

#define CHK(call)                 \
  do {                            \
    int rc_ = (call);             \
    if (rc_ != GMG_OK) return rc_; \
  } while (0)

// a rejected launch (bad grid, too much dynamic LDS, ...) leaves no trace but hipGetLastError: every entry point that
// enqueued kernels ends with this, so the failure surfaces as GMG_ERR_HIP instead of a silently missing result
#define RETURN_LAUNCHED(ctx)                                                      \
  do {                                                                            \
    const hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                       \
      (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(e_);       \
      return GMG_ERR_HIP;                                                         \
    }                                                                             \
    return GMG_OK;                                                                \
  } while (0)

int launch_status(gmg_context *ctx) {
  if (ctx->launch_refused) { ctx->launch_refused = false; return GMG_ERR_UNSUPPORTED; }
  RETURN_LAUNCHED(ctx);
}

int fail(gmg_context *ctx, int code, const char *msg) {
  ctx->err = msg;
  return code;
}

// Host wait on the context's stream by polling: hipStreamSynchronize sleeps on an interrupt and
// wakes ~50-100 us late, which shows up as GPU idle time at every convergence check of the
// coarse CG and every dot product of the outer CG.
inline hipError_t stream_wait(hipStream_t s) {
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
  }
}

constexpr int64_t kTileMaxRowsEarly = 1024;

// level 0 (matrix, coarse CG vectors) is row-partitioned over the ranks
inline bool l0_partitioned(const gmg_context *ctx) { return ctx->dist && ctx->l0_global > 0; }

inline int grid_for(int64_t n) {
  int64_t g = (n + kThreads - 1) / kThreads;
  if (g < 1) g = 1;
  if (g > kMaxPartials) g = kMaxPartials;
  return (int)g;
}

void free_csr(DevCSR &m) {
  if (m.rowptr) (void)hipFree(m.rowptr);
  if (m.col) (void)hipFree(m.col);
  if (m.val) (void)hipFree(m.val);
  if (m.tile_row) (void)hipFree(m.tile_row);
  if (m.slice_ptr) (void)hipFree(m.slice_ptr);
  if (m.sellp_wave_ptr) (void)hipFree(m.sellp_wave_ptr);
  if (m.sellp_wave_rr) (void)hipFree(m.sellp_wave_rr);
  if (m.sellp_rowcls) (void)hipFree(m.sellp_rowcls);
  if (m.sellp_ctab) (void)hipFree(m.sellp_ctab);
  if (m.lat_rowcls) (void)hipFree(m.lat_rowcls);
  if (m.lat_ctab) (void)hipFree(m.lat_ctab);
  if (m.lat_gen) (void)hipFree(m.lat_gen);
  if (m.slice_base) (void)hipFree(m.slice_base);
  if (m.sell_vals) (void)hipFree(m.sell_vals);
  if (m.sell_cols) (void)hipFree(m.sell_cols);
  if (m.sell_dict) (void)hipFree(m.sell_dict);
  if (m.sell_spat) (void)hipFree(m.sell_spat);
  if (m.sell_pat) (void)hipFree(m.sell_pat);

  free_halo(m.halo);
  m = DevCSR();
}

// Host CSR -> device CSR + LDS-window tiling.
// Host-side setup loops (layout conversion of 10^7..10^8 nonzeros) run on a few threads: f(begin, end, chunk)
int g_host_threads = 0;  // option "host_threads" (process-wide); 0 = hardware concurrency, at most 16
inline int host_threads() {
  const int t = g_host_threads > 0 ? g_host_threads : (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(16, t));
}
template <class F>
void parallel_chunks(int64_t n, F f) {
  const int nt = host_threads();
  if (n < 2048 || nt <= 1) { f((int64_t)0, n, 0); return; }
  const int64_t per = (n + nt - 1) / nt;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) {
    const int64_t b = t * per, e = std::min(n, b + per);
    if (b < e) th.emplace_back(f, b, e, t);
  }
  for (auto &x : th) x.join();
}

// Grid of spmv_lattice_kernel: m.lat_C columns x m.lat_K plane steps are cut into S segments per XCD slab -- 2.5 - 3 marching
// waves per SIMD (measured at 121^3: 2 segments 13.0 us, 3 segments 11.8 us on a pure lattice) as long as a wave keeps >= 4
// steps; the register budget admits four waves per SIMD = 1024 workgroups: what the marching waves leave free goes to the
// rows outside the interior (per-entry gathers, ~10 x the cost of an interior row: about one chunk of 64 rows per wave).
// Returns the number of workgroups for those rows.
int lattice_grid(const gmg_context *ctx, DevCSR &m, size_t n_gen) {
  const int slab = std::max(1, m.lat_K / 8);
  int S = (int)std::max<int64_t>(1, (2816 / 8 + m.lat_C / 2) / m.lat_C);
  S = std::min(S, std::max(1, slab / 4));
  if (ctx->lattice_segments > 0) S = std::min(ctx->lattice_segments, slab);
  m.lat_S = S;
  m.lat_fast_blocks = 8 * ((S * m.lat_C + 3) / 4);
  const int n_gen_blocks = n_gen == 0 ? 0 : (int)std::min<size_t>((size_t)std::max(64, 256 * kLatWavesPerSimd - m.lat_fast_blocks), (n_gen + 3) / 4);
  // every workgroup writes one reduction partial: beyond kMaxPartials (lattices above ~360^3) the marching waves take
  // several (segment, column) pairs each
  m.lat_fast_blocks = std::min(m.lat_fast_blocks, (kMaxPartials - n_gen_blocks) / 8 * 8);
  if (ctx->lattice_max_blocks >= 8) m.lat_fast_blocks = std::min(m.lat_fast_blocks, ctx->lattice_max_blocks / 8 * 8);  // (tests: the multi-pass form on a small lattice)
  return n_gen_blocks;
}

// the lattice interior only needs the rows whose columns are owned (< n_rows): any n_cols >= n_rows qualifies
inline int64_t n_cols_owned(int64_t n_rows, int64_t n_cols) { return n_cols >= n_rows ? n_rows : -1; }

int upload_csr(gmg_context *ctx, DevCSR &m, int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int32_t *col,
               const double *val, bool keep_csr = false) {
  if (n_rows < 0 || n_cols < 0 || !rowptr) return fail(ctx, GMG_ERR_INVALID, "upload_csr: bad arguments");
  const int64_t nnz = rowptr[n_rows];
  if (nnz >= (int64_t)1 << 31 || n_rows >= (int64_t)1 << 31 || n_cols >= (int64_t)1 << 31)
    return fail(ctx, GMG_ERR_UNSUPPORTED, "operator needs 64-bit device indices (nnz >= 2^31)");
  HaloPlan keep = m.halo;
  m.halo = HaloPlan();
  free_csr(m);
  m.halo = keep;
  m.n_rows = n_rows; m.n_cols = n_cols; m.nnz = nnz;
  const bool dbg_upload = ctx->debug_upload;
  auto t_phase = std::chrono::steady_clock::now();
  auto phase = [&](const char *what) {
    if (!dbg_upload) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[gmg]   upload %-14s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
    t_phase = now;
  };
  std::vector<int32_t> rp((size_t)n_rows + 1);
  for (int64_t i = 0; i <= n_rows; ++i) {
    if (i && rowptr[i] < rowptr[i - 1]) return fail(ctx, GMG_ERR_INVALID, "rowptr not monotone");
    rp[(size_t)i] = (int32_t)rowptr[i];
  }
  for (int64_t k = 0; k < nnz; ++k)
    if (col[k] < 0 || col[k] >= n_cols) return fail(ctx, GMG_ERR_INVALID, "column index out of range");
  // tiles: consecutive rows whose nonzeros fit the LDS window (window start rounded down to 4)
  std::vector<int32_t> tiles;
  tiles.push_back(0);
  int64_t r = 0;
  while (r < n_rows) {
    const int64_t ka = rowptr[r] & ~(int64_t)3;
    int64_t e = r + 1;  // a tile always holds at least one row (possibly a "long row")
    // rows are capped too: operators with long runs of empty rows (P^T onto the level-0 lattice)
    // would otherwise put 10^5 rows into one workgroup
    while (e < n_rows && rowptr[e + 1] - ka <= kTileNnz && e - r < kTileMaxRowsEarly) ++e;
    if (rowptr[e] - ka > kTileNnz) e = r + 1;
    tiles.push_back((int32_t)e);
    r = e;
  }
  m.n_tiles = (int)tiles.size() - 1;
  m.tiles_per_xcd = (m.n_tiles + 7) / 8;
  int per_xcd = std::min(kMaxPartials / 8, std::max(1, m.tiles_per_xcd));
  m.grid = 8 * per_xcd;
  const size_t pad = 8;
  HIPC(hipMalloc(&m.rowptr, sizeof(int32_t) * ((size_t)n_rows + 1)));
  HIPC(hipMalloc(&m.col, sizeof(int32_t) * ((size_t)nnz + pad)));
  HIPC(hipMalloc(&m.val, sizeof(double) * ((size_t)nnz + pad)));
  HIPC(hipMalloc(&m.tile_row, sizeof(int32_t) * tiles.size()));
  HIPC(hipMemsetAsync(m.col + nnz, 0, sizeof(int32_t) * pad, ctx->stream));
  HIPC(hipMemsetAsync(m.val + nnz, 0, sizeof(double) * pad, ctx->stream));
  HIPC(hipMemcpyAsync(m.rowptr, rp.data(), sizeof(int32_t) * rp.size(), hipMemcpyHostToDevice, ctx->stream));
  if (nnz) {
    HIPC(hipMemcpyAsync(m.col, col, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(m.val, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPC(hipMemcpyAsync(m.tile_row, tiles.data(), sizeof(int32_t) * tiles.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));  // host staging buffers die here
  m.valid = true;
  phase("csr copy");
  // SELL-64 copy when the rows are regular enough (level-0 lattice, active-mesh matrix)
  // the SELL kernels gather through 32-bit byte offsets (c << 3 into a 2 GiB buffer descriptor): columns beyond 2^28
  // stay on the CSR row-window kernel
  if (n_rows >= 1024 && !ctx->disable_sell && n_cols < ((int64_t)1 << 28)) {
    const int64_t n_slices = (n_rows + 63) / 64;
    std::vector<int32_t> sp((size_t)n_slices + 1, 0);
    // column patterns: a slice qualifies when all 64 rows exist and the union of (col - row) over
    // its rows has <= 32 members that are valid columns for every row; rows lacking a member get an
    // explicit +0.0 there (exact: x is finite), which keeps the CSR summation order
    // (a row-partitioned operator has its ghost columns behind the owned ones, n_cols > n_rows: the slices whose columns are all
    // owned still follow the lattice's patterns -- without them the distributed coarse CG ran on per-entry gathers, 99 us per
    // iteration on two ranks against 34 us on one)
    const bool allow_pat = !ctx->disable_patterns && n_cols >= n_rows;
    std::vector<int32_t> spat((size_t)n_slices, -1);
    std::vector<std::vector<int32_t>> patterns;
    std::map<std::vector<int32_t>, int> pattern_id;
    int64_t quads = 0;
    // per slice (parallel): widest row, and the sorted union of (col - row) when the slice qualifies
    std::vector<int32_t> slice_w((size_t)n_slices, 0);
    std::vector<std::vector<int32_t>> slice_delta((size_t)n_slices);
    parallel_chunks(n_slices, [&](int64_t sb, int64_t se, int) {
      for (int64_t sidx = sb; sidx < se; ++sidx) {
        int64_t w = 0;
        for (int64_t r2 = sidx * 64; r2 < std::min<int64_t>(n_rows, sidx * 64 + 64); ++r2) w = std::max(w, rowptr[r2 + 1] - rowptr[r2]);
        slice_w[(size_t)sidx] = (int32_t)w;
        if (!(allow_pat && sidx * 64 + 64 <= n_rows)) continue;
        std::vector<int32_t> delta;
        for (int64_t r2 = sidx * 64; r2 < sidx * 64 + 64 && delta.size() <= 32; ++r2)
          for (int64_t k = rowptr[r2]; k < rowptr[r2 + 1]; ++k) {
            const int32_t d = (int32_t)(col[k] - r2);
            auto it = std::lower_bound(delta.begin(), delta.end(), d);
            if (it == delta.end() || *it != d) delta.insert(it, d);
          }
        bool ok = !delta.empty() && delta.size() <= 32;
        if (ok) ok = sidx * 64 + delta.front() >= 0 && sidx * 64 + 63 + delta.back() < n_cols;
        // columns must be strictly ascending inside each row for the positions to be well defined
        for (int64_t r2 = sidx * 64; ok && r2 < sidx * 64 + 64; ++r2)
          for (int64_t k = rowptr[r2] + 1; k < rowptr[r2 + 1]; ++k) ok = ok && col[k] > col[k - 1];
        if (ok && (int64_t)delta.size() <= ((w + 3) / 4) * 4 + 4) slice_delta[(size_t)sidx] = std::move(delta);
      }
    });
    for (int64_t sidx = 0; sidx < n_slices; ++sidx) {  // pattern ids in slice order
      int64_t w = slice_w[(size_t)sidx];
      if (!slice_delta[(size_t)sidx].empty()) {
        auto ins = pattern_id.emplace(slice_delta[(size_t)sidx], (int)patterns.size());
        if (ins.second) patterns.push_back(slice_delta[(size_t)sidx]);
        spat[(size_t)sidx] = ins.first->second;
        w = (int64_t)slice_delta[(size_t)sidx].size();
      }
      quads += (w + 3) / 4;
      sp[(size_t)sidx + 1] = (int32_t)quads;
    }
    slice_delta.clear();
    if (quads * 256 <= (int64_t)(1.12 * (double)nnz) && quads * 256 < ((int64_t)1 << 31)) {
      phase("patterns");
      // --- value dictionary (bit patterns, so -0.0 / NaN payloads survive)
      const bool allow_comp = !ctx->disable_compression;
      std::vector<double> dict;
      std::unordered_map<uint64_t, int> code_of;
      bool val8 = allow_comp;
      code_of.emplace(0ull, 0);  // +0.0 is the padding value: always code 0
      dict.push_back(0.0);
      if (val8) {
        // distinct bit patterns in order of first occurrence: per chunk in parallel, merged in chunk order
        std::vector<std::vector<uint64_t>> firsts((size_t)host_threads());
        std::vector<char> overflow((size_t)host_threads(), 0);
        parallel_chunks(nnz, [&](int64_t kb, int64_t ke, int t) {
          std::unordered_map<uint64_t, int> seen;
          uint64_t last = ~0ull;
          for (int64_t k = kb; k < ke; ++k) {
            uint64_t bits;
            std::memcpy(&bits, &val[k], 8);
            if (bits == last) continue;
            last = bits;
            if (seen.emplace(bits, 0).second) {
              firsts[(size_t)t].push_back(bits);
              if (firsts[(size_t)t].size() > 256) { overflow[(size_t)t] = 1; return; }
            }
          }
        });
        for (size_t t = 0; t < firsts.size() && val8; ++t) {
          if (overflow[t]) val8 = false;
          for (uint64_t bits : firsts[t]) {
            if (!val8) break;
            if (code_of.emplace(bits, (int)dict.size()).second) {
              double v;
              std::memcpy(&v, &bits, 8);
              dict.push_back(v);
              if (dict.size() > 256) val8 = false;
            }
          }
        }
      }
      // read-only open-addressing table for the packing threads
      std::vector<uint64_t> ht_key(1024, 0);
      std::vector<int16_t> ht_code(1024, -1);
      auto ht_slot = [](uint64_t bits) { return (size_t)((bits * 0x9E3779B97F4A7C15ull) >> 54); };
      if (val8)
        for (const auto &kv : code_of) {
          size_t h = ht_slot(kv.first);
          while (ht_code[h] >= 0) h = (h + 1) & 1023;
          ht_key[h] = kv.first; ht_code[h] = (int16_t)kv.second;
        }
      auto code_lookup = [&](uint64_t bits) -> uint8_t {
        size_t h = ht_slot(bits);
        while (ht_key[h] != bits || ht_code[h] < 0) h = (h + 1) & 1023;
        return (uint8_t)ht_code[h];
      };
      phase("dictionary");
      // --- per-slice column base, 16-bit offsets if every slice spans < 65536 columns
      std::vector<int32_t> sbase((size_t)n_slices, 0);
      std::vector<char> wide((size_t)host_threads(), 0);
      parallel_chunks(n_slices, [&](int64_t sb, int64_t se, int t) {
        for (int64_t sidx = sb; sidx < se; ++sidx) {
          int64_t lo = INT64_MAX, hi = -1;
          const int64_t rend = std::min<int64_t>(n_rows, sidx * 64 + 64);
          for (int64_t r2 = sidx * 64; r2 < rend; ++r2) {
            lo = std::min(lo, r2 < n_cols ? r2 : lo);  // the padding column is the row itself
            hi = std::max(hi, r2 < n_cols ? r2 : hi);
            for (int64_t k = rowptr[r2]; k < rowptr[r2 + 1]; ++k) { lo = std::min<int64_t>(lo, col[k]); hi = std::max<int64_t>(hi, col[k]); }
          }
          if (hi < 0) { lo = hi = 0; }
          sbase[(size_t)sidx] = (int32_t)lo;
          if (hi - lo > 65535) wide[(size_t)t] = 1;
        }
      });
      bool col16 = allow_comp;
      for (char wflag : wide) col16 = col16 && !wflag;
      const size_t n_ent = (size_t)quads * 256;
      int n_pattern_slices = 0;
      for (int32_t v : spat) n_pattern_slices += v >= 0;
      std::vector<double> v2(val8 ? 0 : n_ent, 0.0);
      std::vector<uint8_t> v1(val8 ? n_ent : 0, 0);
      std::vector<int32_t> c4(col16 ? 0 : n_ent, 0);
      std::vector<uint16_t> c2(col16 ? n_ent : 0, 0);
      parallel_chunks(n_slices, [&](int64_t sb, int64_t se, int) {
      for (int64_t sidx = sb; sidx < se; ++sidx) {
        const int64_t q0 = sp[(size_t)sidx], nq = sp[(size_t)sidx + 1] - q0;
        for (int lane = 0; lane < 64; ++lane) {
          const int64_t r2 = sidx * 64 + lane;
          const int64_t k0 = r2 < n_rows ? rowptr[r2] : 0, len = r2 < n_rows ? rowptr[r2 + 1] - k0 : 0;
          const int32_t padcol = r2 < std::min(n_rows, n_cols) ? (int32_t)r2 : sbase[(size_t)sidx];
          const int pidx = spat[(size_t)sidx];
          int64_t kk = 0;  // next CSR entry of this row (pattern slices walk the pattern positions)
          for (int64_t j = 0; j < 4 * nq; ++j) {
            const int64_t qq = q0 + j / 4, e = j & 3;
            const size_t ov = (size_t)(((2 * qq + (e >> 1)) * 64 + lane) * 2 + (e & 1));  // double2-pair layout
            const size_t oc = (size_t)((qq * 64 + lane) * 4 + e);                          // 4-per-lane layout
            int32_t cj;
            double vj;
            if (pidx >= 0) {
              const auto &dl = patterns[(size_t)pidx];
              if (j < (int64_t)dl.size() && kk < len && col[k0 + kk] - r2 == dl[(size_t)j]) { cj = col[k0 + kk]; vj = val[k0 + kk]; ++kk; }
              else { cj = (int32_t)(j < (int64_t)dl.size() ? r2 + dl[(size_t)j] : r2); vj = 0.0; }
            } else {
              cj = j < len ? col[k0 + j] : padcol;
              vj = j < len ? val[k0 + j] : 0.0;
            }
            if (val8) { uint64_t bits; std::memcpy(&bits, &vj, 8); v1[oc] = code_lookup(bits); }
            else v2[ov] = vj;
            if (col16) c2[oc] = (uint16_t)(cj - sbase[(size_t)sidx]);
            else c4[oc] = cj;
          }
        }
      }
      });
      dict.resize(256, 0.0);
      HIPC(hipMalloc(&m.slice_ptr, sizeof(int32_t) * sp.size()));
      HIPC(hipMalloc(&m.slice_base, sizeof(int32_t) * sbase.size()));
      HIPC(hipMalloc(&m.sell_dict, sizeof(double) * 256));
      const size_t vbytes = val8 ? v1.size() : v2.size() * 8, cbytes = col16 ? c2.size() * 2 : c4.size() * 4;
      HIPC(hipMalloc(&m.sell_vals, vbytes + 64));
      HIPC(hipMalloc(&m.sell_cols, cbytes + 64));
      HIPC(hipMemcpyAsync(m.slice_ptr, sp.data(), sizeof(int32_t) * sp.size(), hipMemcpyHostToDevice, ctx->stream));
      HIPC(hipMemcpyAsync(m.slice_base, sbase.data(), sizeof(int32_t) * sbase.size(), hipMemcpyHostToDevice, ctx->stream));
      HIPC(hipMemcpyAsync(m.sell_dict, dict.data(), sizeof(double) * 256, hipMemcpyHostToDevice, ctx->stream));
      HIPC(hipMemcpyAsync(m.sell_vals, val8 ? (const void *)v1.data() : (const void *)v2.data(), vbytes, hipMemcpyHostToDevice, ctx->stream));
      HIPC(hipMemcpyAsync(m.sell_cols, col16 ? (const void *)c2.data() : (const void *)c4.data(), cbytes, hipMemcpyHostToDevice, ctx->stream));
      HIPC(hipStreamSynchronize(ctx->stream));
      m.val8 = val8; m.col16 = col16;
      if (n_pattern_slices > 0) {
        std::vector<int32_t> table(patterns.size() * 32, 0);
        for (size_t pi = 0; pi < patterns.size(); ++pi)
          for (size_t j = 0; j < 32; ++j) table[pi * 32 + j] = j < patterns[pi].size() ? patterns[pi][j] : 0;  // padding: own row, value +0.0
        HIPC(hipMalloc(&m.sell_spat, sizeof(int32_t) * spat.size()));
        HIPC(hipMalloc(&m.sell_pat, sizeof(int32_t) * table.size()));
        HIPC(hipMemcpyAsync(m.sell_spat, spat.data(), sizeof(int32_t) * spat.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipMemcpyAsync(m.sell_pat, table.data(), sizeof(int32_t) * table.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
        m.n_patterns = (int)patterns.size();
        m.n_pattern_slices = n_pattern_slices;
        // the most frequent pattern made of nine runs of three consecutive columns (the x-1, x, x+1
        // neighbours of a 27-point lattice row) is served by spmv_sellp_kernel
        std::vector<size_t> uses(patterns.size(), 0);
        for (size_t sl = 0; sl < n_slices; ++sl)
          if (spat[sl] >= 0) ++uses[(size_t)spat[sl]];
        size_t n_run_slices = 0;
        m.sellp_pid = -1;
        for (size_t pi = 0; pi < patterns.size(); ++pi) {
          const auto &dl = patterns[pi];
          bool ok = dl.size() == 27;
          for (size_t j = 0; ok && j < dl.size(); j += 3) ok = dl[j + 1] == dl[j] + 1 && dl[j + 2] == dl[j] + 2;
          if (!ok || uses[pi] <= n_run_slices) continue;
          n_run_slices = uses[pi];
          m.sellp_pid = (int)pi;
          for (size_t u = 0; u < 9; ++u) m.sellp_centre[u] = dl[3 * u + 1];
        }
        // a streamed slice costs ~4 pattern slices here against ~2.2 in the per-entry kernel: worth it up to ~40 % streamed slices
        m.use_sellp = val8 && !ctx->disable_sellp && n_run_slices * 10 >= n_slices * 7;
      }
      m.sell = true;
      m.n_slices = (int)n_slices;
      m.sell_quads = quads;
      phase("sell pack+copy");
      // one wave per >= 1 slice; at most kMaxPartials workgroups (one reduction partial each)
      const int per_xcd = (int)((n_slices + 7) / 8);
      m.sell_grid = 8 * std::min(kMaxPartials / 8, std::max(1, (per_xcd + 3) / 4));
      if (ctx->sell_grid > 0) m.sell_grid = std::max(8, ctx->sell_grid / 8 * 8);
      if (m.use_sellp) {
        // as many workgroups as are resident at the kernel's register budget: one round
        if (ctx->sell_grid <= 0) m.sell_grid = std::min(m.sell_grid, 256 * kSellpWaves);
        // contiguous slice ranges per wave, balanced by cost: a streamed slice (one memory round trip
        // per quad) costs about four pattern slices
        const double other_cost = ctx->sellp_cost;
        const int n_waves = m.sell_grid * 4;
        std::vector<double> cost(n_slices + 1, 0.0);
        for (size_t sl = 0; sl < n_slices; ++sl)
          cost[sl + 1] = cost[sl] + (spat[sl] == m.sellp_pid ? 1.0 : std::max(0.25, other_cost * (double)(sp[sl + 1] - sp[sl]) / 7.0));
        std::vector<int32_t> wp((size_t)n_waves + 1, 0);
        size_t sl = 0;
        int longest = 0;
        for (int wv = 1; wv <= n_waves; ++wv) {
          const double target = cost[n_slices] * (double)wv / (double)n_waves;
          while (sl < n_slices && cost[sl + 1] <= target + 1e-9) ++sl;
          if (wv == n_waves) sl = n_slices;
          wp[(size_t)wv] = (int32_t)sl;
          longest = std::max(longest, wp[(size_t)wv] - wp[(size_t)wv - 1]);
        }
        if (longest > 64) m.use_sellp = false;  // slice metadata lives in the 64 lanes
        else {
          // ---- row classes of the run-pattern slices (N4): the 27-tuple of value codes of a row; a lattice operator
          // has a few dozen of them.  Too many classes (> kSellpMaxClasses): the per-entry codes stay in use.
          if (!ctx->disable_rowclass) {
            std::vector<uint8_t> rowcls((size_t)n_rows, 0);
            std::map<std::array<uint8_t, 27>, int> cls_of;
            std::vector<double> ctab;
            bool ok = true;
            for (size_t s2 = 0; s2 < (size_t)n_slices && ok; ++s2) {
              if (spat[s2] != m.sellp_pid) continue;
              for (int lane = 0; lane < 64 && ok; ++lane) {
                std::array<uint8_t, 27> key;
                for (int j = 0; j < 27; ++j) key[(size_t)j] = v1[(size_t)((((int64_t)sp[s2] + j / 4) * 64 + lane) * 4 + (j & 3))];
                auto ins = cls_of.emplace(key, (int)cls_of.size());
                if (ins.second) {
                  if ((int)cls_of.size() > kSellpMaxClasses) { ok = false; break; }
                  for (int j = 0; j < 27; ++j) ctab.push_back(dict[key[(size_t)j]]);
                }
                rowcls[s2 * 64 + (size_t)lane] = (uint8_t)ins.first->second;
              }
            }
            if (ok && !cls_of.empty()) {
              HIPC(hipMalloc(&m.sellp_rowcls, rowcls.size()));
              HIPC(hipMalloc(&m.sellp_ctab, sizeof(double) * ctab.size()));
              HIPC(hipMemcpyAsync(m.sellp_rowcls, rowcls.data(), rowcls.size(), hipMemcpyHostToDevice, ctx->stream));
              HIPC(hipMemcpyAsync(m.sellp_ctab, ctab.data(), sizeof(double) * ctab.size(), hipMemcpyHostToDevice, ctx->stream));
              HIPC(hipStreamSynchronize(ctx->stream));
              m.rowclass = true;
              m.n_classes = (int)cls_of.size();
              // waves whose slices are all run-pattern slices need no slice metadata (flag in their wave_ptr entry)
              for (int wv = 0; wv < n_waves; ++wv) {
                bool all = wp[(size_t)wv] < wp[(size_t)wv + 1];
                for (int32_t s2 = wp[(size_t)wv]; s2 < wp[(size_t)wv + 1] && all; ++s2) all = spat[(size_t)s2] == m.sellp_pid;
                if (all) wp[(size_t)wv] |= kSellpFastWave;
              }
              // Large operators: the slices of consecutive fast waves of one XCD are dealt round-robin in pairs, so that at
              // any time the XCD's waves sit inside a window of ~2 W slices (their x working set: three lattice planes
              // instead of the XCD's whole eighth of x, which is then fetched once).  The threshold is a MEASURED
              // crossover in rows, not the L2 size: an XCD's eighth of x leaves its 4 MiB L2 at 4 Mi rows (8 x 4 MiB / 8 B),
              // and below ~3 Mi rows the contiguous order was never slower (121^3 = 1.77 M rows: x/8 = 1.8 MB per XCD stays
              // L2-resident, 17.0 us either way; 5.18 M rows: 50.7 -> 41.0 us; 201^3: PMC traffic 454 -> 197 MB).
              const bool rr = ctx->sellp_rr == 1 || (ctx->sellp_rr == 0 && n_rows > (int64_t)(3 << 20));
              if (rr) {
                const int per_xcd = n_waves / 8;
                std::vector<int4> desc((size_t)n_waves, int4{0, 0, 0, -1});
                for (int x = 0; x < 8; ++x) {
                  int a0 = x * per_xcd;
                  while (a0 < (x + 1) * per_xcd) {
                    if (!(wp[(size_t)a0] & kSellpFastWave)) { ++a0; continue; }
                    int b0 = a0;
                    while (b0 + 1 < (x + 1) * per_xcd && (wp[(size_t)b0 + 1] & kSellpFastWave)) ++b0;
                    const int wn = b0 - a0 + 1;
                    const int32_t S0 = wp[(size_t)a0] & (kSellpFastWave - 1), S1 = wp[(size_t)b0 + 1] & (kSellpFastWave - 1);
                    const int n_pairs = (S1 - S0) / 2;
                    if (wn >= 2 && n_pairs >= wn) {
                      for (int r = 0; r < wn; ++r) {
                        desc[(size_t)(a0 + r)] = int4{S0 + 2 * r, 2 * wn, (n_pairs - r + wn - 1) / wn, -1};
                        wp[(size_t)(a0 + r)] |= kSellpStrided;
                      }
                      if ((S1 - S0) & 1) desc[(size_t)b0].w = S1 - 1;
                    }
                    a0 = b0 + 1;
                  }
                }
                HIPC(hipMalloc(&m.sellp_wave_rr, sizeof(int4) * desc.size()));
                HIPC(hipMemcpyAsync(m.sellp_wave_rr, desc.data(), sizeof(int4) * desc.size(), hipMemcpyHostToDevice, ctx->stream));
              }
            }
          }
          HIPC(hipMalloc(&m.sellp_wave_ptr, sizeof(int32_t) * wp.size()));
          HIPC(hipMemcpyAsync(m.sellp_wave_ptr, wp.data(), sizeof(int32_t) * wp.size(), hipMemcpyHostToDevice, ctx->stream));
          HIPC(hipStreamSynchronize(ctx->stream));
        }
      }
      // ---- lattice interior (gmg_lattice.hpp): the nine runs are the (dz, dy) lines of an nx x ny x nz vertex lattice
      // numbered lexicographically -> the rows whose columns are all owned lattice neighbours are walked plane by plane
      // from a class table; everything else stays on the SELL streams.
      if (val8 && !ctx->disable_lattice && m.sellp_pid >= 0 && n_rows == n_cols_owned(n_rows, n_cols) && n_rows < ((int64_t)1 << 28)) {
        const int64_t nx = m.sellp_centre[5] - m.sellp_centre[4], nxy = m.sellp_centre[7] - m.sellp_centre[4];
        bool lat_ok = nx >= 3 && nxy >= 3 * nx;
        for (int u = 0; u < 9 && lat_ok; ++u) lat_ok = m.sellp_centre[u] == (u / 3 - 1) * nxy + (u % 3 - 1) * nx;
        const int64_t reach = nxy + nx + 1;
        if (lat_ok && n_rows > 4 * reach + 256) {
          // per row: are all stored columns owned lattice neighbours (27 offsets), and is every offset in range?
          std::vector<char> row_ok((size_t)n_rows, 0);
          parallel_chunks(n_rows, [&](int64_t rb2, int64_t re2, int) {
            for (int64_t r2 = rb2; r2 < re2; ++r2) {
              // (lane 0 of a unit holds the two rows in front of the unit's first row: their lowest neighbour is row - 2 - reach;
              // the highest index read is n_rows + 1: every vector has two spare entries)
              bool ok = r2 >= reach + 2 && r2 + reach < n_rows;
              for (int64_t k = rowptr[r2]; k < rowptr[r2 + 1] && ok; ++k) {
                const int64_t t = (int64_t)col[k] - r2 + reach;
                ok = col[k] < n_rows && t >= 0 && t / nxy <= 2 && (t % nxy) / nx <= 2 && (t % nxy) % nx <= 2;
              }
              row_ok[(size_t)r2] = ok ? 1 : 0;
            }
          });
          // longest run of good rows.  A lexicographic numbering gives one run over the whole interior; deal.II's
          // cell-by-cell numbering (the host side's) breaks it at every plane: the first three lines of a plane (and the
          // first three planes) are numbered in first-touch order.  So the interior is a WINDOW of W rows that repeats with
          // the plane stride: rows R0 + k nxy + [0, W), k < K.
          int64_t best_b = 0, best_e = 0, cur_b = -1;
          for (int64_t r2 = 0; r2 <= n_rows; ++r2) {
            const bool ok = r2 < n_rows && row_ok[(size_t)r2];
            if (ok && cur_b < 0) cur_b = r2;
            if (!ok && cur_b >= 0) {
              if (r2 - cur_b > best_e - best_b) { best_b = cur_b; best_e = r2; }
              cur_b = -1;
            }
          }
          int64_t R0 = best_b, R1 = best_e, Wd = std::min<int64_t>(best_e - best_b, nxy), Kp = 0;
          if (best_e - best_b >= nxy) {
            Kp = (best_e - best_b + nxy - 1) / nxy;  // the last plane step may be cut by R1
          } else if (Wd > 0) {
            std::vector<int32_t> bad_before((size_t)n_rows + 1, 0);  // prefix count of bad rows
            for (int64_t r2 = 0; r2 < n_rows; ++r2) bad_before[(size_t)r2 + 1] = bad_before[(size_t)r2] + (row_ok[(size_t)r2] ? 0 : 1);
            auto window_ok = [&](int64_t b2) { return b2 >= 0 && b2 + Wd <= n_rows && bad_before[(size_t)(b2 + Wd)] == bad_before[(size_t)b2]; };
            int64_t lo = best_b, hi = best_b;
            while (window_ok(lo - nxy)) lo -= nxy;
            while (window_ok(hi + nxy)) hi += nxy;
            R0 = lo; Kp = (hi - lo) / nxy + 1; R1 = hi + Wd;
          }
          const int64_t n_fast = Wd == nxy ? R1 - R0 : Kp * Wd;
          if (ctx->debug_upload)
            std::fprintf(stderr, "[gmg]   lattice rows: longest run [%lld, %lld) -> window of %lld rows x %lld planes from row %lld: %lld of %lld rows\n", (long long)best_b, (long long)best_e,
                         (long long)Wd, (long long)Kp, (long long)R0, (long long)n_fast, (long long)n_rows);
          auto is_fast = [&](int64_t r2) { return r2 >= R0 && r2 < R1 && (r2 - R0) % nxy < Wd; };
          // what the marching waves read: rows R0 - 2 - reach .. R1 + 1 + reach - 1 of x (lane 0 / lane 63 of a unit sit two
          // rows outside it): checked here, once, instead of trusting the row test above
          const bool reads_inside = R0 - 2 - reach >= 0 && R1 + reach <= n_rows;
          if (n_fast >= n_rows / 2 && Wd >= 2 && reads_inside) {
            // classes: the 27 value codes of a row by offset position (0 = +0.0 where the row stores nothing)
            std::vector<uint8_t> rowcls((size_t)n_rows, 0);
            const int nt = host_threads();
            std::vector<std::map<std::array<uint8_t, 27>, int>> local_cls((size_t)nt);
            std::vector<std::vector<std::array<uint8_t, 27>>> local_keys((size_t)nt);
            std::vector<std::array<int64_t, 2>> local_rng((size_t)nt, {0, 0});
            std::vector<std::vector<int32_t>> local_id((size_t)nt);
            parallel_chunks(R1 - R0, [&](int64_t cb, int64_t ce, int t) {
              auto &M = local_cls[(size_t)t];
              local_rng[(size_t)t] = {R0 + cb, R0 + ce};
              local_id[(size_t)t].resize((size_t)(ce - cb));
              for (int64_t r2 = R0 + cb; r2 < R0 + ce; ++r2) {
                std::array<uint8_t, 27> key;
                key.fill(0);
                if (!is_fast(r2)) { local_id[(size_t)t][(size_t)(r2 - R0 - cb)] = -1; continue; }
                for (int64_t k = rowptr[r2]; k < rowptr[r2 + 1]; ++k) {
                  const int64_t t2 = (int64_t)col[k] - r2 + reach;
                  const int pos = (int)((t2 / nxy) * 9 + ((t2 % nxy) / nx) * 3 + (t2 % nxy) % nx);
                  uint64_t bits;
                  std::memcpy(&bits, &val[k], 8);
                  key[(size_t)pos] = code_lookup(bits);
                }
                auto ins = M.emplace(key, (int)M.size());
                if (ins.second) local_keys[(size_t)t].push_back(key);
                local_id[(size_t)t][(size_t)(r2 - R0 - cb)] = ins.first->second;
              }
            });
            std::map<std::array<uint8_t, 27>, int> cls_of;
            std::vector<double> ctab;
            bool cls_ok = true;
            for (int t = 0; t < nt && cls_ok; ++t) {
              std::vector<int> remap(local_keys[(size_t)t].size(), 0);
              for (size_t q = 0; q < local_keys[(size_t)t].size() && cls_ok; ++q) {
                auto ins = cls_of.emplace(local_keys[(size_t)t][q], (int)cls_of.size());
                if (ins.second) {
                  if ((int)cls_of.size() > kLatMaxClasses) { cls_ok = false; m.lat_classes = -(int)cls_of.size(); break; }
                  for (int j = 0; j < 27; ++j) ctab.push_back(dict[local_keys[(size_t)t][q][(size_t)j]]);
                }
                remap[q] = ins.first->second;
              }
              if (!cls_ok) break;
              for (int64_t r2 = local_rng[(size_t)t][0]; r2 < local_rng[(size_t)t][1]; ++r2) {
                const int32_t id = local_id[(size_t)t][(size_t)(r2 - local_rng[(size_t)t][0])];
                if (id >= 0) rowcls[(size_t)r2] = (uint8_t)remap[(size_t)id];
              }
            }
            if (cls_ok && !cls_of.empty()) {
              std::vector<int32_t> gen;
              int64_t gen_bytes = 0;
              for (int64_t sl = 0; sl < n_slices; ++sl) {
                bool all_fast = true;  // a slice with any row outside the interior is served from the streams (its interior rows masked)
                for (int64_t r2 = sl * 64; r2 < std::min<int64_t>(n_rows, sl * 64 + 64) && all_fast; ++r2) all_fast = is_fast(r2);
                if (!all_fast) {
                  gen.push_back((int32_t)sl);
                  gen_bytes += (int64_t)(sp[(size_t)sl + 1] - sp[(size_t)sl]) * 256 * (1 + (spat[(size_t)sl] >= 0 ? 0 : (col16 ? 2 : 4))) + 8;
                }
              }
              m.lat_nx = (int)nx; m.lat_nxy = (int)nxy; m.lat_R0 = (int)R0; m.lat_R1 = (int)R1; m.lat_W = (int)Wd;
              m.lat_C = (int)((Wd + kLatRowsPerUnit - 1) / kLatRowsPerUnit);
              m.lat_K = (int)Kp;
              m.lat_fast_rows = n_fast;
              const int gen_blocks = lattice_grid(ctx, m, gen.size());
              m.lat_grid = m.lat_fast_blocks + gen_blocks;
              m.lat_n_gen = (int)gen.size();
              m.lat_gen_bytes = gen_bytes;
              m.lat_classes = (int)cls_of.size();
              if (m.lat_grid <= kMaxPartials) {
                HIPC(hipMalloc(&m.lat_rowcls, rowcls.size() + 64));
                HIPC(hipMalloc(&m.lat_ctab, sizeof(double) * ctab.size()));
                HIPC(hipMalloc(&m.lat_gen, sizeof(int32_t) * std::max<size_t>(gen.size(), 1)));
                HIPC(hipMemcpyAsync(m.lat_rowcls, rowcls.data(), rowcls.size(), hipMemcpyHostToDevice, ctx->stream));
                HIPC(hipMemcpyAsync(m.lat_ctab, ctab.data(), sizeof(double) * ctab.size(), hipMemcpyHostToDevice, ctx->stream));
                if (!gen.empty()) HIPC(hipMemcpyAsync(m.lat_gen, gen.data(), sizeof(int32_t) * gen.size(), hipMemcpyHostToDevice, ctx->stream));
                HIPC(hipStreamSynchronize(ctx->stream));
                m.lattice = true;
                if (ctx->debug_upload)
                  std::fprintf(stderr, "[gmg]   lattice interior: nx %d nxy %d rows [%d, %d) of %lld, %d classes, %d columns x %d steps, %d segments per XCD slab, grid %d + %d, %d slices outside\n",
                               m.lat_nx, m.lat_nxy, m.lat_R0, m.lat_R1, (long long)n_rows, m.lat_classes, m.lat_C, m.lat_K, m.lat_S, m.lat_fast_blocks, gen_blocks, m.lat_n_gen);
              }
            }
          }
        }
        if (ctx->debug_upload && !m.lattice)
          std::fprintf(stderr, "[gmg]   lattice interior not used: nx %lld nxy %lld lattice-shaped %d rows %lld reach %lld classes %d\n", (long long)nx, (long long)nxy, (int)lat_ok,
                       (long long)n_rows, (long long)reach, m.lat_classes);
        phase("lattice plan");
      }
      if (!keep_csr) {  // the CSR copy is only kept where the SGS sweeps need it (levels >= 1)
        (void)hipFree(m.col); (void)hipFree(m.val); (void)hipFree(m.tile_row);
        m.col = nullptr; m.val = nullptr; m.tile_row = nullptr;
      }
    }
  }
  phase("finish");
  if (ctx->debug_upload)
    std::fprintf(stderr, "[gmg] operator %lld x %lld nnz %lld: tiles %d grid %d sell %d val8 %d col16 %d sell_grid %d patterns %d pattern_slices %d/%d\n", (long long)n_rows,
                 (long long)n_cols, (long long)nnz, m.n_tiles, m.grid, (int)m.sell, (int)m.val8, (int)m.col16, m.sell_grid, m.n_patterns, m.n_pattern_slices, m.n_slices);
  if (ctx->debug_upload && m.rowclass) std::fprintf(stderr, "[gmg]   row classes of the run-pattern slices: %d\n", m.n_classes);
  return GMG_OK;
}

// stable transpose: row j of A^T lists the entries of column j in ascending source row,
// i.e. the order in which the sequential scatter of Tvmult / restrict_and_add adds them.
void transpose_host(int64_t n_rows, int64_t n_cols, const int64_t *rp, const int32_t *col, const double *val,
                    std::vector<int64_t> &trp, std::vector<int32_t> &tcol, std::vector<double> &tval) {
  const int64_t nnz = rp[n_rows];
  trp.assign((size_t)n_cols + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) trp[(size_t)col[k] + 1]++;
  for (int64_t j = 0; j < n_cols; ++j) trp[(size_t)j + 1] += trp[(size_t)j];
  tcol.resize((size_t)nnz);
  tval.resize((size_t)nnz);
  std::vector<int64_t> pos(trp.begin(), trp.end() - 1);
  for (int64_t i = 0; i < n_rows; ++i)
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int64_t p = pos[(size_t)col[k]]++;
      tcol[(size_t)p] = (int32_t)i;
      tval[(size_t)p] = val[k];
    }
}

int alloc_vec(gmg_context *ctx, double **p, int64_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIPC(hipMalloc(p, sizeof(double) * (size_t)std::max<int64_t>(n + 2, 2)));
  HIPC(hipMemsetAsync(*p, 0, sizeof(double) * (size_t)std::max<int64_t>(n + 2, 2), ctx->stream));
  return GMG_OK;
}

// ---- SpMV launcher ---------------------------------------------------------------------

// one launcher for both layouts; returns the grid (= number of reduction partials when CG != 0)
// Launch on the context's stream.  When the caller armed ctx->timed_start / timed_stop the events are
// attached to the dispatch itself (hipExtLaunchKernelGGL): hipEventElapsedTime then gives the kernel's own
// begin-to-end time -- what rocprofv3 --kernel-trace reports -- instead of event-record to event-record,
// which includes the launch gap.
template <typename K, typename A>
inline void launch_timed(gmg_context *ctx, K kernel, dim3 grid, dim3 block, size_t lds, const A &args) {
  if (ctx->timed_start) {
    hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, ctx->stream, ctx->timed_start, ctx->timed_stop, 0u, args);
    ctx->timed_start = ctx->timed_stop = nullptr;
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, ctx->stream, args);
  }
}

template <int MODE, int CG>
int launch_op(gmg_context *ctx, const DevCSR &m, const SpmvArgs &a) {
  if (m.sell || m.lat_only) {
    SellArgs sa{m.slice_ptr, m.slice_base, m.sell_vals, m.sell_cols, m.sell_dict, m.sell_spat, m.sell_pat, m.n_slices, (int)m.n_rows, a};
    if constexpr (MODE == kStore && (CG == 0 || CG == 2)) {
      if (m.lattice && !a.init) {
        LatArgs la{};
        la.pa.sa = sa; la.pa.col16 = m.col16 ? 1 : 0;
        la.rowcls = m.lat_rowcls; la.ctab = m.lat_ctab; la.n_classes = m.lat_classes;
        la.nx = m.lat_nx; la.nxy = m.lat_nxy; la.W = m.lat_W; la.R0 = m.lat_R0; la.R1 = m.lat_R1; la.C = m.lat_C; la.K = m.lat_K; la.S = m.lat_S;
        la.fast_blocks = m.lat_fast_blocks; la.gen_slices = m.lat_gen; la.n_gen = m.lat_n_gen;
        la.edge_mode = m.lat_only ? 1 : 0; la.n_rows = (int)m.n_rows;
        if (m.lat_S * m.lat_C > (m.lat_fast_blocks / 8) * 4) launch_timed(ctx, spmv_lattice_kernel<CG, true>, dim3(m.lat_grid), dim3(kThreads), 0, la);
        else launch_timed(ctx, spmv_lattice_kernel<CG, false>, dim3(m.lat_grid), dim3(kThreads), 0, la);
        return m.lat_grid;
      }
    }
    if (m.lat_only) {  // (an operator that exists as a class table only: plain products and the coarse CG's three-kernel iteration)
      ctx->err = "operator set by gmg_set_level_matrix_lattice: only y = A x and the three-kernel coarse CG are available";
      ctx->launch_refused = true;
      return 1;
    }
    if (m.use_sellp) {
      SellPatArgs pa{};
      pa.sa = sa; pa.wave_ptr = m.sellp_wave_ptr; pa.wave_rr = m.sellp_wave_rr; pa.pid0 = m.sellp_pid; pa.col16 = m.col16 ? 1 : 0;
      for (int u = 0; u < 9; ++u) pa.centre[u] = m.sellp_centre[u];
      pa.rowcls = m.sellp_rowcls; pa.ctab = m.sellp_ctab; pa.n_classes = m.n_classes;
      if (m.rowclass) launch_timed(ctx, spmv_sellp_kernel<MODE, CG, true>, dim3(m.sell_grid), dim3(kThreads), 0, pa);
      else launch_timed(ctx, spmv_sellp_kernel<MODE, CG, false>, dim3(m.sell_grid), dim3(kThreads), 0, pa);
      return m.sell_grid;
    }
    if (m.val8 && m.col16) launch_timed(ctx, spmv_sell_kernel<MODE, CG, true, true>, dim3(m.sell_grid), dim3(kThreads), 0, sa);
    else if (m.val8) launch_timed(ctx, spmv_sell_kernel<MODE, CG, true, false>, dim3(m.sell_grid), dim3(kThreads), 0, sa);
    else if (m.col16) launch_timed(ctx, spmv_sell_kernel<MODE, CG, false, true>, dim3(m.sell_grid), dim3(kThreads), 0, sa);
    else launch_timed(ctx, spmv_sell_kernel<MODE, CG, false, false>, dim3(m.sell_grid), dim3(kThreads), 0, sa);
    return m.sell_grid;
  }
  launch_timed(ctx, spmv_tile_kernel<MODE, CG>, dim3(m.grid), dim3(kThreads), 0, a);
  return m.grid;
}
template <int MODE>
void launch_spmv_mode(gmg_context *ctx, const DevCSR &m, const SpmvArgs &a) {
  (void)launch_op<MODE, 0>(ctx, m, a);
}

SpmvArgs base_args(const DevCSR &m, const double *x, double *y) {
  SpmvArgs a{};
  a.rowptr = m.rowptr; a.col = m.col; a.val = m.val; a.tile_row = m.tile_row;
  a.n_tiles = m.n_tiles; a.tiles_per_xcd = m.tiles_per_xcd;
  a.x = x; a.y = y;
  return a;
}

// ghost import of `x` for operator m (Epetra_Import inside every vmult); no-op on one rank
int import_ghosts(gmg_context *ctx, const DevCSR &m, double *x) {
  if (!m.halo.active) return GMG_OK;  // a replicated operator
  int rc = halo_exchange(ctx->comm, m.halo, x, m.n_rows, ctx->stream);
  if (rc) return fail(ctx, GMG_ERR_COMM, "halo exchange failed");
  return GMG_OK;
}

int spmv(gmg_context *ctx, const DevCSR &m, int mode, const double *x, double *y, const double *init = nullptr,
         const double *b = nullptr) {
  if (!m.valid) return fail(ctx, GMG_ERR_INVALID, "operator not set");
  if (m.n_rows == 0) return GMG_OK;
  SpmvArgs a = base_args(m, x, y);
  a.init = init; a.b = b;
  switch (mode) {
    case kStore: launch_spmv_mode<kStore>(ctx, m, a); break;
    case kResid: launch_spmv_mode<kResid>(ctx, m, a); break;
    case kAddTo: launch_spmv_mode<kAddTo>(ctx, m, a); break;
    default: return fail(ctx, GMG_ERR_INVALID, "bad spmv mode");
  }
  return launch_status(ctx);
}

// ---- reductions to the host -------------------------------------------------------------

int fetch_scalars(gmg_context *ctx, int n) {
  HIPC(hipMemcpyAsync(ctx->scal_host, ctx->scal_dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPC(stream_wait(ctx->stream));
  if (*ctx->sgs_abort) return fail(ctx, GMG_ERR_HIP, "SSOR sweep: a wave gave up waiting for its partner (internal protocol error)");
  if (comm_aborted(ctx->comm)) return fail(ctx, GMG_ERR_COMM, "peer transport: a rank gave up waiting for a message or an acknowledgement");
  if (ctx->comm.n_ranks > 1) {
    // sums: slots flagged by the caller; handled in the callers below
  }
  return GMG_OK;
}

int dot_host(gmg_context *ctx, const double *x, const double *y, int64_t n, double *out) {
  const int g = grid_for(n);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, x, y, n, ctx->part_a);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_a, g, 1, 0u,
                     ctx->scal_dev);
  if (ctx->comm.n_ranks > 1) {
    if (allreduce_sum(ctx->comm, ctx->scal_dev, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
  }
  CHK(launch_status(ctx));
  CHK(fetch_scalars(ctx, 1));
  *out = ctx->scal_host[0];
  return GMG_OK;
}

// ---- smoothers (A7) ---------------------------------------------------------------------

// event times of the SSOR sweep launches bracketed so far -> stats (waits for the stream)
void collect_sgs_samples(gmg_context *ctx) {
  if (ctx->ev3_used == 0) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < ctx->ev3_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_e[(size_t)i], ctx->ev_f[(size_t)i]) != hipSuccess) continue;
    ctx->stats.sgs_ms_total += ms;
    ctx->stats.sgs_samples++;
  }
  ctx->ev3_used = 0;
}

// option sgs_profile: the instrumented variant of the sweep; prints where each range of the first blocks spent its cycles
int sgs_profile_launch(gmg_context *ctx, Level &L, SgsWaveArgs p) {
  const size_t nr = (size_t)L.sgs.w_n_ranges;
  unsigned long long *d = nullptr;
  std::vector<unsigned long long> h(4 * nr, 0);
  HIPC(hipMalloc(&d, sizeof(unsigned long long) * 4 * nr));
  p.prof = d;
  p.prof_mode = ctx->sgs_profile_mode;
  hipLaunchKernelGGL(sgs_wave_kernel<true>, dim3(L.sgs.n_blocks), dim3(kSwThreads), (size_t)L.sgs.w_lds_bytes, ctx->stream, p);
  HIPC(hipMemcpyAsync(h.data(), d, sizeof(unsigned long long) * 4 * nr, hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  (void)hipFree(d);
  if (L.sgs.w_steps > 1000) {
    std::fprintf(stderr, "[gmg] SGS sweep profile mode %d (%lld rows, %d ranges; shader cycles: sweep / of it waiting for the ring / load / write-back):", ctx->sgs_profile_mode, (long long)L.n, (int)nr);
    for (size_t i = 0; i < std::min<size_t>(nr, 10); ++i)
      std::fprintf(stderr, " [%llu %llu %llu %llu]", h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
    std::fprintf(stderr, "\n");
  }
  return GMG_OK;
}

int sgs_apply(gmg_context *ctx, Level &L, double *y, const double *r) {
  if (L.sgs.wave) {
    SgsWaveArgs p{};
    p.ranges = L.sgs.w_ranges; p.block_rng = L.sgs.w_block_rng; p.stream = L.sgs.w_stream; p.ws_ci = L.sgs.w_ws_ci;
    p.ci_row = L.sgs.w_ci_row; p.ycur = L.sgs.w_ycur; p.y = y; p.omega = ctx->omega; p.y_slots = L.sgs.w_y_slots;
    p.row_ci = L.sgs.w_row_ci; p.rpos_f = L.sgs.w_rpos_f; p.rpos_b = L.sgs.w_rpos_b; p.iso_diag = L.sgs.w_iso_diag;
    p.iso_invd = L.sgs.w_iso_invd; p.r = r; p.n_rows = L.n; p.abort_flag = ctx->sgs_abort;
    hipLaunchKernelGGL(sgs_wave_prepass_kernel, dim3(grid_for(L.n)), dim3(256), 0, ctx->stream, p);
    // one process per GPU: the blocks are what the reference's ranks sweep -- each rank sweeps its share of them and
    // the pieces of y are all-gathered (levels >= 1 are replicated: every rank needs the whole vector)
    const int n_ranks = ctx->dist ? ctx->comm.n_ranks : 1;
    const bool split = n_ranks > 1 && L.sgs.n_blocks % n_ranks == 0 && !ctx->sgs_profile;
    const int nbl = split ? L.sgs.n_blocks / n_ranks : L.sgs.n_blocks;
    p.block0 = split ? ctx->comm.rank * nbl : 0;
    if (L.sgs.w_n_coupled > 0) {
      if (ctx->sgs_profile && !L.sgs.phased) return sgs_profile_launch(ctx, L, p);
      // sampled like the level-0 SpMV: every prof_every-th sweep launch carries start / stop events; a full pool stops
      // the sampling (no synchronisation ever enters the timed region); stats: launches counted, samples timed
      if (ctx->prof_every > 0 && !ctx->ev_e.empty()) {
        ctx->stats.sgs_launches++;
        if ((ctx->sgs_launch_no++ % (ctx->prof_every | 1)) == 0 && ctx->ev3_used < (int)ctx->ev_e.size()) {  // (odd stride: a V-cycle launches 4 sweeps per level, a stride of 8 would always hit the same one)
          ctx->timed_start = ctx->ev_e[(size_t)ctx->ev3_used]; ctx->timed_stop = ctx->ev_f[(size_t)ctx->ev3_used++];
          ctx->stats.sgs_substeps += L.sgs.w_steps;
          ctx->stats.sgs_stream_bytes += L.sgs.w_stream_bytes;
        }
      }
      const size_t lds = ctx->sgs_lds_bytes_override > 0 ? (size_t)ctx->sgs_lds_bytes_override : (size_t)L.sgs.w_lds_bytes;
      if (L.sgs.phased) {
        SgsPhaseArgs q{};
        q.ranges = L.sgs.p_ranges; q.blk_tab = L.sgs.p_blk_tab; q.block_rng = p.block_rng; q.block0 = p.block0; q.stream = p.stream; q.ws_ci = p.ws_ci; q.ci_row = p.ci_row;
        q.ycur = p.ycur; q.y = p.y; q.omega = p.omega; q.y_slots = p.y_slots; q.prof = nullptr;
        if (ctx->sgs_phase_profile > 0 && n_ranks == 1 && L.sgs.dep) {
          // diagnostics of the one-dependent-wave sweep: per range, cycles of every wave and how many of them it waited
          const size_t nr = (size_t)L.sgs.w_n_ranges;
          unsigned long long *d = nullptr;
          std::vector<unsigned long long> h(12 * nr, 0);
          HIPC(hipMalloc(&d, sizeof(unsigned long long) * 12 * nr));
          q.prof = d;
          hipLaunchKernelGGL(sgs_dep_kernel, dim3(nbl), dim3(kDpThreads), lds, ctx->stream, q, ctx->sgs_abort);
          hipError_t e = hipMemcpyAsync(h.data(), d, sizeof(unsigned long long) * 12 * nr, hipMemcpyDeviceToHost, ctx->stream);
          if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
          (void)hipFree(d);
          if (e != hipSuccess) { ctx->err = std::string("SSOR sweep profile: ") + hipGetErrorString(e); return GMG_ERR_HIP; }
          if (L.sgs.w_steps > 1000 && nbl == 1)
            for (size_t i = 0; i < nr; ++i) {
              const PhRange &P = L.sgs.host_pranges[i];
              std::fprintf(stderr, "[gmg]   %s steps %4d | cycles/step %7.1f | dependent wave waited %5.1f %% | prep waves waited %5.1f %% %5.1f %% %5.1f %%\n", P.backward ? "bwd" : "fwd",
                           P.n_steps, (double)h[12 * i] / std::max(1, P.n_steps), 100.0 * h[12 * i + 1] / std::max<double>(1, h[12 * i]),
                           100.0 * h[12 * i + 3] / std::max<double>(1, h[12 * i + 2]), 100.0 * h[12 * i + 5] / std::max<double>(1, h[12 * i + 4]),
                           100.0 * h[12 * i + 7] / std::max<double>(1, h[12 * i + 6]));
            }
        } else if (ctx->sgs_phase_profile > 0 && n_ranks == 1 && !L.sgs.dep) {  // (several ranks: the option is refused in gmg_set_option; the all-gather below must run)
          // the instrumented variant of the sweep (same arithmetic, same results, s_memtime around every phase); the
          // launch falls through to the common tail like the production one
          const size_t nr = (size_t)L.sgs.w_n_ranges;
          unsigned long long *d = nullptr;
          std::vector<unsigned long long> h(12 * nr, 0);
          HIPC(hipMalloc(&d, sizeof(unsigned long long) * 12 * nr));
          q.prof = d;
#ifdef GMG_EXPERIMENTS
          if (ctx->sgs_chain) hipLaunchKernelGGL(sgs_chain_kernel, dim3(nbl), dim3(kPhThreads), lds, ctx->stream, q, ctx->sgs_abort);
          else
#endif
          if (L.sgs.reg) hipLaunchKernelGGL(sgs_regs_kernel, dim3(nbl), dim3(kPhThreads), lds, ctx->stream, q);
          else hipLaunchKernelGGL(sgs_phase_kernel, dim3(nbl), dim3(kPhThreads), lds, ctx->stream, q);
          hipError_t e = hipMemcpyAsync(h.data(), d, sizeof(unsigned long long) * 12 * nr, hipMemcpyDeviceToHost, ctx->stream);
          if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
          (void)hipFree(d);
          if (e != hipSuccess) { ctx->err = std::string("SSOR sweep profile: ") + hipGetErrorString(e); return GMG_ERR_HIP; }
          if (L.sgs.w_steps > 1000 && nbl == 1) {
            std::fprintf(stderr, "[gmg] four-wave sweep (%s), %lld rows: per range dir steps | cycles/step | load+write-back cycles\n",
                         L.sgs.reg ? "records global -> registers; 'reads' = issuing the loads, 'copy' = none" : ctx->sgs_chain ? "hand-over through an LDS word" : "s_barrier per phase", (long long)L.n);
            for (size_t i = 0; i < nr; ++i) {
              const PhRange &P = L.sgs.host_pranges[i];
              const double turns = std::max(1.0, P.n_steps / (double)kPhWaves);
              std::fprintf(stderr, "[gmg]   %s steps %4d | %7.1f | %llu | wave 0 per turn: wait %.0f reads %.0f copy %.0f P2 %.0f CRIT %.0f barriers / polls %.0f\n", P.backward ? "bwd" : "fwd", P.n_steps,
                           (double)h[12 * i] / std::max(1, P.n_steps), h[12 * i + 2] + h[12 * i + 3], h[12 * i + 4] / turns, h[12 * i + 5] / turns, h[12 * i + 6] / turns, h[12 * i + 7] / turns,
                           h[12 * i + 8] / turns, h[12 * i + 9] / turns);
            }
          }
        } else
        if (L.sgs.dep) {
          if (ctx->timed_start) {  // (two kernel arguments: the one-argument launch_timed does not fit)
            hipExtLaunchKernelGGL(sgs_dep_kernel, dim3(nbl), dim3(kDpThreads), (std::uint32_t)lds, ctx->stream, ctx->timed_start, ctx->timed_stop, 0u, q, ctx->sgs_abort);
            ctx->timed_start = ctx->timed_stop = nullptr;
          } else {
            hipLaunchKernelGGL(sgs_dep_kernel, dim3(nbl), dim3(kDpThreads), lds, ctx->stream, q, ctx->sgs_abort);
          }
        } else
#ifdef GMG_EXPERIMENTS
        if (ctx->sgs_chain) hipLaunchKernelGGL(sgs_chain_kernel, dim3(nbl), dim3(kPhThreads), lds, ctx->stream, q, ctx->sgs_abort);
        else
#endif
        if (L.sgs.reg) launch_timed(ctx, sgs_regs_kernel, dim3(nbl), dim3(kPhThreads), lds, q);
        else launch_timed(ctx, sgs_phase_kernel, dim3(nbl), dim3(kPhThreads), lds, q);
      } else {
        launch_timed(ctx, sgs_wave_kernel<false>, dim3(nbl), dim3(kSwThreads), lds, p);
      }
    }
    if (split) {
      const std::vector<int32_t> &br = L.sgs.host_block_row;
      const int64_t len = L.sgs.w_stage_len;
      auto piece = [&](int r, int64_t *b, int64_t *e) { *b = br[(size_t)(r * nbl)]; *e = br[(size_t)((r + 1) * nbl)]; };
      int64_t b, e;
      piece(ctx->comm.rank, &b, &e);
      double *mine = L.sgs.w_stage + (int64_t)ctx->comm.rank * len;
      if (e > b) HIPC(hipMemcpyAsync(mine, y + b, sizeof(double) * (size_t)(e - b), hipMemcpyDeviceToDevice, ctx->stream));
      if (allgather_chunks(ctx->comm, L.sgs.w_stage, len, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "SSOR: all-gather failed");
      for (int r = 0; r < n_ranks; ++r) {
        if (r == ctx->comm.rank) continue;
        piece(r, &b, &e);
        if (e > b) HIPC(hipMemcpyAsync(y + b, L.sgs.w_stage + (int64_t)r * len, sizeof(double) * (size_t)(e - b), hipMemcpyDeviceToDevice, ctx->stream));
      }
    }
    HIPC(hipGetLastError());
    return GMG_OK;
  }
  HIPC(hipMemsetAsync(y, 0, sizeof(double) * (size_t)L.n, ctx->stream));
  SgsArgs a{};
  a.rowptr = L.A.rowptr; a.col = L.A.col; a.val = L.A.val; a.invd = L.invd;
  a.block_row = L.sgs.block_row; a.block_stage = L.sgs.block_stage;
  a.stage_ptr = L.sgs.stage_ptr; a.stage_rows = L.sgs.stage_rows;
  a.omega = ctx->omega; a.r = r; a.y = y;
  hipLaunchKernelGGL(sgs_sweep_kernel<false>, dim3(L.sgs.n_blocks), dim3(1024), 0, ctx->stream, a);
  hipLaunchKernelGGL(sgs_sweep_kernel<true>, dim3(L.sgs.n_blocks), dim3(1024), 0, ctx->stream, a);
  HIPC(hipGetLastError());
  return GMG_OK;
}

// y = S r for the Chebyshev "preconditioner" (Ifpack_Chebyshev recurrence, zero start).
// Result lands in *out (one of L.w1 / L.w3).
int cheb_apply(gmg_context *ctx, Level &L, const double *r, double **out) {
  const double lmax = ctx->cheb_lmax_user > 0 ? ctx->cheb_lmax_user : L.cheb_lmax;
  const double alpha = lmax / ctx->cheb_ratio, beta = lmax;
  const double delta = 2.0 / (beta - alpha), theta = 0.5 * (beta + alpha), s1 = theta * delta;
  double rhok = 1.0 / s1;
  double *y = L.w1, *yn = L.w3, *w = L.w2;
  hipLaunchKernelGGL(cheb_first_kernel, dim3(grid_for(L.n)), dim3(kThreads), 0, ctx->stream, y, w, r, (const double *)L.invd,
                     theta, L.n);
  for (int deg = 1; deg < ctx->cheb_degree; ++deg) {
    const double rhokp1 = 1.0 / (2.0 * s1 - rhok);
    const double d1 = rhokp1 * rhok, d2 = 2.0 * rhokp1 * delta;
    rhok = rhokp1;
    CHK(import_ghosts(ctx, L.A, y));
    SpmvArgs a = base_args(L.A, y, yn);
    a.b = r; a.invd = L.invd; a.w = w; a.omega = d2; a.c1 = d1;
    launch_spmv_mode<kCheb>(ctx, L.A, a);
    std::swap(y, yn);
  }
  *out = y;
  return GMG_OK;
}

// MGSmootherPrecondition::apply (from_zero) / ::smooth on level l; u and rhs are level vectors.
// Jacobi steps run out of place; *u_io may come back pointing at the level's spare buffer.
int smooth_level(gmg_context *ctx, int l, double **u_io, const double *rhs, bool from_zero, double **spare) {
  Level &L = ctx->lv[(size_t)l];
  double *u = *u_io;
  const int g = grid_for(L.n);
  int first = 0;
  if (from_zero && ctx->steps > 0) {
    first = 1;
    if (ctx->smoother == GMG_SMOOTHER_JACOBI) {
      hipLaunchKernelGGL(vec_scale_mul_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, u, ctx->omega, rhs,
                         (const double *)L.invd, L.n);
    } else if (ctx->smoother == GMG_SMOOTHER_SSOR) {
      CHK(sgs_apply(ctx, L, u, rhs));
    } else {
      double *y = nullptr;
      CHK(cheb_apply(ctx, L, rhs, &y));
      hipLaunchKernelGGL(vec_equ_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, u, 1.0, (const double *)y, L.n);
    }
  }
  for (int s = first; s < ctx->steps; ++s) {
    CHK(import_ghosts(ctx, L.A, u));
    if (ctx->smoother == GMG_SMOOTHER_JACOBI) {
      SpmvArgs a = base_args(L.A, u, *spare);
      a.b = rhs; a.invd = L.invd; a.omega = ctx->omega;
      launch_spmv_mode<kJacobi>(ctx, L.A, a);
      std::swap(u, *spare);
    } else {
      CHK(spmv(ctx, L.A, kResid, u, L.t, nullptr, rhs));  // r = rhs - A u
      if (ctx->smoother == GMG_SMOOTHER_SSOR) {
        CHK(sgs_apply(ctx, L, L.w1, L.t));
        hipLaunchKernelGGL(vec_add_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, u, 1.0, (const double *)L.w1, L.n);
      } else {
        double *y = nullptr;
        CHK(cheb_apply(ctx, L, L.t, &y));
        hipLaunchKernelGGL(vec_add_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, u, 1.0, (const double *)y, L.n);
      }
    }
  }
  *u_io = u;
  return launch_status(ctx);
}

// Sample i brackets iteration i * prof_every of the solve just finished.  Iterations at or beyond the
// converged count returned at once: their event time is the cost of the bracket itself (launch gap +
// an early-exit kernel) and is kept apart, so that bench.py can take it off the live launches.
void collect_profile_samples(gmg_context *ctx) {
  const int iters = ctx->st_final.iters;
  for (int i = 0; i < ctx->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_a[(size_t)i], ctx->ev_b[(size_t)i]) != hipSuccess) continue;
    if (i * ctx->prof_every < iters) {
      ctx->stats.spmv0_ms_total += ms;
      ctx->stats.spmv0_samples++;
    } else {
      ctx->stats.spmv0_noop_ms_total += ms;
      ctx->stats.spmv0_noop_samples++;
    }
  }
  for (int i = 0; i < ctx->ev2_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_c[(size_t)i], ctx->ev_d[(size_t)i]) != hipSuccess) continue;
    if (i * ctx->prof_every < iters) {
      ctx->stats.cgupd_ms_total += ms;
      ctx->stats.cgupd_samples++;
    }
  }
}

constexpr int64_t kUnfusedMinRowsDecl = 200000;

// Enqueues coarse-CG iterations in chunks and watches the 64-byte device state.  The first
// chunk is sized from the previous solve (zero-start solves of one hierarchy need nearly the
// same number of iterations); while the host waits for a chunk's state, the NEXT chunk is
// already queued, so the GPU never idles on the host round trip -- iterations enqueued after
// convergence return at once (the `done` flag, ~1 us each).
// `later_default`: iterations per follow-up chunk; it only has to outlast the host's poll-and-enqueue
// round trip (~30 us), so long iterations (large level 0) take short chunks and waste fewer no-ops.
template <class EnqueueOne>
int run_cg_chunks(gmg_context *ctx, int later_default, EnqueueOne enqueue_one) {
  const int maxit = ctx->coarse_maxit;
  const int last = ctx->last_coarse_iters;
  const int first = ctx->coarse_chunk > 0 ? ctx->coarse_chunk : (last > 8 ? last - (later_default < 6 ? std::max(2, last / 4) : 2) : 16);
  const int later = ctx->coarse_chunk > 0 ? ctx->coarse_chunk : later_default;
  int launched = 0;
  auto launch_chunk = [&](int n, int slot) -> int {
    for (int q = 0; q < n; ++q) {
      const int rc = enqueue_one(launched++);
      if (rc != GMG_OK) return rc;
    }
    CHK(launch_status(ctx));
    HIPC(hipMemcpyAsync(&ctx->st_host[slot], ctx->st, sizeof(CGState), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipEventRecord(ctx->ev_chunk[slot], ctx->stream));
    return GMG_OK;
  };
  int slot = 0;
  CHK(launch_chunk(std::min(first, maxit + 1), slot));
  // over RCCL the no-op iterations after convergence would still run their collectives: not worth it across ranks; over
  // the peer transport they return at once like on one GPU
  const bool speculate = !l0_partitioned(ctx) || ctx->comm.peer;
  for (;;) {
    if (speculate) CHK(launch_chunk(later, slot ^ 1));
    for (;;) {
      const hipError_t e = hipEventQuery(ctx->ev_chunk[slot]);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) { ctx->err = std::string("hipEventQuery: ") + hipGetErrorString(e); return GMG_ERR_HIP; }
    }
    if (ctx->st_host[slot].done) { ctx->st_final = ctx->st_host[slot]; ctx->stats.coarse_enqueued += launched; break; }
    if (launched > maxit + 2 * later + 2) return fail(ctx, GMG_ERR_HIP, "coarse CG state machine did not terminate");
    if (!speculate) CHK(launch_chunk(later, slot ^ 1));
    slot ^= 1;
  }
  return GMG_OK;
}  // == kUnfusedMinRows in gmg_dist.hpp

// ---- coarse solver (A9): device-resident classic CG on level 0 -----------------------------

int coarse_solve_unfused(gmg_context *ctx, double *x, const double *b, int *iters_out, double *res_out);
void collect_profile_samples(gmg_context *ctx);

int coarse_solve(gmg_context *ctx, double *x, const double *b, int *iters_out, double *res_out) {
  Level &L0 = ctx->lv[0];
  const DevCSR &A = L0.A;
  if (!A.valid) return fail(ctx, GMG_ERR_INVALID, "level-0 matrix not set");
  int variant = ctx->cg_variant;
  if (l0_partitioned(ctx) || (variant == 0 && L0.n >= kUnfusedMinRowsDecl) || variant == 2 || A.lat_only)
  {
    ctx->stats.coarse_variant = 2;
    return coarse_solve_unfused(ctx, x, b, iters_out, res_out);
  }
  ctx->stats.coarse_variant = 1;
  const int64_t n = L0.n;
  const int g_upd = grid_for((n / 2 + 0));
  const int g_init = grid_for(n);
  CGInitArgs ia{b, x, ctx->cg_g, ctx->cg_d0, ctx->cg_d1, n, ctx->st, ctx->part_b};
  hipLaunchKernelGGL(cg_init_kernel, dim3(g_init), dim3(kThreads), 0, ctx->stream, ia);
  int n_part_gg = g_init;
  const int maxit = ctx->coarse_maxit;
  ctx->ev_used = 0; ctx->ev2_used = 0;
  CHK(run_cg_chunks(ctx, 6, [&](int launched) -> int {
    const bool odd = launched & 1;
    SpmvArgs a = base_args(A, odd ? ctx->cg_d1 : ctx->cg_d0, ctx->cg_h);
    a.g = ctx->cg_g;
    a.dnew = odd ? ctx->cg_d0 : ctx->cg_d1;
    a.st = ctx->st;
    a.part_in = ctx->part_b; a.n_part_in = n_part_gg;
    a.part_out = ctx->part_a;
    a.tol = ctx->coarse_tol; a.maxit = maxit;
    const bool sample = ctx->prof_every > 0 && (launched % ctx->prof_every) == 0 && ctx->ev_used < (int)ctx->ev_a.size();
    if (sample) { ctx->timed_start = ctx->ev_a[(size_t)ctx->ev_used]; ctx->timed_stop = ctx->ev_b[(size_t)ctx->ev_used++]; }
    const int n_part_dh = launch_op<kStore, 1>(ctx, A, a);
    CGUpdateArgs ua{x, ctx->cg_g, a.dnew, ctx->cg_h, n, ctx->st, ctx->part_a, n_part_dh, ctx->part_b};
    const bool sample2 = sample && ctx->ev2_used < (int)ctx->ev_c.size();
    if (sample2) { ctx->timed_start = ctx->ev_c[(size_t)ctx->ev2_used]; ctx->timed_stop = ctx->ev_d[(size_t)ctx->ev2_used++]; }
    launch_timed(ctx, cg_update_kernel, dim3(g_upd), dim3(kThreads), 0, ua);
    n_part_gg = g_upd;
    return GMG_OK;
  }));
  collect_profile_samples(ctx);
  ctx->last_coarse_iters = ctx->st_final.iters;
  ctx->stats.coarse_solves++;
  ctx->stats.coarse_iterations += ctx->st_final.iters;
  if (iters_out) *iters_out = ctx->st_final.iters;
  if (res_out) *res_out = ctx->st_final.res;
  if (ctx->st_final.status != 0) return fail(ctx, GMG_ERR_COARSE_NOCONV, "coarse CG did not converge within max_it");
  return GMG_OK;
}


// ---- canonical row partition of distributed runs: equal chunks, rank r owns
// [r*chunk, min((r+1)*chunk, n)); chosen so that ncclAllGather can rebuild a replica in place.
inline int64_t part_chunk(int64_t n, int n_ranks) { return (n + n_ranks - 1) / n_ranks; }
inline void part_range(int64_t n, int rank, int n_ranks, int64_t *b, int64_t *e) {
  const int64_t c = part_chunk(n, n_ranks);
  *b = std::min<int64_t>(n, (int64_t)rank * c);
  *e = std::min<int64_t>(n, (int64_t)(rank + 1) * c);
}

// full[0 .. n_global) <- concatenation of every rank's owned slice (local[0 .. n_owned))
int allgather_full(gmg_context *ctx, double *full, const double *local, int64_t n_global) {
  const int64_t c = part_chunk(n_global, ctx->comm.n_ranks);
  int64_t b, e;
  part_range(n_global, ctx->comm.rank, ctx->comm.n_ranks, &b, &e);
  // stage the owned slice at its place; the padded tail of the last chunks is never read
  if (e > b) HIPC(hipMemcpyAsync(full + b, local, sizeof(double) * (size_t)(e - b), hipMemcpyDeviceToDevice, ctx->stream));
  if (allgather_chunks(ctx->comm, full, c, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-gather failed");
  return GMG_OK;
}

// MGCoarseGridBase::operator() as the V-cycle sees it: replicated defect in, replicated solution out
int coarse_level_solve(gmg_context *ctx) {
  Level &L0 = ctx->lv[0];
  if (!l0_partitioned(ctx)) return coarse_solve(ctx, L0.sol, L0.def, nullptr, nullptr);
  int64_t b, e;
  part_range(ctx->l0_global, ctx->comm.rank, ctx->comm.n_ranks, &b, &e);
  if (e > b) HIPC(hipMemcpyAsync(L0.def, L0.def_full + b, sizeof(double) * (size_t)(e - b), hipMemcpyDeviceToDevice, ctx->stream));
  CHK(coarse_solve(ctx, L0.sol, L0.def, nullptr, nullptr));
  return allgather_full(ctx, L0.sol_full, L0.sol, ctx->l0_global);
}

// ---- V-cycle (A5, A6, A8) -----------------------------------------------------------------

int level_v_step(gmg_context *ctx, int l) {
  Level &L = ctx->lv[(size_t)l];
  if (l == 0) return coarse_level_solve(ctx);
  Level &C = ctx->lv[(size_t)l - 1];
  double *c_def = (l == 1) ? C.def_full : C.def;
  const int g = grid_for(L.n);
  CHK(smooth_level(ctx, l, &L.sol, L.def, true, &L.w1));  // pre_smooth->apply (Jacobi may swap sol <-> w1)
  if (L.has_I) {
    CHK(spmv(ctx, L.A, kStore, L.sol, L.t));                 // t = A u
    CHK(spmv(ctx, L.I, kStore, L.sol, L.t, L.t));            // edge_out->vmult_add
    hipLaunchKernelGGL(vec_sadd_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, L.t, -1.0, 1.0, (const double *)L.def, L.n);
  } else {
    CHK(spmv(ctx, L.A, kResid, L.sol, L.t, nullptr, L.def));  // t = defect - A u
  }
  CHK(spmv(ctx, C.Pt, kStore, L.t, c_def, c_def));            // restrict_and_add
  CHK(level_v_step(ctx, l - 1));
  const double *c_sol = (l == 1) ? C.sol_full : C.sol;        // read AFTER the recursion: Jacobi swaps C.sol
  CHK(spmv(ctx, C.P, kAddTo, c_sol, L.sol, nullptr, L.sol));  // u += P u_c
  if (L.has_I) {
    CHK(spmv(ctx, L.It, kResid, L.sol, L.def, nullptr, L.def));  // defect -= I^T u
  }
  CHK(smooth_level(ctx, l, &L.sol, L.def, false, &L.w1));     // post_smooth->smooth
  return GMG_OK;
}

int vcycle(gmg_context *ctx, double *dst, const double *src) {
  if (!ctx->S.valid) return fail(ctx, GMG_ERR_INVALID, "system matrix not set");
  const double *src_v = src;
  double *dst_v = dst;
  int64_t n_sys = ctx->S.n_rows;
  if (ctx->dist) {  // PreconditionMG sees replicas of the outer vectors; only level 0 stays partitioned
    CHK(allgather_full(ctx, ctx->sys_full_a, src, ctx->sys_global));
    src_v = ctx->sys_full_a;
    dst_v = ctx->sys_full_b;
    n_sys = ctx->sys_global;
  }
  for (int l = 0; l < ctx->n_levels; ++l) {  // copy_to_mg
    Level &L = ctx->lv[(size_t)l];
    if (!L.A.valid) return fail(ctx, GMG_ERR_INVALID, "level matrix not set");
    double *def = (l == 0) ? L.def_full : L.def;
    const int64_t nl = (l == 0 && l0_partitioned(ctx)) ? ctx->l0_global : L.n_vec;
    HIPC(hipMemsetAsync(def, 0, sizeof(double) * (size_t)nl, ctx->stream));
    if (l > 0) HIPC(hipMemsetAsync(L.sol, 0, sizeof(double) * (size_t)L.n_vec, ctx->stream));
    if (L.n_copy)
      hipLaunchKernelGGL(gather_scatter_kernel, dim3(grid_for(L.n_copy)), dim3(kThreads), 0, ctx->stream, def,
                         (const int32_t *)L.copy_l, src_v, (const int32_t *)L.copy_g, L.n_copy);
  }
  CHK(level_v_step(ctx, ctx->n_levels - 1));
  HIPC(hipMemsetAsync(dst_v, 0, sizeof(double) * (size_t)n_sys, ctx->stream));  // copy_from_mg: dst = 0
  for (int l = 0; l < ctx->n_levels; ++l) {
    Level &L = ctx->lv[(size_t)l];
    const double *sol = (l == 0) ? L.sol_full : L.sol;
    if (L.n_copy)
      hipLaunchKernelGGL(gather_scatter_kernel, dim3(grid_for(L.n_copy)), dim3(kThreads), 0, ctx->stream, dst_v,
                         (const int32_t *)L.copy_g, sol, (const int32_t *)L.copy_l, L.n_copy);
  }
  if (ctx->dist) {
    int64_t b, e;
    part_range(ctx->sys_global, ctx->comm.rank, ctx->comm.n_ranks, &b, &e);
    if (e > b) HIPC(hipMemcpyAsync(dst, dst_v + b, sizeof(double) * (size_t)(e - b), hipMemcpyDeviceToDevice, ctx->stream));
  }
  ctx->stats.vcycles++;
  return launch_status(ctx);
}

// inverse diagonal + Gershgorin bound of D^-1 A (host), upload
int setup_diag(gmg_context *ctx, int64_t n, const int64_t *rp, const int32_t *col, const double *val, double **invd_dev,
               double *lmax_out) {
  std::vector<double> invd((size_t)std::max<int64_t>(n, 1));
  double lmax = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double aii = 0.0, rs = 0.0;
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
      if (col[k] == i) aii = val[k];
      rs += std::fabs(val[k]);
    }
    invd[(size_t)i] = 1.0 / aii;
    lmax = std::max(lmax, rs / std::fabs(aii));
  }
  if (lmax_out) *lmax_out = lmax;
  CHK(alloc_vec(ctx, invd_dev, n));
  if (n) HIPC(hipMemcpyAsync(*invd_dev, invd.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return GMG_OK;
}

void free_sgs(SgsPlan &g) {
  for (void *p : {(void *)g.stage_ptr, (void *)g.stage_rows, (void *)g.block_row, (void *)g.block_stage, (void *)g.w_ranges,
                  (void *)g.w_block_rng, (void *)g.w_ws_ci, (void *)g.w_ci_row, (void *)g.w_row_ci, (void *)g.w_rpos_f, (void *)g.w_rpos_b,
                  (void *)g.w_stream, (void *)g.w_ycur, (void *)g.w_iso_diag, (void *)g.w_iso_invd, (void *)g.w_stage, (void *)g.p_ranges, (void *)g.p_blk_tab})
    if (p) (void)hipFree(p);
  g = SgsPlan();
}

// Plan of the wavefront sweep (gmg_sgs.hpp): per block the pruned rows, the dependency stages, the steps of both
// sweep directions, their grouping into LDS-sized ranges, and the record stream in consumption order.
int setup_sgs_wave(gmg_context *ctx, Level &L, int64_t n, const int64_t *rp, const int32_t *col, const double *val, int n_blocks,
                   const std::vector<int32_t> &block_row, bool allow_phase = true) {
  SgsPlan &G = L.sgs;
  G.wave = false;
  if (n <= 0 || ctx->sgs_disable_wave) return GMG_OK;
  // four waves in turn (gmg_sgs_phase.hpp) unless switched off; rows wider than its records hold fall back to one wave
  bool ph = allow_phase && !ctx->sgs_disable_phase && !ctx->sgs_profile;
  if (ph) {
    for (int64_t i = 0; i < n && ph; ++i) ph = rp[i + 1] - rp[i] <= 2 * kPhMaxEntries;  // (cheap pre-check; the exact one is per range)
  }
  const bool dep = ph && ctx->sgs_dep;  // one dependent wave + three preparing waves (gmg_sgs_dep.hpp)
  const bool reg = ph && !dep && ctx->sgs_reg && !ctx->sgs_chain;  // four waves, records global -> registers (gmg_sgs_reg.hpp)
  const bool fieldmajor = dep || reg;
  const int late_steps = dep ? kDpLate : 1;
  const int y_max = dep ? kDpYSlots : reg ? kRgYSlots : ph ? kPhYSlots : kSwYSlots;
  int y_cap = ctx->sgs_y_slots > 0 ? ctx->sgs_y_slots : y_max;
  y_cap = std::max(64, std::min(y_cap, y_max)) & ~1;
  std::vector<PhRange> pranges;
  std::vector<uint4> blk_tab;
  std::vector<int64_t> hist_l1(8, 0), hist_l2(4, 0), hist_g(4, 0);  // shapes of the steps (debug print)
  std::vector<int32_t> row_ci((size_t)n, -1), ci_row, rpos_f, rpos_b, ws_ci, block_rng((size_t)n_blocks + 1, 0);
  std::vector<double> iso_diag((size_t)n, 0.0), iso_invd((size_t)n, 1.0);
  std::vector<SwRange> ranges;
  std::vector<char> stream;
  int max_ws = 0;
  int64_t total_steps = 0, total_stages = 0;
  // what the kernels take from the records as ADDRESSES: the forward sweep's global store index (aux, in doubles into the
  // stream) and the LDS byte addresses of y slots.  Their maxima are checked against the allocation / the LDS budget
  // before the plan is accepted (the one fault this code ever produced was a store through an aux field).
  uint64_t max_aux = 0, max_lds_addr = 0;
  for (int b = 0; b < n_blocks; ++b) {
    const int64_t rb = block_row[(size_t)b], re = block_row[(size_t)b + 1];
    const int m = (int)(re - rb);
    block_rng[(size_t)b] = (int32_t)(ph ? pranges.size() : ranges.size());
    if (m == 0) continue;
    // ---- in-block nonzero entries of every row (local column numbers), 1 / a_ii as in setup_diag
    std::vector<int32_t> prp((size_t)m + 1, 0), pcol;
    std::vector<double> pval, invd((size_t)m, 1.0);
    std::vector<char> coupled((size_t)m, 0);
    std::vector<int32_t> low_cnt((size_t)m + 1, 0);
    for (int i = 0; i < m; ++i) {
      double aii = 1.0, dstored = 0.0;
      for (int64_t k = rp[rb + i]; k < rp[rb + i + 1]; ++k) {
        const int64_t c = col[k];
        if (k > rp[rb + i] && c <= col[k - 1]) return GMG_OK;  // the prefix hand-over needs ascending columns: generic sweep
        if (c == rb + i) { aii = val[k]; dstored = val[k]; }
        if (c < rb || c >= re || val[k] == 0.0) continue;
        pcol.push_back((int32_t)(c - rb));
        pval.push_back(val[k]);
        if (c != rb + i) {
          coupled[(size_t)i] = 1; coupled[(size_t)(c - rb)] = 1;
          low_cnt[(size_t)std::max<int64_t>(i, c - rb) + 1]++;
        }
      }
      prp[(size_t)i + 1] = (int32_t)pcol.size();
      invd[(size_t)i] = 1.0 / aii;
      iso_diag[(size_t)(rb + i)] = dstored;
      iso_invd[(size_t)(rb + i)] = 1.0 / aii;
    }
    // ---- stages: stage(i) = 1 + max stage(j) over the j < i coupled to i through a_ij or a_ji
    for (int i = 0; i < m; ++i) low_cnt[(size_t)i + 1] += low_cnt[(size_t)i];
    std::vector<int32_t> low((size_t)low_cnt[(size_t)m]), fill(low_cnt.begin(), low_cnt.end() - 1);
    for (int i = 0; i < m; ++i)
      for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1]; ++k)
        if (pcol[(size_t)k] != i) low[(size_t)fill[(size_t)std::max(i, pcol[(size_t)k])]++] = std::min(i, pcol[(size_t)k]);
    std::vector<int32_t> stage((size_t)m, 0);
    int n_stages = 0;
    std::vector<int32_t> crow;  // coupled rows, ascending
    for (int i = 0; i < m; ++i) {
      if (!coupled[(size_t)i]) continue;
      int st = 0;
      for (int32_t k = low_cnt[(size_t)i]; k < low_cnt[(size_t)i + 1]; ++k) st = std::max(st, stage[(size_t)low[(size_t)k]] + 1);
      stage[(size_t)i] = st;
      n_stages = std::max(n_stages, st + 1);
      crow.push_back(i);
    }
    const size_t ci_base = ci_row.size();
    ci_row.resize(ci_base + crow.size());
    rpos_f.resize(ci_row.size(), 0);
    rpos_b.resize(ci_row.size(), 0);
    if (crow.empty()) continue;
    total_stages += n_stages;
    std::vector<int32_t> sptr((size_t)n_stages + 1, 0), by_stage(crow.size());
    for (int32_t i : crow) sptr[(size_t)stage[(size_t)i] + 1]++;
    for (int t = 0; t < n_stages; ++t) sptr[(size_t)t + 1] += sptr[(size_t)t];
    {
      std::vector<int32_t> pos(sptr.begin(), sptr.end() - 1);
      for (int32_t i : crow) by_stage[(size_t)pos[(size_t)stage[(size_t)i]]++] = i;
    }
    // The compact copy ycur is numbered in SWEEP order (stage by stage): the rows a forward range updates are one contiguous
    // piece of it, those of a backward range a run of stage-long pieces -- the working-set loads and write-backs at the range
    // boundaries (a tenth of the sweep) then touch consecutive addresses instead of one cache line per row.
    for (size_t q = 0; q < by_stage.size(); ++q) {
      row_ci[(size_t)(rb + by_stage[q])] = (int32_t)(ci_base + q);
      ci_row[ci_base + q] = (int32_t)(rb + by_stage[q]);
    }
    std::vector<int32_t> ws_stamp((size_t)m, -1), own_stamp((size_t)m, -1), tmp_stamp((size_t)m, -1), slot_of((size_t)m, 0);
    std::vector<int32_t> prefix_pos((size_t)m, 0);  // index (doubles) of the row's prefix field in the backward records
    std::vector<SwRange> dir_ranges[2];
    std::vector<PhRange> dir_pranges[2];
    std::vector<int32_t> step_of((size_t)m, -1);
    std::vector<int32_t> upd_stamp((size_t)m, -1);  // == stamp_id: the row is updated in the current range
    int stamp_id = 0, tmp_id = 0;
    for (int dir = 1; dir >= 0; --dir) {  // backward first: the forward records point into the backward ones
      // entries of a row in this direction: forward = the columns j < i (y_j = 0 for j >= i); backward = the columns
      // j >= i, continuing the forward sum
      auto in_dir = [&](int i, int c) { return dir == 0 ? c < i : c >= i; };
      auto n_ent = [&](int i) {
        int c = 0;
        for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1]; ++k) c += in_dir(i, pcol[(size_t)k]);
        return c;
      };
      // ---- groups per sub-step for this direction: every step needs ceil(widest row / 8 g) sub-steps, a sub-step of g
      // groups costs about 3 + g units (measured: ~300 + 100 g cycles); steps are roughly the stages
      int g_dir = 2;
      {
        std::vector<int> stage_len((size_t)n_stages, 0);
        for (int32_t i : crow) stage_len[(size_t)stage[(size_t)i]] = std::max(stage_len[(size_t)stage[(size_t)i]], n_ent(i));
        double best = 1e300;
        for (int gg = 1; gg <= 4; ++gg) {
          double cost = 0;
          for (int t = 0; t < n_stages; ++t) cost += (double)std::max(1, (stage_len[(size_t)t] + 8 * gg - 1) / (8 * gg)) * (3.0 + gg);
          if (cost < best) { best = cost; g_dir = gg; }
        }
        if (ctx->sgs_groups > 0) g_dir = std::min(4, ctx->sgs_groups);
      }
      const int w_dir = 8 * g_dir, stride = ph ? ph_stride(3, 12) : sw_stride(g_dir);
      const int max_rows = ph ? kPhMaxRows : 64;
      // ---- steps: <= 64 rows of one stage, records of a sub-step <= kSwMaxBlock bytes, working set <= y_cap
      struct Step { int32_t first, nrows, len; };
      std::vector<int32_t> seq;
      std::vector<Step> steps;
      seq.reserve(crow.size());
      for (int t = 0; t < n_stages; ++t) {
        const int tt = dir == 0 ? t : n_stages - 1 - t;
        Step cur{(int32_t)seq.size(), 0, 0};
        // (rows of a stage are independent: the backward sweep takes them in descending order, so that the rows a backward range
        // updates are a DEscending contiguous piece of ycur, as those of a forward range are an ascending one)
        for (int32_t qq = sptr[(size_t)tt]; qq < sptr[(size_t)tt + 1]; ++qq) {
          const int32_t q = dir == 0 ? qq : sptr[(size_t)tt] + sptr[(size_t)tt + 1] - 1 - qq;
          const int i = by_stage[(size_t)q];
          const int li = n_ent(i);
          const int nl = std::max(cur.len, li);
          const int64_t raw = 16 + (int64_t)(cur.nrows + 1) * stride;
          if (cur.nrows > 0 && (cur.nrows == max_rows || (!ph && raw > kSwMaxBlock) || (cur.nrows + 1) * (nl + 1) > y_cap)) {
            steps.push_back(cur);
            cur = Step{(int32_t)seq.size(), 0, 0};
          }
          cur.len = std::max(cur.len, li);
          cur.nrows++;
          seq.push_back(i);
        }
        if (cur.nrows) steps.push_back(cur);
      }
      // ---- ranges: greedy runs of steps whose rows + referenced rows fit y_cap
      size_t s0 = 0;
      while (s0 < steps.size()) {
        ++stamp_id;
        int ws = 0;
        size_t s1 = s0;
        while (s1 < steps.size()) {
          ++tmp_id;
          int add = 0;
          const Step &S = steps[s1];
          for (int u = 0; u < S.nrows; ++u) {
            const int i = seq[(size_t)(S.first + u)];
            if (ws_stamp[(size_t)i] != stamp_id && tmp_stamp[(size_t)i] != tmp_id) { tmp_stamp[(size_t)i] = tmp_id; ++add; }
            for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1]; ++k) {
              const int c = pcol[(size_t)k];
              if (!in_dir(i, c)) continue;
              if (ws_stamp[(size_t)c] != stamp_id && tmp_stamp[(size_t)c] != tmp_id) { tmp_stamp[(size_t)c] = tmp_id; ++add; }
            }
          }
          if (ws + add > y_cap && s1 > s0) break;
          if (ws + add > y_cap) return fail(ctx, GMG_ERR_UNSUPPORTED, "SGS plan: one step exceeds the LDS working set");
          ws += add;
          for (int u = 0; u < S.nrows; ++u) {
            const int i = seq[(size_t)(S.first + u)];
            ws_stamp[(size_t)i] = stamp_id;
            for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1]; ++k)
              if (in_dir(i, pcol[(size_t)k])) ws_stamp[(size_t)pcol[(size_t)k]] = stamp_id;
          }
          ++s1;
        }
        // slots: rows updated here first (step order), then the rows only read (first touch)
        SwRange R{};
        R.ws_off = (int32_t)ws_ci.size();
        int n_slot = 0;
        for (size_t st = s0; st < s1; ++st)
          for (int u = 0; u < steps[st].nrows; ++u) {
            const int i = seq[(size_t)(steps[st].first + u)];
            own_stamp[(size_t)i] = stamp_id;
            slot_of[(size_t)i] = n_slot++;
            ws_ci.push_back(row_ci[(size_t)(rb + i)]);
          }
        R.n_own = n_slot;
        for (size_t st = s0; st < s1; ++st)
          for (int u = 0; u < steps[st].nrows; ++u) {
            const int i = seq[(size_t)(steps[st].first + u)];
            for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1]; ++k) {
              const int c = pcol[(size_t)k];
              if (!in_dir(i, c) || own_stamp[(size_t)c] == stamp_id) continue;
              own_stamp[(size_t)c] = stamp_id;  // now "has a slot"
              slot_of[(size_t)c] = n_slot++;
              ws_ci.push_back(row_ci[(size_t)(rb + c)]);
            }
          }
        R.n_ws = n_slot;
        max_ws = std::max(max_ws, n_slot);
        R.backward = dir;
        R.groups = g_dir;
        R.n_steps = 0;
        if (ph) {
          // ---- four-wave records.  step_of: the step (of this range) that updates a row; a row's TAIL starts at its first
          // column updated by the step right before its own
          for (size_t st = s0; st < s1; ++st)
            for (int u = 0; u < steps[st].nrows; ++u) {
              step_of[(size_t)seq[(size_t)(steps[st].first + u)]] = (int32_t)(st - s0);
              upd_stamp[(size_t)seq[(size_t)(steps[st].first + u)]] = stamp_id;
            }
          // per row (in step order): entries, index of the first and of the last late column (ne: none)
          struct RowCut { int32_t ne, first, last; };
          std::vector<RowCut> cut;
          for (size_t st = s0; st < s1; ++st)
            for (int u = 0; u < steps[st].nrows; ++u) {
              const int i = seq[(size_t)(steps[st].first + u)];
              int ne = 0, first_late = -1, last_late = -1;
              for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1]; ++k) {
                const int c = pcol[(size_t)k];
                if (!in_dir(i, c)) continue;
                if (c != i && upd_stamp[(size_t)c] == stamp_id && step_of[(size_t)c] >= (int32_t)(st - s0) - late_steps && step_of[(size_t)c] < (int32_t)(st - s0)) {
                  if (first_late < 0) first_late = ne;
                  last_late = ne;
                }
                ++ne;
              }
              if (first_late < 0) { first_late = ne; last_late = ne - 1; }
              cut.push_back(RowCut{ne, first_late, last_late});
            }
          // shape of a step: head (8 G) | T1 (L1: from the first late column to the last, gathered in the dependent phase) | T2 (L2:
          // behind the last late column, products formed ahead).  A row may give the end of its head and the start of its T2 to
          // T1; of the shapes every row of the step fits, the one with the cheapest phase (measured costs, cycles) is taken.
          auto fits = [&](const RowCut &rc, int g, int l1, int l2, int *h0o, int *h1o) {
            const int h1 = std::max(rc.last + 1, rc.ne - l2), h0 = std::max(std::min(rc.first, 8 * g), h1 - l1);
            if (h0 > rc.first || h0 > 8 * g || h0 < 0) return false;
            if (h0o) { *h0o = h0; *h1o = h1; }
            return true;
          };
          struct Shape { int g, l1, l2; };
          std::vector<Shape> shape_of;
          {
            size_t c0 = 0;
            for (size_t st = s0; st < s1; ++st) {
              const int nr = steps[st].nrows;
              double best = 1e300;
              Shape bs{-1, 0, 0};
              for (int g = 0; g <= 3; ++g)
                for (int l1 = 4; l1 <= 28; l1 += 4)
                  for (int l2 = 0; l2 <= 24; l2 += 8) {
                    if (!ph_shape_ok(g, l1, l2) || (ctx->sgs_phase_nosplit && l2 > 0) || (dep && l1 > kDpMaxL1)) continue;
                    bool ok = true;
                    for (int u = 0; u < nr && ok; ++u) ok = fits(cut[c0 + (size_t)u], g, l1, l2, nullptr, nullptr);
                    if (!ok) continue;
                    const int ent = 8 * g + l1 + l2;
                    const double crit = 330.0 + 15.0 * l1 + 4.0 * l2, copy = 45.0 * (16.0 + (double)nr * ph_stride(g, l1 + l2)) / 1024.0,
                                 p1 = 150.0 + 8.0 * (2.0 + 0.75 * ent), p2 = 100.0 + 11.0 * (8 * g + l2);
                    // (dep: the dependent wave's step is what counts -- ~20 cycles per T1 slot, ~8 per T2 add; the rest is the prep waves')
                    // (reg: no copy, no read-back; the record is 2 + 6 g + 0.75 (l1 + l2) load instructions of ~20 cycles)
                    const double load = 60.0 + 20.0 * (2.0 + 6.0 * g + 0.75 * (l1 + l2));
                    const double cost = dep ? 20.0 * l1 + 8.0 * l2 + 2.0 * g
                                        : reg ? std::max(std::max(crit, load), p2) + 0.1 * (crit + load + p2)
                                              : std::max(std::max(crit, copy), std::max(p1, p2)) + 0.1 * (crit + copy + p1 + p2);
                    if (cost < best) { best = cost; bs = Shape{g, l1, l2}; }
                  }
              if (bs.g < 0) return setup_sgs_wave(ctx, L, n, rp, col, val, n_blocks, block_row, false);
              shape_of.push_back(bs);
              c0 += (size_t)nr;
            }
            // A wave pays ~400 cycles when its next step has another shape (another stretch of code): the steps are taken
            // in chunks, one shape per chunk -- the cheapest that holds every row of the chunk.
            const int chunk = ctx->sgs_phase_chunk > 0 ? ctx->sgs_phase_chunk : 32;
            std::vector<size_t> cpos(shape_of.size() + 1, 0);
            for (size_t q = 0; q < shape_of.size(); ++q) cpos[q + 1] = cpos[q] + (size_t)steps[s0 + q].nrows;
            for (size_t q0 = 0; q0 < shape_of.size(); q0 += (size_t)chunk) {
              const size_t q1 = std::min(shape_of.size(), q0 + (size_t)chunk);
              double best = 1e300;
              Shape bs{-1, 0, 0};
              const double avg_rows = (double)(cpos[q1] - cpos[q0]) / (double)(q1 - q0);
              for (int g = 0; g <= 3; ++g)
                for (int l1 = 4; l1 <= 28; l1 += 4)
                  for (int l2 = 0; l2 <= 24; l2 += 8) {
                    if (!ph_shape_ok(g, l1, l2) || (ctx->sgs_phase_nosplit && l2 > 0) || (dep && l1 > kDpMaxL1)) continue;
                    bool ok = true;
                    for (size_t u = cpos[q0]; u < cpos[q1] && ok; ++u) ok = fits(cut[u], g, l1, l2, nullptr, nullptr);
                    if (!ok) continue;
                    const int ent = 8 * g + l1 + l2;
                    const double crit = 330.0 + 15.0 * l1 + 4.0 * l2, copy = 45.0 * (16.0 + avg_rows * ph_stride(g, l1 + l2)) / 1024.0,
                                 p1 = 150.0 + 8.0 * (2.0 + 0.75 * ent), p2 = 100.0 + 11.0 * (8 * g + l2);
                    // (reg: no copy, no read-back; the record is 2 + 6 g + 0.75 (l1 + l2) load instructions of ~20 cycles)
                    const double load = 60.0 + 20.0 * (2.0 + 6.0 * g + 0.75 * (l1 + l2));
                    const double cost = dep ? 20.0 * l1 + 8.0 * l2 + 2.0 * g
                                        : reg ? std::max(std::max(crit, load), p2) + 0.1 * (crit + load + p2)
                                              : std::max(std::max(crit, copy), std::max(p1, p2)) + 0.1 * (crit + copy + p1 + p2);
                    if (cost < best) { best = cost; bs = Shape{g, l1, l2}; }
                  }
              if (bs.g < 0) return setup_sgs_wave(ctx, L, n, rp, col, val, n_blocks, block_row, false);
              for (size_t q = q0; q < q1; ++q) shape_of[q] = bs;
            }
          }
          PhRange P{};
          P.ws_off = R.ws_off; P.n_own = R.n_own; P.n_ws = R.n_ws; P.backward = dir;
          // the rows updated here as a piece of ycur (sweep-order numbering): first index and direction, if they are one
          P.own_ci0 = R.n_own > 0 ? ws_ci[(size_t)R.ws_off] : 0;
          P.own_dir = dir == 0 ? 1 : -1;
          for (int k = 0; k < R.n_own && P.own_dir; ++k)
            if (ws_ci[(size_t)R.ws_off + (size_t)k] != P.own_ci0 + P.own_dir * k) P.own_dir = 0;
          P.n_steps = (int32_t)(s1 - s0);
          const int64_t base = ((int64_t)stream.size() + 1023) / 1024 * 1024;
          P.stream_off = base;
          std::vector<int64_t> boff, bbytes;
          std::vector<uint32_t> step_word;  // shape key | rows << 8 | T1 slots in use << 16
          int64_t off = 0;
          size_t tix = 0;
          for (size_t st = s0; st < s1; ++st) {
            const Step &S = steps[st];
            const Shape sh = shape_of[st - s0];
            const int Gr = sh.g, L1r = sh.l1;
            // T1 slots the step's own rows need of the chunk's shape (the dependent phase stops there)
            int t1_used = 0;
            for (int u = 0; u < S.nrows; ++u) {
              int h0 = 0, h1 = 0;
              (void)fits(cut[tix + (size_t)u], Gr, L1r, sh.l2, &h0, &h1);
              t1_used = std::max(t1_used, h1 - h0);
            }
            t1_used = ctx->sgs_phase_nocascade ? L1r : std::min(L1r, std::max(4, (t1_used + 3) / 4 * 4));
            // (storing the records at the step's OWN width -- head groups, T1 and T2 slots its rows really have -- was measured in
            // round 3: the stream shrinks by 13 % only, a step's widest row is nearly as wide as its chunk's, and the branches that
            // make P2 and CRIT stop at the step's counts cost more than the shorter copy saves: 2.52 against 2.48 ms)
            const int gW = Gr, l1W = L1r;
            const int Lr = sh.l1 + sh.l2, pstride = ph_stride(gW, Lr);
            const int64_t raw = 16 + (int64_t)S.nrows * pstride;
            boff.push_back(off); bbytes.push_back(raw);
            stream.resize((size_t)(base + off + raw), 0);
            char *blk = stream.data() + base + off;
            step_word.push_back((uint32_t)ph::ph_key(Gr, L1r, sh.l2) | ((uint32_t)S.nrows << 8) | ((uint32_t)t1_used << 16));
            for (int u = 0; u < S.nrows; ++u) {
              const int i = seq[(size_t)(S.first + u)];
              const int32_t ci = row_ci[(size_t)(rb + i)];
              // four-wave sweep: records lane-major (one row's fields together); dep: FIELD-major -- unit k (16 bytes) of row u at
              // block + 16 + (k * nrows + u) * 16, so that a preparing wave reads a field of all rows with one coalesced load
              alignas(16) char rec_tmp[32 + 96 * 3 + 12 * kPhMaxEntries + 32];
              char *rec = fieldmajor ? rec_tmp : blk + 16 + (size_t)u * pstride;
              const int64_t rec_pos = fieldmajor ? base + off + 16 + (int64_t)u * 16 : base + off + 16 + (int64_t)u * pstride;
              const int64_t pre_pos = fieldmajor ? base + off + 16 + ((int64_t)S.nrows + u) * 16 : rec_pos + 16;
              (dir == 0 ? rpos_f : rpos_b)[(size_t)ci] = (int32_t)(rec_pos / 8);
              if (dir == 1) prefix_pos[(size_t)i] = (int32_t)(pre_pos / 8);
              double *f = reinterpret_cast<double *>(rec);
              uint32_t *wv = reinterpret_cast<uint32_t *>(rec);
              const uint32_t my = (uint32_t)slot_of[(size_t)i] * 8u;
              f[0] = 0.0; f[1] = invd[(size_t)i]; f[2] = 0.0;
              wv[6] = my;
              wv[7] = dir == 0 ? (uint32_t)prefix_pos[(size_t)i] : 0u;
              max_aux = std::max<uint64_t>(max_aux, wv[7]); max_lds_addr = std::max<uint64_t>(max_lds_addr, my);
              double *hv = f + 4, *tv = f + 4 + 8 * gW;
              uint32_t *ha = reinterpret_cast<uint32_t *>(rec + 32 + 64 * gW + 8 * Lr), *ta = ha + 8 * gW;
              for (int e = 0; e < 8 * gW; ++e) { hv[e] = 0.0; ha[e] = my; }
              for (int e = 0; e < Lr; ++e) { tv[e] = 0.0; ta[e] = my; }
              const RowCut &rc = cut[tix];
              // head [0, h0), T1 [h0, h1), T2 [h1, ne)
              int h0 = 0, h1 = rc.ne;
              (void)fits(rc, Gr, L1r, sh.l2, &h0, &h1);
              int e = 0;
              for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1]; ++k) {
                const int c = pcol[(size_t)k];
                if (!in_dir(i, c)) continue;
                const uint32_t ad = (uint32_t)slot_of[(size_t)c] * 8u;
                max_lds_addr = std::max<uint64_t>(max_lds_addr, ad);
                if (e < h0) { hv[e] = pval[(size_t)k]; ha[e] = ad; }
                else if (e < h1) { tv[e - h0] = pval[(size_t)k]; ta[e - h0] = ad; }
                else { tv[l1W + e - h1] = pval[(size_t)k]; ta[l1W + e - h1] = ad; }
                ++e;
              }
              if (fieldmajor)
                for (int k = 0; k < (32 + 96 * gW + 12 * Lr) / 16; ++k) std::memcpy(blk + 16 + ((size_t)k * S.nrows + (size_t)u) * 16, rec_tmp + 16 * k, 16);
              ++tix;
            }
            off += (raw + 15) / 16 * 16;
          }
          P.blk_tab = (uint32_t)blk_tab.size();
          for (size_t q = 0; q < boff.size(); ++q) {
            const Shape sh = shape_of[q];
            blk_tab.push_back(uint4{(uint32_t)boff[q], (uint32_t)bbytes[q], step_word[q], 0u});
            uint32_t *h = reinterpret_cast<uint32_t *>(stream.data() + base + boff[q]);
            if (q + kPhWaves < boff.size()) {  // the block its reader copies next, and that step's shape
              h[1] = (uint32_t)bbytes[q + kPhWaves]; h[2] = (uint32_t)boff[q + kPhWaves]; h[3] = step_word[q + kPhWaves];
            }
            hist_l1[(size_t)std::min(7, sh.l1 / 4)]++; hist_l2[(size_t)(sh.l2 / 8)]++; hist_g[(size_t)sh.g]++;
          }
          const int64_t padded = (off + 2048 + 1023) / 1024 * 1024;  // the copies read whole KB: up to 1008 bytes beyond a block
          if (padded >= ((int64_t)1 << 31) || (base + padded) / 8 >= ((int64_t)1 << 31)) return setup_sgs_wave(ctx, L, n, rp, col, val, n_blocks, block_row, false);
          stream.resize((size_t)(base + padded), 0);
          P.stream_bytes = (uint32_t)padded;
          // prefetch wave: pf_step bytes per phase starting pf_lead bytes ahead, never behind block p + 12 at phase p
          const int64_t nph = P.n_steps + kPhWaves - 1;
          const int64_t step_b = ((off + nph - 1) / nph + 127) / 128 * 128;
          int64_t lead = 65536;
          for (int64_t q = 0; q < (int64_t)boff.size(); ++q) {
            const int64_t need = boff[(size_t)std::min<int64_t>(q + 12, (int64_t)boff.size() - 1)] + kPhRegion;  // by phase q - 2
            lead = std::max(lead, need - std::max<int64_t>(q - 2, 0) * step_b);
          }
          if (ctx->sgs_pf_lead_kb > 0) lead = std::max<int64_t>(lead, (int64_t)ctx->sgs_pf_lead_kb * 1024);  // (diagnostics: a longer lead / no prefetch at all)
          P.pf_step = (uint32_t)step_b; P.pf_lead = (uint32_t)((lead + 8191) / 8192 * 8192);
          // Measured in round 3: the four-wave sweep with LDS copies gains nothing from the touches (2.455 ms without, 2.485 with:
          // its copies have two phases to land either way), the register variant loses 10 % without them -- so only that one, or
          // an explicit lead, keeps the fifth wave busy.
          if (ctx->sgs_pf_lead_kb < 0 || (ctx->sgs_pf_lead_kb == 0 && !fieldmajor)) P.pf_step = P.pf_lead = 0;
          total_steps += P.n_steps;
          dir_pranges[dir].push_back(P);
          s0 = s1;
          continue;
        }
        // ---- records in consumption order; a block never straddles the end of the ring
        const int64_t base = ((int64_t)stream.size() + kSwChunk - 1) / kSwChunk * kSwChunk;
        R.stream_off = base;
        int64_t off = 0, prev_hdr = -1;
        for (size_t st = s0; st < s1; ++st) {
          const Step &S = steps[st];
          const int nr = S.nrows;
          const int n_sub = std::max(1, (S.len + w_dir - 1) / w_dir);
          const int g = g_dir;
          for (int sub = 0; sub < n_sub; ++sub) {
            const int64_t raw = 16 + (int64_t)nr * stride;
            if (off % kSwRing + raw > kSwRing) off = (off / kSwRing + 1) * kSwRing;
            stream.resize((size_t)(base + off + raw), 0);
            char *blk = stream.data() + base + off;
            SwStepHdr h{};
            h.nrows = (uint16_t)nr;
            h.flags = (uint16_t)((sub == 0 ? 1 : 0) | (sub == n_sub - 1 ? 2 : 0));
            std::memcpy(blk, &h, sizeof h);
            if (prev_hdr >= 0) {
              SwStepHdr ph;
              std::memcpy(&ph, stream.data() + base + prev_hdr, sizeof ph);
              ph.advance = (uint32_t)(off - prev_hdr); ph.next_raw = (uint32_t)raw; ph.next_nrows = (uint16_t)nr;
              std::memcpy(stream.data() + base + prev_hdr, &ph, sizeof ph);
            } else {
              R.first_raw = (int32_t)raw; R.first_nrows = nr;
            }
            prev_hdr = off;
            for (int u = 0; u < nr; ++u) {
              const int i = seq[(size_t)(S.first + u)];
              const int32_t ci = row_ci[(size_t)(rb + i)];
              char *rec = blk + 16 + (size_t)u * stride;
              const int64_t rec_pos = base + off + 16 + (int64_t)u * stride;
              if (sub == n_sub - 1) (dir == 0 ? rpos_f : rpos_b)[(size_t)ci] = (int32_t)(rec_pos / 8);  // the rhs is read when the row is finished
              if (dir == 1 && sub == 0) prefix_pos[(size_t)i] = (int32_t)(rec_pos / 8 + 2);             // the prefix when it is started
              double *f = reinterpret_cast<double *>(rec);
              uint32_t *w = reinterpret_cast<uint32_t *>(rec);
              const uint32_t my = (uint32_t)slot_of[(size_t)i] * 8u;
              f[0] = 0.0; f[1] = invd[(size_t)i]; f[2] = 0.0;
              w[6] = my;
              w[7] = dir == 0 ? (uint32_t)prefix_pos[(size_t)i] : 0u;
              max_aux = std::max<uint64_t>(max_aux, w[7]); max_lds_addr = std::max<uint64_t>(max_lds_addr, my);
              uint32_t *ad = reinterpret_cast<uint32_t *>(rec + 32 + 64 * g);
              int e = 0, seen = 0;
              for (int32_t k = prp[(size_t)i]; k < prp[(size_t)i + 1] && e < 8 * g; ++k) {
                const int c = pcol[(size_t)k];
                if (!in_dir(i, c)) continue;
                if (seen++ < sub * w_dir) continue;  // entries of earlier sub-steps
                f[4 + e] = pval[(size_t)k];
                max_lds_addr = std::max<uint64_t>(max_lds_addr, (uint64_t)slot_of[(size_t)c] * 8u);
                ad[e++] = (uint32_t)slot_of[(size_t)c] * 8u;
              }
              for (; e < 8 * g; ++e) { f[4 + e] = 0.0; ad[e] = my; }
            }
            off += raw;
            R.n_steps++;
          }
        }
        if (prev_hdr >= 0) {  // last step: advance = its own size, nothing follows
          SwStepHdr ph;
          std::memcpy(&ph, stream.data() + base + prev_hdr, sizeof ph);
          ph.advance = (uint32_t)(off - prev_hdr); ph.next_raw = 0;
          std::memcpy(stream.data() + base + prev_hdr, &ph, sizeof ph);
        }
        const int64_t padded = (off + kSwChunk - 1) / kSwChunk * kSwChunk;
        if (padded >= ((int64_t)1 << 31) || (base + padded) / 8 >= ((int64_t)1 << 31)) return GMG_OK;  // generic sweep
        stream.resize((size_t)(base + padded), 0);
        R.stream_bytes = (int32_t)padded;
        total_steps += R.n_steps;
        dir_ranges[dir].push_back(R);
        s0 = s1;
      }
    }
    for (int dir = 0; dir < 2; ++dir) ranges.insert(ranges.end(), dir_ranges[dir].begin(), dir_ranges[dir].end());
    for (int dir = 0; dir < 2; ++dir) pranges.insert(pranges.end(), dir_pranges[dir].begin(), dir_pranges[dir].end());
  }
  block_rng[(size_t)n_blocks] = (int32_t)(ph ? pranges.size() : ranges.size());
  const int y_slots = std::max(2, (max_ws + 1) & ~1);
  // ---- the plan is memory-safe by construction, and checked: every address a record carries lies inside what is allocated
  if (!stream.empty() && (max_aux * 8 + 8 > stream.size() || max_lds_addr + 8 > (uint64_t)y_slots * 8))
    return fail(ctx, GMG_ERR_INVALID, "SSOR plan: a record addresses memory outside the stream / the LDS slots (internal error)");
  for (int32_t ci : ws_ci)
    if (ci < 0 || (size_t)ci >= ci_row.size()) return fail(ctx, GMG_ERR_INVALID, "SSOR plan: working-set entry out of range (internal error)");
  for (size_t q = 0; q < rpos_f.size(); ++q)
    if ((uint64_t)rpos_f[q] * 8 + 8 > stream.size() || (uint64_t)rpos_b[q] * 8 + 8 > stream.size())
      return fail(ctx, GMG_ERR_INVALID, "SSOR plan: rhs position outside the stream (internal error)");
#define SW_UP(dst, vec, T)                                                                                          \
  HIPC(hipMalloc(&dst, sizeof(T) * std::max<size_t>((vec).size(), 1)));                                              \
  if (!(vec).empty()) HIPC(hipMemcpyAsync(dst, (vec).data(), sizeof(T) * (vec).size(), hipMemcpyHostToDevice, ctx->stream));
  SW_UP(G.w_ranges, ranges, SwRange)
  SW_UP(G.p_ranges, pranges, PhRange)
  SW_UP(G.p_blk_tab, blk_tab, uint4)
  SW_UP(G.w_block_rng, block_rng, int32_t)
  SW_UP(G.w_ws_ci, ws_ci, int32_t)
  SW_UP(G.w_ci_row, ci_row, int32_t)
  SW_UP(G.w_row_ci, row_ci, int32_t)
  SW_UP(G.w_rpos_f, rpos_f, int32_t)
  SW_UP(G.w_rpos_b, rpos_b, int32_t)
  SW_UP(G.w_stream, stream, char)
  SW_UP(G.w_iso_diag, iso_diag, double)
  SW_UP(G.w_iso_invd, iso_invd, double)
#undef SW_UP
  HIPC(hipMalloc(&G.w_ycur, sizeof(double) * std::max<size_t>(ci_row.size(), 1)));
  HIPC(hipStreamSynchronize(ctx->stream));
  G.host_block_row = block_row;
  {
    const int n_ranks = ctx->dist ? ctx->comm.n_ranks : 1;
    if (n_ranks > 1 && n_blocks % n_ranks == 0) {
      const int nbl = n_blocks / n_ranks;
      int64_t len = 1;
      for (int r = 0; r < n_ranks; ++r) len = std::max<int64_t>(len, block_row[(size_t)((r + 1) * nbl)] - block_row[(size_t)(r * nbl)]);
      G.w_stage_len = len;
      HIPC(hipMalloc(&G.w_stage, sizeof(double) * (size_t)(len * n_ranks)));
    }
  }
  G.w_y_slots = y_slots;
  G.w_lds_bytes = dep ? y_slots * 8 + kDpSlots * ((kDpSlotBytes + 15) / 16 * 16) + kDpFlagBytes : reg ? y_slots * 8 + kPhJunk : ph ? y_slots * 8 + kPhWaves * kPhRegion + kPhJunk : y_slots * 8 + kSwRing + 32;
  G.dep = dep;
  G.reg = reg;
  G.w_n_ranges = (int)(ph ? pranges.size() : ranges.size());
  G.phased = ph;
  G.host_pranges = pranges;
  G.w_n_coupled = (int64_t)ci_row.size();
  G.w_stream_bytes = (int64_t)stream.size();
  G.w_steps = total_steps; G.w_stages = total_stages;
  HIPC(hipFuncSetAttribute((const void *)sgs_wave_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPC(hipFuncSetAttribute((const void *)sgs_wave_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPC(hipFuncSetAttribute((const void *)sgs_phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPC(hipFuncSetAttribute((const void *)sgs_dep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPC(hipFuncSetAttribute((const void *)sgs_regs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef GMG_EXPERIMENTS
  HIPC(hipFuncSetAttribute((const void *)sgs_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
  // the sweep bakes absolute LDS addresses into its records: its dynamic LDS must start at 0 (no static __shared__)
  hipFuncAttributes fa{};
  HIPC(hipFuncGetAttributes(&fa, ph ? (const void *)sgs_phase_kernel : (const void *)sgs_wave_kernel<false>));
  G.wave = fa.sharedSizeBytes == 0;
  if (dep) {
    HIPC(hipFuncGetAttributes(&fa, (const void *)sgs_dep_kernel));
    G.wave = G.wave && fa.sharedSizeBytes == 0;
  }
  if (reg) {
    HIPC(hipFuncGetAttributes(&fa, (const void *)sgs_regs_kernel));
    G.wave = G.wave && fa.sharedSizeBytes == 0;
  }
#ifdef GMG_EXPERIMENTS
  if (ph) {
    HIPC(hipFuncGetAttributes(&fa, (const void *)sgs_chain_kernel));
    G.wave = G.wave && fa.sharedSizeBytes == 0;
  }
#endif
  if (ctx->debug_upload && ph)
    std::fprintf(stderr, "[gmg] SGS step shapes: G %lld %lld %lld %lld | L1/4 %lld %lld %lld %lld %lld %lld %lld %lld | L2/8 %lld %lld %lld %lld\n", (long long)hist_g[0], (long long)hist_g[1],
                 (long long)hist_g[2], (long long)hist_g[3], (long long)hist_l1[0], (long long)hist_l1[1], (long long)hist_l1[2], (long long)hist_l1[3], (long long)hist_l1[4],
                 (long long)hist_l1[5], (long long)hist_l1[6], (long long)hist_l1[7], (long long)hist_l2[0], (long long)hist_l2[1], (long long)hist_l2[2], (long long)hist_l2[3]);
  if (ctx->debug_upload)
    std::fprintf(stderr, "[gmg] SGS %s plan: %lld rows, %lld coupled, %d blocks, %lld stages, %lld sub-steps, %d ranges, y slots %d, stream %.1f MB\n",
                 ph ? "four-wave" : "one-wave", (long long)n, (long long)G.w_n_coupled, n_blocks, (long long)total_stages, (long long)total_steps, G.w_n_ranges, y_slots,
                 (double)stream.size() / 1e6);
  return GMG_OK;
}


// SGS level schedule on the symmetrised pattern, per block of consecutive rows:
// stage(i) = 1 + max stage(j) over the coupled j < i of the same block.
int setup_sgs(gmg_context *ctx, Level &L, int64_t n, const int64_t *rp, const int32_t *col, const double *val) {
  std::vector<int64_t> trp;
  std::vector<int32_t> tcol;
  std::vector<double> tval, ones((size_t)rp[n], 1.0);
  transpose_host(n, n, rp, col, ones.data(), trp, tcol, tval);
  const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->ssor_blocks, (n + 63) / 64));
  std::vector<int32_t> block_row((size_t)n_blocks + 1), block_stage((size_t)n_blocks + 1, 0);
  for (int b = 0; b <= n_blocks; ++b) block_row[(size_t)b] = (int32_t)(n * b / n_blocks);
  std::vector<int32_t> stage((size_t)std::max<int64_t>(n, 1), 0), sp, rows((size_t)std::max<int64_t>(n, 1));
  int n_stages_max = 0;
  for (int b = 0; b < n_blocks; ++b) {
    const int64_t rb = block_row[(size_t)b], re = block_row[(size_t)b + 1];
    int ns = 0;
    for (int64_t i = rb; i < re; ++i) {
      int s = 0;
      for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
        if (col[k] < i && col[k] >= rb) s = std::max(s, stage[(size_t)col[k]] + 1);
      for (int64_t k = trp[(size_t)i]; k < trp[(size_t)i + 1]; ++k)
        if (tcol[(size_t)k] < i && tcol[(size_t)k] >= rb) s = std::max(s, stage[(size_t)tcol[(size_t)k]] + 1);
      stage[(size_t)i] = s;
      ns = std::max(ns, s + 1);
    }
    if (re == rb) ns = 0;
    // this block's stage offsets (absolute positions in `rows`), appended to sp
    std::vector<int32_t> cnt((size_t)ns + 1, 0);
    for (int64_t i = rb; i < re; ++i) cnt[(size_t)stage[(size_t)i] + 1]++;
    for (int t = 0; t < ns; ++t) cnt[(size_t)t + 1] += cnt[(size_t)t];
    std::vector<int32_t> pos(cnt.begin(), cnt.end());
    for (int64_t i = rb; i < re; ++i) rows[(size_t)(rb + pos[(size_t)stage[(size_t)i]]++)] = (int32_t)i;
    for (int t = 0; t <= ns; ++t) sp.push_back((int32_t)(rb + cnt[(size_t)t]));
    block_stage[(size_t)b + 1] = block_stage[(size_t)b] + ns;
    n_stages_max = std::max(n_stages_max, ns);
  }
  free_sgs(L.sgs);
  L.sgs.n_blocks = n_blocks; L.sgs.n_stages_max = n_stages_max;
  HIPC(hipMalloc(&L.sgs.stage_ptr, sizeof(int32_t) * std::max<size_t>(sp.size(), 1)));
  HIPC(hipMalloc(&L.sgs.stage_rows, sizeof(int32_t) * rows.size()));
  HIPC(hipMalloc(&L.sgs.block_row, sizeof(int32_t) * block_row.size()));
  HIPC(hipMalloc(&L.sgs.block_stage, sizeof(int32_t) * block_stage.size()));
  HIPC(hipMemcpyAsync(L.sgs.stage_ptr, sp.data(), sizeof(int32_t) * sp.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(L.sgs.stage_rows, rows.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(L.sgs.block_row, block_row.data(), sizeof(int32_t) * block_row.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(L.sgs.block_stage, block_stage.data(), sizeof(int32_t) * block_stage.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return setup_sgs_wave(ctx, L, n, rp, col, val, n_blocks, block_row);
}

// frees every operator / work vector but keeps the stream, the reduction scratch and the communicator
void free_cg_ring(gmg_context *ctx) {
  if (ctx->ring_shared[ctx->comm.rank]) {  // (collective: every rank frees its ring at the same point, gmg_set_level_matrix / gmg_reset / gmg_destroy)
    comm_share_free(ctx->comm, ctx->ring_shared);
    for (double *&p : ctx->cg_ring) p = nullptr;
  }
  for (double *&p : ctx->cg_ring)
    if (p) { (void)hipFree(p); p = nullptr; }
  ctx->cg_ring_len = 0;
}

void release_operators(gmg_context *ctx) {
  for (auto &L : ctx->lv) {
    free_csr(L.A); free_csr(L.I); free_csr(L.It); free_csr(L.P); free_csr(L.Pt);
    if (L.sol_full && L.sol_full != L.sol) (void)hipFree(L.sol_full);
    if (L.def_full && L.def_full != L.def) (void)hipFree(L.def_full);
    for (double *p : {L.sol, L.def, L.t, L.w1, L.w2, L.w3, L.invd})
      if (p) (void)hipFree(p);
    if (L.copy_g) (void)hipFree(L.copy_g);
    if (L.copy_l) (void)hipFree(L.copy_l);
    free_sgs(L.sgs);
    L = Level();
  }
  free_csr(ctx->S);
  for (double **p : {&ctx->sys_full_a, &ctx->sys_full_b, &ctx->S_invd, &ctx->S_tmp, &ctx->cg_g, &ctx->cg_d0, &ctx->cg_d1, &ctx->cg_h})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  free_cg_ring(ctx);
}

// tiles of the CSR row-window kernel: consecutive rows whose nonzeros fit the LDS window (as in upload_csr)
template <class RP>
std::vector<int32_t> window_tiles(const RP *rowptr, int64_t n_rows) {
  std::vector<int32_t> tiles;
  tiles.push_back(0);
  int64_t r = 0;
  while (r < n_rows) {
    const int64_t ka = (int64_t)rowptr[r] & ~(int64_t)3;
    int64_t e = r + 1;
    while (e < n_rows && (int64_t)rowptr[e + 1] - ka <= kTileNnz && e - r < kTileMaxRowsEarly) ++e;
    if ((int64_t)rowptr[e] - ka > kTileNnz) e = r + 1;
    tiles.push_back((int32_t)e);
    r = e;
  }
  return tiles;
}

DevCSR *which_matrix(gmg_context *ctx, int which) {
  if (which == GMG_SYSTEM) return &ctx->S;
  if (which < 0 || which >= ctx->n_levels) return nullptr;
  return &ctx->lv[(size_t)which].A;
}

}  // namespace

// The distributed coarse CG lives with the communicator code.
#include "gmg_dist.hpp"

// =============================================================================== C-ABI

extern "C" {

int gmg_create(gmg_context **out, int device_id, int n_levels) {
  if (!out || n_levels < 1) return GMG_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return GMG_ERR_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return GMG_ERR_HIP;
  gmg_context *ctx = new gmg_context();
  ctx->device = device_id;
  ctx->n_levels = n_levels;
  ctx->lv.resize((size_t)n_levels);
  auto bail = [&](int code) { gmg_destroy(ctx); return code; };
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(GMG_ERR_HIP);
  if (hipMalloc(&ctx->st, sizeof(CGState)) != hipSuccess) return bail(GMG_ERR_HIP);
  if (hipHostMalloc((void **)&ctx->st_host, 2 * sizeof(CGState), hipHostMallocDefault) != hipSuccess) return bail(GMG_ERR_HIP);
  for (auto &e : ctx->ev_chunk)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(GMG_ERR_HIP);
  if (hipMalloc(&ctx->part_a, sizeof(double) * 4 * kMaxPartials) != hipSuccess) return bail(GMG_ERR_HIP);
  if (hipMalloc(&ctx->part_b, sizeof(double) * 4 * kMaxPartials) != hipSuccess) return bail(GMG_ERR_HIP);
  if (hipMalloc(&ctx->scal_dev, sizeof(double) * 8) != hipSuccess) return bail(GMG_ERR_HIP);
  if (hipHostMalloc((void **)&ctx->scal_host, sizeof(double) * 8, hipHostMallocDefault) != hipSuccess) return bail(GMG_ERR_HIP);
  if (hipHostMalloc((void **)&ctx->sgs_abort, sizeof(int), hipHostMallocDefault) != hipSuccess) return bail(GMG_ERR_HIP);
  *ctx->sgs_abort = 0;
  (void)hipMemsetAsync(ctx->st, 0, sizeof(CGState), ctx->stream);
  // measurement scripts reach the diagnostic options of a context they do not create themselves through
  // GMG_OPTIONS="key=value,key=value" (same keys as gmg_set_option); unknown keys fail the creation
  if (const char *env = std::getenv("GMG_OPTIONS")) {
    std::string all(env);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos);
      if (end == std::string::npos) end = all.size();
      const std::string item = all.substr(pos, end - pos);
      pos = end + 1;
      const size_t eq = item.find('=');
      if (item.empty()) continue;
      const std::string key = eq == std::string::npos ? item : item.substr(0, eq);
      const double v = eq == std::string::npos ? 1.0 : std::atof(item.c_str() + eq + 1);
      if (gmg_set_option(ctx, key.c_str(), v) != GMG_OK) return bail(GMG_ERR_INVALID);
    }
  }
  *out = ctx;
  return GMG_OK;
}

int gmg_destroy(gmg_context *ctx) {
  if (!ctx) return GMG_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  release_operators(ctx);  // (before the communicator: the shared direction vectors are unmapped collectively)
  if (ctx->peer_push_cnt) (void)hipFree(ctx->peer_push_cnt);
  if (ctx->dens_dev) (void)hipFree(ctx->dens_dev);
  comm_destroy(ctx->comm);
  for (double *p : {ctx->part_a, ctx->part_b, ctx->scal_dev})
    if (p) (void)hipFree(p);
  if (ctx->st) (void)hipFree(ctx->st);
  if (ctx->st_host) (void)hipHostFree(ctx->st_host);
  for (auto &e : ctx->ev_chunk)
    if (e) (void)hipEventDestroy(e);
  if (ctx->scal_host) (void)hipHostFree(ctx->scal_host);
  if (ctx->sgs_abort) (void)hipHostFree(ctx->sgs_abort);
  for (auto *v : {&ctx->ev_a, &ctx->ev_b, &ctx->ev_c, &ctx->ev_d, &ctx->ev_e, &ctx->ev_f})
    for (hipEvent_t e : *v) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return GMG_OK;
}

int gmg_reset(gmg_context *ctx, int n_levels) {
  if (!ctx || n_levels < 1) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  HIPC(hipStreamSynchronize(ctx->stream));
  release_operators(ctx);
  ctx->n_levels = n_levels;
  ctx->lv.assign((size_t)n_levels, Level());
  ctx->last_coarse_iters = 0;
  ctx->stats = gmg_stats{};
  ctx->ev3_used = 0;
  return GMG_OK;
}

const char *gmg_last_error(const gmg_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gmg_synchronize(gmg_context *ctx) {
  if (!ctx) return GMG_ERR_INVALID;
  HIPC(hipStreamSynchronize(ctx->stream));
  return GMG_OK;
}

int gmg_set_system_matrix(gmg_context *ctx, int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int32_t *col,
                          const double *val) {
  if (!ctx) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  CHK(upload_csr(ctx, ctx->S, n_rows, n_cols, rowptr, col, val));
  CHK(setup_diag(ctx, n_rows, rowptr, col, val, &ctx->S_invd, nullptr));
  CHK(alloc_vec(ctx, &ctx->S_tmp, n_cols));
  if (ctx->dist) {
    const int64_t padded = part_chunk(ctx->sys_global, ctx->comm.n_ranks) * ctx->comm.n_ranks;
    CHK(alloc_vec(ctx, &ctx->sys_full_a, padded));
    CHK(alloc_vec(ctx, &ctx->sys_full_b, padded));
    int64_t b, e;
    part_range(ctx->sys_global, ctx->comm.rank, ctx->comm.n_ranks, &b, &e);
    if (e - b != n_rows) return fail(ctx, GMG_ERR_INVALID, "system matrix rows do not match the canonical partition");
  }
  return GMG_OK;
}

// what a level needs besides its operator: work vectors and, on level 0, the coarse CG's vectors and the shape the bench reads
static int finish_level(gmg_context *ctx, int level, int64_t n_rows, int64_t n_cols, int64_t nnz) {
  Level &L = ctx->lv[(size_t)level];
  L.n = n_rows;
  L.n_vec = n_cols;
  for (double **p : {&L.sol, &L.def, &L.t, &L.w1, &L.w2, &L.w3}) CHK(alloc_vec(ctx, p, n_cols));
  if (level == 0) {
    if (l0_partitioned(ctx)) {
      const int64_t padded = part_chunk(ctx->l0_global, ctx->comm.n_ranks) * ctx->comm.n_ranks;
      if (L.sol_full && L.sol_full != L.sol) (void)hipFree(L.sol_full);
      if (L.def_full && L.def_full != L.def) (void)hipFree(L.def_full);
      L.sol_full = L.def_full = nullptr;
      CHK(alloc_vec(ctx, &L.sol_full, padded));
      CHK(alloc_vec(ctx, &L.def_full, padded));
    } else {
      L.sol_full = L.sol;
      L.def_full = L.def;
    }
    ctx->cg_n = n_cols;
    for (double **p : {&ctx->cg_g, &ctx->cg_d0, &ctx->cg_d1, &ctx->cg_h}) CHK(alloc_vec(ctx, p, n_cols));
    free_cg_ring(ctx);  // sized for the previous level 0
    ctx->stats.spmv0_rows = n_rows;
    ctx->stats.spmv0_nnz = nnz;
    ctx->stats.spmv0_layout = L.A.sell ? 1 + (L.A.val8 ? 2 : 0) + (L.A.col16 ? 4 : 0) + (L.A.use_sellp ? 8 : 0) : 0;
    if (L.A.sell) {
      const double frac_stream = L.A.n_slices ? 1.0 - (double)L.A.n_pattern_slices / L.A.n_slices : 1.0;
      const int64_t ent = L.A.sell_quads * 256;
      // row classes: the run-pattern slices stream one byte per row instead of one per entry
      const int64_t val_bytes = L.A.rowclass ? (int64_t)(frac_stream * (double)ent) + (int64_t)L.A.n_pattern_slices * 64 : ent * (L.A.val8 ? 1 : 8);
      ctx->stats.spmv0_matrix_bytes = val_bytes + (int64_t)(frac_stream * (double)ent * (L.A.col16 ? 2 : 4)) + 8 * (int64_t)L.A.n_slices;
      if (L.A.rowclass) ctx->stats.spmv0_layout += 16;
      if (L.A.lattice) {  // one class byte per interior row + the streams of the slices outside
        ctx->stats.spmv0_layout += 32;
        ctx->stats.spmv0_matrix_bytes = L.A.lat_fast_rows + L.A.lat_gen_bytes;
      }
    } else if (L.A.lat_only) {  // class table only: one class byte per row
      ctx->stats.spmv0_layout = 1 + 2 + 4 + 8 + 16 + 32 + 64;
      ctx->stats.spmv0_matrix_bytes = n_rows;
    } else {
      ctx->stats.spmv0_matrix_bytes = 12 * nnz + 4 * (n_rows + 1);
    }
    ctx->stats.spmv0_pattern_slices = L.A.n_pattern_slices;
    ctx->stats.spmv0_slices = L.A.n_slices;
    ctx->last_coarse_iters = 0;
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  return GMG_OK;
}

int gmg_set_level_matrix(gmg_context *ctx, int level, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                         const int32_t *col, const double *val) {
  if (!ctx || level < 0 || level >= ctx->n_levels) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  Level &L = ctx->lv[(size_t)level];
  CHK(upload_csr(ctx, L.A, n_rows, n_cols, rowptr, col, val, level > 0));
  CHK(setup_diag(ctx, n_rows, rowptr, col, val, &L.invd, &L.cheb_lmax));
  if (level > 0) {
    L.n = n_rows;  // (the SGS plan reads it)
    CHK(setup_sgs(ctx, L, n_rows, rowptr, col, val));
  }
  return finish_level(ctx, level, n_rows, n_cols, rowptr[n_rows]);
}

// The level matrix of an UNDIVIDED lattice level formed on the device: what gmg_set_level_matrix would
// have received as CSR from assemble_multigrid (src/step-50.cc:869-889) for nv[0] x nv[1] x nv[2] vertices numbered
// lexicographically, every cell adding the same 8 x 8 matrix Ke, all faces Dirichlet (MGConstrainedDoFs, :704-706):
//   boundary vertex i:            a_ii = sum over its cells of |Ke[a][a]|, stored zeros elsewhere        (:877-879)
//   interior vertex i, column j:  sum over the cells holding both of Ke[a_i][a_j], 0 if j is on the boundary
// -- the sums in cell order (z, y, x ascending), from 0.0, exactly as CSRMatrix::add forms them: the same bits.  A vertex's
// 27 coefficients depend only on its position type per direction (on the low face / next to it / inside / next to the
// high face / on it): 125 classes.  The table is 125 x 27 numbers (host, microseconds); the per-row work -- class byte and
// 1 / a_ii of 10^6..10^7 rows -- is a kernel.  No CSR, no upload.
int gmg_set_level_matrix_lattice(gmg_context *ctx, int level, const int32_t nv[3], const double Ke[64]) {
  if (!ctx || !nv || !Ke || level < 0 || level >= ctx->n_levels) return GMG_ERR_INVALID;
  if (level != 0) return fail(ctx, GMG_ERR_UNSUPPORTED, "gmg_set_level_matrix_lattice: level 0 only (finer levels of an adaptive hierarchy are patches, and their smoothers need the rows)");
  if (l0_partitioned(ctx)) return fail(ctx, GMG_ERR_UNSUPPORTED, "gmg_set_level_matrix_lattice: a row-partitioned level 0 takes its local rows as CSR (gmg_set_level_matrix)");
  const int64_t nx = nv[0], ny = nv[1], nz = nv[2];
  if (nx < 5 || ny < 5 || nz < 5) return fail(ctx, GMG_ERR_INVALID, "gmg_set_level_matrix_lattice: at least 5 vertices per direction");
  const int64_t n = nx * ny * nz, nxy = nx * ny, reach = nxy + nx + 1;
  if (n >= ((int64_t)1 << 28)) return fail(ctx, GMG_ERR_UNSUPPORTED, "gmg_set_level_matrix_lattice: more than 2^28 rows");
  (void)hipSetDevice(ctx->device);
  Level &L = ctx->lv[0];
  DevCSR &m = L.A;
  HaloPlan keep = m.halo;
  m.halo = HaloPlan();
  free_csr(m);
  m.halo = keep;
  // ---- class table
  const int64_t dims[3] = {nx, ny, nz};
  auto rep = [&](int t, int64_t mdim) -> int64_t { return t == 0 ? 0 : t == 1 ? 1 : t == 2 ? 2 : t == 3 ? mdim - 2 : mdim - 1; };
  std::vector<double> ctab((size_t)125 * 27, 0.0);
  double lmax = 0.0;
  for (int cls = 0; cls < 125; ++cls) {
    const int t[3] = {cls % 5, (cls / 5) % 5, cls / 25};
    int64_t v[3];
    bool v_bnd = false;
    for (int d = 0; d < 3; ++d) { v[d] = rep(t[d], dims[d]); v_bnd = v_bnd || v[d] == 0 || v[d] == dims[d] - 1; }
    double rowsum = 0.0;
    for (int j = 0; j < 27; ++j) {
      const int dd[3] = {j % 3 - 1, (j / 3) % 3 - 1, j / 9 - 1};
      int64_t u[3];
      bool exists = true, u_bnd = false;
      for (int d = 0; d < 3; ++d) {
        u[d] = v[d] + dd[d];
        exists = exists && u[d] >= 0 && u[d] < dims[d];
        u_bnd = u_bnd || u[d] == 0 || u[d] == dims[d] - 1;
      }
      double acc = 0.0;
      if (exists && ((v_bnd && j == 13) || (!v_bnd && !u_bnd))) {
        // the cells that hold both vertices, in cell order (z, y, x ascending)
        int64_t lo[3], hi[3];
        for (int d = 0; d < 3; ++d) { lo[d] = std::max<int64_t>(std::max(v[d], u[d]) - 1, 0); hi[d] = std::min<int64_t>(std::min(v[d], u[d]), dims[d] - 2); }
        for (int64_t cz = lo[2]; cz <= hi[2]; ++cz)
          for (int64_t cy = lo[1]; cy <= hi[1]; ++cy)
            for (int64_t cx = lo[0]; cx <= hi[0]; ++cx) {
              const int a = (int)((v[0] - cx) + 2 * (v[1] - cy) + 4 * (v[2] - cz)), b = (int)((u[0] - cx) + 2 * (u[1] - cy) + 4 * (u[2] - cz));
              acc += v_bnd ? std::fabs(Ke[a * 8 + a]) : Ke[a * 8 + b];
            }
      }
      ctab[(size_t)cls * 27 + (size_t)j] = acc;
      rowsum += std::fabs(acc);
    }
    lmax = std::max(lmax, rowsum / std::fabs(ctab[(size_t)cls * 27 + 13]));  // Gershgorin bound of D^-1 A (every class occurs for n >= 5)
  }
  // position types with the same 27 coefficients share a class (most of the 125 do: the table every workgroup copies into
  // its LDS shrinks to ~30 rows)
  LatticeClassMap cmap{};
  std::vector<double> utab;
  int n_unique = 0;
  for (int cls = 0; cls < 125; ++cls) {
    int found = -1;
    for (int q = 0; q < n_unique && found < 0; ++q)
      if (std::memcmp(&utab[(size_t)q * 27], &ctab[(size_t)cls * 27], sizeof(double) * 27) == 0) found = q;
    if (found < 0) {
      found = n_unique++;
      utab.insert(utab.end(), ctab.begin() + (std::ptrdiff_t)cls * 27, ctab.begin() + (std::ptrdiff_t)(cls + 1) * 27);
    }
    cmap.cls[cls] = (uint8_t)found;
  }
  HIPC(hipMalloc(&m.lat_rowcls, (size_t)n + 64));
  HIPC(hipMalloc(&m.lat_ctab, sizeof(double) * utab.size()));
  HIPC(hipMemcpyAsync(m.lat_ctab, utab.data(), sizeof(double) * utab.size(), hipMemcpyHostToDevice, ctx->stream));
  CHK(alloc_vec(ctx, &L.invd, n));
  hipLaunchKernelGGL(lattice_rowclass_kernel, dim3(grid_for(n)), dim3(kThreads), 0, ctx->stream, m.lat_rowcls, L.invd, (const double *)m.lat_ctab, cmap, (int)nx, (int)ny, (int)nz);
  CHK(launch_status(ctx));
  HIPC(hipStreamSynchronize(ctx->stream));  // (utab dies with this scope)
  L.cheb_lmax = lmax;
  m.n_rows = m.n_cols = n;
  m.nnz = (3 * nx - 2) * (3 * ny - 2) * (3 * nz - 2);
  m.valid = true;
  m.lattice = m.lat_only = true;
  m.lat_classes = n_unique;
  m.lat_nx = (int)nx; m.lat_nxy = (int)nxy; m.lat_W = (int)nxy;
  m.lat_R0 = (int)(reach + 2); m.lat_R1 = (int)(n - reach);
  if (m.lat_R1 - m.lat_R0 < nxy) return fail(ctx, GMG_ERR_INVALID, "gmg_set_level_matrix_lattice: fewer than five planes");
  m.lat_C = (int)((nxy + kLatRowsPerUnit - 1) / kLatRowsPerUnit);
  m.lat_K = (int)((m.lat_R1 - m.lat_R0 + nxy - 1) / nxy);
  m.lat_fast_rows = m.lat_R1 - m.lat_R0;
  m.lat_n_gen = (m.lat_R0 + 63) / 64 + (int)((n - m.lat_R1 + 63) / 64);
  m.lat_gen_bytes = (int64_t)m.lat_n_gen * 64;
  m.lat_grid = lattice_grid(ctx, m, (size_t)m.lat_n_gen);
  m.lat_grid += m.lat_fast_blocks;
  if (m.lat_grid > kMaxPartials) return fail(ctx, GMG_ERR_UNSUPPORTED, "gmg_set_level_matrix_lattice: grid exceeds the reduction partials");
  if (ctx->debug_upload)
    std::fprintf(stderr, "[gmg] lattice operator %lld x %lld x %lld formed on the device: %d columns x %d steps, %d segments per XCD slab, grid %d, %d edge chunks\n", (long long)nx,
                 (long long)ny, (long long)nz, m.lat_C, m.lat_K, m.lat_S, m.lat_grid, m.lat_n_gen);
  return finish_level(ctx, 0, n, n, m.nnz);
}

int gmg_set_edge_matrix(gmg_context *ctx, int level, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                        const int32_t *col, const double *val) {
  if (!ctx || level < 0 || level >= ctx->n_levels) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  Level &L = ctx->lv[(size_t)level];
  // stored zeros contribute nothing to vmult_add / Tvmult: prune them (most of I_l is zero)
  std::vector<int64_t> rp((size_t)n_rows + 1, 0);
  std::vector<int32_t> c;
  std::vector<double> v;
  for (int64_t i = 0; i < n_rows; ++i) {
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (val[k] != 0.0) { c.push_back(col[k]); v.push_back(val[k]); }
    rp[(size_t)i + 1] = (int64_t)c.size();
  }
  L.has_I = !c.empty();
  if (!L.has_I) { free_csr(L.I); free_csr(L.It); return GMG_OK; }
  CHK(upload_csr(ctx, L.I, n_rows, n_cols, rp.data(), c.data(), v.data()));
  std::vector<int64_t> trp;
  std::vector<int32_t> tcol;
  std::vector<double> tval;
  transpose_host(n_rows, n_cols, rp.data(), c.data(), v.data(), trp, tcol, tval);
  CHK(upload_csr(ctx, L.It, n_cols, n_rows, trp.data(), tcol.data(), tval.data()));
  return GMG_OK;
}

int gmg_set_prolongation(gmg_context *ctx, int level, int64_t n_fine, int64_t n_coarse, const int64_t *rowptr,
                         const int32_t *col, const double *val) {
  if (!ctx || level < 0 || level >= ctx->n_levels - 1) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  Level &L = ctx->lv[(size_t)level];
  CHK(upload_csr(ctx, L.P, n_fine, n_coarse, rowptr, col, val));
  std::vector<int64_t> trp;
  std::vector<int32_t> tcol;
  std::vector<double> tval;
  transpose_host(n_fine, n_coarse, rowptr, col, val, trp, tcol, tval);
  CHK(upload_csr(ctx, L.Pt, n_coarse, n_fine, trp.data(), tcol.data(), tval.data()));
  L.has_P = true;
  return GMG_OK;
}

// MGTransferPrebuilt::build_matrices on the device (gmg_transfer.hpp): P_level and its transpose from the vertices of the
// DoFs of the two levels.  build_ms (may be null): device time of the build, the part the reference counts in its Solve timer.
int gmg_build_transfer(gmg_context *ctx, int level, int dim, int64_t n_coarse, const uint64_t *coarse_vertex, const uint8_t *coarse_boundary,
                       int64_t n_fine, const uint64_t *fine_vertex, uint64_t fine_spacing, double *build_ms) {
  if (!ctx || level < 0 || level >= ctx->n_levels - 1 || (dim != 2 && dim != 3) || n_coarse <= 0 || n_fine <= 0 || !coarse_vertex || !coarse_boundary ||
      !fine_vertex || fine_spacing == 0 || (fine_spacing & (fine_spacing - 1)) != 0)
    return GMG_ERR_INVALID;
  if (n_coarse >= ((int64_t)1 << 30) || n_fine >= ((int64_t)1 << 28)) return fail(ctx, GMG_ERR_UNSUPPORTED, "gmg_build_transfer: level too large for 32-bit row pointers");
  (void)hipSetDevice(ctx->device);
  Level &L = ctx->lv[(size_t)level];
  for (DevCSR *m : {&L.P, &L.Pt}) {
    HaloPlan keep = m->halo;
    m->halo = HaloPlan();
    free_csr(*m);
    m->halo = keep;
  }
  unsigned long long *d_fv = nullptr, *d_cv = nullptr, *d_fk = nullptr, *d_ck = nullptr;
  uint8_t *d_cb = nullptr;
  int32_t *d_fd = nullptr, *d_cd = nullptr;
  auto cleanup = [&]() {
    for (void *q : {(void *)d_fv, (void *)d_cv, (void *)d_fk, (void *)d_ck, (void *)d_cb, (void *)d_fd, (void *)d_cd})
      if (q) (void)hipFree(q);
  };
  auto table_size = [](int64_t n) { unsigned long long t = 1024; while ((int64_t)t < 2 * n) t <<= 1; return t; };
  const unsigned long long ft = table_size(n_fine), ct = table_size(n_coarse);
#define TRC(call) do { if ((call) != hipSuccess) { cleanup(); return fail(ctx, GMG_ERR_HIP, "gmg_build_transfer: " #call " failed"); } } while (0)
  TRC(hipMalloc(&d_fv, sizeof(uint64_t) * (size_t)n_fine));
  TRC(hipMalloc(&d_cv, sizeof(uint64_t) * (size_t)n_coarse));
  TRC(hipMalloc(&d_cb, (size_t)n_coarse));
  TRC(hipMalloc(&d_fk, sizeof(uint64_t) * ft));
  TRC(hipMalloc(&d_ck, sizeof(uint64_t) * ct));
  TRC(hipMalloc(&d_fd, sizeof(int32_t) * ft));
  TRC(hipMalloc(&d_cd, sizeof(int32_t) * ct));
  TRC(hipMemcpyAsync(d_fv, fine_vertex, sizeof(uint64_t) * (size_t)n_fine, hipMemcpyHostToDevice, ctx->stream));
  TRC(hipMemcpyAsync(d_cv, coarse_vertex, sizeof(uint64_t) * (size_t)n_coarse, hipMemcpyHostToDevice, ctx->stream));
  TRC(hipMemcpyAsync(d_cb, coarse_boundary, (size_t)n_coarse, hipMemcpyHostToDevice, ctx->stream));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  TRC(hipEventCreate(&e0));
  TRC(hipEventCreate(&e1));
  TRC(hipEventRecord(e0, ctx->stream));
  TRC(hipMemsetAsync(d_fk, 0xff, sizeof(uint64_t) * ft, ctx->stream));
  TRC(hipMemsetAsync(d_ck, 0xff, sizeof(uint64_t) * ct, ctx->stream));
  hipLaunchKernelGGL(tr_table_build_kernel, dim3(grid_for(n_fine)), dim3(kThreads), 0, ctx->stream, (const unsigned long long *)d_fv, n_fine, d_fk, d_fd, ft - 1);
  hipLaunchKernelGGL(tr_table_build_kernel, dim3(grid_for(n_coarse)), dim3(kThreads), 0, ctx->stream, (const unsigned long long *)d_cv, n_coarse, d_ck, d_cd, ct - 1);
  TransferArgs a{};
  a.fine_vertex = d_fv; a.coarse_vertex = d_cv; a.coarse_boundary = d_cb; a.n_fine = n_fine; a.n_coarse = n_coarse;
  a.fkeys = d_fk; a.ckeys = d_ck; a.fdof = d_fd; a.cdof = d_cd; a.fmask = ft - 1; a.cmask = ct - 1; a.dim = dim; a.half = fine_spacing;
  // one operator after the other: count, scan, allocate, fill
  auto build = [&](DevCSR &m, bool transposed) -> int {
    const int64_t nr = transposed ? n_coarse : n_fine, nc = transposed ? n_fine : n_coarse;
    HIPC(hipMalloc(&m.rowptr, sizeof(int32_t) * ((size_t)nr + 1)));
    a.rowptr = m.rowptr;
    if (transposed) hipLaunchKernelGGL(tr_restriction_kernel<false>, dim3(grid_for(nr)), dim3(kThreads), 0, ctx->stream, a);
    else hipLaunchKernelGGL(tr_prolongation_kernel<false>, dim3(grid_for(nr)), dim3(kThreads), 0, ctx->stream, a);
    hipLaunchKernelGGL(tr_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, m.rowptr, nr);
    int32_t nnz = 0;
    HIPC(hipMemcpyAsync(&nnz, m.rowptr + nr, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    const size_t pad = 8;
    HIPC(hipMalloc(&m.col, sizeof(int32_t) * ((size_t)nnz + pad)));
    HIPC(hipMalloc(&m.val, sizeof(double) * ((size_t)nnz + pad)));
    HIPC(hipMemsetAsync(m.col + nnz, 0, sizeof(int32_t) * pad, ctx->stream));
    HIPC(hipMemsetAsync(m.val + nnz, 0, sizeof(double) * pad, ctx->stream));
    a.col = m.col; a.val = m.val;
    if (transposed) hipLaunchKernelGGL(tr_restriction_kernel<true>, dim3(grid_for(nr)), dim3(kThreads), 0, ctx->stream, a);
    else hipLaunchKernelGGL(tr_prolongation_kernel<true>, dim3(grid_for(nr)), dim3(kThreads), 0, ctx->stream, a);
    m.n_rows = nr; m.n_cols = nc; m.nnz = nnz;
    return launch_status(ctx);
  };
  int rc = build(L.P, false);
  if (rc == GMG_OK) rc = build(L.Pt, true);
  if (rc == GMG_OK && hipEventRecord(e1, ctx->stream) != hipSuccess) rc = GMG_ERR_HIP;
  // the row-window tiling of the two operators (host: a pass over the row pointers)
  for (DevCSR *m : {&L.P, &L.Pt}) {
    if (rc != GMG_OK) break;
    std::vector<int32_t> rp((size_t)m->n_rows + 1);
    if (hipMemcpyAsync(rp.data(), m->rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = GMG_ERR_HIP; break; }
    const std::vector<int32_t> tiles = window_tiles(rp.data(), m->n_rows);
    m->n_tiles = (int)tiles.size() - 1;
    m->tiles_per_xcd = (m->n_tiles + 7) / 8;
    m->grid = 8 * std::min(kMaxPartials / 8, std::max(1, m->tiles_per_xcd));
    if (hipMalloc(&m->tile_row, sizeof(int32_t) * tiles.size()) != hipSuccess ||
        hipMemcpyAsync(m->tile_row, tiles.data(), sizeof(int32_t) * tiles.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = GMG_ERR_HIP; break; }
    m->valid = true;
  }
  float ms = 0.f;
  if (rc == GMG_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = 0.f;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#undef TRC
  cleanup();
  if (rc != GMG_OK) { if (ctx->err.empty()) ctx->err = "gmg_build_transfer failed"; return rc; }
  L.has_P = true;
  ctx->stats.build_matrices_ms += ms;
  if (build_ms) *build_ms = ms;
  if (ctx->debug_upload)
    std::fprintf(stderr, "[gmg] transfer %d -> %d built on the device: P %lld x %lld nnz %lld, P^T nnz %lld, %.3f ms\n", level, level + 1, (long long)n_fine, (long long)n_coarse,
                 (long long)L.P.nnz, (long long)L.Pt.nnz, ms);
  return GMG_OK;
}

// CSR of P_level (transposed = 0) or its transpose as the device holds it (tests): sizes first (arrays null), then the copy
int gmg_get_transfer(gmg_context *ctx, int level, int transposed, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int64_t *rowptr, int32_t *col, double *val) {
  if (!ctx || level < 0 || level >= ctx->n_levels - 1) return GMG_ERR_INVALID;
  const DevCSR &m = transposed ? ctx->lv[(size_t)level].Pt : ctx->lv[(size_t)level].P;
  if (!m.valid || !m.rowptr || !m.col || !m.val) return fail(ctx, GMG_ERR_UNSUPPORTED, "gmg_get_transfer: no CSR copy of this operator on the device");
  if (n_rows) *n_rows = m.n_rows;
  if (n_cols) *n_cols = m.n_cols;
  if (nnz) *nnz = m.nnz;
  if (!rowptr) return GMG_OK;
  std::vector<int32_t> rp((size_t)m.n_rows + 1);
  HIPC(hipMemcpyAsync(rp.data(), m.rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, ctx->stream));
  if (m.nnz) {
    HIPC(hipMemcpyAsync(col, m.col, sizeof(int32_t) * (size_t)m.nnz, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(val, m.val, sizeof(double) * (size_t)m.nnz, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < rp.size(); ++i) rowptr[i] = rp[i];
  return GMG_OK;
}

int gmg_set_copy_indices(gmg_context *ctx, int level, int64_t n, const int32_t *global_idx, const int32_t *level_idx) {
  if (!ctx || level < 0 || level >= ctx->n_levels || n < 0) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  Level &L = ctx->lv[(size_t)level];
  if (L.copy_g) { (void)hipFree(L.copy_g); L.copy_g = nullptr; }
  if (L.copy_l) { (void)hipFree(L.copy_l); L.copy_l = nullptr; }
  L.n_copy = n;
  if (n == 0) return GMG_OK;
  HIPC(hipMalloc(&L.copy_g, sizeof(int32_t) * (size_t)n));
  HIPC(hipMalloc(&L.copy_l, sizeof(int32_t) * (size_t)n));
  HIPC(hipMemcpyAsync(L.copy_g, global_idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(L.copy_l, level_idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return GMG_OK;
}

int gmg_set_smoother(gmg_context *ctx, int kind, double omega, int steps, int cheb_degree, double cheb_ratio,
                     double cheb_lmax) {
  if (!ctx || kind < 0 || kind > 2 || steps < 0) return GMG_ERR_INVALID;
  ctx->smoother = kind; ctx->omega = omega; ctx->steps = steps;
  if (cheb_degree > 0) ctx->cheb_degree = cheb_degree;
  if (cheb_ratio > 0) ctx->cheb_ratio = cheb_ratio;
  ctx->cheb_lmax_user = cheb_lmax;
  return GMG_OK;
}

int gmg_set_coarse(gmg_context *ctx, double abs_tol, int max_it) {
  if (!ctx || max_it < 1) return GMG_ERR_INVALID;
  ctx->coarse_tol = abs_tol; ctx->coarse_maxit = max_it;
  return GMG_OK;
}

// ---- vectors ----

int gmg_vec_alloc(gmg_context *ctx, int64_t n, double **dptr) {
  if (!ctx || !dptr || n < 0) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  *dptr = nullptr;
  return alloc_vec(ctx, dptr, n);
}
int gmg_vec_free(gmg_context *ctx, double *dptr) {
  if (!ctx) return GMG_ERR_INVALID;
  if (dptr) { HIPC(hipStreamSynchronize(ctx->stream)); HIPC(hipFree(dptr)); }
  return GMG_OK;
}
int gmg_vec_upload(gmg_context *ctx, double *dst, const double *src, int64_t n) {
  if (!ctx || n < 0) return GMG_ERR_INVALID;
  if (n) HIPC(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return GMG_OK;
}
int gmg_vec_download(gmg_context *ctx, double *dst, const double *src, int64_t n) {
  if (!ctx || n < 0) return GMG_ERR_INVALID;
  if (n) HIPC(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return GMG_OK;
}
int gmg_vec_set_zero(gmg_context *ctx, double *x, int64_t n) {
  if (!ctx || n < 0) return GMG_ERR_INVALID;
  if (n) HIPC(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, ctx->stream));
  return GMG_OK;
}
int gmg_vec_equ(gmg_context *ctx, double *y, double a, const double *x, int64_t n) {
  if (!ctx || n < 0) return GMG_ERR_INVALID;
  if (n) hipLaunchKernelGGL(vec_equ_kernel, dim3(grid_for(n)), dim3(kThreads), 0, ctx->stream, y, a, x, n);
  RETURN_LAUNCHED(ctx);
}
int gmg_vec_add(gmg_context *ctx, double *y, double a, const double *x, int64_t n) {
  if (!ctx || n < 0) return GMG_ERR_INVALID;
  if (n) hipLaunchKernelGGL(vec_add_kernel, dim3(grid_for(n)), dim3(kThreads), 0, ctx->stream, y, a, x, n);
  RETURN_LAUNCHED(ctx);
}
int gmg_vec_sadd(gmg_context *ctx, double *y, double s, double a, const double *x, int64_t n) {
  if (!ctx || n < 0) return GMG_ERR_INVALID;
  if (n) hipLaunchKernelGGL(vec_sadd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, ctx->stream, y, s, a, x, n);
  RETURN_LAUNCHED(ctx);
}
int gmg_vec_dot(gmg_context *ctx, const double *x, const double *y, int64_t n, double *out) {
  if (!ctx || !out || n < 0) return GMG_ERR_INVALID;
  return dot_host(ctx, x, y, n, out);
}
int gmg_vec_norms(gmg_context *ctx, const double *x, int64_t n, double *l1, double *l2, double *linf) {
  if (!ctx || n < 0) return GMG_ERR_INVALID;
  const int g = grid_for(n);
  hipLaunchKernelGGL(norms_partial_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, x, n, ctx->part_a);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_a, g, 4, 4u,
                     ctx->scal_dev);
  if (ctx->comm.n_ranks > 1) {
    if (allreduce_sum(ctx->comm, ctx->scal_dev, 2, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
    if (allreduce_max(ctx->comm, ctx->scal_dev + 2, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
    if (allreduce_sum(ctx->comm, ctx->scal_dev + 3, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
  }
  CHK(launch_status(ctx));
  CHK(fetch_scalars(ctx, 4));
  if (l1) *l1 = ctx->scal_host[0];
  if (l2) *l2 = std::sqrt(ctx->scal_host[1]);
  if (linf) *linf = ctx->scal_host[2];
  return GMG_OK;
}
int gmg_vec_all_zero(gmg_context *ctx, const double *x, int64_t n, int *out) {
  if (!ctx || !out) return GMG_ERR_INVALID;
  double a, b, c;
  CHK(gmg_vec_norms(ctx, x, n, &a, &b, &c));
  *out = (ctx->scal_host[3] == 0.0) ? 1 : 0;
  return GMG_OK;
}

// ---- concepts ----

int gmg_spmv(gmg_context *ctx, int which, double *dst, const double *src) {
  if (!ctx) return GMG_ERR_INVALID;
  DevCSR *m = which_matrix(ctx, which);
  if (!m || !m->valid) return fail(ctx, GMG_ERR_INVALID, "gmg_spmv: operator not set");
  if (m->halo.active) {
    // the caller's src holds owned entries only: stage it in a buffer with a ghost tail
    double *tmp = (which == GMG_SYSTEM) ? ctx->S_tmp : ctx->lv[(size_t)which].w3;
    HIPC(hipMemcpyAsync(tmp, src, sizeof(double) * (size_t)m->n_rows, hipMemcpyDeviceToDevice, ctx->stream));
    CHK(import_ghosts(ctx, *m, tmp));
    return spmv(ctx, *m, kStore, tmp, dst);
  }
  return spmv(ctx, *m, kStore, src, dst);
}

int gmg_precondition(gmg_context *ctx, double *dst, const double *src) {
  if (!ctx) return GMG_ERR_INVALID;
  return vcycle(ctx, dst, src);
}

int gmg_precondition_jacobi(gmg_context *ctx, double omega, double *dst, const double *src) {
  if (!ctx || !ctx->S.valid) return GMG_ERR_INVALID;
  hipLaunchKernelGGL(vec_scale_mul_kernel, dim3(grid_for(ctx->S.n_rows)), dim3(kThreads), 0, ctx->stream, dst, omega, src,
                     (const double *)ctx->S_invd, ctx->S.n_rows);
  RETURN_LAUNCHED(ctx);
}

int gmg_coarse_solve(gmg_context *ctx, double *dst, const double *src, int *iterations, double *residual) {
  if (!ctx) return GMG_ERR_INVALID;
  Level &L0 = ctx->lv[0];
  if (!L0.A.valid) return fail(ctx, GMG_ERR_INVALID, "level-0 matrix not set");
  // run on the level's own vectors (ghost tails), then hand the owned part back
  HIPC(hipMemcpyAsync(L0.def, src, sizeof(double) * (size_t)L0.n, hipMemcpyDeviceToDevice, ctx->stream));
  int rc = coarse_solve(ctx, L0.sol, L0.def, iterations, residual);
  HIPC(hipMemcpyAsync(dst, L0.sol, sizeof(double) * (size_t)L0.n, hipMemcpyDeviceToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return rc;
}

int gmg_smoother_step(gmg_context *ctx, int level, double *u, const double *rhs, int from_zero) {
  if (!ctx || level < 0 || level >= ctx->n_levels) return GMG_ERR_INVALID;
  Level &L = ctx->lv[(size_t)level];
  if (!L.A.valid) return fail(ctx, GMG_ERR_INVALID, "level matrix not set");
  HIPC(hipMemcpyAsync(L.sol, u, sizeof(double) * (size_t)L.n, hipMemcpyDeviceToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(L.def, rhs, sizeof(double) * (size_t)L.n, hipMemcpyDeviceToDevice, ctx->stream));
  CHK(smooth_level(ctx, level, &L.sol, L.def, from_zero != 0, &L.w1));
  HIPC(hipMemcpyAsync(u, L.sol, sizeof(double) * (size_t)L.n, hipMemcpyDeviceToDevice, ctx->stream));
  return GMG_OK;
}

int gmg_prolongate(gmg_context *ctx, int level, double *dst_fine, const double *src_coarse) {
  if (!ctx || level < 0 || level >= ctx->n_levels - 1) return GMG_ERR_INVALID;
  Level &L = ctx->lv[(size_t)level];
  if (!L.has_P) return fail(ctx, GMG_ERR_INVALID, "prolongation not set");
  if (L.P.halo.active) {
    HIPC(hipMemcpyAsync(L.w3, src_coarse, sizeof(double) * (size_t)L.n, hipMemcpyDeviceToDevice, ctx->stream));
    CHK(import_ghosts(ctx, L.P, L.w3));
    return spmv(ctx, L.P, kStore, L.w3, dst_fine);
  }
  return spmv(ctx, L.P, kStore, src_coarse, dst_fine);
}

int gmg_restrict_and_add(gmg_context *ctx, int level, double *dst_coarse, const double *src_fine) {
  if (!ctx || level < 0 || level >= ctx->n_levels - 1) return GMG_ERR_INVALID;
  Level &L = ctx->lv[(size_t)level];
  if (!L.has_P) return fail(ctx, GMG_ERR_INVALID, "prolongation not set");
  if (L.Pt.halo.active) {
    Level &F = ctx->lv[(size_t)level + 1];
    HIPC(hipMemcpyAsync(F.w3, src_fine, sizeof(double) * (size_t)F.n, hipMemcpyDeviceToDevice, ctx->stream));
    CHK(import_ghosts(ctx, L.Pt, F.w3));
    return spmv(ctx, L.Pt, kStore, F.w3, dst_coarse, dst_coarse);
  }
  return spmv(ctx, L.Pt, kStore, src_fine, dst_coarse, dst_coarse);
}

// SolverCG<vector_t>::solve, deal.II operation order (see oracle/gmg_oracle.c:cg_solve).
int gmg_cg_solve(gmg_context *ctx, double *x, const double *b, double rel_tol, int max_it, int precond, int *iterations,
                 double *starting_value, double *convergence_value) {
  if (!ctx || !ctx->S.valid) return GMG_ERR_INVALID;
  const int64_t n = ctx->S.n_rows;
  double *g = nullptr, *d = nullptr, *h = nullptr;
  CHK(alloc_vec(ctx, &g, ctx->S.n_cols));
  CHK(alloc_vec(ctx, &d, ctx->S.n_cols));
  CHK(alloc_vec(ctx, &h, ctx->S.n_cols));
  auto cleanup = [&]() { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(g); (void)hipFree(d); (void)hipFree(h); };
  auto apply_precond = [&](double *dst, const double *src) -> int {
    if (precond == GMG_PRECOND_GMG) return vcycle(ctx, dst, src);
    if (precond == GMG_PRECOND_JACOBI) return gmg_precondition_jacobi(ctx, 0.6, dst, src);
    return GMG_ERR_INVALID;
  };
  int rc = GMG_OK;
  double bb = 0.0, res = 0.0, gh = 0.0, alpha = 0.0, beta = 0.0, tmp = 0.0;
  int it = 0, zero = 0;
  do {
    if ((rc = dot_host(ctx, b, b, n, &bb))) break;
    const double tol = rel_tol * std::sqrt(bb);
    if ((rc = gmg_vec_all_zero(ctx, x, n, &zero))) break;
    if (!zero) {
      if ((rc = gmg_spmv(ctx, GMG_SYSTEM, g, x))) break;
      gmg_vec_add(ctx, g, -1.0, b, n);
    } else {
      gmg_vec_equ(ctx, g, -1.0, b, n);
    }
    if ((rc = dot_host(ctx, g, g, n, &tmp))) break;
    res = std::sqrt(tmp);
    if (starting_value) *starting_value = res;
    if (res <= tol) break;
    if (precond != GMG_PRECOND_IDENTITY) {
      if ((rc = apply_precond(h, g))) break;
      gmg_vec_equ(ctx, d, -1.0, h, n);
      if ((rc = dot_host(ctx, g, h, n, &gh))) break;
    } else {
      gmg_vec_equ(ctx, d, -1.0, g, n);
      gh = res * res;
    }
    for (;;) {
      ++it;
      if ((rc = gmg_spmv(ctx, GMG_SYSTEM, h, d))) break;
      if ((rc = dot_host(ctx, d, h, n, &alpha))) break;
      alpha = gh / alpha;
      gmg_vec_add(ctx, x, alpha, d, n);
      gmg_vec_add(ctx, g, alpha, h, n);
      if ((rc = dot_host(ctx, g, g, n, &tmp))) break;
      res = std::sqrt(tmp);
      if (res <= tol) break;
      if (it >= max_it || res != res) { rc = GMG_ERR_OUTER_NOCONV; ctx->err = "outer CG did not converge within max_it"; break; }
      if (precond != GMG_PRECOND_IDENTITY) {
        if ((rc = apply_precond(h, g))) break;
        beta = gh;
        if ((rc = dot_host(ctx, g, h, n, &gh))) break;
        beta = gh / beta;
        gmg_vec_sadd(ctx, d, beta, -1.0, h, n);
      } else {
        beta = gh;
        gh = res * res;
        beta = gh / beta;
        gmg_vec_sadd(ctx, d, beta, -1.0, g, n);
      }
    }
  } while (0);
  if (iterations) *iterations = it;
  if (convergence_value) *convergence_value = res;
  cleanup();
  return rc;
}

// ---- N1: charge density ----

int gmg_charge_density(gmg_context *ctx, int64_t n_cells, const double *cell_lo, const double *cell_h,
                       const double *root_lo, double root_h, int64_t n_atoms, const double *atom_xyz,
                       const double *atom_q, double r_c, double cutoff, int use_lists, int nq,
                       const double *quadrature_points, double *dens) {
  if (!ctx || n_cells < 0 || n_atoms < 0 || nq < 1 || n_cells >= ((int64_t)1 << 31)) return GMG_ERR_INVALID;
  if (n_cells == 0) return GMG_OK;
  (void)hipSetDevice(ctx->device);
  // bins of edge `cutoff` over the atoms' bounding box (host; 64 k atoms: microseconds)
  DensityArgs a{};
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int d = 0; d < 3; ++d) { lo[d] = 1e300; hi[d] = -1e300; }
  for (int64_t i = 0; i < n_atoms; ++i)
    for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], atom_xyz[3 * i + d]); hi[d] = std::max(hi[d], atom_xyz[3 * i + d]); }
  if (n_atoms == 0) { for (int d = 0; d < 3; ++d) lo[d] = hi[d] = 0; }
  const double bs = std::max(cutoff, 1e-12);
  int bn[3];
  int64_t total = 1;
  for (int d = 0; d < 3; ++d) { bn[d] = std::max(1, (int)std::floor((hi[d] - lo[d]) / bs) + 1); total *= bn[d]; }
  if (total > ((int64_t)1 << 28)) return fail(ctx, GMG_ERR_UNSUPPORTED, "atom bin grid too large");
  std::vector<int32_t> bptr((size_t)total + 1, 0), bitems((size_t)std::max<int64_t>(n_atoms, 1));
  auto bin_of = [&](int64_t i) {
    int b[3];
    for (int d = 0; d < 3; ++d) b[d] = std::min(bn[d] - 1, std::max(0, (int)std::floor((atom_xyz[3 * i + d] - lo[d]) / bs)));
    return (int64_t)b[0] + bn[0] * ((int64_t)b[1] + (int64_t)bn[1] * b[2]);
  };
  for (int64_t i = 0; i < n_atoms; ++i) bptr[(size_t)bin_of(i) + 1]++;
  for (int64_t b = 0; b < total; ++b) bptr[(size_t)b + 1] += bptr[(size_t)b];
  {
    std::vector<int32_t> pos(bptr.begin(), bptr.end() - 1);
    for (int64_t i = 0; i < n_atoms; ++i) bitems[(size_t)pos[(size_t)bin_of(i)]++] = (int32_t)i;
  }
  double *d_lo = nullptr, *d_h = nullptr, *d_root = nullptr, *d_xyz = nullptr, *d_q = nullptr, *d_qp = nullptr, *d_dens = nullptr;
  int32_t *d_bptr = nullptr, *d_bitems = nullptr;
  auto cleanup = [&]() {
    for (void *p : {(void *)d_lo, (void *)d_h, (void *)d_root, (void *)d_xyz, (void *)d_q, (void *)d_qp, (void *)d_dens, (void *)d_bptr, (void *)d_bitems})
      if (p) (void)hipFree(p);
  };
#define UP(dst, src, bytes)                                                                   \
  do {                                                                                        \
    if (hipMalloc(&dst, std::max<size_t>((bytes), 8)) != hipSuccess ||                         \
        ((bytes) && hipMemcpyAsync(dst, src, (bytes), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)) { \
      cleanup();                                                                              \
      return fail(ctx, GMG_ERR_HIP, "gmg_charge_density: upload failed");                     \
    }                                                                                         \
  } while (0)
  UP(d_lo, cell_lo, sizeof(double) * 3 * (size_t)n_cells);
  UP(d_h, cell_h, sizeof(double) * (size_t)n_cells);
  UP(d_root, root_lo, sizeof(double) * 3 * (size_t)n_cells);
  UP(d_xyz, atom_xyz, sizeof(double) * 3 * (size_t)n_atoms);
  UP(d_q, atom_q, sizeof(double) * (size_t)n_atoms);
  UP(d_qp, quadrature_points, sizeof(double) * 3 * (size_t)nq);
  UP(d_bptr, bptr.data(), sizeof(int32_t) * bptr.size());
  UP(d_bitems, bitems.data(), sizeof(int32_t) * bitems.size());
#undef UP
  if (hipMalloc(&d_dens, sizeof(double) * (size_t)n_cells * (size_t)nq) != hipSuccess) { cleanup(); return fail(ctx, GMG_ERR_HIP, "gmg_charge_density: out of memory"); }
  if (ctx->dens_dev) { (void)hipFree(ctx->dens_dev); ctx->dens_dev = nullptr; ctx->dens_cells = 0; ctx->dens_nq = 0; }
  a.cell_lo = d_lo; a.cell_h = d_h; a.root_lo = d_root; a.root_h = root_h;
  a.atom_xyz = d_xyz; a.atom_q = d_q; a.n_atoms = (int)n_atoms;
  a.bin_lo0 = lo[0]; a.bin_lo1 = lo[1]; a.bin_lo2 = lo[2]; a.bin_size = bs;
  a.bin_n0 = bn[0]; a.bin_n1 = bn[1]; a.bin_n2 = bn[2];
  a.bin_ptr = d_bptr; a.bin_items = d_bitems;
  a.cutoff = cutoff; a.r_c = r_c; a.use_lists = use_lists;
  a.qp = d_qp; a.nq = nq; a.n_cells = (int)n_cells; a.dens = d_dens;
  hipLaunchKernelGGL(charge_density_kernel, dim3((unsigned)((n_cells + 3) / 4)), dim3(kThreads), 0, ctx->stream, a);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && dens) e = hipMemcpyAsync(dens, d_dens, sizeof(double) * (size_t)n_cells * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && !dens) {  // the densities stay in HBM for gmg_rhs_assemble (gmg_get_charge_density copies them out on demand)
    ctx->dens_dev = d_dens; ctx->dens_cells = n_cells; ctx->dens_nq = nq;
    d_dens = nullptr;
  }
  cleanup();
  if (e != hipSuccess) { ctx->err = std::string("gmg_charge_density: ") + hipGetErrorString(e); return GMG_ERR_HIP; }
  return GMG_OK;
}

int gmg_get_charge_density(gmg_context *ctx, int64_t n_cells, int nq, double *dens) {
  if (!ctx || !dens) return GMG_ERR_INVALID;
  if (!ctx->dens_dev || ctx->dens_cells != n_cells || ctx->dens_nq != nq) return fail(ctx, GMG_ERR_INVALID, "gmg_get_charge_density: no densities of this shape on the device");
  HIPC(hipMemcpyAsync(dens, ctx->dens_dev, sizeof(double) * (size_t)n_cells * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return GMG_OK;
}

// assemble_system's right-hand side (src/step-50.cc:813-828) from densities that never left the device: per cell
// F_i = sum_q phi_i(x_q) rho(x_q) w_q JxW (:813-820), the Dirichlet terms -K_ij g_j (:825-828) as a list of subtractions, and
// the scatter into the DoFs as a per-DoF gather over the (cell, vertex) slots in the reference's cell order with the
// hanging-node weights -- sequential per DoF: deterministic, no fp64 atomics, the same bits as the host loop.
int gmg_rhs_assemble(gmg_context *ctx, int64_t n_cells, int nq, int dim, const double *shape, const double *weight, const uint8_t *cell_level,
                     const double *jxw_of_level, int64_t n_terms, const int32_t *term_slot, const double *term_value, int64_t n_dofs,
                     const int64_t *dof_ptr, const int32_t *entry_slot, const uint8_t *entry_coef, const double *coef_table, double *rhs) {
  if (!ctx || n_cells < 0 || nq < 1 || nq > 512 || (dim != 2 && dim != 3) || !shape || !weight || !cell_level || !jxw_of_level || n_terms < 0 || n_dofs < 0 ||
      !dof_ptr || !coef_table || !rhs)
    return GMG_ERR_INVALID;
  if (!ctx->dens_dev || ctx->dens_cells != n_cells || ctx->dens_nq != nq)
    return fail(ctx, GMG_ERR_INVALID, "gmg_rhs_assemble: call gmg_charge_density(..., dens = NULL) for these cells first");
  const int nv = 1 << dim;
  const int64_t n_slots = n_cells * nv, n_ent = dof_ptr[n_dofs];
  if (n_slots >= ((int64_t)1 << 31) || n_ent >= ((int64_t)1 << 31)) return fail(ctx, GMG_ERR_UNSUPPORTED, "gmg_rhs_assemble: more than 2^31 slots");
  (void)hipSetDevice(ctx->device);
  double *d_F = nullptr, *d_tv = nullptr;
  uint8_t *d_lv = nullptr, *d_ec = nullptr;
  int32_t *d_ts = nullptr, *d_es = nullptr, *d_ptr = nullptr;
  RhsArgs a{};
  auto cleanup = [&]() {
    for (void *q : {(void *)d_F, (void *)d_tv, (void *)d_lv, (void *)d_ec, (void *)d_ts, (void *)d_es, (void *)d_ptr})
      if (q) (void)hipFree(q);
  };
#define RHC(call) do { if ((call) != hipSuccess) { cleanup(); return fail(ctx, GMG_ERR_HIP, "gmg_rhs_assemble: " #call " failed"); } } while (0)
  std::vector<int32_t> ptr32((size_t)n_dofs + 1);
  for (int64_t i = 0; i <= n_dofs; ++i) {
    if (i && dof_ptr[i] < dof_ptr[i - 1]) return fail(ctx, GMG_ERR_INVALID, "gmg_rhs_assemble: dof_ptr not monotone");
    ptr32[(size_t)i] = (int32_t)dof_ptr[i];
  }
  for (int64_t e = 0; e < n_ent; ++e)
    if (entry_slot[e] < 0 || entry_slot[e] >= n_slots) return fail(ctx, GMG_ERR_INVALID, "gmg_rhs_assemble: slot out of range");
  for (int64_t t = 0; t < n_terms; ++t)
    if (term_slot[t] < 0 || term_slot[t] >= n_slots || (t && term_slot[t] < term_slot[t - 1])) return fail(ctx, GMG_ERR_INVALID, "gmg_rhs_assemble: term slots must ascend inside the slot range");
  RHC(hipMalloc(&d_F, sizeof(double) * (size_t)std::max<int64_t>(n_slots, 1)));
  RHC(hipMalloc(&d_lv, (size_t)std::max<int64_t>(n_cells, 1)));
  RHC(hipMalloc(&d_ptr, sizeof(int32_t) * ptr32.size()));
  RHC(hipMalloc(&d_es, sizeof(int32_t) * (size_t)std::max<int64_t>(n_ent, 1)));
  RHC(hipMalloc(&d_ec, (size_t)std::max<int64_t>(n_ent, 1)));
  RHC(hipMalloc(&d_ts, sizeof(int32_t) * (size_t)std::max<int64_t>(n_terms, 1)));
  RHC(hipMalloc(&d_tv, sizeof(double) * (size_t)std::max<int64_t>(n_terms, 1)));
  if (n_cells) RHC(hipMemcpyAsync(d_lv, cell_level, (size_t)n_cells, hipMemcpyHostToDevice, ctx->stream));
  RHC(hipMemcpyAsync(d_ptr, ptr32.data(), sizeof(int32_t) * ptr32.size(), hipMemcpyHostToDevice, ctx->stream));
  if (n_ent) {
    RHC(hipMemcpyAsync(d_es, entry_slot, sizeof(int32_t) * (size_t)n_ent, hipMemcpyHostToDevice, ctx->stream));
    RHC(hipMemcpyAsync(d_ec, entry_coef, (size_t)n_ent, hipMemcpyHostToDevice, ctx->stream));
  }
  if (n_terms) {
    RHC(hipMemcpyAsync(d_ts, term_slot, sizeof(int32_t) * (size_t)n_terms, hipMemcpyHostToDevice, ctx->stream));
    RHC(hipMemcpyAsync(d_tv, term_value, sizeof(double) * (size_t)n_terms, hipMemcpyHostToDevice, ctx->stream));
  }
  a.dens = ctx->dens_dev; a.n_cells = n_cells; a.nq = nq; a.nv = nv; a.cell_level = d_lv; a.F = d_F;
  for (int q = 0; q < nq; ++q) {
    a.weight[q] = weight[q];
    for (int i = 0; i < nv; ++i) a.shape[q * 8 + i] = shape[q * nv + i];
  }
  for (int l = 0; l < 16; ++l) a.jxw[l] = jxw_of_level[l];
  a.n_terms = n_terms; a.term_slot = d_ts; a.term_value = d_tv;
  a.n_dofs = n_dofs; a.dof_ptr = d_ptr; a.entry_slot = d_es; a.entry_coef = d_ec; a.rhs = rhs;
  for (int c = 0; c < 256; ++c) a.coef[c] = coef_table[c];
  RhsArgs *d_a = nullptr;  // (the argument block is 39 KB: it travels through memory, not through the kernel-argument segment)
  RHC(hipMalloc(&d_a, sizeof(RhsArgs)));
  hipError_t e = hipMemcpyAsync(d_a, &a, sizeof(RhsArgs), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    if (n_cells) hipLaunchKernelGGL(rhs_cell_kernel, dim3(grid_for(n_cells)), dim3(kThreads), 0, ctx->stream, (const RhsArgs *)d_a);
    if (n_terms) hipLaunchKernelGGL(rhs_terms_kernel, dim3(grid_for(n_terms)), dim3(kThreads), 0, ctx->stream, (const RhsArgs *)d_a);
    if (n_dofs) hipLaunchKernelGGL(rhs_gather_kernel, dim3(grid_for(n_dofs)), dim3(kThreads), 0, ctx->stream, (const RhsArgs *)d_a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_a);
#undef RHC
  cleanup();
  if (e != hipSuccess) { ctx->err = std::string("gmg_rhs_assemble: ") + hipGetErrorString(e); return GMG_ERR_HIP; }
  return GMG_OK;
}

// ---- distributed ----

int gmg_comm_unique_id(void *out_id) { return comm_unique_id(out_id) ? GMG_ERR_COMM : GMG_OK; }

int gmg_comm_init(gmg_context *ctx, int rank, int n_ranks, const void *id) {
  if (!ctx || rank < 0 || n_ranks < 1 || rank >= n_ranks) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  if (comm_init(ctx->comm, rank, n_ranks, id)) {
    ctx->err = ctx->comm.why[0] ? std::string(ctx->comm.why) : std::string("communicator start-up failed (ncclCommInitRank / peer mailbox mapping)");
    return GMG_ERR_COMM;
  }
  ctx->dist = true;
  return GMG_OK;
}

int gmg_comm_info(gmg_context *ctx, int64_t out[8]) {
  if (!ctx || !out) return GMG_ERR_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  out[0] = ctx->dist ? ctx->comm.n_ranks : 1;
  out[1] = !ctx->dist ? 0 : (ctx->comm.peer ? 2 : 1);
  out[2] = ctx->comm.peer && ctx->comm.box_fine ? 1 : 0;
  out[3] = ctx->comm.peer ? ctx->comm.ring_fine : -1;
  out[4] = ctx->dist ? ctx->comm.n_devices : 1;
  out[5] = l0_partitioned(ctx) ? 1 : 0;
  return GMG_OK;
}

int gmg_comm_barrier(gmg_context *ctx) {
  if (!ctx) return GMG_ERR_INVALID;
  if (!ctx->dist) return GMG_OK;
  HIPC(hipStreamSynchronize(ctx->stream));
  if (comm_host_barrier(ctx->comm)) return fail(ctx, GMG_ERR_COMM, "host barrier: a rank did not arrive");
  return GMG_OK;
}

int gmg_set_global_sizes(gmg_context *ctx, int64_t n_system_global, int64_t n_level0_global) {
  if (!ctx || n_system_global < 0 || n_level0_global < 0) return GMG_ERR_INVALID;
  if (!ctx->dist) return fail(ctx, GMG_ERR_INVALID, "gmg_set_global_sizes: call gmg_comm_init first");
  ctx->sys_global = n_system_global;
  ctx->l0_global = n_level0_global;
  return GMG_OK;
}

int gmg_partition_range(int64_t n_global, int rank, int n_ranks, int64_t *begin, int64_t *end) {
  if (n_global < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks || !begin || !end) return GMG_ERR_INVALID;
  part_range(n_global, rank, n_ranks, begin, end);
  return GMG_OK;
}

int gmg_vec_allgather(gmg_context *ctx, int64_t n_global, double *dst_full, const double *src_local) {
  if (!ctx || n_global < 0) return GMG_ERR_INVALID;
  if (!ctx->dist) {
    if (n_global) HIPC(hipMemcpyAsync(dst_full, src_local, sizeof(double) * (size_t)n_global, hipMemcpyDeviceToDevice, ctx->stream));
    return GMG_OK;
  }
  return allgather_full(ctx, dst_full, src_local, n_global);
}

int gmg_set_halo_plan(gmg_context *ctx, int which, int n_neighbors, const int32_t *neighbor_rank,
                      const int32_t *send_count, const int32_t *send_idx, const int32_t *recv_count) {
  if (!ctx) return GMG_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  // `which`: GMG_SYSTEM, level l (A_l, I_l share it), 1000+l (P_l), 2000+l (P_l^T), 3000+l (I_l^T)
  std::vector<DevCSR *> targets;
  if (which == GMG_SYSTEM) targets = {&ctx->S};
  else if (which >= 0 && which < ctx->n_levels) targets = {&ctx->lv[(size_t)which].A, &ctx->lv[(size_t)which].I};
  else if (which >= 1000 && which < 1000 + ctx->n_levels) targets = {&ctx->lv[(size_t)which - 1000].P};
  else if (which >= 2000 && which < 2000 + ctx->n_levels) targets = {&ctx->lv[(size_t)which - 2000].Pt};
  else if (which >= 3000 && which < 3000 + ctx->n_levels) targets = {&ctx->lv[(size_t)which - 3000].It};
  else return fail(ctx, GMG_ERR_INVALID, "gmg_set_halo_plan: bad operator id");
  for (DevCSR *m : targets) {
    free_halo(m->halo);
    if (build_halo(m->halo, n_neighbors, neighbor_rank, send_count, send_idx, recv_count, ctx->stream))
      return fail(ctx, GMG_ERR_HIP, "halo plan upload failed");
  }
  return GMG_OK;
}

// ---- measurement ----

int gmg_stats_reset(gmg_context *ctx) {
  if (!ctx) return GMG_ERR_INVALID;
  collect_sgs_samples(ctx);
  const gmg_stats keep = ctx->stats;
  ctx->stats = gmg_stats{};
  ctx->stats.spmv0_rows = keep.spmv0_rows; ctx->stats.spmv0_nnz = keep.spmv0_nnz; ctx->stats.coarse_variant = keep.coarse_variant;
  ctx->stats.spmv0_layout = keep.spmv0_layout; ctx->stats.spmv0_matrix_bytes = keep.spmv0_matrix_bytes;
  ctx->stats.spmv0_pattern_slices = keep.spmv0_pattern_slices; ctx->stats.spmv0_slices = keep.spmv0_slices;
  return GMG_OK;
}
int gmg_stats_get(gmg_context *ctx, gmg_stats *out) {
  if (!ctx || !out) return GMG_ERR_INVALID;
  collect_sgs_samples(ctx);
  *out = ctx->stats;
  return GMG_OK;
}
int gmg_set_profiling(gmg_context *ctx, int sample_every) {
  if (!ctx || sample_every < 0) return GMG_ERR_INVALID;
  ctx->prof_every = sample_every;
  if (sample_every > 0 && ctx->ev_a.empty()) {
    for (auto *v : {&ctx->ev_a, &ctx->ev_b, &ctx->ev_c, &ctx->ev_d, &ctx->ev_e, &ctx->ev_f}) {
      v->resize(256);
      for (auto &e : *v) HIPC(hipEventCreate(&e));
    }
  }
  return GMG_OK;
}
int gmg_calibrate_hbm(gmg_context *ctx, int64_t n_bytes, int reps, double *read_gbps, double *copy_gbps) {
  if (!ctx || n_bytes < (1 << 20) || reps < 1) return GMG_ERR_INVALID;
  const int64_t n2 = n_bytes / 16;
  double2 *a = nullptr, *b = nullptr;
  HIPC(hipMalloc(&a, (size_t)n2 * 16));
  HIPC(hipMalloc(&b, (size_t)n2 * 16));
  HIPC(hipMemsetAsync(a, 0, (size_t)n2 * 16, ctx->stream));
  HIPC(hipMemsetAsync(b, 0, (size_t)n2 * 16, ctx->stream));
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  float ms = 0.f;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms up
    HIPC(hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL(stream_read_kernel, dim3(kMaxPartials), dim3(kThreads), 0, ctx->stream, (const double2 *)a, n2, ctx->part_a);
    HIPC(hipEventRecord(e1, ctx->stream));
    HIPC(hipEventSynchronize(e1));
    HIPC(hipEventElapsedTime(&ms, e0, e1));
  }
  if (read_gbps) *read_gbps = (double)n2 * 16 * reps / (ms * 1e-3) / 1e9;
  for (int pass = 0; pass < 2; ++pass) {
    HIPC(hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL(stream_copy_kernel, dim3(kMaxPartials * 2), dim3(kThreads), 0, ctx->stream, (const double2 *)a, b, n2);
    HIPC(hipEventRecord(e1, ctx->stream));
    HIPC(hipEventSynchronize(e1));
    HIPC(hipEventElapsedTime(&ms, e0, e1));
  }
  if (copy_gbps) *copy_gbps = (double)n2 * 32 * reps / (ms * 1e-3) / 1e9;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  return GMG_OK;
}

int gmg_set_option(gmg_context *ctx, const char *key, double value) {
  if (!ctx || !key) return GMG_ERR_INVALID;
  const std::string k(key);
  const bool on = value != 0.0;
  if (k == "host_threads") g_host_threads = (int)value;
  else if (k == "debug_upload") ctx->debug_upload = on;
  else if (k == "disable_sell") ctx->disable_sell = on;
  else if (k == "disable_patterns") ctx->disable_patterns = on;
  else if (k == "disable_compression") ctx->disable_compression = on;
  else if (k == "disable_sellp") ctx->disable_sellp = on;
  else if (k == "disable_rowclass") ctx->disable_rowclass = on;
  else if (k == "disable_lattice") ctx->disable_lattice = on;
  else if (k == "lattice_segments") ctx->lattice_segments = (int)value;
  else if (k == "lattice_max_blocks") ctx->lattice_max_blocks = (int)value;
  else if (k == "sell_grid") ctx->sell_grid = (int)value;
  else if (k == "sellp_cost") ctx->sellp_cost = value;
  else if (k == "sellp_rr") ctx->sellp_rr = (int)value;
  else if (k == "cg_variant") ctx->cg_variant = (int)value;
  else if (k == "coarse_chunk") ctx->coarse_chunk = (int)value;
  else if (k == "sgs_y_slots") ctx->sgs_y_slots = (int)value;
  else if (k == "sgs_disable_wave") ctx->sgs_disable_wave = on;
  else if (k == "sgs_disable_phase") ctx->sgs_disable_phase = on;
  else if (k == "sgs_dep") ctx->sgs_dep = on;
  else if (k == "sgs_reg") ctx->sgs_reg = on;
  else if (k == "sgs_pf_lead_kb") ctx->sgs_pf_lead_kb = (int)value;
  else if (k == "sgs_chain") {
#ifndef GMG_EXPERIMENTS
    if (on) return fail(ctx, GMG_ERR_UNSUPPORTED, "sgs_chain needs a -DGMG_EXPERIMENTS build (tools/build_experiments.sh)");
#endif
    ctx->sgs_chain = on;
  }
  else if (k == "sgs_phase_profile") {
    if (on && ctx->dist && ctx->comm.n_ranks > 1) return fail(ctx, GMG_ERR_INVALID, "sgs_phase_profile: one rank only (the instrumented sweep is a single-GPU measurement)");
    ctx->sgs_phase_profile = (int)value;
  }
  else if (k == "sgs_phase_nosplit") ctx->sgs_phase_nosplit = on;
  else if (k == "sgs_phase_nocascade") ctx->sgs_phase_nocascade = on;
  else if (k == "sgs_phase_chunk") ctx->sgs_phase_chunk = (int)value;
  else if (k == "sgs_groups") ctx->sgs_groups = (int)value;
  else if (k == "sgs_lds_bytes_override") ctx->sgs_lds_bytes_override = (int)value;
  else if (k == "sgs_profile") {
    if (on && ctx->dist && ctx->comm.n_ranks > 1) return fail(ctx, GMG_ERR_INVALID, "sgs_profile: one rank only");
    ctx->sgs_profile = on;
    ctx->sgs_profile_mode = (int)value - 1;
#ifndef GMG_EXPERIMENTS
    // the timing modes that skip work (and give wrong results) are not in this library: tools/build_experiments.sh
    if (ctx->sgs_profile_mode > 0) return fail(ctx, GMG_ERR_UNSUPPORTED, "sgs_profile modes > 1 need a -DGMG_EXPERIMENTS build (tools/build_experiments.sh)");
#endif
  }
  else return fail(ctx, GMG_ERR_INVALID, "gmg_set_option: unknown key");
  return GMG_OK;
}

int gmg_set_tuning(gmg_context *ctx, int coarse_chunk, int cg_variant) {
  if (!ctx || coarse_chunk < 0 || cg_variant < 0 || cg_variant > 2) return GMG_ERR_INVALID;
  ctx->coarse_chunk = coarse_chunk;
  ctx->cg_variant = cg_variant;
  return GMG_OK;
}

int gmg_set_ssor_blocks(gmg_context *ctx, int n_blocks) {
  if (!ctx || n_blocks < 1) return GMG_ERR_INVALID;
  ctx->ssor_blocks = n_blocks;
  return GMG_OK;
}

}  // extern "C"
