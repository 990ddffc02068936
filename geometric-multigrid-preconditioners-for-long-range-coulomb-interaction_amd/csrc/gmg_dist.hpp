// gmg_dist.hpp -- coarse CG with the direction update as its own kernel.
//
// Same SolverCG operation order as the fused path (gmg_device.hpp), used in two situations:
//   * row-partitioned level 0 (one process per GPU): the ghost entries of d must be imported
//     between the direction update and the SpMV; every reduction goes workgroup partials ->
//     one device scalar -> ncclAllReduce (sum, fp64, count 1) -> read by the next kernel as a
//     one-element "partials" array.  No host synchronisation inside a chunk of iterations.
//   * large single-GPU level 0 (>= kUnfusedMinRows): the fused kernel reads d AND g for every
//     operand, which costs more vector-load issue than streaming 24 N bytes once (81^3 level 0:
//     22.3 vs 15.9 ms per solve); small problems keep the fused 2-kernel iteration (launch-bound).
// The update kernel of this variant touches g only; x += alpha d is applied kXRing iterations at
// a time from a ring of direction vectors (cg_xflush_kernel): same additions in the same order.
// Included by gmg_coulomb.hip.
#pragma once

namespace {

constexpr int64_t kUnfusedMinRows = 200000;

int coarse_solve_unfused(gmg_context *ctx, double *x, const double *b, int *iters_out, double *res_out) {
  Level &L0 = ctx->lv[0];
  DevCSR &A = L0.A;
  const bool comm = l0_partitioned(ctx);
  const int64_t n = L0.n;
  const int g_vec = grid_for(n);
  const int g_upd = grid_for(n / 2);
  double *s_gg = ctx->scal_dev + 4, *s_dh = ctx->scal_dev + 5;
  // ring of direction vectors (with ghost tails): x += alpha d is applied kXRing iterations at a time
  if (!ctx->cg_ring[0]) {
    for (int r = 0; r < kXRing; ++r) CHK(alloc_vec(ctx, &ctx->cg_ring[r], A.n_cols));
    ctx->cg_ring_len = A.n_cols;
  }
  if (ctx->cg_ring_len < A.n_cols) return fail(ctx, GMG_ERR_INVALID, "coarse CG ring sized for a smaller operator");

  // the direction read by iteration 0 (beta = 0) must be finite: the init kernel zeroes it
  CGInitArgs ia{b, x, ctx->cg_g, ctx->cg_ring[kXRing - 1], ctx->cg_ring[kXRing - 1], n, ctx->st, ctx->part_b};
  hipLaunchKernelGGL(cg_init_kernel, dim3(g_vec), dim3(kThreads), 0, ctx->stream, ia);
  // where the consumers find the reduced scalars: all-reduced single values, or the raw partials
  const double *gg_src = ctx->part_b;
  int gg_n = g_vec;
  if (comm) {
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_b, g_vec, 1, 0u, s_gg);
    if (allreduce_sum(ctx->comm, s_gg, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
    gg_src = s_gg; gg_n = 1;
  }

  const int maxit = ctx->coarse_maxit;
  ctx->ev_used = 0; ctx->ev2_used = 0;
  auto flush_x = [&](int lo, int upto) {
    CGXFlushArgs fa{};
    fa.x = x; fa.n = n; fa.st = ctx->st; fa.lo = lo; fa.upto = upto;
    for (int r = 0; r < kXRing; ++r) fa.ring[r] = ctx->cg_ring[r];
    hipLaunchKernelGGL(cg_xflush_kernel, dim3(g_upd), dim3(kThreads), 0, ctx->stream, fa);
  };
  int launched_total = 0;
  CHK(run_cg_chunks(ctx, (!comm && L0.n >= kUnfusedMinRows) ? 3 : 6, [&](int launched) -> int {
    launched_total = launched + 1;
    double *d = ctx->cg_ring[launched % kXRing];
    CGDirArgs da{d, ctx->cg_ring[(launched + kXRing - 1) % kXRing], ctx->cg_g, n, ctx->st, gg_src, gg_n, ctx->coarse_tol, maxit};
    hipLaunchKernelGGL(cg_direction_kernel, dim3(g_vec), dim3(kThreads), 0, ctx->stream, da);
    if (comm && halo_exchange(ctx->comm, A.halo, d, n, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "halo exchange failed");
    SpmvArgs a = base_args(A, d, ctx->cg_h);
    a.st = ctx->st;
    a.part_out = ctx->part_a;
    const bool sample = ctx->prof_every > 0 && (launched % ctx->prof_every) == 0 && ctx->ev_used < (int)ctx->ev_a.size();
    if (sample) { ctx->timed_start = ctx->ev_a[(size_t)ctx->ev_used]; ctx->timed_stop = ctx->ev_b[(size_t)ctx->ev_used++]; }
    const int n_part_dh = launch_op<kStore, 2>(ctx, A, a);
    const double *dh_src = ctx->part_a;
    int dh_n = n_part_dh;
    if (comm) {
      hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_a, n_part_dh, 1, 0u, s_dh);
      if (allreduce_sum(ctx->comm, s_dh, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
      dh_src = s_dh; dh_n = 1;
    }
    CGUpdateGArgs ua{ctx->cg_g, ctx->cg_h, n, ctx->st, dh_src, dh_n, ctx->part_b};
    const bool sample2 = sample && ctx->ev2_used < (int)ctx->ev_c.size();
    if (sample2) { ctx->timed_start = ctx->ev_c[(size_t)ctx->ev2_used]; ctx->timed_stop = ctx->ev_d[(size_t)ctx->ev2_used++]; }
    launch_timed(ctx, cg_update_g_kernel, dim3(g_upd), dim3(kThreads), 0, ua);
    if ((launched + 1) % kXRing == 0) flush_x(launched + 1 - kXRing, launched + 1);
    if (comm) {
      hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_b, g_upd, 1, 0u, s_gg);
      if (allreduce_sum(ctx->comm, s_gg, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
    } else {
      gg_src = ctx->part_b; gg_n = g_upd;
    }
    return GMG_OK;
  }));
  // iterations not covered by an enqueued pass: [kXRing * floor(launched / kXRing), completed)
  flush_x(launched_total / kXRing * kXRing, INT_MAX);
  collect_profile_samples(ctx);
  ctx->last_coarse_iters = ctx->st_final.iters;
  ctx->stats.coarse_solves++;
  ctx->stats.coarse_iterations += ctx->st_final.iters;
  if (iters_out) *iters_out = ctx->st_final.iters;
  if (res_out) *res_out = ctx->st_final.res;
  if (ctx->st_final.status != 0) return fail(ctx, GMG_ERR_COARSE_NOCONV, "coarse CG did not converge within max_it");
  return GMG_OK;
}

}  // namespace
