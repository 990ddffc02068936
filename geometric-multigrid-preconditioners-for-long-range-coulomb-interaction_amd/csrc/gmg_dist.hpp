// gmg_dist.hpp -- coarse CG on a row-partitioned level-0 operator (one process per GPU).
//
// Same SolverCG operation order as the single-GPU path (gmg_device.hpp), but the direction
// update is its own kernel so that the ghost entries of d can be imported between it and the
// SpMV, and every reduction goes workgroup partials -> one device scalar -> ncclAllReduce
// (sum, fp64, count 1) -> read by the next kernel as a one-element "partials" array.  No host
// synchronisation inside a chunk of iterations.  Included by gmg_coulomb.hip.
#pragma once

namespace {

int coarse_solve_distributed(gmg_context *ctx, double *x, const double *b, int *iters_out, double *res_out) {
  Level &L0 = ctx->lv[0];
  DevCSR &A = L0.A;
  const int64_t n = L0.n;
  const int g_vec = grid_for(n);
  const int g_upd = grid_for(n / 2);
  double *s_gg = ctx->scal_dev + 4, *s_dh = ctx->scal_dev + 5;
  double *d = ctx->cg_d0;

  CGInitArgs ia{b, x, ctx->cg_g, ctx->cg_d0, ctx->cg_d1, n, ctx->st, ctx->part_b};
  hipLaunchKernelGGL(cg_init_kernel, dim3(g_vec), dim3(kThreads), 0, ctx->stream, ia);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_b, g_vec, 1, 0u, s_gg);
  if (allreduce_sum(ctx->comm, s_gg, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");

  const int maxit = ctx->coarse_maxit;
  int launched = 0;
  int chunk = ctx->coarse_chunk > 0 ? ctx->coarse_chunk : (ctx->last_coarse_iters > 8 ? ctx->last_coarse_iters - 2 : 16);
  for (;;) {
    int todo = std::min(chunk, maxit + 1 - launched);
    if (todo <= 0) todo = 1;
    for (int q = 0; q < todo; ++q, ++launched) {
      CGDirArgs da{d, ctx->cg_g, n, ctx->st, s_gg, 1, ctx->coarse_tol, maxit};
      hipLaunchKernelGGL(cg_direction_kernel, dim3(g_vec), dim3(kThreads), 0, ctx->stream, da);
      if (halo_exchange(ctx->comm, A.halo, d, n, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "halo exchange failed");
      SpmvArgs a = base_args(A, d, ctx->cg_h);
      a.st = ctx->st;
      a.part_out = ctx->part_a;
      const int n_part_dh = launch_op<kStore, 2>(ctx, A, a);
      hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_a, n_part_dh, 1, 0u, s_dh);
      if (allreduce_sum(ctx->comm, s_dh, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
      CGUpdateArgs ua{x, ctx->cg_g, d, ctx->cg_h, n, ctx->st, s_dh, 1, ctx->part_b};
      hipLaunchKernelGGL(cg_update_kernel, dim3(g_upd), dim3(kThreads), 0, ctx->stream, ua);
      hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_b, g_upd, 1, 0u, s_gg);
      if (allreduce_sum(ctx->comm, s_gg, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
    }
    HIPC(hipMemcpyAsync(ctx->st_host, ctx->st, sizeof(CGState), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    if (ctx->st_host->done) break;
    if (launched > maxit + 1) return fail(ctx, GMG_ERR_HIP, "coarse CG state machine did not terminate");
    chunk = ctx->coarse_chunk > 0 ? ctx->coarse_chunk : 4;
  }
  ctx->last_coarse_iters = ctx->st_host->iters;
  ctx->stats.coarse_solves++;
  ctx->stats.coarse_iterations += ctx->st_host->iters;
  if (iters_out) *iters_out = ctx->st_host->iters;
  if (res_out) *res_out = ctx->st_host->res;
  if (ctx->st_host->status != 0) return fail(ctx, GMG_ERR_COARSE_NOCONV, "coarse CG did not converge within max_it");
  return GMG_OK;
}

}  // namespace
