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
  const bool peer = comm && ctx->comm.peer;  // sums and halo entries travel inside the iteration's own kernels (gmg_device.hpp: PeerCG)
  if (!ctx->cg_ring[0]) {
    if (peer) {
      // one shared allocation: the neighbours' direction kernels store their halo entries of d straight into my vectors
      ctx->ring_stride = (A.n_cols + 2 + 15) / 16 * 16;
      if (comm_share_alloc(ctx->comm, sizeof(double) * (size_t)(ctx->ring_stride * kXRing), ctx->ring_shared))
        return fail(ctx, GMG_ERR_COMM, "coarse CG: shared direction vectors could not be mapped");
      for (int r = 0; r < kXRing; ++r) ctx->cg_ring[r] = reinterpret_cast<double *>(ctx->ring_shared[ctx->comm.rank]) + (int64_t)r * ctx->ring_stride;
      // what a neighbour must know to address its segment of my ghost tail: stride, owned rows, offset of every source
      int64_t mine[4 + kPeerMaxRanks];
      mine[0] = ctx->ring_stride; mine[1] = n; mine[2] = mine[3] = 0;
      for (int r = 0; r < kPeerMaxRanks; ++r) mine[4 + r] = -1;
      int64_t ro = 0;
      for (int i = 0; i < A.halo.n_neighbors; ++i) { mine[4 + A.halo.rank[(size_t)i]] = ro; ro += A.halo.recv_count[(size_t)i]; }
      if (comm_exchange_meta(ctx->comm, mine, 4 + kPeerMaxRanks, ctx->peer_meta)) return fail(ctx, GMG_ERR_COMM, "coarse CG: meta exchange failed");
      if (!ctx->peer_push_cnt) {
        HIPC(hipMalloc(&ctx->peer_push_cnt, sizeof(unsigned int)));
        HIPC(hipMemsetAsync(ctx->peer_push_cnt, 0, sizeof(unsigned int), ctx->stream));
      }
    } else {
      for (int r = 0; r < kXRing; ++r) CHK(alloc_vec(ctx, &ctx->cg_ring[r], A.n_cols));
    }
    ctx->cg_ring_len = A.n_cols;
  }
  if (ctx->cg_ring_len < A.n_cols) return fail(ctx, GMG_ERR_INVALID, "coarse CG ring sized for a smaller operator");
  PeerCG pc{};
  if (peer) {
    pc.area = ctx->comm.box[ctx->comm.rank] + kPeerCgOffset;
    for (int r = 0; r < ctx->comm.n_ranks; ++r) pc.peer_area[r] = ctx->comm.box[r] + kPeerCgOffset;
    pc.n_ranks = ctx->comm.n_ranks; pc.me = ctx->comm.rank;
    for (int i = 0; i < A.halo.n_neighbors; ++i)
      if (A.halo.recv_count[(size_t)i] > 0) pc.nb_mask |= 1u << A.halo.rank[(size_t)i];
    pc.tag0 = ctx->peer_tag0;
    ctx->peer_tag0 += (unsigned long long)ctx->coarse_maxit + 16ull;  // the same stride on every rank
    pc.abort_flag = ctx->comm.abort_host;
  }
  auto peer_sum = [&](const double *part, int n_part, int kind, int from_init, double *out) {
    PeerSumArgs ps{pc, part, n_part, kind, from_init, ctx->st, out};
    hipLaunchKernelGGL(peer_allsum_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, ps);
  };

  // the direction read by iteration 0 (beta = 0) must be finite: the init kernel zeroes it
  CGInitArgs ia{b, x, ctx->cg_g, ctx->cg_ring[kXRing - 1], ctx->cg_ring[kXRing - 1], n, ctx->st, ctx->part_b};
  hipLaunchKernelGGL(cg_init_kernel, dim3(g_vec), dim3(kThreads), 0, ctx->stream, ia);
  // where the consumers find the reduced scalars: all-reduced single values, or the raw partials
  const double *gg_src = ctx->part_b;
  int gg_n = g_vec;
  if (peer) {
    peer_sum(ctx->part_b, g_vec, 0, 1, s_gg);  // |g|^2 that opens iteration 0: every rank's sum to every rank, total in s_gg
    gg_src = s_gg; gg_n = 1;
  } else if (comm) {
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
    CGDirArgs da{};
    da.d = d; da.d_old = ctx->cg_ring[(launched + kXRing - 1) % kXRing]; da.g = ctx->cg_g; da.n = n; da.st = ctx->st;
    da.part_in = gg_src; da.n_part_in = gg_n; da.tol = ctx->coarse_tol; da.maxit = maxit;
    if (peer) {
      da.pc = pc;
      da.send_idx = A.halo.send_idx; da.cnt = ctx->peer_push_cnt;
      int so = 0;
      for (int i = 0; i < A.halo.n_neighbors; ++i) {
        if (A.halo.send_count[(size_t)i] > 0) {
          const int k = da.n_nb++, pr = A.halo.rank[(size_t)i];
          da.nb_rank[k] = pr; da.send_off[k] = so; da.send_off[k + 1] = so + A.halo.send_count[(size_t)i];
          da.peer_stride[k] = ctx->peer_meta[pr][0];
          da.peer_ghost[k] = reinterpret_cast<double *>(ctx->ring_shared[pr]) + ctx->peer_meta[pr][1] + ctx->peer_meta[pr][4 + ctx->comm.rank];
          if (ctx->peer_meta[pr][4 + ctx->comm.rank] < 0) return fail(ctx, GMG_ERR_INVALID, "coarse CG: the neighbours' halo plans disagree");
        }
        so += A.halo.send_count[(size_t)i];
      }
      da.n_push_wg = da.n_nb ? std::min(g_vec, 16) : 0;
    }
    hipLaunchKernelGGL(cg_direction_kernel, dim3(g_vec), dim3(kThreads), 0, ctx->stream, da);
    if (comm && !peer && halo_exchange(ctx->comm, A.halo, d, n, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "halo exchange failed");
    if (peer && pc.nb_mask) hipLaunchKernelGGL(peer_wait_halo_kernel, dim3(1), dim3(64), 0, ctx->stream, pc, (const CGState *)ctx->st);
    SpmvArgs a = base_args(A, d, ctx->cg_h);
    a.st = ctx->st;
    a.part_out = ctx->part_a;
    const bool sample = ctx->prof_every > 0 && (launched % ctx->prof_every) == 0 && ctx->ev_used < (int)ctx->ev_a.size();
    if (sample) { ctx->timed_start = ctx->ev_a[(size_t)ctx->ev_used]; ctx->timed_stop = ctx->ev_b[(size_t)ctx->ev_used++]; }
    const int n_part_dh = launch_op<kStore, 2>(ctx, A, a);
    const double *dh_src = ctx->part_a;
    int dh_n = n_part_dh;
    if (peer) {
      peer_sum(ctx->part_a, n_part_dh, 1, 0, s_dh);
      dh_src = s_dh; dh_n = 1;
    } else if (comm) {
      hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, (const double *)ctx->part_a, n_part_dh, 1, 0u, s_dh);
      if (allreduce_sum(ctx->comm, s_dh, 1, ctx->stream)) return fail(ctx, GMG_ERR_COMM, "all-reduce failed");
      dh_src = s_dh; dh_n = 1;
    }
    CGUpdateGArgs ua{ctx->cg_g, ctx->cg_h, n, ctx->st, dh_src, dh_n, ctx->part_b};
    const bool sample2 = sample && ctx->ev2_used < (int)ctx->ev_c.size();
    if (sample2) { ctx->timed_start = ctx->ev_c[(size_t)ctx->ev2_used]; ctx->timed_stop = ctx->ev_d[(size_t)ctx->ev2_used++]; }
    launch_timed(ctx, cg_update_g_kernel, dim3(g_upd), dim3(kThreads), 0, ua);
    if ((launched + 1) % kXRing == 0) flush_x(launched + 1 - kXRing, launched + 1);
    if (peer) {
      peer_sum(ctx->part_b, g_upd, 0, 0, s_gg);
    } else if (comm) {
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
  if (ctx->st_final.status != 0) {
    if (comm_aborted(ctx->comm)) return fail(ctx, GMG_ERR_COMM, "coarse CG: a rank gave up waiting for its neighbours' sums or halo entries");
    ctx->err = "coarse CG did not converge within max_it (stopped at iteration " + std::to_string(ctx->st_final.iters) + ", residual " + std::to_string(ctx->st_final.res) + ")";
    return GMG_ERR_COARSE_NOCONV;
  }
  return GMG_OK;
}

}  // namespace
