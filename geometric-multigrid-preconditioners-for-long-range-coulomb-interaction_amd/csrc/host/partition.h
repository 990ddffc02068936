// partition.h -- row partition of an operator for one-process-per-GPU runs: the counterpart of
// what Epetra_Map / Epetra_Import give the reference (locally_owned_dofs, :656-657, and the
// ghost import inside every vmult).  Every rank holds the whole host-side operator (setup is
// replicated) and cuts out its own rows, so all ranks derive consistent plans by construction.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "laplace_problem.h"

namespace step50 {

struct HaloPlanHost {
  std::vector<int32_t> neighbor_rank, send_count, send_idx, recv_count;
};

struct LocalOperator {
  CSRMatrix A;  // owned rows, columns renumbered [owned | ghosts grouped by owner rank, ascending]
  HaloPlanHost halo;
  int64_t row_begin = 0, row_end = 0;
  std::vector<int64_t> ghost_global;  // global index of each ghost column
};

inline void canonical_range(int64_t n, int rank, int n_ranks, int64_t *b, int64_t *e) {
  const int64_t c = (n + n_ranks - 1) / n_ranks;  // == gmg_partition_range
  *b = std::min<int64_t>(n, (int64_t)rank * c);
  *e = std::min<int64_t>(n, (int64_t)(rank + 1) * c);
}
inline int owner_of(int64_t j, int64_t n, int n_ranks) {
  const int64_t c = (n + n_ranks - 1) / n_ranks;
  return (int)(j / c);
}

// Square operator, rows and columns in the same canonical partition.
inline LocalOperator localize(const CSRMatrix &G, int rank, int n_ranks) {
  const int64_t n = G.n_rows;
  LocalOperator L;
  canonical_range(n, rank, n_ranks, &L.row_begin, &L.row_end);
  // needed[s] = off-rank columns referenced by the rows of rank s (ascending, unique)
  std::vector<std::vector<int64_t>> needed((size_t)n_ranks);
  std::vector<char> mark((size_t)n, 0);
  for (int s = 0; s < n_ranks; ++s) {
    int64_t b, e;
    canonical_range(n, s, n_ranks, &b, &e);
    auto &v = needed[(size_t)s];
    for (int64_t i = b; i < e; ++i)
      for (int64_t k = G.rowptr[(size_t)i]; k < G.rowptr[(size_t)i + 1]; ++k) {
        const int64_t j = G.col[(size_t)k];
        if ((j < b || j >= e) && !mark[(size_t)j]) { mark[(size_t)j] = 1; v.push_back(j); }
      }
    std::sort(v.begin(), v.end());
    for (int64_t j : v) mark[(size_t)j] = 0;
  }
  // my ghosts, grouped by owner (ascending global index == ascending owner)
  L.ghost_global = needed[(size_t)rank];
  std::vector<int32_t> recv_by_rank((size_t)n_ranks, 0), send_by_rank((size_t)n_ranks, 0);
  for (int64_t j : L.ghost_global) recv_by_rank[(size_t)owner_of(j, n, n_ranks)]++;
  std::vector<std::vector<int32_t>> send_lists((size_t)n_ranks);
  for (int s = 0; s < n_ranks; ++s) {
    if (s == rank) continue;
    for (int64_t j : needed[(size_t)s])
      if (j >= L.row_begin && j < L.row_end) send_lists[(size_t)s].push_back((int32_t)(j - L.row_begin));
    send_by_rank[(size_t)s] = (int32_t)send_lists[(size_t)s].size();
  }
  for (int s = 0; s < n_ranks; ++s) {
    if (s == rank || (send_by_rank[(size_t)s] == 0 && recv_by_rank[(size_t)s] == 0)) continue;
    L.halo.neighbor_rank.push_back(s);
    L.halo.send_count.push_back(send_by_rank[(size_t)s]);
    L.halo.recv_count.push_back(recv_by_rank[(size_t)s]);
    L.halo.send_idx.insert(L.halo.send_idx.end(), send_lists[(size_t)s].begin(), send_lists[(size_t)s].end());
  }
  // local matrix
  const int64_t n_owned = L.row_end - L.row_begin;
  L.A.n_rows = n_owned;
  L.A.n_cols = n_owned + (int64_t)L.ghost_global.size();
  L.A.rowptr.assign((size_t)n_owned + 1, 0);
  const int64_t k0 = G.rowptr[(size_t)L.row_begin], k1 = G.rowptr[(size_t)L.row_end];
  L.A.col.resize((size_t)(k1 - k0));
  L.A.val.assign(G.val.begin() + k0, G.val.begin() + k1);
  for (int64_t i = 0; i < n_owned; ++i) L.A.rowptr[(size_t)i + 1] = G.rowptr[(size_t)(L.row_begin + i) + 1] - k0;
  for (int64_t k = k0; k < k1; ++k) {
    const int64_t j = G.col[(size_t)k];
    int32_t lj;
    if (j >= L.row_begin && j < L.row_end) lj = (int32_t)(j - L.row_begin);
    else lj = (int32_t)(n_owned + (std::lower_bound(L.ghost_global.begin(), L.ghost_global.end(), j) - L.ghost_global.begin()));
    L.A.col[(size_t)(k - k0)] = lj;
  }
  return L;
}

}  // namespace step50
