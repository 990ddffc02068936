// forest.h -- hexahedral / quadrilateral forest with hanging nodes: the host-side counterpart
// of what the reference gets from parallel::distributed::Triangulation (p4est) + DoFHandler.
//
// Reference: every level-0 lattice cell is its own tree (GridGenerator::
// subdivided_hyper_rectangle, src/step-50.cc:1504-1526) or the tree is one cell refined
// globally (hyper_cube + refine_global, :1496-1497); mesh smoothing
// limit_level_difference_at_vertices + construct_multigrid_hierarchy (:120-122).
// Nothing of deal.II / p4est is used or copied: cells are integer boxes, vertices are integer
// points on the finest addressable lattice, DoFs (Q1) sit on vertices.
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <unordered_map>
#include <vector>

namespace step50 {

constexpr int kMaxLevelShift = 12;  // vertex keys address level <= 12 below the root lattice

struct Cell {
  int32_t c[3];         // integer coordinates in units of this level's cell size
  int32_t parent;       // index in the previous level (-1 on level 0)
  int32_t first_child;  // index of child 0 in the next level, children contiguous; -1 = active
};

inline uint64_t pack3(uint64_t x, uint64_t y, uint64_t z) { return x | (y << 21) | (z << 42); }

template <int dim>
class Forest {
 public:
  static constexpr int n_children = 1 << dim;
  static constexpr int n_vertices = 1 << dim;
  static constexpr int n_faces = 2 * dim;

  int n0 = 0;         // root cells per direction
  double origin = 0;  // lower-left corner (same in every direction)
  double h0 = 0;      // root cell size
  std::vector<std::vector<Cell>> levels;
  std::vector<std::unordered_map<uint64_t, int32_t>> index;  // per level >= 1: packed coords -> cell

  void create_lattice(int n_root, double origin_, double h_root) {
    n0 = n_root; origin = origin_; h0 = h_root;
    levels.assign(1, {});
    index.assign(1, {});
    const int nz = dim == 3 ? n0 : 1;
    levels[0].reserve((size_t)n0 * n0 * nz);
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < n0; ++j)
        for (int i = 0; i < n0; ++i) levels[0].push_back(Cell{{i, j, k}, -1, -1});
  }

  int n_levels() const { return (int)levels.size(); }
  double cell_size(int level) const { return h0 / double(1 << level); }

  int32_t find(int level, int x, int y, int z) const {
    const int n = n0 << level;
    if (x < 0 || y < 0 || x >= n || y >= n) return -1;
    if (dim == 3 && (z < 0 || z >= n)) return -1;
    if (level >= n_levels()) return -1;
    if (level == 0) return x + n0 * (y + n0 * (dim == 3 ? z : 0));
    auto it = index[(size_t)level].find(pack3((uint64_t)x, (uint64_t)y, (uint64_t)z));
    return it == index[(size_t)level].end() ? -1 : it->second;
  }

  bool active(int level, int32_t c) const { return levels[(size_t)level][(size_t)c].first_child < 0; }

  // split one cell; children in deal.II order (x fastest)
  void split(int level, int32_t ci) {
    if ((int)levels.size() <= level + 1) { levels.emplace_back(); index.emplace_back(); }
    Cell &p = levels[(size_t)level][(size_t)ci];
    if (p.first_child >= 0) return;
    auto &next = levels[(size_t)level + 1];
    p.first_child = (int32_t)next.size();
    const Cell pc = p;
    for (int a = 0; a < n_children; ++a) {
      Cell ch{{2 * pc.c[0] + (a & 1), 2 * pc.c[1] + ((a >> 1) & 1), dim == 3 ? 2 * pc.c[2] + ((a >> 2) & 1) : 0}, ci, -1};
      index[(size_t)level + 1][pack3((uint64_t)ch.c[0], (uint64_t)ch.c[1], (uint64_t)ch.c[2])] = (int32_t)next.size();
      next.push_back(ch);
    }
  }

  void refine_global(int times) {
    for (int t = 0; t < times; ++t) {
      const int L = n_levels() - 1;
      const size_t n = levels[(size_t)L].size();
      for (size_t c = 0; c < n; ++c) split(L, (int32_t)c);
    }
  }

  int64_t n_active_cells() const {
    int64_t n = 0;
    for (auto &lv : levels)
      for (auto &c : lv) n += c.first_child < 0;
    return n;
  }

  // vertex a (bit d set = upper side in direction d) of a cell, on the finest addressable lattice
  uint64_t vertex_key(int level, const Cell &c, int a) const {
    const int s = kMaxLevelShift - level;
    return pack3((uint64_t)(c.c[0] + (a & 1)) << s, (uint64_t)(c.c[1] + ((a >> 1) & 1)) << s,
                 dim == 3 ? (uint64_t)(c.c[2] + ((a >> 2) & 1)) << s : 0);
  }
  static void unpack(uint64_t key, uint64_t v[3]) {
    v[0] = key & 0x1FFFFF; v[1] = (key >> 21) & 0x1FFFFF; v[2] = (key >> 42) & 0x1FFFFF;
  }
  void vertex_coords(uint64_t key, double x[3]) const {
    uint64_t v[3];
    unpack(key, v);
    const double hf = h0 / double(1 << kMaxLevelShift);
    for (int d = 0; d < 3; ++d) x[d] = d < dim ? origin + hf * (double)v[d] : 0.0;
  }
  bool vertex_on_boundary(uint64_t key) const {
    uint64_t v[3];
    unpack(key, v);
    const uint64_t hi = (uint64_t)n0 << kMaxLevelShift;
    for (int d = 0; d < dim; ++d)
      if (v[d] == 0 || v[d] == hi) return true;
    return false;
  }
  void cell_origin(int level, const Cell &c, double x[3]) const {
    const double h = cell_size(level);
    for (int d = 0; d < 3; ++d) x[d] = d < dim ? origin + h * c.c[d] : 0.0;
  }

  // Refine the flagged active cells, then restore the 2:1 balance over vertices (deal.II
  // limit_level_difference_at_vertices == p4est full-connectivity balance).  flags[level][cell].
  // Returns the number of cells split.
  int64_t refine_flagged(std::vector<std::vector<char>> &flags) {
    int64_t n_split = 0;
    flags.resize(levels.size());
    for (size_t l = 0; l < levels.size(); ++l) flags[l].resize(levels[l].size(), 0);
    // closure: a flagged cell at level l needs every vertex-neighbour region to be at level >= l,
    // i.e. active neighbours at level l-1 must be refined too.  Iterate from fine to coarse.
    bool changed = true;
    while (changed) {
      changed = false;
      for (int l = n_levels() - 1; l >= 1; --l) {
        for (size_t ci = 0; ci < levels[(size_t)l].size(); ++ci) {
          const Cell &c = levels[(size_t)l][ci];
          const bool will_refine = c.first_child < 0 && flags[(size_t)l][ci];
          if (!will_refine) continue;
          // all 3^dim - 1 neighbours at level l must exist after refinement => their parents
          // (level l-1) must be refined or flagged
          for (int dz = (dim == 3 ? -1 : 0); dz <= (dim == 3 ? 1 : 0); ++dz)
            for (int dy = -1; dy <= 1; ++dy)
              for (int dx = -1; dx <= 1; ++dx) {
                if (!dx && !dy && !dz) continue;
                const int x = c.c[0] + dx, y = c.c[1] + dy, z = c.c[2] + dz;
                const int n = n0 << l;
                if (x < 0 || y < 0 || z < 0 || x >= n || y >= n || (dim == 3 && z >= n)) continue;
                if (find(l, x, y, z) >= 0) continue;
                const int32_t p = find(l - 1, x >> 1, y >> 1, dim == 3 ? z >> 1 : 0);
                if (p < 0) { std::abort(); }  // balance invariant broken
                if (!flags[(size_t)l - 1][(size_t)p]) { flags[(size_t)l - 1][(size_t)p] = 1; changed = true; }
              }
        }
      }
    }
    for (int l = n_levels() - 1; l >= 0; --l) {
      const size_t n = levels[(size_t)l].size();
      for (size_t ci = 0; ci < n; ++ci)
        if (flags[(size_t)l][ci] && levels[(size_t)l][ci].first_child < 0) { split(l, (int32_t)ci); ++n_split; }
    }
    return n_split;
  }
};

}  // namespace step50
