// flat_map.h -- vertex key -> value, open addressing with linear probing.
//
// The host mirror of LaplaceProblem (laplace_problem.h) numbers DoFs by looking vertices up by their packed lattice
// coordinates (forest.h: pack3, 63 bits): two million insertions per level and cycle.  A node-based std::unordered_map spends
// most of distribute_dofs() in malloc and pointer chasing; this table is one array of {key, value}, at most half full.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

namespace step50 {

template <class V>
class FlatMap {
 public:
  static constexpr uint64_t kEmpty = ~0ull;  // no vertex key has all bits set (pack3 uses 63)
  void clear() { slots_.clear(); n_ = 0; shift_ = 64; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  // room for n keys without growing
  void reserve(size_t n) {
    size_t cap = 16;
    int bits = 4;
    while (cap < 2 * n) { cap <<= 1; ++bits; }
    if (cap > slots_.size()) rehash(cap, bits);
  }
  // {pointer to the value, true if the key is new}
  std::pair<V *, bool> emplace(uint64_t key, const V &v) {
    if (2 * (n_ + 1) > slots_.size()) reserve(n_ ? 2 * n_ : 8);
    size_t i = home(key);
    for (;; i = (i + 1) & (slots_.size() - 1)) {
      Slot &s = slots_[i];
      if (s.key == key) return {&s.value, false};
      if (s.key == kEmpty) { s.key = key; s.value = v; ++n_; return {&s.value, true}; }
    }
  }
  V *find(uint64_t key) {
    if (slots_.empty()) return nullptr;
    for (size_t i = home(key);; i = (i + 1) & (slots_.size() - 1)) {
      Slot &s = slots_[i];
      if (s.key == key) return &s.value;
      if (s.key == kEmpty) return nullptr;
    }
  }
  const V *find(uint64_t key) const { return const_cast<FlatMap *>(this)->find(key); }
  size_t count(uint64_t key) const { return find(key) ? 1 : 0; }
  const V &at(uint64_t key) const {
    const V *p = find(key);
    if (!p) throw std::out_of_range("FlatMap::at: unknown vertex");
    return *p;
  }
  V &operator[](uint64_t key) { return *emplace(key, V()).first; }

 private:
  struct Slot { uint64_t key; V value; };
  size_t home(uint64_t key) const { return (size_t)((key * 0x9E3779B97F4A7C15ull) >> shift_); }
  void rehash(size_t cap, int bits) {
    std::vector<Slot> old;
    old.swap(slots_);
    slots_.assign(cap, Slot{kEmpty, V()});
    shift_ = 64 - bits;
    for (const Slot &s : old)
      if (s.key != kEmpty) {
        size_t i = home(s.key);
        while (slots_[i].key != kEmpty) i = (i + 1) & (cap - 1);
        slots_[i] = s;
      }
  }
  std::vector<Slot> slots_;
  size_t n_ = 0;
  int shift_ = 64;
};

// vertex key -> DoF for the meshes of this program: vertices of the undivided lattice (level 0; the bulk of every mesh of
// the adaptive loop: 1.77 of 1.93 million at 64 k atoms) index a dense array by their lattice position, a cell loop walks
// it almost sequentially; only the vertices the refinement added go through the hash table.
// Keys are forest.h's pack3 (21 bits per direction) of coordinates on the finest addressable lattice: a level-0 vertex has
// the low `shift` bits of every coordinate clear.
class VertexMap {
 public:
  // lattice of n[0] x n[1] x n[2] vertices with spacing 1 << shift (n[d] = 0: no dense part)
  void reset(int shift, const int n[3]) {
    shift_ = shift; mask_ = (1ull << shift) - 1;
    for (int d = 0; d < 3; ++d) n_[d] = (uint64_t)n[d];
    dense_.assign((size_t)(n_[0] * n_[1] * n_[2]), -1);
    sparse_.clear();
    size_ = 0;
  }
  void clear() { dense_.clear(); sparse_.clear(); n_[0] = n_[1] = n_[2] = 0; size_ = 0; }
  void reserve_sparse(size_t n) { sparse_.reserve(n); }
  size_t size() const { return size_; }
  std::pair<int32_t *, bool> emplace(uint64_t key, int32_t v) {
    const int64_t i = dense_index(key);
    if (i >= 0) {
      int32_t &s = dense_[(size_t)i];
      if (s >= 0) return {&s, false};
      s = v; ++size_;
      return {&s, true};
    }
    const auto r = sparse_.emplace(key, v);
    size_ += r.second;
    return r;
  }
  const int32_t *find(uint64_t key) const {
    const int64_t i = dense_index(key);
    if (i >= 0) return dense_[(size_t)i] >= 0 ? &dense_[(size_t)i] : nullptr;
    return sparse_.find(key);
  }
  size_t count(uint64_t key) const { return find(key) ? 1 : 0; }
  int32_t at(uint64_t key) const {
    const int32_t *p = find(key);
    if (!p) throw std::out_of_range("VertexMap::at: unknown vertex");
    return *p;
  }
  // dense lattice position of a key (-1: not a vertex of the lattice)
  int64_t dense_index(uint64_t key) const {
    if (dense_.empty()) return -1;
    const uint64_t x = key & 0x1FFFFF, y = (key >> 21) & 0x1FFFFF, z = (key >> 42) & 0x1FFFFF;
    if ((x | y | z) & mask_) return -1;
    const uint64_t i = x >> shift_, j = y >> shift_, k = z >> shift_;
    if (i >= n_[0] || j >= n_[1] || k >= n_[2]) return -1;
    return (int64_t)(i + n_[0] * (j + n_[1] * k));
  }

 private:
  int shift_ = 0;
  uint64_t mask_ = 0, n_[3] = {0, 0, 0};
  std::vector<int32_t> dense_;
  FlatMap<int32_t> sparse_;
  size_t size_ = 0;
};

}  // namespace step50
