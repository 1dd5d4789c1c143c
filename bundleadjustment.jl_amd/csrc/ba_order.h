// Host-only: camera graph, fill-reducing camera orderings and the tile-level symbolic factorisation of the reduced camera
// system (no HIP in here: ba_order.cpp compiles with any C++17 compiler).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

// ordering methods: what `perm` selects in the reference (src/lm.jl:84-88, src/LevenbergMarquardt.jl:106-110)
enum { BA_ORDER_AMD = 0, BA_ORDER_METIS = 1, BA_ORDER_NATURAL = 2 };

// tile pattern of a block-sparse reduced camera system after the symbolic factorisation
struct TilePattern {
  int64_t nt = 0;
  std::vector<int> prow_ptr, prow;  // pair q: [k+1] + the tile rows of its pattern (k = 2q), ascending
  std::vector<int> lcol_ptr, lcol;  // tile row i: the tile columns j < i of its pattern, ascending
  std::vector<int> lpair_ptr, lpair;  // pair q of the backward sweep (tile rows 2q + 1, 2q): union of their pattern columns < 2q
  double tile_fill = 1.0;           // pattern tiles (with fill) / all lower tiles
  double flop_fill = 1.0;           // trailing-update tiles of the pattern / of the dense factorisation
  // Two independent chains (elimination from both ends of a profile-ordered sequence, see cam_order): the pairs [0, a_clean)
  // and [split, split + b_clean) touch disjoint tiles and no other pair before `split` touches the second group's columns, so
  // the two runs may factor side by side; all other pairs follow in index order.  0 / 0: one chain.
  int split = 0, a_clean = 0, b_clean = 0;
};

// symbolic factorisation of the tile occupancy `occ` (nt x nt, lower, row-major; receives the fill) per tile column pair
// split_pair > 0: the tile column pair at which a second, independent group of columns starts (hint from the ordering; what can
// really run side by side is worked out from the pattern: TilePattern::a_clean, b_clean)
void tile_pattern_build(int64_t nt, std::vector<unsigned char> &occ, TilePattern *out, int split_pair = 0);

// camera graph: cameras adjacent when they share a point; one bit row per camera (no self loops)
struct CamGraph {
  int64_t n = 0, W = 0;           // cameras, 64-bit words per row
  std::vector<uint64_t> bits;     // n x W
  bool test(int64_t a, int64_t b) const { return (bits[(size_t)(a * W + (b >> 6))] >> (b & 63)) & 1u; }
  void set(int64_t a, int64_t b) { bits[(size_t)(a * W + (b >> 6))] |= (uint64_t)1 << (b & 63); }
  int64_t edges() const;          // unordered camera pairs
};
// from the observation lists grouped by point: pt_ptr (npnts + 1), pt_obs (observation ids), cam0 (camera of an observation)
void cam_graph_build(int64_t ncams, int64_t npnts, const int *pt_ptr, const int *pt_obs, const int *cam0, CamGraph *g);

// position -> camera ("perm[k] is the camera at block row k of S"); method BA_ORDER_*; nb = tile size in scalars.
// Deterministic.  chosen (optional): name of the candidate sequence that won (static storage).
// split_pair (optional): see tile_pattern_build
void cam_order(const CamGraph &g, int method, int nb, std::vector<int> *perm, const char **chosen, int *split_pair = nullptr);
double tile_pattern_cost(const TilePattern &pat);

// tile occupancy (nt x nt lower, before fill) of S when camera c sits at block row pos[c]; NB = tile size in scalars
void cam_tile_occupancy(const CamGraph &g, const std::vector<int> &pos, int64_t nt, int nb, std::vector<unsigned char> *occ);
