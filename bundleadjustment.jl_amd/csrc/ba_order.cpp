// Fill-reducing ordering of the cameras of the reduced camera system (host only).
//
// Reference: the solvers order the augmented matrix K = [[I J];[J' -lambda I]] once, with AMD.jl (`amd(A)`) or Metis.jl
// (`Metis.permutation(A' + A)`), src/lm.jl:84-88, src/LevenbergMarquardt.jl:106-110, and hand the permutation to
// `ldl_analyse`, src/ldl_aux.jl:246-283.  Both are third-party C libraries (SuiteSparse AMD, METIS 5.1.0) that are not in
// the image: what is restated here is their ROLE, by the published algorithms they implement -- parity unpinned (an
// ordering changes fill and rounding, never the step in exact arithmetic).
//
// On the device the residual rows and point columns of K are eliminated in closed form, in exactly the order AMD gives them
// (degree-12 residual rows first, then the points: SURVEY Appendix C); what is left to order is the camera block, whose
// graph has one node per camera and an edge for every camera pair that shares a point -- the 9 x 9 blocks of the Schur
// complement S.  The factorisation of S works on 128 x 128 tiles (14.2 cameras) in tile-column pairs, so an ordering is
// judged by the TILE pattern it leaves (tile_pattern_build), not by scalar fill:
//   :AMD   -> minimum degree on the camera graph (exact external degrees on bit rows: the graph has at most a few 10^4
//             nodes), followed by a postorder of the elimination tree so that every subtree -- a set of cameras that only
//             interact among themselves and with their ancestors -- is contiguous and lands in the same tiles;
//   :Metis -> nested dissection: recursive bisection by level-structure separators (George), separators last,
//             leaves and separators ordered by the same minimum-degree routine.
#include "ba_order.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>  // std::iota, std::gcd

namespace {

inline int popc(uint64_t x) { return __builtin_popcountll(x); }

template <typename F>
inline void for_bits(const uint64_t *row, int64_t W, F f) {
  for (int64_t w = 0; w < W; w++) {
    uint64_t m = row[w];
    while (m) {
      const int b = __builtin_ctzll(m);
      m &= m - 1;
      f((int)(w * 64 + b));
    }
  }
}

// induced subgraph on `verts` (local ids = positions in verts); loc: scratch of size g.n, -1 outside
void induced(const CamGraph &g, const std::vector<int> &verts, std::vector<int> &loc, CamGraph *s) {
  const int64_t k = (int64_t)verts.size();
  s->n = k;
  s->W = (k + 63) / 64;
  s->bits.assign((size_t)(k * s->W), 0);
  for (int64_t i = 0; i < k; i++) loc[(size_t)verts[(size_t)i]] = (int)i;
  for (int64_t i = 0; i < k; i++)
    for_bits(&g.bits[(size_t)((int64_t)verts[(size_t)i] * g.W)], g.W, [&](int u) {
      const int lu = loc[(size_t)u];
      if (lu >= 0) s->set(i, lu);
    });
  for (int64_t i = 0; i < k; i++) loc[(size_t)verts[(size_t)i]] = -1;
}

// Minimum degree with exact external degrees.  Eliminating v makes its neighbours a clique: row(u) |= row(v) for every
// neighbour u; rows only ever hold uneliminated nodes, so a node's degree is the population count of its row.  Ties go to
// the lowest index (deterministic).  The column structures at elimination time give the elimination tree; the result is
// its postorder (children by ascending subtree size, so a parent follows its largest child): same fill, subtrees contiguous.
void order_md(const CamGraph &g, std::vector<int> *out) {
  const int64_t n = g.n, W = g.W;
  out->clear();
  if (n == 0) return;
  std::vector<uint64_t> rows(g.bits), L((size_t)(n * W));
  std::vector<int> deg((size_t)n), pos((size_t)n, -1), elim((size_t)n);
  std::vector<char> alive((size_t)n, 1);
  for (int64_t i = 0; i < n; i++) {
    int d = 0;
    for (int64_t w = 0; w < W; w++) d += popc(rows[(size_t)(i * W + w)]);
    deg[(size_t)i] = d;
  }
  for (int64_t step = 0; step < n; step++) {
    int v = -1, best = INT32_MAX;
    for (int64_t i = 0; i < n; i++)
      if (alive[(size_t)i] && deg[(size_t)i] < best) {
        best = deg[(size_t)i];
        v = (int)i;
      }
    elim[(size_t)step] = v;
    pos[(size_t)v] = (int)step;
    alive[(size_t)v] = 0;
    const uint64_t *rv = &rows[(size_t)((int64_t)v * W)];
    std::memcpy(&L[(size_t)((int64_t)v * W)], rv, (size_t)W * sizeof(uint64_t));
    if (best == (int)(n - step - 1)) {
      // v is adjacent to every remaining node: so is every remaining node of the same degree after this step -- the rest is
      // a clique in the making; eliminating it in index order costs nothing more (and skips the quadratic tail on dense graphs)
      bool clique = true;
      for (int64_t i = 0; i < n && clique; i++)
        if (alive[(size_t)i] && deg[(size_t)i] != best) clique = false;
      if (clique) {
        std::vector<int> rest;
        for (int64_t i = 0; i < n; i++)
          if (alive[(size_t)i]) rest.push_back((int)i);
        for (size_t a = 0; a < rest.size(); a++) {
          const int u = rest[a];
          elim[(size_t)(step + 1 + (int64_t)a)] = u;
          pos[(size_t)u] = (int)(step + 1 + (int64_t)a);
          alive[(size_t)u] = 0;
          uint64_t *lu = &L[(size_t)((int64_t)u * W)];
          std::memset(lu, 0, (size_t)W * sizeof(uint64_t));
          for (size_t b = a + 1; b < rest.size(); b++) lu[rest[b] >> 6] |= (uint64_t)1 << (rest[b] & 63);
        }
        break;
      }
    }
    for_bits(rv, W, [&](int u) {
      uint64_t *ru = &rows[(size_t)((int64_t)u * W)];
      int d = 0;
      for (int64_t w = 0; w < W; w++) ru[w] |= rv[w];
      ru[u >> 6] &= ~((uint64_t)1 << (u & 63));
      ru[v >> 6] &= ~((uint64_t)1 << (v & 63));
      for (int64_t w = 0; w < W; w++) d += popc(ru[w]);
      deg[(size_t)u] = d;
    });
  }
  // elimination tree: the parent of v is the first of its neighbours at elimination time to be eliminated after it
  std::vector<int> parent((size_t)n, -1), size((size_t)n, 1);
  for (int64_t v = 0; v < n; v++) {
    int bestp = INT32_MAX, pv = -1;
    for_bits(&L[(size_t)(v * W)], W, [&](int u) {
      if (pos[(size_t)u] < bestp) {
        bestp = pos[(size_t)u];
        pv = u;
      }
    });
    parent[(size_t)v] = pv;
  }
  for (int64_t s = 0; s < n; s++) {  // elimination order is a topological order of the tree (children first)
    const int v = elim[(size_t)s];
    if (parent[(size_t)v] >= 0) size[(size_t)parent[(size_t)v]] += size[(size_t)v];
  }
  std::vector<std::vector<int>> kids((size_t)n);
  std::vector<int> roots;
  for (int64_t s = 0; s < n; s++) {
    const int v = elim[(size_t)s];
    if (parent[(size_t)v] >= 0) kids[(size_t)parent[(size_t)v]].push_back(v);
    else roots.push_back(v);
  }
  auto by_size = [&](int a, int b) { return size[(size_t)a] != size[(size_t)b] ? size[(size_t)a] < size[(size_t)b] : pos[(size_t)a] < pos[(size_t)b]; };
  std::sort(roots.begin(), roots.end(), by_size);
  for (auto &k : kids) std::sort(k.begin(), k.end(), by_size);
  std::vector<std::pair<int, size_t>> stack;
  for (int r : roots) {
    stack.emplace_back(r, 0);
    while (!stack.empty()) {
      auto &top = stack.back();
      if (top.second < kids[(size_t)top.first].size()) {
        const int c = kids[(size_t)top.first][top.second++];
        stack.emplace_back(c, 0);
      } else {
        out->push_back(top.first);
        stack.pop_back();
      }
    }
  }
}

// breadth-first level structure of the region `id` (label[v] == id) from root; returns the number of levels
int bfs_levels(const CamGraph &g, const std::vector<int> &label, int id, int root, std::vector<int> &level, std::vector<int> &queue) {
  queue.clear();
  queue.push_back(root);
  level[(size_t)root] = 0;
  int nlev = 1;
  for (size_t h = 0; h < queue.size(); h++) {
    const int v = queue[h];
    for_bits(&g.bits[(size_t)((int64_t)v * g.W)], g.W, [&](int u) {
      if (label[(size_t)u] == id && level[(size_t)u] < 0) {
        level[(size_t)u] = level[(size_t)v] + 1;
        nlev = level[(size_t)u] + 1;
        queue.push_back(u);
      }
    });
  }
  return nlev;
}

// Reverse Cuthill-McKee on the nodes with keep[v] != 0 (the others -- "hubs": cameras that see a large part of the scene and
// would put everything into two or three levels -- are appended at the end, by ascending degree): components by ascending
// size, each from a pseudo-peripheral root, neighbours by ascending degree, the whole sequence reversed.
void order_rcm(const CamGraph &g, const std::vector<char> &keep, std::vector<int> *out) {
  const int64_t n = g.n;
  out->clear();
  std::vector<int> label((size_t)n, 0), level((size_t)n, -1), queue, deg((size_t)n, 0), seq;
  for (int64_t v = 0; v < n; v++) {
    label[(size_t)v] = keep[(size_t)v] ? 1 : 0;
  }
  for (int64_t v = 0; v < n; v++)
    if (keep[(size_t)v]) for_bits(&g.bits[(size_t)(v * g.W)], g.W, [&](int u) { deg[(size_t)v] += keep[(size_t)u] != 0; });
  std::vector<std::vector<int>> comps;
  for (int64_t v = 0; v < n; v++)
    if (keep[(size_t)v] && level[(size_t)v] < 0) {
      bfs_levels(g, label, 1, (int)v, level, queue);
      comps.emplace_back(queue);
    }
  std::stable_sort(comps.begin(), comps.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a.size() < b.size(); });
  std::vector<int> nb;
  for (auto &c : comps) {
    int root = c[0];
    for (int v : c)
      if (deg[(size_t)v] < deg[(size_t)root]) root = v;
    int nlev = 0;
    for (int it = 0; it < 8; it++) {
      for (int v : c) level[(size_t)v] = -1;
      const int nl = bfs_levels(g, label, 1, root, level, queue);
      if (nl <= nlev) break;
      nlev = nl;
      int cand = -1;
      for (int v : queue)
        if (level[(size_t)v] == nl - 1 && (cand < 0 || deg[(size_t)v] < deg[(size_t)cand])) cand = v;
      if (cand < 0 || cand == root) break;
      root = cand;
    }
    // Cuthill-McKee numbering from root
    for (int v : c) level[(size_t)v] = -1;
    const size_t first = seq.size();
    seq.push_back(root);
    level[(size_t)root] = 0;
    for (size_t h = first; h < seq.size(); h++) {
      const int v = seq[h];
      nb.clear();
      for_bits(&g.bits[(size_t)((int64_t)v * g.W)], g.W, [&](int u) {
        if (label[(size_t)u] == 1 && level[(size_t)u] < 0) {
          level[(size_t)u] = 0;
          nb.push_back(u);
        }
      });
      std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[(size_t)a] < deg[(size_t)b]; });
      for (int u : nb) seq.push_back(u);
    }
    for (int v : c) label[(size_t)v] = 2;  // done
  }
  std::reverse(seq.begin(), seq.end());
  *out = seq;
  std::vector<int> hubs;
  for (int64_t v = 0; v < n; v++)
    if (!keep[(size_t)v]) hubs.push_back((int)v);
  std::vector<int> fdeg((size_t)n, 0);
  for (int v : hubs)
    for (int64_t w = 0; w < g.W; w++) fdeg[(size_t)v] += popc(g.bits[(size_t)((int64_t)v * g.W + w)]);
  std::stable_sort(hubs.begin(), hubs.end(), [&](int a, int b) { return fdeg[(size_t)a] < fdeg[(size_t)b]; });
  for (int v : hubs) out->push_back(v);
}

struct ND {
  const CamGraph &g;
  std::vector<int> label, level, queue, loc;
  std::vector<int> *out;
  int next_id = 1;
  int leaf;
  explicit ND(const CamGraph &g_, std::vector<int> *o, int leaf_) : g(g_), label((size_t)g_.n, 0), level((size_t)g_.n, -1), loc((size_t)g_.n, -1), out(o), leaf(leaf_) {}

  void md_append(const std::vector<int> &verts) {
    CamGraph s;
    induced(g, verts, loc, &s);
    std::vector<int> ord;
    order_md(s, &ord);
    for (int l : ord) out->push_back(verts[(size_t)l]);
  }

  int region_degree(int v, int id) const {
    int d = 0;
    for_bits(&g.bits[(size_t)((int64_t)v * g.W)], g.W, [&](int u) { d += label[(size_t)u] == id; });
    return d;
  }

  void run(std::vector<int> verts) {
    if (verts.empty()) return;
    if ((int)verts.size() <= leaf) return md_append(verts);
    const int id = next_id++;
    for (int v : verts) label[(size_t)v] = id;
    // connected components first: each is dissected on its own
    {
      std::vector<std::vector<int>> comps;
      for (int v : verts) level[(size_t)v] = -1;
      for (int v : verts)
        if (level[(size_t)v] < 0) {
          bfs_levels(g, label, id, v, level, queue);
          comps.emplace_back(queue);
        }
      if (comps.size() > 1) {
        std::stable_sort(comps.begin(), comps.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a.size() < b.size(); });
        for (auto &c : comps) run(c);
        return;
      }
    }
    // pseudo-peripheral root: start from a node of minimum degree, move to a minimum-degree node of the last level while
    // the structure gets deeper
    int root = verts[0], rd = region_degree(root, id);
    for (int v : verts) {
      const int d = region_degree(v, id);
      if (d < rd) {
        rd = d;
        root = v;
      }
    }
    int nlev = 0;
    for (int it = 0; it < 8; it++) {
      for (int v : verts) level[(size_t)v] = -1;
      const int nl = bfs_levels(g, label, id, root, level, queue);
      if (nl <= nlev) {
        nlev = nl;
        break;
      }
      nlev = nl;
      int cand = -1, cd = INT32_MAX;
      for (int v : queue)
        if (level[(size_t)v] == nl - 1) {
          const int d = region_degree(v, id);
          if (d < cd) {
            cd = d;
            cand = v;
          }
        }
      if (cand < 0 || cand == root) break;
      root = cand;
    }
    for (int v : verts) level[(size_t)v] = -1;
    nlev = bfs_levels(g, label, id, root, level, queue);
    if (nlev < 3) return md_append(verts);  // (nearly) a clique: nothing to dissect
    // separator: the level that splits best -- smallest level among those leaving at least a quarter on either side (else the
    // most balanced one)
    std::vector<int64_t> cnt((size_t)nlev, 0);
    for (int v : verts) cnt[(size_t)level[(size_t)v]]++;
    const int64_t n = (int64_t)verts.size();
    int best = -1;
    double best_cost = 1e300;
    int64_t left = cnt[0];
    for (int l = 1; l + 1 < nlev; l++) {
      const int64_t right = n - left - cnt[(size_t)l], small = std::min(left, right);
      // separator size relative to the smaller side, a mild preference for balance
      const double cost = (double)cnt[(size_t)l] / (double)std::max<int64_t>(1, small) + (small * 4 < n ? 10.0 : 0.0) + (small * 8 < n ? 100.0 : 0.0);
      if (cost < best_cost) {
        best_cost = cost;
        best = l;
      }
      left += cnt[(size_t)l];
    }
    std::vector<int> A, B, S;
    for (int v : verts) {
      const int l = level[(size_t)v];
      if (l < best) A.push_back(v);
      else if (l > best) B.push_back(v);
      else {
        // a node of the separator level without a neighbour one level further out separates nothing: it joins the inner part
        bool touches = false;
        for_bits(&g.bits[(size_t)((int64_t)v * g.W)], g.W, [&](int u) { touches |= (label[(size_t)u] == id && level[(size_t)u] == best + 1); });
        (touches ? S : A).push_back(v);
      }
    }
    for (int v : verts) level[(size_t)v] = -1;
    run(A);
    run(B);
    md_append(S);
  }
};

void order_nd(const CamGraph &g, std::vector<int> *out) {
  out->clear();
  ND nd(g, out, 28);
  std::vector<int> all((size_t)g.n);
  std::iota(all.begin(), all.end(), 0);
  nd.run(all);
}

}  // namespace

int64_t CamGraph::edges() const {
  int64_t e = 0;
  for (uint64_t w : bits) e += popc(w);
  return e / 2;
}

void cam_graph_build(int64_t ncams, int64_t npnts, const int *pt_ptr, const int *pt_obs, const int *cam0, CamGraph *g) {
  g->n = ncams;
  g->W = (ncams + 63) / 64;
  g->bits.assign((size_t)(ncams * g->W), 0);
  std::vector<uint64_t> mask((size_t)g->W, 0);
  for (int64_t pt = 0; pt < npnts; pt++) {
    const int q0 = pt_ptr[pt], q1 = pt_ptr[pt + 1];
    if (q1 - q0 < 2) continue;
    if (q1 - q0 <= 8) {  // the common case: set the pairs one by one
      for (int a = q0; a < q1; a++)
        for (int b = q0; b < q1; b++)
          if (a != b) g->set(cam0[pt_obs[a]], cam0[pt_obs[b]]);
    } else {  // a long track: OR its camera mask into every row
      for (int a = q0; a < q1; a++) mask[(size_t)(cam0[pt_obs[a]] >> 6)] |= (uint64_t)1 << (cam0[pt_obs[a]] & 63);
      for (int a = q0; a < q1; a++) {
        uint64_t *row = &g->bits[(size_t)((int64_t)cam0[pt_obs[a]] * g->W)];
        for (int64_t w = 0; w < g->W; w++) row[w] |= mask[(size_t)w];
      }
      for (int a = q0; a < q1; a++) mask[(size_t)(cam0[pt_obs[a]] >> 6)] = 0;
    }
  }
  for (int64_t c = 0; c < ncams; c++) g->bits[(size_t)(c * g->W + (c >> 6))] &= ~((uint64_t)1 << (c & 63));
}

// What the device pays for a pattern, in trailing-update tiles: per tile column pair the tiles of its update -- at least the
// few hundred a launch of the update kernel costs however short its list -- plus its panel solves (four workgroups per tile
// row, a small fraction of a tile update each).  The chain of diagonal tiles is the same for every ordering.
double tile_pattern_cost(const TilePattern &pat) {
  double cost = 0;
  for (size_t q = 0; q + 1 < pat.prow_ptr.size(); q++) {
    const double c1 = (double)(pat.prow_ptr[q + 1] - pat.prow_ptr[q]), u = c1 > 0 ? c1 - 1 : 0;
    cost += std::max(u * (u + 1) / 2, 256.0) + 8.0 * u;
  }
  // ... and the chain: a pair's two diagonal tiles, panel solves and column update cost what ~640 update tiles cost
  // (~100 us against 0.15 us per tile) whatever its row list; pairs of two independent runs share that time
  const double npairs = (double)pat.prow_ptr.size() - 1.0;
  cost += 640.0 * (npairs - (double)std::min(pat.a_clean, pat.b_clean));
  return cost;
}

// The ordering `method` stands for, chosen -- as CHOLMOD chooses between the orderings it is given -- by the symbolic
// factorisation itself: every candidate's tile pattern is built and the cheapest kept.  Candidates for either method: the
// numbering the caller gave (a BAL file of a sequential capture is already well numbered; it also wins ties, so such a
// problem keeps its numbering), reverse Cuthill-McKee with the cameras of more than 2 / 4 / 8 times the median degree
// deferred to the end (or none); then minimum degree for :AMD, nested dissection for :Metis.  At BAL sizes (10^2..10^4
// cameras, 14 of them per tile) profile-reducing sequences usually win: the unit of fill is a 128 x 128 tile, and scattered
// scalar fill costs whole tiles.
void cam_order(const CamGraph &g, int method, int nb, std::vector<int> *perm, const char **chosen, int *split_pair) {
  const int64_t n = g.n;
  perm->resize((size_t)n);
  std::iota(perm->begin(), perm->end(), 0);
  if (chosen) *chosen = "natural";
  if (split_pair) *split_pair = 0;
  if (method == BA_ORDER_NATURAL || n < 2) return;
  const int64_t nt = std::max<int64_t>(1, (9 * n + nb - 1) / nb);
  std::vector<int> pos((size_t)n), cand;
  std::vector<unsigned char> occ;
  TilePattern pat;
  int cand_split = 0;  // split hint of the candidate being offered
  auto cost_of = [&](const std::vector<int> &order) {
    for (int64_t k = 0; k < n; k++) pos[(size_t)order[(size_t)k]] = (int)k;
    cam_tile_occupancy(g, pos, nt, nb, &occ);
    tile_pattern_build(nt, occ, &pat, cand_split);
    return tile_pattern_cost(pat);
  };
  const bool verbose = getenv("BA_ORDER_VERBOSE") != nullptr;
  double best = cost_of(*perm);
  if (verbose) fprintf(stderr, "[ba_order] natural: cost %.0f (tile fill %.4f, update fill %.5f)\n", best, pat.tile_fill, pat.flop_fill);
  auto offer = [&](const char *name) {
    const double c = cost_of(cand);
    if (verbose) fprintf(stderr, "[ba_order] %s: cost %.0f (tile fill %.4f, update fill %.5f)\n", name, c, pat.tile_fill, pat.flop_fill);
    if (c < best * (1.0 - 1e-9)) {
      best = c;
      *perm = cand;
      if (chosen) *chosen = name;
      if (split_pair) *split_pair = (pat.a_clean > 0 && pat.b_clean > 0) ? cand_split : 0;
    }
  };
  std::vector<int> deg((size_t)n, 0), sorted;
  for (int64_t v = 0; v < n; v++)
    for (int64_t w = 0; w < g.W; w++) deg[(size_t)v] += popc(g.bits[(size_t)(v * g.W + w)]);
  sorted = deg;
  std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
  const int median = std::max(1, sorted[(size_t)(n / 2)]);
  std::vector<char> keep((size_t)n);
  int64_t last_hubs = -1;
  static const char *const rcm_names[4] = {"rcm", "rcm-hubs8", "rcm-hubs4", "rcm-hubs2"};
  const int factors[4] = {0, 8, 4, 2};
  for (int f = 0; f < 4; f++) {
    int64_t hubs = 0;
    for (int64_t v = 0; v < n; v++) {
      keep[(size_t)v] = factors[f] == 0 || deg[(size_t)v] <= (int64_t)factors[f] * median;
      hubs += !keep[(size_t)v];
    }
    if (hubs == last_hubs || (f > 0 && hubs == 0) || hubs > n / 4) continue;  // same split as before / no hub / not hubs any more
    last_hubs = hubs;
    order_rcm(g, keep, &cand);
    offer(rcm_names[f]);
  }
  if (method == BA_ORDER_AMD) {
    order_md(g, &cand);
    offer("minimum-degree");
  } else {
    order_nd(g, &cand);
    offer("nested-dissection");
  }
  // ... and the same algorithm at the granularity the factorisation works in: the best sequence so far cut into the camera
  // groups of its tile column pairs, the groups as the nodes of a quotient graph (adjacent when any of their cameras are),
  // ordered by minimum degree / nested dissection; the cameras of a group keep their relative order
  {
    const std::vector<int> base = *perm;
    const int64_t pair_rows = 2 * (int64_t)nb, ng = (9 * n + pair_rows - 1) / pair_rows;
    std::vector<int> group((size_t)n);
    for (int64_t k = 0; k < n; k++) group[(size_t)base[(size_t)k]] = (int)(9 * k / pair_rows);
    CamGraph q;
    q.n = ng;
    q.W = (ng + 63) / 64;
    q.bits.assign((size_t)(ng * q.W), 0);
    for (int64_t a = 0; a < n; a++)
      for_bits(&g.bits[(size_t)(a * g.W)], g.W, [&](int b) {
        if (group[(size_t)a] != group[(size_t)b]) q.set(group[(size_t)a], group[(size_t)b]);
      });
    std::vector<int> gorder;
    if (method == BA_ORDER_AMD) order_md(q, &gorder);
    else order_nd(q, &gorder);
    std::vector<std::vector<int>> members((size_t)ng);
    for (int64_t k = 0; k < n; k++) members[(size_t)group[(size_t)base[(size_t)k]]].push_back(base[(size_t)k]);
    cand.clear();
    for (int gi : gorder)
      for (int c : members[(size_t)gi]) cand.push_back(c);
    offer(method == BA_ORDER_AMD ? "minimum-degree on tile pairs" : "nested-dissection on tile pairs");
  }
  // ... and the best sequence eliminated FROM BOTH ENDS: the first m cameras (A), then the cameras behind them that are not
  // adjacent to A, LAST FIRST (B, from the far end inwards), then the frontier between the two (S).  A and B share no edge, so
  // their tile column pairs factor side by side as two chains of half the length, and a profile-ordered sequence pays no fill
  // for it (a band eliminated from both ends fills its band, as from one end).  m is a multiple of 256 cameras: only there does
  // a camera boundary (9 rows) coincide with a tile-pair boundary (2 nb rows), and B must start on one.
  {
    const std::vector<int> base = *perm;
    const int64_t unit = 2 * (int64_t)nb / std::gcd((int64_t)9, 2 * (int64_t)nb) ;  // cameras per common boundary (256 for nb = 128)
    const int64_t m = (n / 2 / unit) * unit;
    if (m >= unit && n - m >= 2 * unit) {
      std::vector<char> inA((size_t)n, 0), inS((size_t)n, 0);
      for (int64_t k = 0; k < m; k++) inA[(size_t)base[(size_t)k]] = 1;
      for (int64_t k = m; k < n; k++) {
        const int c = base[(size_t)k];
        bool adj = false;
        for_bits(&g.bits[(size_t)((int64_t)c * g.W)], g.W, [&](int u) { adj |= inA[(size_t)u] != 0; });
        inS[(size_t)c] = adj;
      }
      cand.assign(base.begin(), base.begin() + m);
      for (int64_t k = n - 1; k >= m; k--)
        if (!inS[(size_t)base[(size_t)k]]) cand.push_back(base[(size_t)k]);
      const int64_t nB = (int64_t)cand.size() - m;
      for (int64_t k = m; k < n; k++)
        if (inS[(size_t)base[(size_t)k]]) cand.push_back(base[(size_t)k]);
      if (nB >= unit) {
        cand_split = (int)(9 * m / (2 * (int64_t)nb));
        offer("two-ended");
        cand_split = 0;
      }
    }
  }
}

void cam_tile_occupancy(const CamGraph &g, const std::vector<int> &pos, int64_t nt, int nb, std::vector<unsigned char> *occ) {
  occ->assign((size_t)(nt * nt), 0);
  auto mark = [&](int64_t pa, int64_t pb) {
    if (pa < pb) std::swap(pa, pb);
    const int64_t r0 = 9 * pa, c0 = 9 * pb;
    for (int64_t ti = r0 / nb; ti <= (r0 + 8) / nb; ti++)
      for (int64_t tj = c0 / nb; tj <= (c0 + 8) / nb; tj++)
        if (ti >= tj) (*occ)[(size_t)(ti * nt + tj)] = 1;
  };
  for (int64_t a = 0; a < g.n; a++) {
    mark(pos[(size_t)a], pos[(size_t)a]);
    for_bits(&g.bits[(size_t)(a * g.W)], g.W, [&](int b) {
      if (b < a) mark(pos[(size_t)a], pos[(size_t)b]);
    });
  }
}

// Symbolic factorisation per tile column PAIR (the unit of the schedule): the rows U_q of pair q are the tile rows below it
// with a pattern tile in either column, and every tile (i, j), i >= j, i, j in U_q, joins the pattern (fill).  Taking the
// union of the two columns' rows keeps ONE row list per pair; a tile whose operands are structurally zero receives a zero
// update (correct, a little wasted work when the two columns differ).  Counterpart of src/ldl_aux.jl:82-119.
void tile_pattern_build(int64_t nt, std::vector<unsigned char> &occ /* nt x nt, lower, row-major; gets the fill */, TilePattern *out,
                        int split_pair) {
  const int npairs = (int)((nt + 1) / 2);
  out->nt = nt;
  out->prow_ptr.assign(1, 0);
  out->prow.clear();
  double tiles_sparse = 0, tiles_dense = 0;
  std::vector<int> U;
  for (int q = 0; q < npairs; q++) {
    const int k = 2 * q;
    U.clear();
    for (int64_t i = k + 2; i < nt; i++)
      if (occ[(size_t)(i * nt + k)] || (k + 1 < nt && occ[(size_t)(i * nt + k + 1)])) U.push_back((int)i);
    for (size_t a = 0; a < U.size(); a++)
      for (size_t b = 0; b <= a; b++) occ[(size_t)((int64_t)U[a] * nt + U[b])] = 1;
    if (k + 1 < nt) {
      occ[(size_t)((int64_t)(k + 1) * nt + k)] = 1;
      out->prow.push_back(k + 1);  // the list of pair q starts with tile row k+1 (the panel solve of column k needs it)
      for (int i : U) {
        occ[(size_t)((int64_t)i * nt + k)] = 1;  // union of the two columns' rows
        occ[(size_t)((int64_t)i * nt + k + 1)] = 1;
      }
    }
    for (int i : U) out->prow.push_back(i);
    out->prow_ptr.push_back((int)out->prow.size());
    const double m = (double)(nt - k - 2 > 0 ? nt - k - 2 : 0), u = (double)U.size();
    tiles_sparse += u * (u + 1) / 2;
    tiles_dense += m * (m + 1) / 2;
  }
  out->lcol_ptr.assign(1, 0);
  out->lcol.clear();
  int64_t ntiles = 0;
  for (int64_t i = 0; i < nt; i++) {
    for (int64_t j = 0; j < i; j++)
      if (occ[(size_t)(i * nt + j)]) {
        out->lcol.push_back((int)j);
        ntiles++;
      }
    out->lcol_ptr.push_back((int)out->lcol.size());
  }
  // backward sweep, the two tile rows of a tile column pair per launch (rows 2q + 1, 2q; a last single row when nt is odd takes
  // its lcol list): the union of their pattern columns < 2q
  out->lpair_ptr.assign(1, 0);
  out->lpair.clear();
  for (int64_t q = 0; q < npairs; q++) {
    const int64_t k0 = 2 * q, k1 = 2 * q + 1;
    for (int64_t j = 0; j < k0; j++)
      if (occ[(size_t)(k0 * nt + j)] || (k1 < nt && occ[(size_t)(k1 * nt + j)])) out->lpair.push_back((int)j);
    out->lpair_ptr.push_back((int)out->lpair.size());
  }
  out->tile_fill = (double)(ntiles + nt) / ((double)nt * (double)(nt + 1) / 2);
  out->flop_fill = tiles_dense > 0 ? tiles_sparse / tiles_dense : 1.0;
  // two chains: the longest run of pairs from 0 whose rows stay below the split (nothing of theirs touches the second group),
  // and the longest run from the split whose own columns no pair of the first group touches (what the late pairs of the first
  // group still update -- the frontier -- must not have been eliminated yet)
  out->split = out->a_clean = out->b_clean = 0;
  if (split_pair > 0 && split_pair < npairs) {
    const int a = split_pair;
    std::vector<unsigned char> dirty((size_t)nt, 0);  // tile rows >= 2a that a pair of the first group touches
    int ac = -1;
    for (int q = 0; q < a; q++) {
      bool clean = true;
      for (int l = out->prow_ptr[(size_t)q]; l < out->prow_ptr[(size_t)q + 1]; l++)
        if (out->prow[(size_t)l] >= 2 * a) {
          dirty[(size_t)out->prow[(size_t)l]] = 1;
          clean = false;
        }
      if (!clean && ac < 0) ac = q;
    }
    if (ac < 0) ac = a;
    int bc = 0;
    while (a + bc < npairs && !dirty[(size_t)(2 * (a + bc))] && (2 * (a + bc) + 1 >= nt || !dirty[(size_t)(2 * (a + bc) + 1)])) bc++;
    if (std::min(ac, bc) >= 4) {
      out->split = a;
      out->a_clean = ac;
      out->b_clean = bc;
    }
  }
}

// From the index arrays of a BAL problem (1-based, any observation order): camera ordering by `method`, and the fill of the
// tile pattern it leaves.  perm1 (ncams, may be null): perm1[k] = the (1-based) camera at block row k of S.
int schur_ordering_host(int64_t ncams, int64_t npnts, int64_t nobs, const int64_t *cam_idx1, const int64_t *pnt_idx1, int method,
                        int nb, int64_t *perm1, double *tile_fill, double *flop_fill, double *block_fill) {
  if (ncams <= 0 || npnts < 0 || nobs < 0 || (nobs > 0 && (!cam_idx1 || !pnt_idx1))) return 1;
  std::vector<int> ptr((size_t)npnts + 1, 0), obs((size_t)nobs), cam0((size_t)nobs);
  for (int64_t k = 0; k < nobs; k++) {
    if (cam_idx1[k] < 1 || cam_idx1[k] > ncams || pnt_idx1[k] < 1 || pnt_idx1[k] > npnts) return 1;
    ptr[(size_t)pnt_idx1[k]]++;
    cam0[(size_t)k] = (int)(cam_idx1[k] - 1);
  }
  for (int64_t i = 0; i < npnts; i++) ptr[(size_t)i + 1] += ptr[(size_t)i];
  {
    std::vector<int> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t k = 0; k < nobs; k++) obs[(size_t)cur[(size_t)(pnt_idx1[k] - 1)]++] = (int)k;
  }
  CamGraph g;
  cam_graph_build(ncams, npnts, ptr.data(), obs.data(), cam0.data(), &g);
  std::vector<int> perm, pos((size_t)ncams);
  int split = 0;
  cam_order(g, method, nb, &perm, nullptr, &split);
  if ((int64_t)perm.size() != ncams) return 2;
  for (int64_t k = 0; k < ncams; k++) pos[(size_t)perm[(size_t)k]] = (int)k;
  const int64_t nt = std::max<int64_t>(1, (9 * ncams + nb - 1) / nb);
  std::vector<unsigned char> occ;
  cam_tile_occupancy(g, pos, nt, nb, &occ);
  TilePattern pat;
  tile_pattern_build(nt, occ, &pat, split);
  if (perm1)
    for (int64_t k = 0; k < ncams; k++) perm1[k] = perm[(size_t)k] + 1;
  if (tile_fill) *tile_fill = pat.tile_fill;
  if (flop_fill) *flop_fill = pat.flop_fill;
  if (block_fill) *block_fill = (double)(g.edges() + ncams) / ((double)ncams * (double)(ncams + 1) / 2);
  return 0;
}
