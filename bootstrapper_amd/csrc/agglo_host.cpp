// Hierarchical agglomeration with the histogram-quantile scorers on the host.
//
// The reference's non-blockwise `bs segment --ws` lets the user pick one of ten
// OneMinus<HistogramQuantileAffinity<RegionGraphType, Q, ScoreValue, 256, InitWithMax>> scorers besides the mean
// (/root/reference/bootstrapper/post/watershed.py:230-243) and hands the whole loop to waterz, host C++.  Here the device
// builds the region graph and the per-edge affinity histograms (seg.hip: agg_edges_kernel, agg_hist_kernel -- the
// bandwidth-bound part) and relabels the volume; the merge loop itself is a sequential priority-queue algorithm over a
// graph that shrinks as it goes, with a 256-bin histogram added per contraction, and runs here as waterz's does.  Only the
// mean scorer, the one the blockwise path is restricted to, has a device merge loop (seg.hip: agg_merge_kernel).
//
// Algorithm: IterativeRegionMerging::mergeUntil / mergeRegions as specified in oracle/seg_ref.c (the same rule the device
// loop follows): pop by the STORED score, a stale edge is re-scored and pushed back, on a merge the dearer of two edges to a
// shared neighbour is folded into the cheaper one, every edge at the survivor becomes stale; ties by the edge's initial key.
// Score of an edge: pivot = Q * total / 100 + 1 (1-based), the first bin whose running count reaches it, 1 - (bin + 0.5) / 256.
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <queue>
#include <unordered_map>
#include <vector>

#include "../../include/bsmi.h"

namespace bsmi {

namespace {

struct QItem {
  float score;
  uint64_t key0;
  uint32_t e;
};
struct QLater {  // std::priority_queue keeps the LARGEST on top: order by "comes later"
  bool operator()(const QItem& a, const QItem& b) const {
    if (a.score != b.score) return a.score > b.score;
    return a.key0 > b.key0;
  }
};

inline float quantile_score(const uint32_t* hist, int quantile) {
  uint64_t total = 0;
  for (int b = 0; b < 256; ++b) total += hist[b];
  const uint64_t pivot = (uint64_t)quantile * total / 100 + 1;
  uint64_t run = 0;
  int bin = 0;
  for (; bin < 256; ++bin) {
    run += hist[bin];
    if (run >= pivot) break;
  }
  if (bin > 255) bin = 255;
  return 1.0f - ((float)bin + 0.5f) / 256.0f;
}

}  // namespace

// nn nodes (ranks in ascending id order), ne edges with initial endpoints eu[e] < ev[e] (each pair once), hist [ne][256]
// (modified: histograms are added up as edges merge).  init_with_max: every edge starts from ONE entry, its largest affinity.
// roots_out [nthr][nn]: rank of the node's cluster after mergeUntil(thresholds[t]) (a cluster is named by its smallest rank).
void host_agglomerate_hist(uint32_t nn, uint32_t ne, const uint32_t* eu, const uint32_t* ev, uint32_t* hist, int quantile,
                           int init_with_max, const float* thresholds, int nthr, uint32_t* roots_out) {
  std::vector<uint32_t> u(eu, eu + ne), v(ev, ev + ne);
  std::vector<uint64_t> key0(ne);
  std::vector<float> stored(ne);
  std::vector<uint8_t> deleted(ne, 0), stale(ne, 0);
  std::vector<std::unordered_map<uint32_t, uint32_t>> adj(nn);  // neighbour -> live edge
  std::priority_queue<QItem, std::vector<QItem>, QLater> queue;
  for (uint32_t e = 0; e < ne; ++e) {
    uint32_t* h = hist + (std::size_t)e * 256;
    if (init_with_max) {
      int top = 255;
      while (top > 0 && !h[top]) --top;
      for (int b = 0; b < 256; ++b) h[b] = 0;
      h[top] = 1;
    }
    key0[e] = ((uint64_t)u[e] << 32) | v[e];
    stored[e] = quantile_score(h, quantile);
    adj[u[e]][v[e]] = e;
    adj[v[e]][u[e]] = e;
    queue.push(QItem{stored[e], key0[e], e});
  }
  std::vector<uint32_t> parent(nn);
  for (uint32_t i = 0; i < nn; ++i) parent[i] = i;

  for (int t = 0; t < nthr; ++t) {
    const float thr = thresholds[t];
    while (!queue.empty() && queue.top().score < thr) {
      const uint32_t e = queue.top().e;
      queue.pop();
      if (deleted[e]) continue;
      if (stale[e]) {
        stale[e] = 0;
        stored[e] = quantile_score(hist + (std::size_t)e * 256, quantile);
        queue.push(QItem{stored[e], key0[e], e});
        continue;
      }
      const uint32_t a = u[e] < v[e] ? u[e] : v[e], b = u[e] < v[e] ? v[e] : u[e];
      for (auto& kv : adj[a]) stale[kv.second] = 1;
      for (auto& kv : adj[b]) {
        const uint32_t n = kv.first, f = kv.second;
        if (f == e) continue;
        adj[n].erase(b);
        auto it = adj[a].find(n);
        if (it != adj[a].end() && stored[f] > stored[it->second]) {  // shared neighbour, f is the dearer edge: folded into g
          const uint32_t g = it->second;
          uint32_t *hg = hist + (std::size_t)g * 256, *hf = hist + (std::size_t)f * 256;
          for (int k = 0; k < 256; ++k) hg[k] += hf[k];
          deleted[f] = 1;
          stale[g] = 1;
          continue;
        }
        if (it != adj[a].end()) {  // shared neighbour, g is the dearer (or an equal) edge: folded into f, which moves to {a, n}
          const uint32_t g = it->second;
          uint32_t *hg = hist + (std::size_t)g * 256, *hf = hist + (std::size_t)f * 256;
          for (int k = 0; k < 256; ++k) hf[k] += hg[k];
          deleted[g] = 1;
        }
        u[f] = a < n ? a : n;
        v[f] = a < n ? n : a;
        stale[f] = 1;
        adj[a][n] = f;
        adj[n][a] = f;
      }
      adj[a].erase(b);
      adj[b].clear();
      deleted[e] = 1;
      parent[b] = a;
    }
    uint32_t* roots = roots_out + (std::size_t)t * nn;
    for (uint32_t i = 0; i < nn; ++i) {
      uint32_t r = i;
      while (parent[r] != r) r = parent[r];
      roots[i] = r;
    }
  }
}

}  // namespace bsmi

extern "C" int bsmi_agglomerate_hist_graph(uint32_t n_nodes, uint32_t n_edges, const uint32_t* edge_u, const uint32_t* edge_v,
                                           uint32_t* hist, int quantile, int init_with_max, const float* thresholds,
                                           int n_thresholds, uint32_t* roots_out) {
  if ((n_edges && (!edge_u || !edge_v || !hist)) || !thresholds || !roots_out || n_thresholds < 1 || quantile < 0 || quantile > 100)
    return BSMI_ERR_INVALID;
  for (uint32_t e = 0; e < n_edges; ++e)
    if (edge_u[e] >= edge_v[e] || edge_v[e] >= n_nodes) return BSMI_ERR_INVALID;
  for (int t = 1; t < n_thresholds; ++t)
    if (thresholds[t] < thresholds[t - 1]) return BSMI_ERR_INVALID;
  bsmi::host_agglomerate_hist(n_nodes, n_edges, edge_u, edge_v, hist, quantile, init_with_max, thresholds, n_thresholds, roots_out);
  return BSMI_OK;
}

// ---- blockwise RAG scoring on the host -------------------------------------------------------------------------------
// post/blockwise/waterz_agglom.py:106-170 scores the edges of a block's region graph by agglomerating it to the end
// (waterz, OneMinus<MeanAffinity>, discretize_queue bins) and reading every edge's score off the merge tree.  The device
// builds the graph (seg.hip: rag_ids / agg_edges / rag_compact kernels -- the pass over the 160^3 read box); the merge
// loop is a sequential queue algorithm over a few thousand edges, which one wave replays in 12 ms per block (seg.hip:
// rag_merge_kernel, kept: bsmi_rag_merge_scores_u8) and a host core in a fraction of a millisecond.  The block pipeline
// (volume.SlabSegmenter) needs the scores on the host anyway -- the global connected components run there -- so it exports
// the graphs (bsmi_rag_graph_u8) and calls this for all blocks of a slab at once, blocks side by side on host threads.
//
// Algorithm: exactly rag_merge_body / agg_contract / rag_scores_kernel of seg.hip (which are bit-exact to oracle/seg_ref.c):
// scores 1 - sum / (255 cnt) in double, rounded to float; bin queue of `nbins` FIFO bins, bin = (int)(score * (nbins - 1));
// initial pushes in edge order; a popped edge whose endpoint merged after its last scoring is re-scored and re-queued; a
// merge absorbs the larger rank into the smaller, folds the dearer (by STORED score) of two parallel edges into the cheaper;
// tree node nn + m for merge m; an edge's score is that of the lowest common ancestor of its two fragments (NaN: none).
#include <atomic>
#include <cmath>
#include <thread>

namespace bsmi {

namespace {

constexpr uint32_t kNoEdge = 0xffffffffu;

// (node pair) -> edge: open addressing, linear probing, deletion by backward shift (no tombstones: the merge loop erases and
// inserts a key per relinked edge).  std::unordered_map here was most of the loop's time (a heap node per key).
struct PairMap {
  static constexpr uint64_t kEmpty = ~0ull;
  std::vector<uint64_t> k;
  std::vector<uint32_t> v;
  uint32_t mask;
  int shift;
  explicit PairMap(uint64_t n) {
    int bits = 4;
    while ((1ull << bits) < 2 * n + 8) ++bits;
    k.assign((size_t)1 << bits, kEmpty);
    v.resize((size_t)1 << bits);
    mask = (uint32_t)((1ull << bits) - 1);
    shift = 64 - bits;
  }
  uint32_t home(uint64_t key) const { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift); }
  // slot of `key`, or the empty slot where it would go
  uint32_t slot(uint64_t key) const {
    uint32_t i = home(key);
    while (k[i] != kEmpty && k[i] != key) i = (i + 1) & mask;
    return i;
  }
  void erase_slot(uint32_t i) {
    uint32_t j = i;
    for (;;) {
      j = (j + 1) & mask;
      if (k[j] == kEmpty) break;
      const uint32_t h = home(k[j]);
      // the entry at j may move to the hole at i unless its home lies cyclically in (i, j]
      const bool stays = i <= j ? (h > i && h <= j) : (h > i || h <= j);
      if (!stays) { k[i] = k[j]; v[i] = v[j]; i = j; }
    }
    k[i] = kEmpty;
  }
  void erase(uint64_t key) {
    const uint32_t i = slot(key);
    if (k[i] == key) erase_slot(i);
  }
};

void rag_merge_scores_one(uint64_t ne, const uint64_t* edges, const uint64_t* sums, const uint32_t* cnts, float threshold, int nbins,
                          int bin_rule, float* scores) {
  if (!ne) return;
  // The order in which edges leave the bin queue is the algorithm (and is replayed as the device loop and the oracle do it); how
  // the graph is held is not.  Nodes are numbered as they appear (nothing below looks at their order); the edge lists of a node
  // are intrusive (an edge is linked once through each end); of two merging nodes the one with the shorter list is relinked.
  // Which list is walked is free, the TIE RULE between two parallel edges is not: the specification (oracle/seg_ref.c,
  // agg_contract in seg.hip) folds the edge of the ABSORBED region (the larger fragment id) into the survivor's only when it is
  // strictly dearer; on equal stored scores the absorbed region's edge lives on, with its place in its bin.  So every node carries
  // the smallest fragment id of its region, and when the walked list belongs to the region that survives by id the comparison is
  // taken from the other side (>= instead of >).  (Round 3 compared one way whichever list it walked: with tied scores -- u8
  // affinities saturate at 0 / 255, ties are the normal case -- a different edge survived and the merge order diverged.)
  // An edge that meets a parallel one is absorbed by it or absorbs it, and the score of an input edge is the score at which the
  // edge that absorbed it -- through however many steps -- was merged: there is exactly one live edge between two regions, so the
  // merge that joins an input edge's two regions pops that representative.  (The device loop finds the same value by walking a
  // merge tree.)
  PairMap node_of(2 * ne);
  uint32_t nn = 0;
  std::vector<uint64_t> region_id;  // per node: the smallest fragment id of the region it stands for
  region_id.reserve(2 * ne);
  auto number = [&](uint64_t id) {
    const uint32_t sl = node_of.slot(id);
    if (node_of.k[sl] != id) { node_of.k[sl] = id; node_of.v[sl] = nn++; region_id.push_back(id); }
    return node_of.v[sl];
  };
  std::vector<uint32_t> eu(ne), ev(ne), cnt(ne), etime(ne, 0), qnext(ne, kNoEdge), nu(ne), nv(ne), into(ne, kNoEdge);
  std::vector<uint64_t> sum(ne);
  std::vector<float> score(ne), merged_at(ne, std::nanf(""));
  std::vector<uint8_t> dead(ne, 0);
  for (uint32_t e = 0; e < ne; ++e) {
    eu[e] = number(edges[2 * (size_t)e]);
    ev[e] = number(edges[2 * (size_t)e + 1]);
  }
  std::vector<uint32_t> head(nn, kNoEdge), deg(nn, 0), ntime(nn, 0);
  PairMap by_key(ne);
  auto key_of = [](uint32_t a, uint32_t b) { return a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a; };
  auto edge_score = [](uint64_t s, uint32_t c) { return 1.0f - (float)((double)s / (255.0 * (double)c)); };
  for (uint32_t e = 0; e < ne; ++e) {
    const uint32_t a = eu[e], b = ev[e];
    sum[e] = sums[e];
    cnt[e] = cnts[e];
    score[e] = edge_score(sum[e], cnt[e]);
    nu[e] = head[a]; head[a] = e; ++deg[a];
    nv[e] = head[b]; head[b] = e; ++deg[b];
    const uint32_t sl = by_key.slot(key_of(a, b));
    by_key.k[sl] = key_of(a, b);
    by_key.v[sl] = e;
  }
  std::vector<uint32_t> bhead(nbins, kNoEdge), btail(nbins, kNoEdge);
  int minbin = nbins;
  // bin of a score: BSMI_QUEUE_BINS_N_MINUS_1 (the specification, oracle/seg_ref.c) bin = (int)(score * (N - 1)), or
  // BSMI_QUEUE_BINS_N bin = min(N - 1, (int)(score * N)) -- upstream's binning cannot be checked here (waterz absent)
  const float scale = bin_rule == BSMI_QUEUE_BINS_N ? (float)nbins : (float)(nbins - 1);
  auto push = [&](uint32_t e, float sc) {
    int b = (int)(sc * scale);
    b = b < 0 ? 0 : (b > nbins - 1 ? nbins - 1 : b);
    qnext[e] = kNoEdge;
    if (bhead[b] == kNoEdge) bhead[b] = e; else qnext[btail[b]] = e;
    btail[b] = e;
    if (b < minbin) minbin = b;
  };
  for (uint32_t e = 0; e < ne; ++e)
    if (score[e] < threshold) push(e, score[e]);
  uint32_t clock = 0;
  for (;;) {
    uint32_t pick = kNoEdge;
    for (;;) {
      while (minbin < nbins && bhead[minbin] == kNoEdge) ++minbin;
      if (minbin >= nbins) break;
      const uint32_t e = bhead[minbin];
      bhead[minbin] = qnext[e];
      if (dead[e]) continue;
      const uint32_t tu = ntime[eu[e]], tv = ntime[ev[e]];
      if (etime[e] < (tu > tv ? tu : tv)) {
        const float sc = edge_score(sum[e], cnt[e]);
        score[e] = sc;
        etime[e] = clock;
        if (sc < threshold) push(e, sc);
        continue;
      }
      pick = e;
      break;
    }
    if (pick == kNoEdge) break;
    const uint32_t e = pick;
    uint32_t a = eu[e], b = ev[e];
    if (deg[a] < deg[b]) { const uint32_t t = a; a = b; b = t; }
    // b's list is walked.  The specification absorbs the region with the larger id: is that b?
    const bool b_absorbed = region_id[b] > region_id[a];
    for (uint32_t f = head[b]; f != kNoEdge;) {
      const bool uside = eu[f] == b;
      const uint32_t next = uside ? nu[f] : nv[f];
      if (f != e && !dead[f]) {
        const uint32_t nb = uside ? ev[f] : eu[f];
        by_key.erase(key_of(b, nb));
        const uint64_t gkey = key_of(a, nb);
        const uint32_t sl = by_key.slot(gkey);
        const bool found = by_key.k[sl] == gkey;
        bool move_f = true;
        if (found) {
          const uint32_t g = by_key.v[sl];
          if (b_absorbed ? score[f] > score[g] : score[f] >= score[g]) {
            sum[g] += sum[f];
            cnt[g] += cnt[f];
            dead[f] = 1;
            into[f] = g;
            move_f = false;
          } else {
            sum[f] += sum[g];
            cnt[f] += cnt[g];
            dead[g] = 1;
            into[g] = f;
            by_key.v[sl] = f;
          }
        }
        if (move_f) {
          if (uside) { eu[f] = a; nu[f] = head[a]; } else { ev[f] = a; nv[f] = head[a]; }
          head[a] = f;
          ++deg[a];
          if (!found) { by_key.k[sl] = gkey; by_key.v[sl] = f; }
        }
      }
      f = next;
    }
    by_key.erase(key_of(a, b));
    dead[e] = 1;
    merged_at[e] = score[e];
    head[b] = kNoEdge;
    deg[b] = 0;
    if (b_absorbed == false) region_id[a] = region_id[b];
    ntime[a] = ++clock;
  }
  for (uint32_t e = 0; e < ne; ++e) {
    uint32_t r = e;
    while (into[r] != kNoEdge) {
      const uint32_t up = into[r];
      if (into[up] != kNoEdge) into[r] = into[up];  // path halving
      r = up;
    }
    scores[e] = merged_at[r];
  }
}

}  // namespace

}  // namespace bsmi

namespace bsmi {
namespace {
// a graph as bsmi_rag_graph_u8 leaves it (edge hash table order) -> ascending (id, id) order, in place: the order the merge loop's
// queue is filled in, and the one the device loop and the oracle number their edges in
void sort_graph(uint64_t ne, uint64_t* edges, uint64_t* sums, uint32_t* cnts) {
  bool sorted = true;
  for (uint64_t e = 1; e < ne && sorted; ++e)
    sorted = edges[2 * e - 2] < edges[2 * e] || (edges[2 * e - 2] == edges[2 * e] && edges[2 * e - 1] <= edges[2 * e + 1]);
  if (sorted) return;
  std::vector<uint32_t> perm(ne);
  for (uint64_t e = 0; e < ne; ++e) perm[e] = (uint32_t)e;
  std::sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) {
    return edges[2 * (size_t)x] != edges[2 * (size_t)y] ? edges[2 * (size_t)x] < edges[2 * (size_t)y] : edges[2 * (size_t)x + 1] < edges[2 * (size_t)y + 1];
  });
  std::vector<uint64_t> e2(2 * ne), s2(ne);
  std::vector<uint32_t> c2(ne);
  for (uint64_t e = 0; e < ne; ++e) {
    e2[2 * e] = edges[2 * (size_t)perm[e]];
    e2[2 * e + 1] = edges[2 * (size_t)perm[e] + 1];
    s2[e] = sums[perm[e]];
    c2[e] = cnts[perm[e]];
  }
  std::copy(e2.begin(), e2.end(), edges);
  std::copy(s2.begin(), s2.end(), sums);
  std::copy(c2.begin(), c2.end(), cnts);
}
}  // namespace
}  // namespace bsmi

extern "C" int bsmi_rag_merge_scores_host(int n_graphs, const uint64_t* n_edges, uint64_t* const* edges, uint64_t* const* sums,
                                          uint32_t* const* counts, float threshold, int discretize_queue, float* const* scores,
                                          int n_threads) {
  return bsmi_rag_merge_scores_host_rule(n_graphs, n_edges, edges, sums, counts, threshold, discretize_queue, BSMI_QUEUE_BINS_N_MINUS_1,
                                         scores, n_threads);
}

extern "C" int bsmi_rag_merge_scores_host_rule(int n_graphs, const uint64_t* n_edges, uint64_t* const* edges, uint64_t* const* sums,
                                               uint32_t* const* counts, float threshold, int discretize_queue, int bin_rule,
                                               float* const* scores, int n_threads) {
  if (bin_rule != BSMI_QUEUE_BINS_N_MINUS_1 && bin_rule != BSMI_QUEUE_BINS_N) return BSMI_ERR_INVALID;
  if (n_graphs < 0 || (n_graphs && (!n_edges || !edges || !sums || !counts || !scores)) || discretize_queue < 1 || discretize_queue > 1024 ||
      !(threshold > 0.f))
    return BSMI_ERR_INVALID;
  for (int g = 0; g < n_graphs; ++g) {
    if (n_edges[g] >= 0xffffffffull) return BSMI_ERR_INVALID;
    if (n_edges[g] && (!edges[g] || !sums[g] || !counts[g] || !scores[g])) return BSMI_ERR_INVALID;
    for (uint64_t e = 0; e < n_edges[g]; ++e)
      if (!counts[g][e] || edges[g][2 * e] == edges[g][2 * e + 1]) return BSMI_ERR_INVALID;
  }
  // largest graphs first: the threads take them from one queue, and the last ones taken decide when all are done
  std::vector<int> order(n_graphs);
  for (int g = 0; g < n_graphs; ++g) order[g] = g;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return n_edges[x] > n_edges[y]; });
  std::atomic<int> next{0};
  auto work = [&] {
    for (int i = next.fetch_add(1); i < n_graphs; i = next.fetch_add(1)) {
      const int g = order[i];
      bsmi::sort_graph(n_edges[g], edges[g], sums[g], counts[g]);
      bsmi::rag_merge_scores_one(n_edges[g], edges[g], sums[g], counts[g], threshold, discretize_queue, bin_rule, scores[g]);
    }
  };
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  const int nt = std::max(1, std::min(std::min(n_threads > 0 ? n_threads : 24, hw), n_graphs));
  if (nt <= 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work);
    for (auto& x : th) x.join();
  }
  return BSMI_OK;
}
