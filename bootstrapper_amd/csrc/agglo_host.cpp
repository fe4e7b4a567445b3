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
