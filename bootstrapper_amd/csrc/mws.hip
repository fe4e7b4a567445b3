// Mutex watershed (reference post/mws.py:51-56 `mwatershed.agglom`, post/watershed_mutex.py:153-160 volara `AffAgglom`
// and `GraphMWS`; the packages themselves are [EXT], absent from /root/reference: restated from the published
// algorithm, Wolf et al., "The Mutex Watershed", ECCV 2018 -- parity unpinned, see oracle/mws_ref.py).
//
//   edges  = every (voxel p, p + offset_k) inside the volume (subsampled by stride_k), weight w = affs[k][p]
//   order  = |w| descending; ties in the order (k, p) ascending (a stable sort of the edges listed k-major)
//   sweep  = for an attractive edge (w > 0): unite the two clusters unless a mutex constraint separates them;
//            for a repulsive edge (w < 0): add a mutex constraint between the two clusters; w == 0: nothing
//   label  = 1 + the smallest voxel index of the cluster
//
// The sweep is sequential by definition (each decision depends on all earlier ones), so the work is split as the
// data suggests: the device lists and sorts the edges (one key kernel + a hipcub radix sort of 8-byte keys with 8-byte
// values), the host walks them with a union-find whose clusters carry their mutex constraints as sorted lists of
// constraint ids (a check is an intersection test, a union a merge of two sorted lists).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/bsmi.h"
#include "common.h"

#include "dev_guard.h"  // last: routes hipMalloc / hipFree through the guarded allocator (BSMI_GUARD_MB)

namespace bsmi {
namespace {

struct MwsGeom {
  int K;
  int D, H, W;
};

// key = bit pattern of |w| (monotonic for non-negative doubles), 0 for edges that do not exist;
// value = edge id (k * N + p) with the sign of w in bit 63
__global__ void mws_keys_kernel(const double* affs, const int32_t* offs, const int32_t* strides, uint32_t rand_seed, MwsGeom g,
                                uint64_t* keys, uint64_t* vals) {
  const size_t N = (size_t)g.D * g.H * g.W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * g.K) return;
  const int k = (int)(i / N);
  const size_t p = i - (size_t)k * N;
  const int x = (int)(p % g.W), y = (int)((p / g.W) % g.H), z = (int)(p / ((size_t)g.W * g.H));
  const int oz = offs[3 * k], oy = offs[3 * k + 1], ox = offs[3 * k + 2];
  bool ok = z + oz >= 0 && z + oz < g.D && y + oy >= 0 && y + oy < g.H && x + ox >= 0 && x + ox < g.W;
  if (ok && strides) {
    const int sz = strides[3 * k], sy = strides[3 * k + 1], sx = strides[3 * k + 2];
    if (rand_seed) {  // randomized strides: keep an edge with probability 1 / (sz sy sx), counter-based hash of (seed, edge)
      uint64_t h = (i + 1) * 0x9E3779B97F4A7C15ull ^ ((uint64_t)rand_seed << 32);
      h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
      const uint64_t vol = (uint64_t)sz * sy * sx;
      ok = vol <= 1 || h % vol == 0;
    } else {
      ok = z % sz == 0 && y % sy == 0 && x % sx == 0;
    }
  }
  const double w = ok ? affs[i] : 0.0;
  const double a = fabs(w);
  uint64_t key = 0;
  if (a > 0.0) memcpy(&key, &a, 8);  // NaN compares false: no edge
  keys[i] = key;
  vals[i] = (uint64_t)i | (w < 0.0 ? (1ull << 63) : 0ull);
}

// union-find over clusters that carry their mutex constraints
struct MutexForest {
  std::vector<uint32_t> parent;
  std::vector<uint8_t> rank;
  std::vector<std::vector<uint32_t>> mutex;  // per root: ids of the constraints it takes part in, ascending
  uint32_t next_id = 0;

  explicit MutexForest(size_t n) : parent(n), rank(n, 0), mutex(n) { std::iota(parent.begin(), parent.end(), 0u); }
  uint32_t find(uint32_t x) {
    while (parent[x] != x) {
      parent[x] = parent[parent[x]];
      x = parent[x];
    }
    return x;
  }
  static bool intersects(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    size_t i = 0, j = 0;
    while (i < a.size() && j < b.size()) {
      if (a[i] == b[j]) return true;
      if (a[i] < b[j]) ++i; else ++j;
    }
    return false;
  }
  void edge(uint32_t u, uint32_t v, bool repulsive) {
    uint32_t a = find(u), b = find(v);
    if (a == b) return;
    if (repulsive) {
      if (intersects(mutex[a], mutex[b])) return;  // already separated
      mutex[a].push_back(next_id);                  // ids grow with the sweep: the lists stay sorted
      mutex[b].push_back(next_id);
      ++next_id;
      return;
    }
    if (intersects(mutex[a], mutex[b])) return;
    if (rank[a] < rank[b]) std::swap(a, b);
    parent[b] = a;
    if (rank[a] == rank[b]) ++rank[a];
    if (!mutex[b].empty()) {
      if (mutex[a].empty()) {
        mutex[a].swap(mutex[b]);
      } else {
        std::vector<uint32_t> m(mutex[a].size() + mutex[b].size());
        std::merge(mutex[a].begin(), mutex[a].end(), mutex[b].begin(), mutex[b].end(), m.begin());
        mutex[a].swap(m);
        std::vector<uint32_t>().swap(mutex[b]);
      }
    }
  }
  // label = 1 + smallest member index of the cluster
  void labels(uint64_t* out) {
    const size_t n = parent.size();
    std::vector<uint32_t> lowest(n, 0xffffffffu);
    for (size_t i = 0; i < n; ++i) {
      const uint32_t r = find((uint32_t)i);
      if (lowest[r] == 0xffffffffu) lowest[r] = (uint32_t)i;  // i ascending: the first member seen is the smallest
    }
    for (size_t i = 0; i < n; ++i) out[i] = (uint64_t)lowest[find((uint32_t)i)] + 1;
  }
};

}  // namespace
}  // namespace bsmi

using namespace bsmi;

extern "C" {

int bsmi_mws_agglom_f64(int device, const double* affs_dev, int n_offsets, const int32_t* offsets, const int32_t* strides,
                        uint32_t random_seed, const int64_t shape[3], uint64_t* labels_dev, void* stream) {
  if (!affs_dev || !offsets || !shape || !labels_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n_offsets < 1 || n_offsets > 1024) BSMI_FAIL(BSMI_ERR_INVALID, "%d offsets", n_offsets);
  for (int d = 0; d < 3; ++d)
    if (shape[d] <= 0) BSMI_FAIL(BSMI_ERR_INVALID, "bad shape (%lld,%lld,%lld)", (long long)shape[0], (long long)shape[1], (long long)shape[2]);
  const size_t N = (size_t)shape[0] * shape[1] * shape[2];
  if (N >= 0xffffffffull) BSMI_FAIL(BSMI_ERR_INVALID, "volume of %zu voxels: the cluster indices are 32-bit", N);
  const size_t E = N * (size_t)n_offsets;
  if (E >= (size_t)1 << 31) BSMI_FAIL(BSMI_ERR_INVALID, "%zu candidate edges: more than one radix sort takes (cut the volume into blocks)", E);
  if (strides)
    for (int i = 0; i < 3 * n_offsets; ++i)
      if (strides[i] < 1) BSMI_FAIL(BSMI_ERR_INVALID, "stride %d", strides[i]);
  BSMI_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  int32_t* d_offs = nullptr;
  int32_t* d_str = nullptr;
  uint64_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  int rc = BSMI_OK;
  std::vector<uint64_t> order;
  std::vector<uint64_t> skeys;
  auto cleanup = [&]() {
    (void)hipFree(d_offs); (void)hipFree(d_str); (void)hipFree(keys); (void)hipFree(vals); (void)hipFree(keys2); (void)hipFree(vals2); (void)hipFree(tmp);
  };
#define MWS_HIP(x)                                                                       \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      cleanup();                                                                         \
      BSMI_FAIL(BSMI_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_));                      \
    }                                                                                    \
  } while (0)
  MWS_HIP(hipMalloc((void**)&d_offs, sizeof(int32_t) * 3 * n_offsets));
  MWS_HIP(hipMemcpyAsync(d_offs, offsets, sizeof(int32_t) * 3 * n_offsets, hipMemcpyHostToDevice, s));
  if (strides) {
    MWS_HIP(hipMalloc((void**)&d_str, sizeof(int32_t) * 3 * n_offsets));
    MWS_HIP(hipMemcpyAsync(d_str, strides, sizeof(int32_t) * 3 * n_offsets, hipMemcpyHostToDevice, s));
  }
  MWS_HIP(hipMalloc((void**)&keys, E * 8));
  MWS_HIP(hipMalloc((void**)&vals, E * 8));
  MWS_HIP(hipMalloc((void**)&keys2, E * 8));
  MWS_HIP(hipMalloc((void**)&vals2, E * 8));
  const MwsGeom g{n_offsets, (int)shape[0], (int)shape[1], (int)shape[2]};
  hipLaunchKernelGGL(mws_keys_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, affs_dev, d_offs, d_str, strides ? random_seed : 0u, g,
                     keys, vals);
  MWS_HIP(hipGetLastError());
  MWS_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, keys, keys2, vals, vals2, (int)E, 0, 64, s));
  MWS_HIP(hipMalloc(&tmp, tb));
  MWS_HIP(hipcub::DeviceRadixSort::SortPairsDescending(tmp, tb, keys, keys2, vals, vals2, (int)E, 0, 64, s));  // stable: ties stay (k, p) ascending
  order.resize(E);
  skeys.resize(E);
  MWS_HIP(hipMemcpyAsync(order.data(), vals2, E * 8, hipMemcpyDeviceToHost, s));
  MWS_HIP(hipMemcpyAsync(skeys.data(), keys2, E * 8, hipMemcpyDeviceToHost, s));
  MWS_HIP(hipStreamSynchronize(s));
  {
    MutexForest f(N);
    std::vector<int64_t> lin(n_offsets);
    for (int k = 0; k < n_offsets; ++k)
      lin[k] = ((int64_t)offsets[3 * k] * shape[1] + offsets[3 * k + 1]) * shape[2] + offsets[3 * k + 2];
    for (size_t i = 0; i < E && skeys[i] != 0; ++i) {
      const uint64_t v = order[i];
      const uint64_t e = v & ~(1ull << 63);
      const size_t k = e / N, p = e - k * N;
      f.edge((uint32_t)p, (uint32_t)((int64_t)p + lin[k]), (v >> 63) != 0);
    }
    std::vector<uint64_t> lab(N);
    f.labels(lab.data());
    MWS_HIP(hipMemcpyAsync(labels_dev, lab.data(), N * 8, hipMemcpyHostToDevice, s));
    MWS_HIP(hipStreamSynchronize(s));
  }
#undef MWS_HIP
  cleanup();
  return rc;
}

int bsmi_mws_cluster(uint64_t n_nodes, const uint64_t* edges, const double* scores, uint64_t m, uint64_t* labels_out) {
  if ((m && (!edges || !scores)) || (n_nodes && !labels_out)) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n_nodes >= 0xffffffffull) BSMI_FAIL(BSMI_ERR_INVALID, "%llu nodes: the cluster indices are 32-bit", (unsigned long long)n_nodes);
  for (uint64_t i = 0; i < m; ++i)
    if (edges[2 * i] >= n_nodes || edges[2 * i + 1] >= n_nodes)
      BSMI_FAIL(BSMI_ERR_INVALID, "edge %llu names node %llu / %llu of %llu", (unsigned long long)i, (unsigned long long)edges[2 * i],
                (unsigned long long)edges[2 * i + 1], (unsigned long long)n_nodes);
  std::vector<uint64_t> order(m);
  std::iota(order.begin(), order.end(), (uint64_t)0);
  std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return std::fabs(scores[a]) > std::fabs(scores[b]); });
  MutexForest f(n_nodes);
  for (uint64_t i : order) {
    const double w = scores[i];
    if (!(std::fabs(w) > 0.0)) continue;  // 0 and NaN: no edge
    f.edge((uint32_t)edges[2 * i], (uint32_t)edges[2 * i + 1], w < 0.0);
  }
  f.labels(labels_out);
  return BSMI_OK;
}

}  // extern "C"

// ---- mean affinity between adjacent fragments (volara AffAgglom [EXT]; reference post/watershed_mutex.py:143-153) ----
namespace bsmi {
namespace {

// pair key = (smaller dense index << 32 | larger), 0 = no pair; value = the affinity byte
__global__ void pair_keys_kernel(const uint8_t* affs, const uint64_t* dense, const int32_t* offs, MwsGeom g, uint64_t* keys, uint32_t* vals) {
  const size_t N = (size_t)g.D * g.H * g.W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * g.K) return;
  const int k = (int)(i / N);
  const size_t p = i - (size_t)k * N;
  const int x = (int)(p % g.W), y = (int)((p / g.W) % g.H), z = (int)(p / ((size_t)g.W * g.H));
  const int oz = offs[3 * k], oy = offs[3 * k + 1], ox = offs[3 * k + 2];
  uint64_t key = 0;
  if (z + oz >= 0 && z + oz < g.D && y + oy >= 0 && y + oy < g.H && x + ox >= 0 && x + ox < g.W) {
    const uint64_t a = dense[p], b = dense[((size_t)(z + oz) * g.H + (y + oy)) * g.W + (x + ox)];
    if (a && b && a != b) key = a < b ? (a << 32 | b) : (b << 32 | a);
  }
  keys[i] = key;
  vals[i] = affs[i];
}

struct Widen {
  const uint32_t* in;
  __host__ __device__ uint64_t operator()(size_t i) const { return in[i]; }
};

__global__ void ones_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 1u;
}

}  // namespace
}  // namespace bsmi

extern "C" {

int bsmi_frag_pair_affinity_u8(int device, const uint8_t* affs_dev, int n_offsets, const int32_t* offsets, const uint64_t* dense_dev,
                               const int64_t shape[3], uint64_t capacity, uint64_t* pairs_out, uint64_t* sums_out, uint64_t* counts_out,
                               uint64_t* n_pairs, void* stream) {
  if (!affs_dev || !offsets || !dense_dev || !shape || !n_pairs) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (n_offsets < 1 || n_offsets > 1024) BSMI_FAIL(BSMI_ERR_INVALID, "%d offsets", n_offsets);
  const size_t N = (size_t)shape[0] * shape[1] * shape[2];
  const size_t E = N * (size_t)n_offsets;
  if (N == 0 || E >= (size_t)1 << 31) BSMI_FAIL(BSMI_ERR_INVALID, "%zu candidate pairs: out of range for one sort", E);
  BSMI_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  int32_t* d_offs = nullptr;
  uint64_t *keys = nullptr, *keys2 = nullptr, *ukeys = nullptr, *usum = nullptr;
  uint32_t *vals = nullptr, *vals2 = nullptr, *ones = nullptr, *ucnt = nullptr;
  int* d_n = nullptr;
  void* tmp = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_offs); (void)hipFree(keys); (void)hipFree(keys2); (void)hipFree(ukeys); (void)hipFree(usum); (void)hipFree(vals);
    (void)hipFree(vals2); (void)hipFree(ones); (void)hipFree(ucnt); (void)hipFree(d_n); (void)hipFree(tmp);
  };
#define PA_HIP(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      cleanup();                                                                         \
      BSMI_FAIL(BSMI_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_));                      \
    }                                                                                    \
  } while (0)
  PA_HIP(hipMalloc((void**)&d_offs, sizeof(int32_t) * 3 * n_offsets));
  PA_HIP(hipMemcpyAsync(d_offs, offsets, sizeof(int32_t) * 3 * n_offsets, hipMemcpyHostToDevice, s));
  PA_HIP(hipMalloc((void**)&keys, E * 8));
  PA_HIP(hipMalloc((void**)&keys2, E * 8));
  PA_HIP(hipMalloc((void**)&vals, E * 4));
  PA_HIP(hipMalloc((void**)&vals2, E * 4));
  PA_HIP(hipMalloc((void**)&ones, E * 4));
  PA_HIP(hipMalloc((void**)&ukeys, E * 8));
  PA_HIP(hipMalloc((void**)&usum, E * 8));
  PA_HIP(hipMalloc((void**)&ucnt, E * 4));
  PA_HIP(hipMalloc((void**)&d_n, sizeof(int)));
  const MwsGeom g{n_offsets, (int)shape[0], (int)shape[1], (int)shape[2]};
  const unsigned grid = (unsigned)((E + 255) / 256);
  hipLaunchKernelGGL(pair_keys_kernel, dim3(grid), dim3(256), 0, s, affs_dev, dense_dev, d_offs, g, keys, vals);
  hipLaunchKernelGGL(ones_kernel, dim3(grid), dim3(256), 0, s, ones, E);
  PA_HIP(hipGetLastError());
  // sums of up to 2^31 bytes: the sorted values are widened to 64 bits on the fly
  hipcub::CountingInputIterator<size_t> cnt(0);
  hipcub::TransformInputIterator<uint64_t, Widen, hipcub::CountingInputIterator<size_t>> wide(cnt, Widen{vals2});
  size_t tb = 0, tb2 = 0, tb3 = 0;
  PA_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, keys2, vals, vals2, (int)E, 0, 64, s));
  PA_HIP(hipcub::DeviceReduce::ReduceByKey(nullptr, tb2, keys2, ukeys, wide, usum, d_n, hipcub::Sum(), (int)E, s));
  PA_HIP(hipcub::DeviceReduce::ReduceByKey(nullptr, tb3, keys2, ukeys, ones, ucnt, d_n, hipcub::Sum(), (int)E, s));
  tb = std::max(tb, std::max(tb2, tb3));
  PA_HIP(hipMalloc(&tmp, tb));
  PA_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys2, vals, vals2, (int)E, 0, 64, s));
  PA_HIP(hipcub::DeviceReduce::ReduceByKey(tmp, tb, keys2, ukeys, wide, usum, d_n, hipcub::Sum(), (int)E, s));
  PA_HIP(hipcub::DeviceReduce::ReduceByKey(tmp, tb, keys2, ukeys, ones, ucnt, d_n, hipcub::Sum(), (int)E, s));
  int nuniq = 0;
  PA_HIP(hipMemcpyAsync(&nuniq, d_n, sizeof(int), hipMemcpyDeviceToHost, s));
  PA_HIP(hipStreamSynchronize(s));
  // the run of key 0 (no pair) sorts first
  std::vector<uint64_t> hk((size_t)nuniq), hs((size_t)nuniq);
  std::vector<uint32_t> hc((size_t)nuniq);
  if (nuniq) {
    PA_HIP(hipMemcpy(hk.data(), ukeys, (size_t)nuniq * 8, hipMemcpyDeviceToHost));
    PA_HIP(hipMemcpy(hs.data(), usum, (size_t)nuniq * 8, hipMemcpyDeviceToHost));
    PA_HIP(hipMemcpy(hc.data(), ucnt, (size_t)nuniq * 4, hipMemcpyDeviceToHost));
  }
#undef PA_HIP
  cleanup();
  const size_t first = (nuniq && hk[0] == 0) ? 1 : 0;
  const uint64_t np = (uint64_t)nuniq - first;
  *n_pairs = np;
  if (np > capacity) BSMI_FAIL(BSMI_ERR_INVALID, "%llu fragment pairs, room for %llu", (unsigned long long)np, (unsigned long long)capacity);
  if (np && (!pairs_out || !sums_out || !counts_out)) BSMI_FAIL(BSMI_ERR_INVALID, "null output");
  for (uint64_t i = 0; i < np; ++i) {
    pairs_out[2 * i] = hk[first + i] >> 32;
    pairs_out[2 * i + 1] = hk[first + i] & 0xffffffffull;
    sums_out[i] = hs[first + i];
    counts_out[i] = hc[first + i];
  }
  return BSMI_OK;
}

}  // extern "C"
