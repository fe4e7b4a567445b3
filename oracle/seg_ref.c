/* ORACLE (test infrastructure, not product code): plain-C CPU restatement of the
 * segmentation half of the hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; bootstrapper_amd/ never does.
 *
 * Restated reference code (paths relative to /root/reference/bootstrapper):
 *   post/ws.py:8-35    watershed_from_boundary_distance
 *   post/ws.py:38-112  watershed_from_affinities
 *   post/watershed.py:333-338 waterz.agglomerate(affs, thresholds, fragments,
 *                      "OneMinus<MeanAffinity<RegionGraphType, ScoreValue>>")
 * and the third-party routines those call, restated from their observed behaviour
 * (SURVEY.md Appendix A): scipy.ndimage distance_transform_edt / maximum_filter / label
 * and skimage.segmentation.watershed (0.18.3).
 *
 * PINNING: seg_ws_fragments_u8 is pinned bit-for-bit by tests/golden/ws_cases.npz, which
 * tools/gen_goldens_ws.py produced by running the reference post/ws.py itself.
 * seg_agglomerate_mean_u8 restates waterz (ZettaAI/waterz, unpinned git dependency of the
 * reference, pyproject.toml:52-56), whose source is NOT in /root/reference and which is not
 * installed here: **parity unpinned** for that function.  Its queue order is specified
 * here (total order (score, initial edge key)) and the HIP kernels are held to this file.
 *
 * Integer formulation used throughout (exactly equivalent to the float one for uint8
 * affinities): xy foreground mask a_y + a_x >= 256; 3-D mask
 * a_z + a_y + a_x >= 383; distances are kept as exact squared integers d2 (the reference's
 * float64 sqrt and max(D) - D are strictly monotone in d2 for d2 < 2^24).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BIG 0x3fffffff

/* ---- exact squared EDT -------------------------------------------------------------- */
/* 1-D lower envelope pass (Felzenszwalb & Huttenlocher) over the finite samples of
 * f[0..n); BIG marks "no site".  Exact for integer inputs of this size: an intersection
 * abscissa is either an integer (both parabolas tie there) or at least 1/(2n) away from one. */
static void dt1d(const int64_t *f, int n, int64_t *d, int *v, double *z) {
  int k = -1;
  for (int q = 0; q < n; q++) {
    if (f[q] >= BIG) continue;
    if (k < 0) { k = 0; v[0] = q; z[0] = -1e30; z[1] = 1e30; continue; }
    double s;
    for (;;) {
      s = ((double)(f[q] + (int64_t)q * q) - (double)(f[v[k]] + (int64_t)v[k] * v[k])) / (2.0 * (q - v[k]));
      if (s <= z[k]) k--; else break; /* z[0] = -inf keeps k >= 0 */
    }
    k++;
    v[k] = q;
    z[k] = s;
    z[k + 1] = 1e30;
  }
  if (k < 0) { for (int q = 0; q < n; q++) d[q] = BIG; return; }
  k = 0;
  for (int q = 0; q < n; q++) {
    while (z[k + 1] < q) k++;
    const int64_t dq = q - v[k];
    d[q] = dq * dq + f[v[k]];
  }
}

static void edt_axis(int64_t *d2, int n0, int n1, int n2, int axis) {
  /* in-place min over the given axis of d2[..j..] + (i-j)^2 ; array dims (n0,n1,n2) */
  const int dims[3] = {n0, n1, n2};
  const int64_t strides[3] = {(int64_t)n1 * n2, n2, 1};
  const int n = dims[axis];
  int64_t *line = (int64_t *)malloc(sizeof(int64_t) * n);
  int64_t *outl = (int64_t *)malloc(sizeof(int64_t) * n);
  int *v = (int *)malloc(sizeof(int) * (n + 1));
  double *z = (double *)malloc(sizeof(double) * (n + 2));
  const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
  for (int i = 0; i < dims[a1]; i++)
    for (int j = 0; j < dims[a2]; j++) {
      int64_t base = i * strides[a1] + j * strides[a2];
      for (int q = 0; q < n; q++) line[q] = d2[base + q * strides[axis]];
      dt1d(line, n, outl, v, z);
      for (int q = 0; q < n; q++) d2[base + q * strides[axis]] = outl[q];
    }
  free(line); free(outl); free(v); free(z);
}

/* scipy.ndimage.distance_transform_edt(mask) squared.  With no background voxel at all,
 * scipy behaves as if the only background voxel sat at index (-1, 0[, 0]) of the array
 * (measured, scipy 1.7.1 and 1.15.3): d2 = (i0+1)^2 + i1^2 (+ i2^2). */
static void edt_sq(const uint8_t *mask, int n0, int n1, int n2, int32_t *out) {
  const int64_t n = (int64_t)n0 * n1 * n2;
  int any_bg = 0;
  for (int64_t i = 0; i < n; i++) any_bg |= !mask[i];
  if (!any_bg) {
    for (int a = 0; a < n0; a++)
      for (int b = 0; b < n1; b++)
        for (int c = 0; c < n2; c++) {
          /* leading axis of the array scipy sees: for a 2-D slice (n0 == 1) it is axis 1 */
          int64_t d;
          if (n0 == 1) d = (int64_t)(b + 1) * (b + 1) + (int64_t)c * c;
          else d = (int64_t)(a + 1) * (a + 1) + (int64_t)b * b + (int64_t)c * c;
          out[((int64_t)a * n1 + b) * n2 + c] = (int32_t)d;
        }
    return;
  }
  int64_t *d2 = (int64_t *)malloc(sizeof(int64_t) * n);
  for (int64_t i = 0; i < n; i++) d2[i] = mask[i] ? BIG : 0;
  edt_axis(d2, n0, n1, n2, 2);
  edt_axis(d2, n0, n1, n2, 1);
  if (n0 > 1) edt_axis(d2, n0, n1, n2, 0);
  for (int64_t i = 0; i < n; i++) out[i] = (int32_t)d2[i];
  free(d2);
}

/* ---- scipy.ndimage.maximum_filter(size), mode='reflect' ------------------------------- */
static inline int reflect_dup(int i, int n) {
  /* (d c b a | a b c d | d c b a): edge value duplicated, period 2n */
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i < n ? i : p - 1 - i;
}

static void maxfilter_axis(int32_t *a, int n0, int n1, int n2, int axis, int size) {
  const int dims[3] = {n0, n1, n2};
  const int64_t strides[3] = {(int64_t)n1 * n2, n2, 1};
  const int n = dims[axis];
  const int left = size / 2, right = size - 1 - size / 2; /* window [i-left, i+right] */
  int32_t *line = (int32_t *)malloc(sizeof(int32_t) * n);
  const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
  for (int i = 0; i < dims[a1]; i++)
    for (int j = 0; j < dims[a2]; j++) {
      int64_t base = i * strides[a1] + j * strides[a2];
      for (int q = 0; q < n; q++) line[q] = a[base + q * strides[axis]];
      for (int q = 0; q < n; q++) {
        int32_t m = INT32_MIN;
        for (int k = q - left; k <= q + right; k++) {
          const int32_t v = line[reflect_dup(k, n)];
          if (v > m) m = v;
        }
        a[base + q * strides[axis]] = m;
      }
    }
  free(line);
}

/* ---- scipy.ndimage.label (4-/6-connectivity), ids in raster order of first voxel ------- */
static int uf_find(int32_t *p, int x) {
  while (p[x] != x) {
    p[x] = p[p[x]];
    x = p[x];
  }
  return x;
}

static int label_cc(const uint8_t *fg, int n0, int n1, int n2, int32_t *lab) {
  const int64_t n = (int64_t)n0 * n1 * n2;
  int32_t *p = (int32_t *)malloc(sizeof(int32_t) * n);
  for (int64_t i = 0; i < n; i++) p[i] = (int32_t)i;
  for (int a = 0; a < n0; a++)
    for (int b = 0; b < n1; b++)
      for (int c = 0; c < n2; c++) {
        const int64_t i = ((int64_t)a * n1 + b) * n2 + c;
        if (!fg[i]) continue;
        if (c > 0 && fg[i - 1]) { int x = uf_find(p, (int)i), y = uf_find(p, (int)(i - 1)); if (x != y) p[x > y ? x : y] = x > y ? y : x; }
        if (b > 0 && fg[i - n2]) { int x = uf_find(p, (int)i), y = uf_find(p, (int)(i - n2)); if (x != y) p[x > y ? x : y] = x > y ? y : x; }
        if (a > 0 && fg[i - (int64_t)n1 * n2]) { int x = uf_find(p, (int)i), y = uf_find(p, (int)(i - (int64_t)n1 * n2)); if (x != y) p[x > y ? x : y] = x > y ? y : x; }
      }
  /* roots are the minimum raster index of their component: number them in raster order */
  int count = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!fg[i]) { lab[i] = 0; continue; }
    const int r = uf_find(p, (int)i);
    if (r == i) lab[i] = ++count; else lab[i] = lab[r];
  }
  free(p);
  return count;
}

/* ---- skimage.segmentation.watershed(image, markers, mask=mask), connectivity 1 --------- */
/* Sequential priority flood, SURVEY.md Appendix A.1.  `prio` = d2 (larger d2 = smaller
 * image value max(D)-D).  lab: in = markers (already multiplied by mask), out = labels. */
typedef struct { int32_t d2; uint32_t age; int32_t idx; } hitem;

static inline int h_smaller(const hitem *a, const hitem *b) {
  if (a->d2 != b->d2) return a->d2 > b->d2;
  return a->age < b->age;
}

static void flood(const int32_t *d2, const uint8_t *mask, int n0, int n1, int n2, int32_t *lab) {
  const int64_t n = (int64_t)n0 * n1 * n2;
  hitem *heap = (hitem *)malloc(sizeof(hitem) * (n + 1));
  int64_t items = 0;
#define PUSH(it) do { int64_t c_ = items++; heap[c_] = (it); \
    while (c_ > 0) { int64_t p_ = (c_ + 1) / 2 - 1; if (h_smaller(&heap[c_], &heap[p_])) { hitem t_ = heap[c_]; heap[c_] = heap[p_]; heap[p_] = t_; c_ = p_; } else break; } } while (0)
  for (int64_t i = 0; i < n; i++)
    if (lab[i] != 0) { hitem it = {d2[i], 0u, (int32_t)i}; PUSH(it); }
  uint32_t age = 0;
  const int64_t s0 = (int64_t)n1 * n2, s1 = n2;
  while (items > 0) {
    const hitem e = heap[0];
    items--;
    heap[0] = heap[items];
    { /* sift down */
      int64_t i = 0;
      for (;;) {
        int64_t c1 = 2 * i + 1, c2 = c1 + 1, sm = i;
        if (c1 < items && h_smaller(&heap[c1], &heap[sm])) sm = c1;
        if (c2 < items && h_smaller(&heap[c2], &heap[sm])) sm = c2;
        if (sm == i) break;
        hitem t = heap[i]; heap[i] = heap[sm]; heap[sm] = t;
        i = sm;
      }
    }
    const int64_t idx = e.idx;
    const int c = (int)(idx % n2), b = (int)((idx / n2) % n1), a = (int)(idx / s0);
    /* neighbour order [-HW, -W, -1, +1, +W, +HW] (2-D: [-W, -1, +1, +W]) */
    const int64_t nb[6] = {idx - s0, idx - s1, idx - 1, idx + 1, idx + s1, idx + s0};
    const int ok[6] = {a > 0, b > 0, c > 0, c < n2 - 1, b < n1 - 1, a < n0 - 1};
    for (int k = 0; k < 6; k++) {
      if (!ok[k]) continue;
      const int64_t q = nb[k];
      if (!mask[q] || lab[q] != 0) continue;
      age++;
      lab[q] = lab[idx];
      hitem it = {d2[q], age, (int32_t)q};
      PUSH(it);
    }
  }
#undef PUSH
  free(heap);
}

/* post/ws.py:8-35 on one domain (a 2-D slice when n0 == 1, else a 3-D volume).
 * Returns n (number of maxima components, including those outside the mask). */
static int ws_domain(const uint8_t *mask, int n0, int n1, int n2, int msd, uint64_t id_offset,
                     uint64_t *frags, uint64_t *seeds_out) {
  const int64_t n = (int64_t)n0 * n1 * n2;
  int32_t *d2 = (int32_t *)malloc(sizeof(int32_t) * n);
  int32_t *mf = (int32_t *)malloc(sizeof(int32_t) * n);
  uint8_t *mx = (uint8_t *)malloc(n);
  int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * n);
  edt_sq(mask, n0, n1, n2, d2);
  memcpy(mf, d2, sizeof(int32_t) * n);
  maxfilter_axis(mf, n0, n1, n2, 2, msd);
  maxfilter_axis(mf, n0, n1, n2, 1, msd);
  if (n0 > 1) maxfilter_axis(mf, n0, n1, n2, 0, msd);
  for (int64_t i = 0; i < n; i++) mx[i] = mf[i] == d2[i];
  const int cnt = label_cc(mx, n0, n1, n2, lab);
  /* ws.py:21-22: n == 0 -> zeros (cannot happen for a non-empty array: the global maximum
   * always equals its own filter response) */
  if (seeds_out)
    for (int64_t i = 0; i < n; i++) seeds_out[i] = lab[i] ? (uint64_t)lab[i] + id_offset : 0;
  /* markers outside the mask vanish inside skimage (markers * mask) */
  for (int64_t i = 0; i < n; i++) if (!mask[i]) lab[i] = 0;
  flood(d2, mask, n0, n1, n2, lab);
  for (int64_t i = 0; i < n; i++) frags[i] = lab[i] ? (uint64_t)lab[i] + id_offset : 0;
  free(d2); free(mf); free(mx); free(lab);
  return cnt;
}

/* post/ws.py:38-112 for uint8 affinities [3][D][H][W] (max_affinity_value 255). */
int seg_ws_fragments_u8(const uint8_t *affs, int D, int H, int W, int fragments_in_xy,
                        int min_seed_distance, uint64_t *frags, uint64_t *max_id, uint64_t *seeds) {
  const int64_t hw = (int64_t)H * W, n = hw * D;
  const uint8_t *az = affs, *ay = affs + n, *ax = affs + 2 * n;
  uint8_t *mask = (uint8_t *)malloc(n);
  if (fragments_in_xy) {
    /* 0.5*(affs[-1]+affs[-2]) > 0.5  <=>  a_y + a_x >= 256 */
    for (int64_t i = 0; i < n; i++) mask[i] = (int)ay[i] + (int)ax[i] >= 256;
    uint64_t id_offset = 0;
    for (int z = 0; z < D; z++) {
      const int cnt = ws_domain(mask + z * hw, 1, H, W, min_seed_distance, id_offset, frags + z * hw,
                                seeds ? seeds + z * hw : NULL);
      id_offset += (uint64_t)cnt;
    }
    *max_id = id_offset;
  } else {
    /* mean(affs, axis=0) > 0.5  <=>  a_z + a_y + a_x >= 383 */
    for (int64_t i = 0; i < n; i++) mask[i] = (int)az[i] + (int)ay[i] + (int)ax[i] >= 383;
    const int cnt = ws_domain(mask, D, H, W, min_seed_distance, 0, frags, seeds);
    *max_id = (uint64_t)cnt;
  }
  free(mask);
  return 0;
}

/* ---- mean-affinity hierarchical agglomeration (waterz restatement; parity unpinned) ---- */
/* waterz (ZettaAI/waterz, unpinned git dependency of the reference, not in /root/reference) cannot be read or run
 * here.  This restates its published algorithm -- IterativeRegionMerging::mergeUntil / mergeRegions with the
 * MeanAffinity provider and OneMinus -- as far as it is specified; what it leaves to its containers is fixed below
 * and listed in DESIGN.md section 2.  It is also the contract of the HIP kernels.
 *  nodes   = distinct non-zero fragment ids.
 *  edges   = unordered pairs {u < v} of nodes that are 6-adjacent somewhere; each adjacent
 *            voxel pair (p, p - e_d), d in {z,y,x}, contributes affinity affs[d][p]
 *            (channel d at the higher-index voxel) to (sum, count) of its edge.
 *  score   = 1.0f - (float)((double)sum / (255.0 * count))          (OneMinus<MeanAffinity>)
 *  stored  = the score an edge had when it was last (re)scored: its position in the queue.
 *  queue   = min-queue over the total order (stored score, key0) where key0 = (u0 << 32 | v0) packs
 *            the edge's INITIAL endpoints as ranks in the sorted node list (seg_agglomerate_mean_u8), or
 *            discretize_queue = N FIFO bins, bin = (int)(stored * (N - 1)) (seg_rag_merge_scores_u8).
 *  mergeUntil(t), for each threshold t (ascending), continuing from the previous state:
 *    while queue not empty:
 *      e = top; if stored(e) >= t: break            (the STORED score of the popped edge decides)
 *      pop; if e deleted: continue
 *      if e stale: stale = false; rescore (stored = fresh score), reinsert; continue
 *      mergeRegions(e): a = min(u,v), b = max(u,v) of e's current endpoints; b is absorbed by a
 *        every live edge incident to a becomes stale
 *        for every live edge f != e incident to b with other endpoint n:
 *          if no live edge {a, n} exists: f becomes {a, n}; f stale                 (exclusive neighbour)
 *          else g = {a, n}:                                                          (shared neighbour)
 *            if stored(f) > stored(g): g.sum += f.sum; g.count += f.count; delete f; g stale
 *            else:                     f.sum += g.sum; f.count += g.count; delete g; f becomes {a, n}; f stale
 *            (the dearer edge is merged into the cheaper one, which keeps its place in the queue: a stale
 *             edge therefore never waits behind a score larger than its true one)
 *        delete e; parent[b] = a
 *    seg_t[p] = id of root(frag[p])   (0 stays 0)
 */
typedef struct { float score; uint32_t e; } qitem;
typedef struct {
  uint32_t u, v;      /* current endpoints (node ranks) */
  uint64_t key0;      /* initial (u << 32 | v) */
  uint64_t sum;
  uint32_t cnt;
  uint8_t deleted, stale;
  float stored;       /* score at the last (re)scoring */
  uint32_t *hist;     /* histogram-quantile scoring: 256 bins of this edge's affinities (NULL: mean scoring) */
} edge_t;

static __thread edge_t *g_edges; /* per-thread: the cpu_baseline leg runs one volume per core */
static inline int q_less(const qitem *a, const qitem *b) {
  if (a->score != b->score) return a->score < b->score;
  return g_edges[a->e].key0 < g_edges[b->e].key0;
}
/* OneMinus<HistogramQuantileAffinity<RegionGraphType, Q, ScoreValue, 256, InitWithMax>> (reference post/watershed.py:230-243
 * enables Q in {10, 25, 50, 75, 90}, each with InitWithMax false / true), restated from upstream's HistogramQuantileProvider
 * (parity unpinned like the rest of waterz):
 *   bin of an affinity value x in [0, 1]: min(255, (int)(x * 256)); for the uint8 / 255 values of this path that is the
 *     uint8 value itself;
 *   an edge's histogram: every affinity of its voxel pairs (InitWithMax false) or ONE entry, the largest of them
 *     (InitWithMax true); when two edges are merged their histograms are added;
 *   quantile value: pivot = Q * total / 100 + 1 (integer arithmetic, 1-based); the first bin whose running count reaches the
 *     pivot; value = (bin + 0.5) / 256;  score = 1 - value. */
static __thread int g_quantile = -1;  /* < 0: mean scoring */
/* bin rule of the discretized queue: 0 (the specification) bin = (int)(score * (N - 1)); 1: min(N - 1, (int)(score * N)).
 * waterz's own rule is unpinned; the alternative exists so that tools/gen_goldens_waterz.py's vectors can decide. */
static int g_bin_rule = 0;
void seg_set_bin_rule(int rule) { g_bin_rule = rule ? 1 : 0; }
static inline float edge_score(const edge_t *e) {
  if (e->hist) {
    uint64_t total = 0;
    for (int b = 0; b < 256; b++) total += e->hist[b];
    const uint64_t pivot = (uint64_t)g_quantile * total / 100 + 1;
    uint64_t run = 0;
    int bin = 0;
    for (bin = 0; bin < 256; bin++) { run += e->hist[bin]; if (run >= pivot) break; }
    if (bin > 255) bin = 255;
    return 1.0f - ((float)bin + 0.5f) / 256.0f;
  }
  return 1.0f - (float)((double)e->sum / (255.0 * (double)e->cnt));
}

/* open-addressing hash map (u64 key -> u32 value) with tombstones */
typedef struct { uint64_t *keys; uint32_t *vals; uint64_t cap; } hmap;
#define HEMPTY 0xffffffffffffffffull
#define HTOMB 0xfffffffffffffffeull
static inline uint64_t hmix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
static void hm_init(hmap *m, uint64_t n) { m->cap = 64; while (m->cap < 4 * n + 16) m->cap <<= 1; m->keys = (uint64_t *)malloc(8 * m->cap); m->vals = (uint32_t *)malloc(4 * m->cap); memset(m->keys, 0xff, 8 * m->cap); }
static void hm_free(hmap *m) { free(m->keys); free(m->vals); }
static int64_t hm_find(const hmap *m, uint64_t k) { uint64_t i = hmix(k) & (m->cap - 1); for (;;) { if (m->keys[i] == k) return (int64_t)i; if (m->keys[i] == HEMPTY) return -1; i = (i + 1) & (m->cap - 1); } }
static void hm_put(hmap *m, uint64_t k, uint32_t v) { uint64_t i = hmix(k) & (m->cap - 1); while (m->keys[i] != HEMPTY && m->keys[i] != HTOMB && m->keys[i] != k) i = (i + 1) & (m->cap - 1); m->keys[i] = k; m->vals[i] = v; }
static void hm_del(hmap *m, uint64_t k) { int64_t i = hm_find(m, k); if (i >= 0) m->keys[i] = HTOMB; }

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }

/* per-node growable lists of edge ids (entries of deleted or moved-away edges stay and are skipped) */
typedef struct { uint32_t **adj; uint32_t *adjn, *adjc; } adj_t;
static void adj_push(adj_t *A, uint32_t node, uint32_t e) {
  if (A->adjn[node] == A->adjc[node]) {
    A->adjc[node] = A->adjc[node] ? 2 * A->adjc[node] : 4;
    A->adj[node] = (uint32_t *)realloc(A->adj[node], 4 * A->adjc[node]);
  }
  A->adj[node][A->adjn[node]++] = e;
}

/* IterativeRegionMerging::mergeRegions restated (see the specification above).  em: live edges by current key. */
static void merge_regions(edge_t *E, uint32_t ei, hmap *em, adj_t *A, uint32_t *a_out, uint32_t *b_out) {
  edge_t *e = &E[ei];
  const uint32_t a = e->u < e->v ? e->u : e->v, b = e->u < e->v ? e->v : e->u;
  for (uint32_t k = 0; k < A->adjn[a]; k++) {
    edge_t *g = &E[A->adj[a][k]];
    if (!g->deleted && (g->u == a || g->v == a)) g->stale = 1;
  }
  for (uint32_t k = 0; k < A->adjn[b]; k++) {
    const uint32_t fi = A->adj[b][k];
    edge_t *f = &E[fi];
    if (fi == ei || f->deleted) continue;
    if (f->u != b && f->v != b) continue; /* moved away earlier (cannot happen: b only dies once) */
    const uint32_t nb = f->u == b ? f->v : f->u;
    const uint32_t gu = a < nb ? a : nb, gv = a < nb ? nb : a;
    const uint64_t gkey = ((uint64_t)gu << 32) | gv;
    const uint64_t fkey = ((uint64_t)(f->u) << 32) | f->v;
    const int64_t s = hm_find(em, gkey);
    hm_del(em, fkey);
    if (s >= 0 && f->stored > E[em->vals[s]].stored) {
      edge_t *g = &E[em->vals[s]];
      g->sum += f->sum; g->cnt += f->cnt; g->stale = 1;
      if (g->hist) for (int b = 0; b < 256; b++) g->hist[b] += f->hist[b];
      f->deleted = 1;
      continue;
    }
    if (s >= 0) {
      edge_t *g = &E[em->vals[s]];
      f->sum += g->sum; f->cnt += g->cnt;
      if (f->hist) for (int b = 0; b < 256; b++) f->hist[b] += g->hist[b];
      g->deleted = 1;
      hm_del(em, gkey);
    }
    f->u = gu; f->v = gv; f->stale = 1;
    hm_put(em, gkey, fi);
    adj_push(A, a, fi);
  }
  hm_del(em, ((uint64_t)e->u << 32) | e->v);
  e->deleted = 1;
  *a_out = a;
  *b_out = b;
}

/* quantile < 0: OneMinus<MeanAffinity>; else OneMinus<HistogramQuantileAffinity<., quantile, ., 256, init_with_max>> */
int seg_agglomerate_u8(const uint8_t *affs, const uint64_t *frags, int D, int H, int W,
                       const float *thresholds, int nthr, uint64_t *segs, int quantile, int init_with_max) {
  const int64_t hw = (int64_t)H * W, n = hw * D;
  g_quantile = quantile;
  /* nodes: sorted distinct non-zero ids */
  uint64_t *ids = (uint64_t *)malloc(8 * (n + 1));
  int64_t nid = 0;
  for (int64_t i = 0; i < n; i++) if (frags[i]) ids[nid++] = frags[i];
  qsort(ids, nid, 8, cmp_u64);
  int64_t nn = 0;
  for (int64_t i = 0; i < nid; i++) if (i == 0 || ids[i] != ids[i - 1]) ids[nn++] = ids[i];
  /* rank of each voxel */
  uint32_t *rank = (uint32_t *)malloc(4 * n);
  {
    hmap m; hm_init(&m, nn);
    for (int64_t i = 0; i < nn; i++) hm_put(&m, ids[i], (uint32_t)i);
    for (int64_t i = 0; i < n; i++) rank[i] = frags[i] ? m.vals[hm_find(&m, frags[i])] : 0xffffffffu;
    hm_free(&m);
  }
  /* edges */
  int64_t ecap = 1024, ne = 0;
  edge_t *E = (edge_t *)malloc(sizeof(edge_t) * ecap);
  hmap em; hm_init(&em, 1024);
  int64_t em_n = 0;
  const int64_t strides[3] = {hw, W, 1};
  for (int z = 0; z < D; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const int64_t p = (z * (int64_t)H + y) * W + x;
        const uint32_t r1 = rank[p];
        if (r1 == 0xffffffffu) continue;
        const int ok[3] = {z > 0, y > 0, x > 0};
        for (int d = 0; d < 3; d++) {
          if (!ok[d]) continue;
          const uint32_t r2 = rank[p - strides[d]];
          if (r2 == 0xffffffffu || r2 == r1) continue;
          const uint32_t u = r1 < r2 ? r1 : r2, v = r1 < r2 ? r2 : r1;
          const uint64_t key = ((uint64_t)u << 32) | v;
          int64_t s = hm_find(&em, key);
          uint32_t e;
          if (s < 0) {
            if (ne == ecap) { ecap *= 2; E = (edge_t *)realloc(E, sizeof(edge_t) * ecap); }
            if (4 * (em_n + 1) + 16 > (int64_t)em.cap) { /* grow */
              hmap nm; hm_init(&nm, 4 * em_n + 64);
              for (uint64_t i = 0; i < em.cap; i++) if (em.keys[i] != HEMPTY && em.keys[i] != HTOMB) hm_put(&nm, em.keys[i], em.vals[i]);
              hm_free(&em); em = nm;
            }
            e = (uint32_t)ne++;
            E[e].u = u; E[e].v = v; E[e].key0 = key; E[e].sum = 0; E[e].cnt = 0; E[e].deleted = 0; E[e].stale = 0; E[e].stored = 0.f;
            E[e].hist = quantile >= 0 ? (uint32_t *)calloc(256, 4) : NULL;
            hm_put(&em, key, e); em_n++;
          } else e = em.vals[s];
          E[e].sum += affs[(int64_t)d * n + p];
          E[e].cnt += 1;
          if (E[e].hist) E[e].hist[affs[(int64_t)d * n + p]] += 1;
        }
      }
  if (quantile >= 0 && init_with_max)
    for (int64_t e = 0; e < ne; e++) {
      int top = 255;
      while (top > 0 && !E[e].hist[top]) top--;
      memset(E[e].hist, 0, 256 * 4);
      E[e].hist[top] = 1;
    }
  g_edges = E;
  adj_t A;
  A.adj = (uint32_t **)calloc(nn ? nn : 1, sizeof(uint32_t *));
  A.adjn = (uint32_t *)calloc(nn ? nn : 1, 4);
  A.adjc = (uint32_t *)calloc(nn ? nn : 1, 4);
  for (int64_t e = 0; e < ne; e++) { adj_push(&A, E[e].u, (uint32_t)e); adj_push(&A, E[e].v, (uint32_t)e); }
  /* queue: binary heap (any correct priority queue gives the same result: the order is total).  An edge is rescored
   * at most once per pop, so the heap never holds more than ne + (number of pushes in flight) entries: grown on demand. */
  int64_t hcap = ne + 1;
  qitem *heap = (qitem *)malloc(sizeof(qitem) * hcap);
  int64_t hn = 0;
#define QPUSH(it) do { if (hn == hcap) { hcap *= 2; heap = (qitem *)realloc(heap, sizeof(qitem) * hcap); } int64_t c_ = hn++; heap[c_] = (it); while (c_ > 0) { int64_t p_ = (c_ - 1) / 2; if (q_less(&heap[c_], &heap[p_])) { qitem t_ = heap[c_]; heap[c_] = heap[p_]; heap[p_] = t_; c_ = p_; } else break; } } while (0)
  for (int64_t e = 0; e < ne; e++) { E[e].stored = edge_score(&E[e]); qitem it = {E[e].stored, (uint32_t)e}; QPUSH(it); }
  uint32_t *parent = (uint32_t *)malloc(4 * (nn ? nn : 1));
  for (int64_t i = 0; i < nn; i++) parent[i] = (uint32_t)i;

  for (int t = 0; t < nthr; t++) {
    const float thr = thresholds[t];
    while (hn > 0 && heap[0].score < thr) {
      const qitem top = heap[0];
      hn--;
      heap[0] = heap[hn];
      for (int64_t i = 0;;) {
        int64_t c1 = 2 * i + 1, c2 = c1 + 1, sm = i;
        if (c1 < hn && q_less(&heap[c1], &heap[sm])) sm = c1;
        if (c2 < hn && q_less(&heap[c2], &heap[sm])) sm = c2;
        if (sm == i) break;
        qitem tt = heap[i]; heap[i] = heap[sm]; heap[sm] = tt; i = sm;
      }
      edge_t *e = &E[top.e];
      if (e->deleted) continue;
      if (e->stale) { e->stale = 0; e->stored = edge_score(e); qitem it = {e->stored, top.e}; QPUSH(it); continue; }
      uint32_t a, b;
      merge_regions(E, top.e, &em, &A, &a, &b);
      parent[b] = a;
    }
    uint64_t *seg = segs + (int64_t)t * n;
    for (int64_t i = 0; i < n; i++) {
      if (rank[i] == 0xffffffffu) { seg[i] = 0; continue; }
      uint32_t r = rank[i];
      while (parent[r] != r) r = parent[r];
      seg[i] = ids[r];
    }
  }
  for (int64_t i = 0; i < nn; i++) free(A.adj[i]);
  if (quantile >= 0) for (int64_t e = 0; e < ne; e++) free(E[e].hist);
  g_quantile = -1;
  free(A.adj); free(A.adjn); free(A.adjc); free(heap); free(parent); free(E); hm_free(&em); free(rank); free(ids);
  return 0;
}

int seg_agglomerate_mean_u8(const uint8_t *affs, const uint64_t *frags, int D, int H, int W,
                            const float *thresholds, int nthr, uint64_t *segs) {
  return seg_agglomerate_u8(affs, frags, D, H, W, thresholds, nthr, segs, -1, 0);
}

/* number of distinct non-zero labels (test helper) */
int64_t seg_count_labels(const uint64_t *lab, int64_t n) {
  uint64_t *ids = (uint64_t *)malloc(8 * (n + 1));
  int64_t k = 0, c = 0;
  for (int64_t i = 0; i < n; i++) if (lab[i]) ids[k++] = lab[i];
  qsort(ids, k, 8, cmp_u64);
  for (int64_t i = 0; i < k; i++) if (i == 0 || ids[i] != ids[i - 1]) c++;
  free(ids);
  return c;
}

/* ======================================================================================
 * Blockwise fragment post-processing (reference post/blockwise/watershed_frags.py).
 * The reference file cannot be imported here (volara / funlib absent); these functions restate
 * the library calls it makes and are pinned by tests/golden/blockwise_cases.npz, produced by
 * running those calls (scipy.ndimage.mean, skimage.morphology.remove_small_objects,
 * skimage.measure.label; skimage 0.18.3) in tools/gen_goldens_blockwise.py.
 * ====================================================================================== */

/* watershed_frags.py:148-156 filter_avg_fragments on uint8 affinities taken through the
 * float64 path of watershed_in_block (:198-205): average = mean(affs[0:3] / 255, axis=0),
 * means = ndi.mean(average, fragments, ids).  ids must be sorted ascending (np.unique). */
void seg_fragment_means_u8(const uint8_t *affs, const uint64_t *frags, int64_t n, const uint64_t *ids, int64_t m,
                           double *means) {
  double *sum = (double *)calloc(m ? m : 1, sizeof(double));
  int64_t *cnt = (int64_t *)calloc(m ? m : 1, sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) {
    /* binary search of the label */
    int64_t lo = 0, hi = m - 1, k = -1;
    while (lo <= hi) { int64_t mid = (lo + hi) / 2; if (ids[mid] == frags[i]) { k = mid; break; } if (ids[mid] < frags[i]) lo = mid + 1; else hi = mid - 1; }
    if (k < 0) continue;
    const double a0 = (double)affs[i] / 255.0, a1 = (double)affs[n + i] / 255.0, a2 = (double)affs[2 * n + i] / 255.0;
    sum[k] += ((a0 + a1) + a2) / 3.0;
    cnt[k]++;
  }
  for (int64_t k = 0; k < m; k++) means[k] = cnt[k] ? sum[k] / (double)cnt[k] : 0.0;
  free(sum); free(cnt);
}

/* filter (mean < filter_value -> 0; skipped if filter_value <= 0) then remove_small_objects
 * (label count < min_size -> 0; skipped if min_size <= 0), in place.  watershed_frags.py:181-192. */
void seg_filter_fragments_u8(const uint8_t *affs, uint64_t *frags, int64_t n, double filter_value, int64_t min_size) {
  uint64_t *ids = (uint64_t *)malloc(8 * (n + 1));
  int64_t k = 0, m = 0;
  for (int64_t i = 0; i < n; i++) ids[k++] = frags[i];
  qsort(ids, k, 8, cmp_u64);
  for (int64_t i = 0; i < k; i++) if (i == 0 || ids[i] != ids[i - 1]) ids[m++] = ids[i];
  if (filter_value > 0) {
    double *means = (double *)malloc(8 * (m ? m : 1));
    seg_fragment_means_u8(affs, frags, n, ids, m, means);
    for (int64_t i = 0; i < n; i++) {
      int64_t lo = 0, hi = m - 1;
      while (lo <= hi) { int64_t mid = (lo + hi) / 2; if (ids[mid] == frags[i]) { if (means[mid] < filter_value) frags[i] = 0; break; } if (ids[mid] < frags[i]) lo = mid + 1; else hi = mid - 1; }
    }
    free(means);
  }
  if (min_size > 0) {
    int64_t *cnt = (int64_t *)calloc(m ? m : 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) {
      int64_t lo = 0, hi = m - 1;
      while (lo <= hi) { int64_t mid = (lo + hi) / 2; if (ids[mid] == frags[i]) { cnt[mid]++; break; } if (ids[mid] < frags[i]) lo = mid + 1; else hi = mid - 1; }
    }
    for (int64_t i = 0; i < n; i++) {
      if (!frags[i]) continue;
      int64_t lo = 0, hi = m - 1;
      while (lo <= hi) { int64_t mid = (lo + hi) / 2; if (ids[mid] == frags[i]) { if (cnt[mid] < min_size) frags[i] = 0; break; } if (ids[mid] < frags[i]) lo = mid + 1; else hi = mid - 1; }
    }
    free(cnt);
  }
  free(ids);
}

/* skimage.measure.label(x, return_num=True): components of equal non-zero value under full
 * (26-) connectivity, numbered 1.. in raster order of their first voxel.  watershed_frags.py:222. */
int64_t seg_label26(const uint64_t *x, int D, int H, int W, uint32_t *lab) {
  const int64_t n = (int64_t)D * H * W;
  int32_t *p = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
  for (int64_t i = 0; i < n; i++) p[i] = (int32_t)i;
  for (int z = 0; z < D; z++)
    for (int y = 0; y < H; y++)
      for (int xx = 0; xx < W; xx++) {
        const int64_t i = ((int64_t)z * H + y) * W + xx;
        if (!x[i]) continue;
        for (int dz = -1; dz <= 0; dz++)
          for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
              if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0))) continue; /* only preceding neighbours */
              const int zz = z + dz, yy = y + dy, x2 = xx + dx;
              if (zz < 0 || yy < 0 || yy >= H || x2 < 0 || x2 >= W) continue;
              const int64_t j = ((int64_t)zz * H + yy) * W + x2;
              if (x[j] != x[i]) continue;
              int a = uf_find(p, (int)i), b = uf_find(p, (int)j);
              if (a != b) p[a > b ? a : b] = a > b ? b : a;
            }
      }
  int64_t count = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!x[i]) { lab[i] = 0; continue; }
    const int r = uf_find(p, (int)i);
    if (r == i) lab[i] = (uint32_t)(++count); else lab[i] = lab[r];
  }
  free(p);
  return count;
}

/* ======================================================================================
 * Per-block RAG edge scoring (reference post/blockwise/waterz_agglom.py:106-170).
 *
 * The reference relabels the block's fragments densely in ascending id order
 * (funlib.segment.arrays.relabel, :116-120), runs waterz.agglomerate(thresholds=[0, 1.0],
 * discretize_queue=256, return_merge_history, return_region_graph) (:131-139), takes the
 * region graph of the first item as the initial RAG (:141-146), replays the merge history of
 * the second into a MergeTree (:153-160) and stores for every RAG edge the score of the merge
 * in which its two fragments first joined one segment, None if they never do (:162-170).
 * waterz is absent (see the header): **parity unpinned**; the restatement, to which the HIP
 * kernels are held bit for bit, is
 *   nodes   ranks of the distinct non-zero ids in ascending order
 *   edges   as in seg_agglomerate_mean_u8, numbered in ascending (u, v) order
 *   queue   `nbins` FIFO bins, bin = (int)(score * (nbins - 1)) in float32; an edge enters the
 *           queue only while its score < threshold; initial pushes in edge order
 *   loop    take the front of the lowest non-empty bin; deleted -> drop; stale -> rescore and
 *           push to the back of its new bin; else merge exactly as seg_agglomerate_mean_u8 and
 *           record (a, b, score); until the queue is empty
 *   score   of RAG edge {u, v} = score recorded by the merge that first put u and v in one
 *           cluster (the lowest common ancestor of the merge tree, post/merge_tree.py:5-27), NaN if none.
 * Returns the number of edges (edges_out[2e], edges_out[2e+1] = fragment ids, u < v, ascending);
 * merges_out (a = surviving id, b) and mscores_out hold the merge history in order.
 * ====================================================================================== */
typedef struct { uint64_t key; uint8_t aff; } epair;
static int cmp_epair(const void *a, const void *b) { uint64_t x = ((const epair *)a)->key, y = ((const epair *)b)->key; return x < y ? -1 : x > y; }

int64_t seg_rag_merge_scores_u8(const uint8_t *affs, const uint64_t *frags, int D, int H, int W, float threshold, int nbins,
                                uint64_t *edges_out, float *scores_out, int64_t cap, uint64_t *merges_out,
                                float *mscores_out, int64_t *nmerges_out) {
  const int64_t hw = (int64_t)H * W, n = hw * D;
  uint64_t *ids = (uint64_t *)malloc(8 * (n + 1));
  int64_t nid = 0, nn = 0;
  for (int64_t i = 0; i < n; i++) if (frags[i]) ids[nid++] = frags[i];
  qsort(ids, nid, 8, cmp_u64);
  for (int64_t i = 0; i < nid; i++) if (i == 0 || ids[i] != ids[i - 1]) ids[nn++] = ids[i];
  uint32_t *rank = (uint32_t *)malloc(4 * (n ? n : 1));
  {
    hmap m; hm_init(&m, nn);
    for (int64_t i = 0; i < nn; i++) hm_put(&m, ids[i], (uint32_t)i);
    for (int64_t i = 0; i < n; i++) rank[i] = frags[i] ? m.vals[hm_find(&m, frags[i])] : 0xffffffffu;
    hm_free(&m);
  }
  /* voxel-pair contributions, sorted by edge key */
  epair *P = (epair *)malloc(sizeof(epair) * (3 * n + 1));
  int64_t np_ = 0;
  const int64_t strides[3] = {hw, W, 1};
  for (int z = 0; z < D; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const int64_t p = (z * (int64_t)H + y) * W + x;
        const uint32_t r1 = rank[p];
        if (r1 == 0xffffffffu) continue;
        const int ok[3] = {z > 0, y > 0, x > 0};
        for (int d = 0; d < 3; d++) {
          if (!ok[d]) continue;
          const uint32_t r2 = rank[p - strides[d]];
          if (r2 == 0xffffffffu || r2 == r1) continue;
          const uint32_t u = r1 < r2 ? r1 : r2, v = r1 < r2 ? r2 : r1;
          P[np_].key = ((uint64_t)u << 32) | v;
          P[np_++].aff = affs[(int64_t)d * n + p];
        }
      }
  qsort(P, np_, sizeof(epair), cmp_epair);
  int64_t ne = 0;
  for (int64_t i = 0; i < np_; i++) if (i == 0 || P[i].key != P[i - 1].key) ne++;
  edge_t *E = (edge_t *)calloc(ne ? ne : 1, sizeof(edge_t));
  hmap em; hm_init(&em, ne);
  {
    int64_t e = -1;
    for (int64_t i = 0; i < np_; i++) {
      if (i == 0 || P[i].key != P[i - 1].key) {
        e++;
        E[e].u = (uint32_t)(P[i].key >> 32); E[e].v = (uint32_t)P[i].key; E[e].key0 = P[i].key;
        hm_put(&em, P[i].key, (uint32_t)e);
      }
      E[e].sum += P[i].aff; E[e].cnt += 1;
    }
  }
  free(P);
  adj_t A;
  A.adj = (uint32_t **)calloc(nn ? nn : 1, sizeof(uint32_t *));
  A.adjn = (uint32_t *)calloc(nn ? nn : 1, 4);
  A.adjc = (uint32_t *)calloc(nn ? nn : 1, 4);
  for (int64_t e = 0; e < ne; e++) { adj_push(&A, E[e].u, (uint32_t)e); adj_push(&A, E[e].v, (uint32_t)e); }
  /* bin queue: FIFO lists threaded through qnext (an edge is queued at most once) */
  uint32_t *qnext = (uint32_t *)malloc(4 * (ne ? ne : 1));
  uint32_t *bhead = (uint32_t *)malloc(4 * nbins), *btail = (uint32_t *)malloc(4 * nbins);
  for (int b = 0; b < nbins; b++) bhead[b] = btail[b] = 0xffffffffu;
  int minbin = nbins;
#define BPUSH(e_, sc_) do { int b_ = (int)((sc_) * (float)(g_bin_rule ? nbins : nbins - 1)); if (b_ < 0) b_ = 0; if (b_ > nbins - 1) b_ = nbins - 1; \
    qnext[e_] = 0xffffffffu; if (bhead[b_] == 0xffffffffu) bhead[b_] = (e_); else qnext[btail[b_]] = (e_); btail[b_] = (e_); \
    if (b_ < minbin) minbin = b_; } while (0)
  /* an edge whose stored score is not below the threshold can never be popped before the loop ends: it is left out
   * of the queue (upstream queues it and stops at it), its stored score still takes part in the comparisons */
  for (int64_t e = 0; e < ne; e++) { E[e].stored = edge_score(&E[e]); if (E[e].stored < threshold) BPUSH((uint32_t)e, E[e].stored); }
  /* merge tree: leaves 0..nn-1, merge i creates node nn+i */
  uint32_t *tnext = (uint32_t *)malloc(4 * (2 * nn + 1)), *cur = (uint32_t *)malloc(4 * (nn ? nn : 1));
  float *tscore = (float *)calloc(2 * nn + 1, 4);
  for (int64_t i = 0; i < 2 * nn + 1; i++) tnext[i] = 0xffffffffu;
  for (int64_t i = 0; i < nn; i++) cur[i] = (uint32_t)i;
  int64_t nm = 0;
  for (;;) {
    while (minbin < nbins && bhead[minbin] == 0xffffffffu) minbin++;
    if (minbin >= nbins) break;
    const uint32_t ei = bhead[minbin];
    bhead[minbin] = qnext[ei];
    edge_t *e = &E[ei];
    if (e->deleted) continue;
    if (e->stale) { e->stale = 0; e->stored = edge_score(e); if (e->stored < threshold) BPUSH(ei, e->stored); continue; }
    const float sc = e->stored;
    uint32_t a, b;
    merge_regions(E, ei, &em, &A, &a, &b);
    const uint32_t t = (uint32_t)(nn + nm);
    tnext[cur[a]] = t; tnext[cur[b]] = t; cur[a] = t; tscore[t] = sc;
    if (merges_out) { merges_out[2 * nm] = ids[a]; merges_out[2 * nm + 1] = ids[b]; }
    if (mscores_out) mscores_out[nm] = sc;
    nm++;
  }
  if (nmerges_out) *nmerges_out = nm;
  for (int64_t e = 0; e < ne && e < cap; e++) {
    uint32_t x = (uint32_t)(E[e].key0 >> 32), y = (uint32_t)E[e].key0;
    edges_out[2 * e] = ids[x]; edges_out[2 * e + 1] = ids[y];
    float s = NAN;
    for (;;) {
      if (x == y) { s = tscore[x]; break; }
      if (x < y) { if (tnext[x] == 0xffffffffu) break; x = tnext[x]; }
      else { if (tnext[y] == 0xffffffffu) break; y = tnext[y]; }
    }
    scores_out[e] = s;
  }
  for (int64_t i = 0; i < nn; i++) free(A.adj[i]);
  free(A.adj); free(A.adjn); free(A.adjc); free(qnext); free(bhead); free(btail); free(tnext); free(cur); free(tscore);
  free(E); hm_free(&em); free(rank); free(ids);
  return ne;
}

/* funlib.segment.graphs.impl.connected_components(nodes, edges, scores, threshold)
 * (post/watershed.py:182; package absent: **parity unpinned**).  Choice documented in DESIGN.md:
 * an edge joins its endpoints when score <= threshold; every component is named by its
 * smallest node id.  nodes must be ascending; endpoints that are not nodes are ignored. */
void seg_connected_components(const uint64_t *nodes, int64_t n, const uint64_t *edges, const float *scores, int64_t m,
                              float threshold, uint64_t *components) {
  int32_t *p = (int32_t *)malloc(4 * (n ? n : 1));
  for (int64_t i = 0; i < n; i++) p[i] = (int32_t)i;
  for (int64_t e = 0; e < m; e++) {
    if (!(scores[e] <= threshold)) continue;
    int64_t idx[2];
    int ok = 1;
    for (int k = 0; k < 2; k++) {
      int64_t lo = 0, hi = n - 1, f = -1;
      while (lo <= hi) { int64_t mid = (lo + hi) / 2; if (nodes[mid] == edges[2 * e + k]) { f = mid; break; } if (nodes[mid] < edges[2 * e + k]) lo = mid + 1; else hi = mid - 1; }
      if (f < 0) ok = 0;
      idx[k] = f;
    }
    if (!ok) continue;
    int a = uf_find(p, (int)idx[0]), b = uf_find(p, (int)idx[1]);
    if (a == b) continue;
    if (a < b) p[b] = a; else p[a] = b;
  }
  for (int64_t i = 0; i < n; i++) components[i] = nodes[uf_find(p, (int)i)];
  free(p);
}

/* ======================================================================================
 * Thresholded-affinity connected components (reference post/cc.py:7-74, call site
 * post/connected_components.py:77-80).  hard[d][p] = affs[d][p] > cut links voxel p with p + e_d;
 * every voxel that has a link of its own, in raster order, starts a depth-first fill unless already
 * visited; ids 1.. in that order.  Pinned by tests/golden/cc_cases.npz (tools/gen_goldens_cc.py runs the
 * reference function).  The float comparison of the call site (u8 / 255 as float32 > threshold) is an
 * integer cut for uint8 data: seg_cc_cut().
 * ====================================================================================== */
int seg_cc_cut(double threshold) {
  /* largest u8 value v with !((float)v / 255.0f > (float)threshold); -1 if even 0 passes */
  const float t = (float)threshold;
  int cut = -1;
  for (int v = 0; v < 256; v++) if (!(((float)v / 255.0f) > t)) cut = v;
  return cut;
}

int64_t seg_cc_affs_u8(const uint8_t *affs, int D, int H, int W, int cut, uint32_t *seg) {
  const int64_t hw = (int64_t)H * W, n = hw * D;
  uint8_t *visited = (uint8_t *)calloc(n ? n : 1, 1);
  int64_t *stack = (int64_t *)malloc(8 * (n ? n : 1));
  memset(seg, 0, 4 * n);
  uint32_t cur = 1;
#define HARD(d_, p_) ((int)affs[(int64_t)(d_) * n + (p_)] > cut)
  for (int64_t s = 0; s < n; s++) {
    if (visited[s] || !(HARD(0, s) || HARD(1, s) || HARD(2, s))) continue;
    int64_t top = 0;
    stack[top++] = s;
    visited[s] = 1;
    while (top) {
      const int64_t p = stack[--top];
      seg[p] = cur;
      const int z = (int)(p / hw), y = (int)((p / W) % H), x = (int)(p % W);
      if (z + 1 < D && HARD(0, p) && !visited[p + hw]) { stack[top++] = p + hw; visited[p + hw] = 1; }
      if (y + 1 < H && HARD(1, p) && !visited[p + W]) { stack[top++] = p + W; visited[p + W] = 1; }
      if (x + 1 < W && HARD(2, p) && !visited[p + 1]) { stack[top++] = p + 1; visited[p + 1] = 1; }
      if (z - 1 >= 0 && HARD(0, p - hw) && !visited[p - hw]) { stack[top++] = p - hw; visited[p - hw] = 1; }
      if (y - 1 >= 0 && HARD(1, p - W) && !visited[p - W]) { stack[top++] = p - W; visited[p - W] = 1; }
      if (x - 1 >= 0 && HARD(2, p - 1) && !visited[p - 1]) { stack[top++] = p - 1; visited[p - 1] = 1; }
    }
    cur++;
  }
#undef HARD
  free(visited); free(stack);
  return (int64_t)cur - 1;
}
