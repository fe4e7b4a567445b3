// The 3-D seeded watershed flood on the host.
//
// fragments_in_xy = false (reference post/ws.py:98-110) floods a whole block from ONE priority queue; skimage's heap order
// decides ties, so the loop is sequential by definition.  The device replay of it (seg.hip: ws3_flood_kernel) is one wave
// walking that loop at a global-memory round trip per pop: 12.8 s for a 128^3 block, where a host core takes 0.16 s
// (tools/probe_ws3.py).  So the device computes mask, distance transform, maxima and markers (the data-parallel part), this
// file floods, the device continues -- also in the block pipeline, whose 16 lanes made the device loop 0.8 s per block.
// bsmi_seg_set_host_flood(h, 0) selects the device loop (asynchronous; kept and tested).
//
// Algorithm: exactly ws3_flood_kernel's (and oracle/seg_ref.c's) -- entries (MAXD2 - d2) << 46 | age << 23 | voxel, ordered by
// (value, age) only; seeds pushed in raster order with age 0; heappush sifts up while strictly smaller than the parent,
// heappop moves the last entry to the root and sifts it down towards the smaller child; neighbours in the order
// [-HW, -W, -1, +1, +W, +HW], each unlabelled masked neighbour labelled and pushed with the next age.
#include <cstddef>
#include <cstdint>
#include <vector>

namespace bsmi {

void host_flood3(int D, int H, int W, const uint8_t* mask, const int32_t* d2, int32_t* lab) {
  const size_t n = (size_t)D * H * W, hw = (size_t)H * W;
  constexpr uint64_t MAXD2 = (1u << 18) - 1;
  std::vector<uint64_t> heap;
  heap.reserve(n / 4 + 1024);
  auto smaller = [](uint64_t a, uint64_t b) { return (a >> 23) < (b >> 23); };
  auto push = [&](uint64_t it) {
    size_t c = heap.size();
    heap.push_back(it);
    while (c > 0) {
      const size_t p = (c + 1) / 2 - 1;
      const uint64_t pv = heap[p];
      if (smaller(it, pv)) { heap[c] = pv; c = p; } else break;
    }
    heap[c] = it;
  };
  for (size_t j = 0; j < n; ++j)
    if (lab[j] != 0) push(((MAXD2 - (uint64_t)d2[j]) << 46) | (uint64_t)j);
  uint64_t age = 0;
  while (!heap.empty()) {
    const uint64_t e = heap[0];
    const size_t items = heap.size() - 1;
    const size_t idx = (size_t)(e & 0x7fffffu);
    const int32_t l = lab[idx];
    if (items > 0) {
      const uint64_t last = heap[items];
      size_t i = 0;
      for (;;) {
        const size_t c1 = 2 * i + 1, c2 = c1 + 1;
        if (c1 >= items) break;
        size_t sm = i;
        uint64_t smv = last;
        const uint64_t v1 = heap[c1];
        if (smaller(v1, smv)) { sm = c1; smv = v1; }
        if (c2 < items) {
          const uint64_t v2 = heap[c2];
          if (smaller(v2, smv)) { sm = c2; smv = v2; }
        }
        if (sm == i) break;
        heap[i] = smv;
        i = sm;
      }
      heap[i] = last;
    }
    heap.pop_back();
    const int x = (int)(idx % W), y = (int)((idx / W) % H), z = (int)(idx / hw);
    const bool ok[6] = {z > 0, y > 0, x > 0, x < W - 1, y < H - 1, z < D - 1};
    const long long dq[6] = {-(long long)hw, -(long long)W, -1, 1, (long long)W, (long long)hw};
    for (int k = 0; k < 6; ++k) {
      if (!ok[k]) continue;
      const size_t q = (size_t)((long long)idx + dq[k]);
      if (!mask[q] || lab[q] != 0) continue;
      ++age;
      lab[q] = l;
      push(((MAXD2 - (uint64_t)d2[q]) << 46) | (age << 23) | (uint64_t)q);
    }
  }
}

}  // namespace bsmi
