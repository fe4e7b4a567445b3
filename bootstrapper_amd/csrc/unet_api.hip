// Host side of the U-Net engine: network description, weight packing, shape planning and
// the C-ABI entry points declared in include/bsmi.h.
//
// The planner restates the control flow of the reference UNet.rec_forward
// (models/3d_affs/unet.py:440-469) once per input shape and lowers it to a flat list of
// launches: every ConvPass stage (unet.py:7-76) becomes ONE implicit-GEMM launch whose
// K-steps cover the kernel taps of all concatenated inputs plus, for the last stage, the
// cropped 1x1x1 residual branch; max-pool and trilinear-upsample+crop are separate
// memory-bound launches; the crop of the skip connection (unet.py:203-213) is pure index
// arithmetic folded into the K-step offsets.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include <chrono>
#include <thread>

#include "unet_internal.h"
#include "unet_ops.h"

#include "dev_guard.h"  // last: routes hipMalloc / hipFree through the guarded allocator (BSMI_GUARD_MB)

namespace bsmi {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

}  // namespace bsmi

using namespace bsmi;

namespace bsmi {

int esize(int prec) { return prec == BSMI_PREC_F32 ? 4 : 2; }
// Which form of the split mode a layer takes: the fused kernel (conv_x3_body) or three listed K-steps through the bf16
// kernels.  BSMI_X3_FUSED (dev / tests): 0 = listed form everywhere, 1 (default) / 2 = fused wherever the 8-wave kernels
// run, 3 = fused except on 256 x 320 tiles (the two forms side by side in one forward).
static int x3_fused_mode() {
  static const int m = [] { const char* e = getenv("BSMI_X3_FUSED"); return e ? atoi(e) : 1; }();
  return m;
}
bool x3_fused_for(TileCfg tile) {
  if (!two_waves_per_simd()) return false;  // BSMI_WAVES8=0 (tests): the fused wide kernels exist as 8-wave kernels only
  return x3_fused_mode() == 1 || x3_fused_mode() == 2 || (x3_fused_mode() == 3 && tile != TILE_256x320);
}
int ksplit(int prec) { return prec == BSMI_PREC_BF16X3 ? 3 : 1; }  // listed form; PackedConv::ks holds the layer's own value
static inline float host_bf16_to_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
int bke(int prec) { return kStepRowBytes / esize(prec); }
int sube(int prec) { return 32 / esize(prec); }

static void expect_weight(bsmi_unet* h, const std::string& key, std::vector<int64_t> shape) {
  HostWeight hw;
  hw.shape = std::move(shape);
  h->weights[key] = std::move(hw);
}

static void register_pass(bsmi_unet* h, const PassSite& p) {
  int cin_total = p.cin[0] + (p.nslots > 1 ? p.cin[1] : 0);
  int cin = cin_total;
  for (int i = 0; i < p.nconv; ++i) {
    const std::string base = p.prefix + ".conv_pass." + std::to_string(2 * i);
    expect_weight(h, base + ".weight", {p.cout, cin, p.k[i][0], p.k[i][1], p.k[i][2]});
    expect_weight(h, base + ".bias", {p.cout});
    cin = p.cout;
  }
  expect_weight(h, p.prefix + ".residual.0.weight", {p.cout, cin_total, 1, 1, 1});
  expect_weight(h, p.prefix + ".residual.0.bias", {p.cout});
}

// Build the unit list of stage `ci` of a ConvPass.  kUnitsPerStep consecutive entries = one
// K-step.  Order: for every source slot, 32-channel chunk major with the kernel taps inside (one
// tap x 32 channels per K-step, i.e. 64 contiguous bytes per gathered row; a 16-channel tensor
// packs two taps per K-step), then, for the last stage, the cropped 1x1x1 residual in 32-channel
// chunks.  Each (slot, 32-channel chunk) is also one PackPhase:
// LONG = all kernel taps of the chunk, SHORT = its residual tap.
void build_entries(const PassSite& p, int ci, int prec, std::vector<PackEntry>& out, std::vector<PackPhase>* phases_out) {
  const int SUB = sube(prec);
  const bool last = ci == p.nconv - 1;
  std::vector<PackPhase> phases;
  auto close_phase = [&](PackPhase ph) {
    while ((out.size() - ph.first_unit) % kUnitsPerStep) out.push_back(PackEntry{ph.slot, 0, 0, 0, 0, 0, 0, 0, 0, true, (int)phases.size()});
    ph.nunits = (int)out.size() - ph.first_unit;
    phases.push_back(ph);
  };
  auto add_taps = [&](int slot, const int* k, int cin_base, int creal) {
    const int cpad = round_up(creal, kChanPad);
    if (cpad == SUB) {
      // a voxel is one 32-byte unit: a K-step takes two x-adjacent taps (dx, dx+1), which are 64
      // contiguous bytes of the channels-last tensor (conv_rh.hip reads them as one halo row)
      PackPhase ph{slot, 0, 0, (int)out.size(), 0};
      for (int z = 0; z < k[0]; ++z)
        for (int y = 0; y < k[1]; ++y)
          for (int x = 0; x < k[2]; x += kUnitsPerStep)
            for (int j = 0; j < kUnitsPerStep; ++j) {
              if (x + j < k[2])
                out.push_back(PackEntry{slot, z, y, x + j, 0, 0, (z * k[1] + y) * k[2] + x + j, cin_base, creal, false, (int)phases.size()});
              else
                out.push_back(PackEntry{slot, 0, 0, 0, 0, 0, 0, 0, 0, true, (int)phases.size()});
            }
      close_phase(ph);
      return;
    }
    for (int c32 = 0; c32 < cpad; c32 += kUnitsPerStep * SUB) {
      PackPhase ph{slot, c32, 0, (int)out.size(), 0};
      for (int z = 0; z < k[0]; ++z)
        for (int y = 0; y < k[1]; ++y)
          for (int x = 0; x < k[2]; ++x) {
            for (int c0 = c32; c0 < std::min(cpad, c32 + kUnitsPerStep * SUB); c0 += SUB)
              out.push_back(PackEntry{slot, z, y, x, c0, 0, (z * k[1] + y) * k[2] + x, cin_base, creal, false, (int)phases.size()});
            while ((out.size() - ph.first_unit) % kUnitsPerStep)  // half chunk: the K-step's second unit is padding
              out.push_back(PackEntry{slot, 0, 0, 0, 0, 0, 0, 0, 0, true, (int)phases.size()});
          }
      close_phase(ph);
    }
  };
  auto add_residual = [&](int slot, const int* crop, int cin_base, int creal) {
    const int cpad = round_up(creal, kChanPad);
    for (int c32 = 0; c32 < cpad; c32 += kUnitsPerStep * SUB) {
      PackPhase ph{slot, c32, 1, (int)out.size(), 0};
      for (int c0 = c32; c0 < std::min(cpad, c32 + kUnitsPerStep * SUB); c0 += SUB)
        out.push_back(PackEntry{slot, crop[0] / 2, crop[1] / 2, crop[2] / 2, c0, 1, 0, cin_base, creal, false, (int)phases.size()});
      close_phase(ph);
    }
  };
  const int* k = p.k[ci];
  if (ci == 0) {
    int base = 0;
    for (int s = 0; s < p.nslots; ++s) {
      add_taps(s, k, base, p.cin[s]);
      base += p.cin[s];
    }
  } else {
    add_taps(0, k, 0, p.cout);
  }
  if (last) {
    int crop[3] = {0, 0, 0};
    for (int i = 0; i < p.nconv; ++i)
      for (int d = 0; d < 3; ++d) crop[d] += p.k[i][d] - 1;
    const int first_slot = ci == 0 ? 0 : 1;
    int base = 0;
    for (int s = 0; s < p.nslots; ++s) {
      add_residual(first_slot + s, crop, base, p.cin[s]);
      base += p.cin[s];
    }
  }
  // an even number of K-steps: the 16x16x32 kernel body is unrolled by two
  if ((out.size() / kUnitsPerStep) % 2)
    for (int j = 0; j < kUnitsPerStep; ++j) out.push_back(PackEntry{0, 0, 0, 0, 0, 0, 0, 0, 0, true, (int)phases.size() - 1});
  if (phases_out) *phases_out = phases;
}

static int pack_conv(bsmi_unet* h, PassSite& p, int ci, int prec) {
  PackedConv& pc = p.packed[prec][ci];
  if (pc.ready) return BSMI_OK;
  pc.entries.clear();
  build_entries(p, ci, prec, pc.entries, &pc.phases);
  pc.tile = choose_tile(p.cout);
  pc.Npad = round_up(p.cout, tile_bn(pc.tile));
  const int BKE = bke(prec);
  const bool last = ci == p.nconv - 1;
  const HostWeight& wm = h->weights[p.prefix + ".conv_pass." + std::to_string(2 * ci) + ".weight"];
  const HostWeight& bm = h->weights[p.prefix + ".conv_pass." + std::to_string(2 * ci) + ".bias"];
  const HostWeight& wr = h->weights[p.prefix + ".residual.0.weight"];
  const HostWeight& br = h->weights[p.prefix + ".residual.0.bias"];
  const int64_t cin_m = wm.shape[1], ntap_m = wm.shape[2] * wm.shape[3] * wm.shape[4];
  const int64_t cin_r = wr.shape[1];
  const int SUB = sube(prec);
  // Split mode: logical K-step s becomes the kernel's K-steps 3s (hi activations x hi weights), 3s + 1 (lo
  // activations x hi weights) and 3s + 2 (hi activations x lo weights); the weight image carries a row block for each
  const bool fused = prec == BSMI_PREC_BF16X3 && x3_fused_for(pc.tile);
  const int KS = pc.ks = fused ? 1 : ksplit(prec);
  const size_t nsteps = pc.entries.size() / kUnitsPerStep * KS;
  const size_t nelem = (nsteps * (size_t)pc.Npad + kWeightRowSlack) * BKE;  // slack rows: padded tile loads
  std::vector<float> bias(pc.Npad, 0.f);
  for (int n = 0; n < p.cout; ++n) bias[n] = bm.data[n] + (last ? br.data[n] : 0.f);

  pc.lo_image_bytes = fused ? nelem * esize(prec) : 0;
  std::vector<uint8_t> packed(nelem * esize(prec) * (fused ? 2 : 1), 0);
  // one output channel per task: its weights (cin x taps floats, contiguous) are read once and stay in the core's cache, its
  // rows of the image(s) are whole 64-byte lines of its own (unit-major, as this loop ran before, every element was a cache
  // miss 160 KB from the last: loading the 95 M parameters of the 3d_affs net took 1.34 s of `bs predict`'s start, 0.71 s now;
  // tools/probe_load.py)
  host_parallel_for((size_t)p.cout, [&](size_t n_) {
    const int n = (int)n_;
    for (size_t u = 0; u < pc.entries.size(); ++u) {
      const PackEntry& e = pc.entries[u];
      if (e.dummy) continue;
      const size_t s = u / kUnitsPerStep, j = u % kUnitsPerStep;
      for (int kk = 0; kk < SUB; ++kk) {
        const int c = e.c0 + kk;
        if (c >= e.creal) break;
        float v;
        if (e.wsrc == 0) v = wm.data[((size_t)n * cin_m + (e.cin_base + c)) * ntap_m + e.tap];
        else v = wr.data[(size_t)n * cin_r + (e.cin_base + c)];
        const size_t idx = (s * KS * pc.Npad + n) * BKE + j * SUB + kk;
        if (prec == BSMI_PREC_F32) {
          ((float*)packed.data())[idx] = v;
        } else if (prec == BSMI_PREC_BF16) {
          ((uint16_t*)packed.data())[idx] = host_f32_to_bf16(v);
        } else {
          const uint16_t hi = host_f32_to_bf16(v);
          const uint16_t lo = host_f32_to_bf16(v - host_bf16_to_f32(hi));
          uint16_t* w16 = (uint16_t*)packed.data();
          if (fused) {  // hi image, then the lo image of the same layout
            w16[idx] = hi;
            w16[idx + nelem] = lo;
          } else {
            const size_t blk = (size_t)pc.Npad * BKE;
            w16[idx] = hi;
            w16[idx + blk] = hi;
            w16[idx + 2 * blk] = lo;
          }
        }
      }
    }
  });
  BSMI_HIP(hipMalloc(&pc.w, packed.size()));
  BSMI_HIP(hipMemcpy(pc.w, packed.data(), packed.size(), hipMemcpyHostToDevice));
  BSMI_HIP(hipMalloc((void**)&pc.bias, bias.size() * sizeof(float)));
  BSMI_HIP(hipMemcpy(pc.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  pc.ready = true;
  return BSMI_OK;
}

// Which 3x3x3 stages of the split-bf16 mode run in their Winograd form (wino.hip).  BSMI_WINO: 0 = none, unset / 1 = where it
// was measured to win on the 128^3 block -- the stages with at least 256 channels on both sides, whose matrix work outweighs
// the transform passes over the layer's tensors (ms direct -> Winograd: 1500 -> 1500 12.19 -> 7.07, 1800 -> 300 10.55 -> 7.63,
// 300 -> 1500 3.37 -> 2.27, 300 -> 300 3.22 -> 2.90 and 2.23 -> 2.03; narrower stages lose: 360 -> 60 2.48 -> 3.44,
// 60 -> 300 0.91 -> 1.20) --, 2 = every stage the kernels can take (tests: small nets).
static int wino_mode() {
  static const int m = [] { const char* e = getenv("BSMI_WINO"); return e ? atoi(e) : 1; }();
  return m;
}
static int wino4_mode();
static bool wino_eligible(const PassSite& p, int ci) {
  if (wino_mode() == 0 || p.k[ci][0] != 3 || p.k[ci][1] != 3 || p.k[ci][2] != 3) return false;
  if (ci == 0 && p.nslots > kWinoMaxSrc) return false;
  if (!x3_fused_for(choose_tile(p.cout))) return false;
  if (wino_mode() >= 2) return true;
  int cin = p.cout;
  if (ci == 0) {
    cin = 0;
    for (int s = 0; s < p.nslots; ++s) cin += p.cin[s];
  }
  if (cin >= 256 && p.cout >= 256) return true;
  // with F(4x4) tiles (V 2.25x the input instead of 4x, 2.25 sums per output instead of 4) two narrower stages win as well,
  // measured on the 128^3 block: 60 -> 300 0.88 -> 0.80 ms, 360 -> 60 2.09 -> 1.94 (60 -> 60 0.87 -> 1.23 and the 12-channel
  // stages lose)
  return wino4_mode() == 1 && (int64_t)cin * p.cout >= 16384 && cin >= 32 && p.cout >= 32;
}

// Which Winograd stages use the F(4x4, 3x3) tile (round 4): 36 products per 16 outputs instead of 16 per 4.  BSMI_WINO4: 0 =
// none (F(2x2) as in round 3), unset / 1 = every Winograd stage, 2 = the same (kept for the tests' spelling).  Measured on the
// 128^3 block (ms F(2x2) -> F(4x4)): 1500 -> 1500 6.99 -> 4.27, 1800 -> 300 7.07 -> 4.67, 300 -> 300 2.90 -> 1.82 and 1.90 ->
// 1.38, 300 -> 1500 2.22 -> 1.50; block 28.3 -> 20.5 ms.  Error of the whole net against the CPU fp32 oracle 8.6e-6 -> 9.0e-6
// (gate 1e-4): the interpolation points 0, +-1/sqrt2, +-sqrt2, inf keep a layer's error at four times F(2x2)'s where the
// textbook points 0, +-1, +-2 make it fifteen times (wino.hip), and a layer's error reaches the sigmoid outputs attenuated.
static int wino4_mode() {
  static const int m = [] { const char* e = getenv("BSMI_WINO4"); return e ? atoi(e) : 1; }();
  return m;
}
static int wino_tile_edge(const PassSite& p, int ci) {
  (void)p; (void)ci;
  return wino4_mode() == 0 ? 2 : 4;
}

static int pack_wino(bsmi_unet* h, PassSite& p, int ci) {
  PackedWino& pw = p.wino[ci];
  if (pw.w) { (void)hipFree(pw.w); pw.w = nullptr; }
  if (pw.res_w) { (void)hipFree(pw.res_w); pw.res_w = nullptr; }
  for (int part = 0; part < 2; ++part)
    if (pw.res_part_w[part]) { (void)hipFree(pw.res_part_w[part]); pw.res_part_w[part] = nullptr; }
  pw.ready = false;
  if (!wino_eligible(p, ci)) return BSMI_OK;
  const HostWeight& wm = h->weights[p.prefix + ".conv_pass." + std::to_string(2 * ci) + ".weight"];
  const HostWeight& wr = h->weights[p.prefix + ".residual.0.weight"];
  const int cin_m = (int)wm.shape[1];
  // channels of V: every source tensor keeps its own padding to kChanPad
  pw.cin_of_v.clear();
  if (ci == 0) {
    int base = 0;
    for (int s = 0; s < p.nslots; ++s) {
      for (int c = 0; c < round_up(p.cin[s], kChanPad); ++c) pw.cin_of_v.push_back(c < p.cin[s] ? base + c : -1);
      base += p.cin[s];
    }
  } else {
    for (int c = 0; c < round_up(p.cout, kChanPad); ++c) pw.cin_of_v.push_back(c < p.cout ? c : -1);
  }
  pw.Cv = (int)pw.cin_of_v.size();
  pw.m = wino_tile_edge(p, ci);
  pw.tile = choose_tile(p.cout);
  pw.Npad = round_up(p.cout, tile_bn(pw.tile));
  wino_units(pw.Cv, pw.units);
  {
    size_t image_elems = 0, batch_elems = 0;
    static const bool on_host = [] { const char* e = getenv("BSMI_WINO_PACK_HOST"); return e && e[0] == '1'; }();
    if (on_host) {   // (the host form of the transform: kept as the cross-check of the device one, tests/test_fullsize_gpu.py)
      std::vector<uint16_t> packed;
      wino_pack_weights(wm.data.data(), p.cout, cin_m, pw.cin_of_v, pw.Npad, pw.units, packed, image_elems, batch_elems, pw.m);
      BSMI_HIP(hipMalloc(&pw.w, packed.size() * 2));
      BSMI_HIP(hipMemcpy(pw.w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    } else {
      const int rc = wino_pack_weights_dev(wm.data.data(), p.cout, cin_m, pw.cin_of_v, pw.Npad, pw.units, pw.m, &pw.w, image_elems, batch_elems);
      if (rc) return rc;
    }
    pw.lo_image_bytes = image_elems * 2;
    pw.batch_bytes = batch_elems * 2;
  }
  // the cropped 1x1x1 residual branch of the last stage: its K-steps alone
  pw.res_entries.clear();
  if (ci == p.nconv - 1) {
    std::vector<PackEntry> all;
    build_entries(p, ci, BSMI_PREC_BF16X3, all);
    for (size_t u = 0; u + 1 < all.size(); u += kUnitsPerStep) {
      bool res = false;
      for (int j = 0; j < kUnitsPerStep; ++j) res |= !all[u + j].dummy && all[u + j].wsrc == 1;
      if (res)
        for (int j = 0; j < kUnitsPerStep; ++j) pw.res_entries.push_back(all[u + j]);
    }
    if ((pw.res_entries.size() / kUnitsPerStep) % 2)
      for (int j = 0; j < kUnitsPerStep; ++j) pw.res_entries.push_back(PackEntry{0, 0, 0, 0, 0, 0, 0, 0, 0, true, 0});
    const int64_t cin_r = wr.shape[1];
    auto pack_res = [&](const std::vector<PackEntry>& ents, void** w_out, size_t* lo_out) -> int {
      const size_t nsteps = ents.size() / kUnitsPerStep;
      const size_t nelem = (nsteps * (size_t)pw.Npad + kWeightRowSlack) * 32;
      std::vector<uint16_t> packed(2 * nelem, 0);
      for (size_t u = 0; u < ents.size(); ++u) {
        const PackEntry& e = ents[u];
        if (e.dummy) continue;
        const size_t s = u / kUnitsPerStep, j = u % kUnitsPerStep;
        for (int n = 0; n < p.cout; ++n)
          for (int kk = 0; kk < 16; ++kk) {
            const int c = e.c0 + kk;
            if (c >= e.creal) break;
            const float v = wr.data[(size_t)n * cin_r + (e.cin_base + c)];
            const size_t idx = (s * pw.Npad + n) * 32 + j * 16 + kk;
            const uint16_t hi = host_f32_to_bf16(v);
            packed[idx] = hi;
            packed[nelem + idx] = host_f32_to_bf16(v - host_bf16_to_f32(hi));
          }
      }
      *lo_out = nelem * 2;
      BSMI_HIP(hipMalloc(w_out, packed.size() * 2));
      BSMI_HIP(hipMemcpy(*w_out, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
      return BSMI_OK;
    };
    int rc = pack_res(pw.res_entries, &pw.res_w, &pw.res_lo_image_bytes);
    if (rc) return rc;
    for (int part = 0; part < 2; ++part) pw.res_part[part].clear();
    if (p.nslots == 2) {  // the branch cut by source: the skip connection, the upsampled map (the last slot of the launch)
      const int up_slot = (ci == 0 ? 0 : 1) + 1;
      for (size_t u = 0; u + 1 < pw.res_entries.size(); u += kUnitsPerStep) {
        int slot = -1;
        for (int j = 0; j < kUnitsPerStep; ++j)
          if (!pw.res_entries[u + j].dummy) slot = pw.res_entries[u + j].slot;
        if (slot < 0) continue;  // the padding K-step of the whole list
        for (int j = 0; j < kUnitsPerStep; ++j) pw.res_part[slot == up_slot ? 1 : 0].push_back(pw.res_entries[u + j]);
      }
      for (int part = 0; part < 2; ++part) {
        if ((pw.res_part[part].size() / kUnitsPerStep) % 2)
          for (int j = 0; j < kUnitsPerStep; ++j) pw.res_part[part].push_back(PackEntry{0, 0, 0, 0, 0, 0, 0, 0, 0, true, 0});
        if ((rc = pack_res(pw.res_part[part], &pw.res_part_w[part], &pw.res_part_lo[part]))) return rc;
      }
    }
  }
  pw.ready = true;
  return BSMI_OK;
}

// Which stages run in the halo-resident form (conv_h16.hip).  BSMI_H16: 0 = none, unset / 1 = the stages with at most 64 output
// channels whose launch is large enough to fill the chip, 2 = every stage the kernel can take (tests: small nets).
static int h16_mode() {
  static const int m = [] { const char* e = getenv("BSMI_H16"); return e ? atoi(e) : 1; }();
  return m;
}

// unit list of the halo-resident form: per source slot, 16-channel chunk and z tap a phase; its in-plane taps two per K-step
static void build_entries_h16(const PassSite& p, int ci, std::vector<PackEntry>& out, std::vector<H16PhaseHost>& phases) {
  const bool last = ci == p.nconv - 1;
  const int* k = p.k[ci];
  const PackEntry dummy{0, 0, 0, 0, 0, 0, 0, 0, 0, true, 0};
  auto add_taps = [&](int slot, int cin_base, int creal) {
    const int cpad = round_up(creal, kChanPad);
    // z tap outermost: two consecutive 16-channel chunks of a voxel share a 128-byte line, so every other phase finds its
    // rows in L2
    for (int z = 0; z < k[0]; ++z)
      for (int c0 = 0; c0 < cpad; c0 += 16) {
        H16PhaseHost ph{slot, c0, z, 0, (int)(out.size() / kUnitsPerStep), 0};
        const int ntap = k[1] * k[2];
        for (int i = 0; i < ntap; i += 2) {
          for (int j = 0; j < 2; ++j) {
            if (i + j < ntap) {
              const int y = (i + j) / k[2], x = (i + j) % k[2];
              out.push_back(PackEntry{slot, z, y, x, c0, 0, (z * k[1] + y) * k[2] + x, cin_base, creal, false, (int)phases.size()});
            } else {
              out.push_back(dummy);
            }
          }
          ++ph.nsteps;
        }
        phases.push_back(ph);
      }
  };
  if (ci == 0) {
    int base = 0;
    for (int s = 0; s < p.nslots; ++s) {
      add_taps(s, base, p.cin[s]);
      base += p.cin[s];
    }
  } else {
    add_taps(0, 0, p.cout);
  }
  if (last) {
    int crop[3] = {0, 0, 0};
    for (int i = 0; i < p.nconv; ++i)
      for (int d = 0; d < 3; ++d) crop[d] += p.k[i][d] - 1;
    const int first_slot = ci == 0 ? 0 : 1;
    int base = 0;
    for (int s = 0; s < p.nslots; ++s) {
      const int cpad = round_up(p.cin[s], kChanPad);
      for (int c0 = 0; c0 < cpad; c0 += 16) {
        phases.push_back(H16PhaseHost{first_slot + s, c0, crop[0] / 2, 1, (int)(out.size() / kUnitsPerStep), 1});
        out.push_back(PackEntry{first_slot + s, crop[0] / 2, crop[1] / 2, crop[2] / 2, c0, 1, 0, base, p.cin[s], false, (int)phases.size() - 1});
        out.push_back(dummy);
      }
      base += p.cin[s];
    }
  }
}

static int pack_h16(bsmi_unet* h, PassSite& p, int ci) {
  PackedH16& ph = p.h16[ci];
  if (ph.w) { (void)hipFree(ph.w); ph.w = nullptr; }
  ph.ready = false;
  if (h16_mode() == 0 || p.cout > 64) return BSMI_OK;
  ph.entries.clear();
  ph.phases.clear();
  build_entries_h16(p, ci, ph.entries, ph.phases);
  ph.Npad = p.cout <= 16 ? 16 : 64;
  const HostWeight& wm = h->weights[p.prefix + ".conv_pass." + std::to_string(2 * ci) + ".weight"];
  const HostWeight& wr = h->weights[p.prefix + ".residual.0.weight"];
  const int64_t cin_m = wm.shape[1], ntap_m = wm.shape[2] * wm.shape[3] * wm.shape[4], cin_r = wr.shape[1];
  const size_t nsteps = ph.entries.size() / kUnitsPerStep;
  const size_t nelem = (nsteps * (size_t)ph.Npad + kWeightRowSlack) * 32;
  std::vector<uint16_t> packed(2 * nelem, 0);
  host_parallel_for((size_t)p.cout, [&](size_t n_) {   // one output channel per task (pack_conv)
    const int n = (int)n_;
    for (size_t u = 0; u < ph.entries.size(); ++u) {
      const PackEntry& e = ph.entries[u];
      if (e.dummy) continue;
      const size_t s = u / kUnitsPerStep, j = u % kUnitsPerStep;
      for (int kk = 0; kk < 16; ++kk) {
        const int c = e.c0 + kk;
        if (c >= e.creal) break;
        const float v = e.wsrc == 0 ? wm.data[((size_t)n * cin_m + (e.cin_base + c)) * ntap_m + e.tap] : wr.data[(size_t)n * cin_r + (e.cin_base + c)];
        const size_t idx = (s * ph.Npad + n) * 32 + j * 16 + kk;
        const uint16_t hi = host_f32_to_bf16(v);
        packed[idx] = hi;
        packed[nelem + idx] = host_f32_to_bf16(v - host_bf16_to_f32(hi));
      }
    }
  });
  ph.lo_image_bytes = nelem * 2;
  BSMI_HIP(hipMalloc(&ph.w, packed.size() * 2));
  BSMI_HIP(hipMemcpy(ph.w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
  ph.ready = true;
  return BSMI_OK;
}

// An upsampled map whose only readers are Winograd stages of the ConvPass that follows it (its first stage's taps, its
// last stage's residual branch): the stages read the LOW-resolution tensor and interpolate on the fly, the UP step is skipped.
struct UpFuse {
  TDesc low;    // the tensor below the upsampling
  int f[3];     // factors (1, f, f)
  int o[3];     // crop offset of the upsampled map inside the full upsampling (Upsample.crop_to_factor, unet.py:177-183)
};

struct Planner {
  bsmi_unet* h;
  int prec;
  Plan* plan;
  bool dry;  // shape / flop arithmetic only: no allocation, no device traffic
  const UpFuse* fuse_up = nullptr;  // set while the ConvPass behind a fused upsampling is planned

  int alloc(TDesc& t) {
    t.Cpad = round_up(t.C, kChanPad);
    const size_t plane = (size_t)t.D * t.H * t.W * t.Cpad * esize(prec);
    const bool split = prec == BSMI_PREC_BF16X3;
    // split mode: each 16-byte vector of 8 hi values is followed by the 16 bytes of their lo values (conv_dev.h act_index)
    t.lo_off = split ? 16 : 0;
    const size_t bytes = split ? 2 * plane : plane;
    plan->bytes += bytes;
    if (bytes >= ((size_t)1 << 31))
      BSMI_FAIL(BSMI_ERR_INVALID, "activation tensor of %zu bytes exceeds the 31-bit byte offsets of the conv kernel", bytes);
    if (dry) return BSMI_OK;
    // slack: the raster-halo kernel stages whole rows of the input raster, and the rows it drops
    // (xx >= Wo, yy >= Ho) may lie a few lines past the end of a cropped source tensor
    const size_t slack = (size_t)8 * t.W * t.Cpad * esize(prec) * (split ? 2 : 1) + 4096;
    BSMI_HIP(hipMalloc(&t.ptr, bytes + slack));
    BSMI_HIP(hipMemsetAsync((char*)t.ptr + bytes, 0, slack, nullptr));
    plan->allocs.push_back(t.ptr);
    return BSMI_OK;
  }

  // Raster-halo launch of one ConvPass stage (conv_rh.hip) when the tile's halo buffer holds
  // 256 + (ky-1)*Win + (kx-1) rows; otherwise st.use_rh stays false (gather kernel, conv_igemm.hip).
  int plan_rh(const PassSite& p, int ci, const PackedConv& pc, const TDesc* slots, const int (*so)[3], int nsl,
              const TDesc& o, PlanStep& st) {
    st.use_rh = false;
    // BSMI_USE_RH: 1 = wherever the halo fits, 0 = never, unset = only where it was measured to win on the
    // 128^3 block: small-Cout tiles with a long K loop (the 360->60 channel layer: 1.43 -> 1.04 ms); the rows the
    // raster-halo form computes and drops cost the small-plane layers 8-17 %, and short-K layers do not amortise it
    static const int mode = [] { const char* e = getenv("BSMI_USE_RH"); return !e ? 2 : (e[0] == '1' ? 1 : 0); }();
    // split mode: never (a form that listed the K-steps per plane through the bf16 kernel lost to the fused gather kernel on every
    // layer -- 360 -> 60 channels: 3.16 against 2.41 ms -- and was removed in round 3, as was its fused variant; the halo form of the split mode is plan_h16 below)
    const bool enabled = prec != BSMI_PREC_BF16X3 && (mode == 1 || (mode == 2 && tile_bn(st.tile) <= 64 && pc.entries.size() / kUnitsPerStep >= 200));
    const int* k = p.k[ci];
    const int Hin = o.H + k[1] - 1, Win = o.W + k[2] - 1;
    if (!enabled || !two_waves_per_simd() || !rh_supported(st.tile, Win, k[1], k[2])) return BSMI_OK;
    const int64_t es = esize(prec);  // bytes per channel of a row
    const int SUB = sube(prec);
    const size_t nsteps = pc.entries.size() / kUnitsPerStep;
    std::vector<RhStep> steps(nsteps);
    std::vector<RhPhase> phases;
    std::vector<int> first_step;  // per phase
    // phase key of the previous K-step
    int pslot = -1, pc32 = -1, pz = -1, pkind = -1;
    for (size_t s = 0; s < nsteps; ++s) {
      const PackEntry& e0 = pc.entries[kUnitsPerStep * s];
      const PackEntry& e1 = pc.entries[kUnitsPerStep * s + 1];
      if (e0.dummy) {  // the all-zero K-step that makes the count even: any resident halo row will do
        if (phases.empty() || !e1.dummy) BSMI_FAIL(BSMI_ERR_STATE, "raster-halo plan: K-step %zu starts with a padding unit", s);
        steps[s] = RhStep{0, (int32_t)(((phases.size() - 1) & 1) | ((phases.size() - 1) << 8)), 0, -1};
        continue;
      }
      const TDesc& t = slots[e0.slot];
      const int c32 = e0.c0 / (kUnitsPerStep * SUB) * (kUnitsPerStep * SUB);
      const int kind = e0.wsrc;
      if (!e1.dummy) {
        const bool same_tap = e1.dz == e0.dz && e1.dy == e0.dy && e1.dx == e0.dx && e1.c0 == e0.c0 + SUB && e1.slot == e0.slot;
        const bool x_pair = t.Cpad == SUB && e1.slot == e0.slot && e1.dz == e0.dz && e1.dy == e0.dy && e1.dx == e0.dx + 1 &&
                            e1.c0 == e0.c0 && kind == 0 && e1.wsrc == 0;
        if (!same_tap && !x_pair) return BSMI_OK;  // unit order the halo rows cannot serve: gather kernel
      }
      if (e0.slot != pslot || c32 != pc32 || e0.dz != pz || kind != pkind) {
        RhPhase ph;
        ph.tensor = e0.slot;
        const int oz = so[e0.slot][0] + e0.dz;
        const int oy = so[e0.slot][1] + (kind ? e0.dy : 0);
        const int ox = so[e0.slot][2] + (kind ? e0.dx : 0);
        ph.delta = (int32_t)(((((int64_t)oz * t.H + oy) * t.W + ox) * t.Cpad + c32) * es);
        ph.buf = (int32_t)(phases.size() & 1);
        ph.issue_step = -1;
        phases.push_back(ph);
        first_step.push_back((int)s);
        pslot = e0.slot; pc32 = c32; pz = e0.dz; pkind = kind;
      }
      RhStep& r = steps[s];
      r.rowoff = kind ? 0 : e0.dy * Win + e0.dx;
      r.buf_phase = (int32_t)(((phases.size() - 1) & 1) | ((phases.size() - 1) << 8));
      r.wait = 0;
      r.issue = -1;
    }
    const size_t nlisted = steps.size();
    const int np = (int)phases.size();
    // the halo of phase p >= 2 is staged with the weights of K-step issue_step + 4, as soon as
    // phase p-2 (same buffer) has been read; phases 0 and 1 in the prologue
    for (int q = 2; q < np; ++q) {
      const int is = first_step[q - 1] - 1;
      phases[q].issue_step = is;
      if (steps[is].issue >= 0) BSMI_FAIL(BSMI_ERR_STATE, "raster-halo plan: two halos on one staging slot");
      steps[is].issue = q;
    }
    // in-order queue of one wave: H0 [H1] W0 W1 W2 W3 {[H(issue[h])] W(h+4)}...
    std::vector<int> posW(nlisted + 8, 0), posH(np, 0);
    std::vector<char> isH;  // per queue position
    auto push = [&](bool halo) { isH.push_back(halo ? 1 : 0); return (int)isH.size() - 1; };
    posH[0] = push(true);
    if (np > 1) posH[1] = push(true);
    for (int w = 0; w < 4; ++w) posW[w] = push(false);
    for (size_t hh = 0; hh < nlisted; ++hh) {
      if (steps[hh].issue >= 0) posH[steps[hh].issue] = push(true);
      posW[hh + 4] = push(false);
    }
    for (size_t hh = 0; hh < nlisted; ++hh) {
      const int upto = posW[hh + 3];
      int need = posW[hh + 1];
      if (hh + 1 < nlisted) need = std::max(need, posH[steps[hh + 1].buf_phase >> 8]);
      if (need > upto) BSMI_FAIL(BSMI_ERR_STATE, "raster-halo plan: halo of K-step %zu staged too late", hh + 1);
      int aw = 0, bh = 0;
      for (int q = need + 1; q <= upto; ++q) (isH[q] ? bh : aw)++;
      if (aw > 2 || bh > 2) BSMI_FAIL(BSMI_ERR_STATE, "raster-halo plan: wait code out of range");
      steps[hh].wait = aw * 3 + bh;
    }
    RhArgs& a = st.rh;
    memset(&a, 0, sizeof a);
    for (int sl = 0; sl < kMaxConvTensors; ++sl) a.t[sl] = st.conv.t[sl];
    RhStep* dsteps = nullptr;
    RhPhase* dphases = nullptr;
    BSMI_HIP(hipMalloc((void**)&dsteps, steps.size() * sizeof(RhStep)));
    plan->allocs.push_back(dsteps);
    BSMI_HIP(hipMalloc((void**)&dphases, phases.size() * sizeof(RhPhase)));
    plan->allocs.push_back(dphases);
    BSMI_HIP(hipMemcpy(dsteps, steps.data(), steps.size() * sizeof(RhStep), hipMemcpyHostToDevice));
    BSMI_HIP(hipMemcpy(dphases, phases.data(), phases.size() * sizeof(RhPhase), hipMemcpyHostToDevice));
    a.steps = dsteps;
    a.phases = dphases;
    a.nsteps = (int)nlisted;
    a.nphases = np;
    a.w = pc.w;
    a.bias = pc.bias;
    a.out = o.ptr;
    a.Do = o.D; a.Ho = o.H; a.Wo = o.W; a.Co = o.Cpad;
    a.Hin = Hin; a.Win = Win;
    a.Q = o.D * Hin * Win;
    a.Npad = pc.Npad;
    a.relu = 1;
    st.use_rh = true;
    (void)nsl;
    return BSMI_OK;
  }

  // Halo-resident launch of a stage with at most 64 output channels (conv_h16.hip), split-bf16 mode.
  int plan_h16(PassSite& p, int ci, const PackedConv& pc, const TDesc* slots, const int (*so)[3], const TDesc& o, PlanStep& st) {
    st.use_h16 = false;
    const PackedH16& ph = p.h16[ci];
    const int* k = p.k[ci];
    const int Hin = o.H + k[1] - 1, Win = o.W + k[2] - 1;
    if (prec != BSMI_PREC_BF16X3 || !ph.ready || h16_mode() == 0 || pc.bias == nullptr) return BSMI_OK;
    static const int only_npad = [] { const char* e = getenv("BSMI_H16_NPAD"); return e ? atoi(e) : 0; }();  // dev: one column count only
    if (only_npad && only_npad != ph.Npad) return BSMI_OK;
    int max_r = 0;
    const int rows = h16_halo_rows(Win, k[1], k[2], ph.Npad, &max_r);
    if (!rows || Hin > 2047 || Win > 2047 || o.D > 1023) return BSMI_OK;
    const int64_t Q = (int64_t)o.D * Hin * Win;
    if (Q >= ((int64_t)1 << 31)) return BSMI_OK;
    // a launch that cannot fill the chip (two workgroups of 512 rows per CU) stays with the gather kernel's 256-row tiles
    if (h16_mode() == 1 && Q < (int64_t)2 * 256 * kH16TileRows) return BSMI_OK;
    // A residual branch costs this form one phase -- a halo load -- per 16 channels for half a K-step's worth of multiplies.
    // The second stage of unet.r_conv.0.1 (60 -> 60 channels, residual from 360): 23 such phases beside 12 of nine taps,
    // 0.60 ms against 0.56 for the gather kernel.
    if (h16_mode() == 1 && ph.Npad == 64) {
      int main_steps = 0, res_steps = 0;
      for (const H16PhaseHost& hp : ph.phases) (hp.kind ? res_steps : main_steps) += hp.nsteps;
      if (3 * res_steps > main_steps) return BSMI_OK;
    }
    const int64_t es = 4;  // bytes per channel of a row: (hi, lo) interleaved
    std::vector<H16Phase> phases(ph.phases.size());
    std::vector<H16Step> steps(ph.entries.size() / kUnitsPerStep);
    for (size_t q = 0; q < ph.phases.size(); ++q) {
      const H16PhaseHost& hp = ph.phases[q];
      const TDesc& t = slots[hp.slot];
      const PackEntry& e0 = ph.entries[(size_t)kUnitsPerStep * hp.first_step];
      // kind 0: the taps' (dy, dx) are row offsets into the halo; kind 1 (the residual's crop centre): folded into the origin
      const int oz = so[hp.slot][0] + hp.dz;
      const int oy = so[hp.slot][1] + (hp.kind ? e0.dy : 0);
      const int ox = so[hp.slot][2] + (hp.kind ? e0.dx : 0);
      const int64_t delta = ((((int64_t)oz * t.H + oy) * t.W + ox) * t.Cpad + hp.c0) * es;
      if (delta >= ((int64_t)1 << 31)) return BSMI_OK;
      phases[q] = H16Phase{hp.slot, (int32_t)delta, hp.kind ? kH16TileRows : kH16TileRows + (k[1] - 1) * Win + (k[2] - 1), hp.nsteps};
      for (int s = hp.first_step; s < hp.first_step + hp.nsteps; ++s) {
        const PackEntry& ea = ph.entries[(size_t)kUnitsPerStep * s];
        const PackEntry& eb = ph.entries[(size_t)kUnitsPerStep * s + 1];
        const int offa = hp.kind ? 0 : ea.dy * Win + ea.dx;
        const int offb = (hp.kind || eb.dummy) ? offa : eb.dy * Win + eb.dx;
        steps[s] = H16Step{offa, offb, 0, 0};
      }
    }
    H16Args& a = st.h16;
    memset(&a, 0, sizeof a);
    for (int sl = 0; sl < kMaxConvTensors; ++sl) a.t[sl] = st.conv.t[sl];
    H16Phase* dphases = nullptr;
    H16Step* dsteps = nullptr;
    BSMI_HIP(hipMalloc((void**)&dphases, phases.size() * sizeof(H16Phase)));
    plan->allocs.push_back(dphases);
    BSMI_HIP(hipMalloc((void**)&dsteps, steps.size() * sizeof(H16Step)));
    plan->allocs.push_back(dsteps);
    BSMI_HIP(hipMemcpy(dphases, phases.data(), phases.size() * sizeof(H16Phase), hipMemcpyHostToDevice));
    BSMI_HIP(hipMemcpy(dsteps, steps.data(), steps.size() * sizeof(H16Step), hipMemcpyHostToDevice));
    a.phases = dphases;
    a.steps = dsteps;
    a.nphases = (int)phases.size();
    a.nsteps = (int)steps.size();
    a.w = ph.w;
    a.w_lo = (const char*)ph.w + ph.lo_image_bytes;
    a.bias = pc.bias;  // (pc.Npad >= ph.Npad rows)
    a.out = o.ptr;
    a.Do = o.D; a.Ho = o.H; a.Wo = o.W; a.Co = o.Cpad;
    a.Hin = Hin; a.Win = Win;
    a.Q = (int)Q;
    a.Npad = ph.Npad;
    a.relu = 1;
    {
      static const bool balanced = [] { const char* e = getenv("BSMI_H16_BALANCE"); return !(e && e[0] == '0'); }();
      int n_cus = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
      h16_tiling(Q, ph.Npad, rows, max_r, balanced ? n_cus : 0, &a.ntiles, &a.n_big, &a.r_small);
    }
    st.h16_rows = rows;
    st.use_h16 = true;
    return BSMI_OK;
  }

  // Box-halo launch of one ConvPass stage (conv_box.hip): bf16, 3x3x3, at most 16 output channels.
  // BSMI_USE_BOX: 0 = never, unset / 1 = wherever it applies.
  int plan_box(const PassSite& p, int ci, const TDesc* slots, const int (*so)[3], int nsl, const TDesc& o, PlanStep& st) {
    st.use_box = false;
    static const bool enabled = [] { const char* e = getenv("BSMI_USE_BOX"); return !(e && e[0] == '0'); }();
    const int* k = p.k[ci];
    if (!enabled || prec != BSMI_PREC_BF16 || k[0] != 3 || k[1] != 3 || k[2] != 3 || !box_supported(p.cout)) return BSMI_OK;
    const bool last = ci == p.nconv - 1;
    const int NB = 1;  // one block of 16 output channels
    const HostWeight& wm = h->weights[p.prefix + ".conv_pass." + std::to_string(2 * ci) + ".weight"];
    const HostWeight& bm = h->weights[p.prefix + ".conv_pass." + std::to_string(2 * ci) + ".bias"];
    const HostWeight& wr = h->weights[p.prefix + ".residual.0.weight"];
    const HostWeight& br = h->weights[p.prefix + ".residual.0.bias"];
    const int64_t cin_m = wm.shape[1], cin_r = wr.shape[1];
    int crop[3] = {0, 0, 0};
    for (int i = 0; i < p.nconv; ++i)
      for (int d = 0; d < 3; ++d) crop[d] += p.k[i][d] - 1;

    struct Src { int slot, cin_base, creal; };
    std::vector<Src> full, center;
    if (ci == 0) {
      int base = 0;
      for (int s = 0; s < p.nslots; ++s) { full.push_back({s, base, p.cin[s]}); base += p.cin[s]; }
    } else {
      full.push_back({0, 0, p.cout});
    }
    if (last) {
      const int first_slot = ci == 0 ? 0 : 1;
      int base = 0;
      for (int s = 0; s < p.nslots; ++s) { center.push_back({first_slot + s, base, p.cin[s]}); base += p.cin[s]; }
    }
    std::vector<BoxChunk> chunks;
    std::vector<uint32_t> wimg;
    auto chunk_of = [&](const TDesc& t, const int* org, int c0) {
      BoxChunk ck;
      ck.sx = t.Cpad * 2; ck.sy = t.W * ck.sx; ck.sz = t.H * ck.sy;
      ck.base = (uint64_t)(uintptr_t)t.ptr + (uint64_t)org[0] * ck.sz + (uint64_t)org[1] * ck.sy + (uint64_t)org[2] * ck.sx + (uint64_t)c0 * 2;
      ck.D = t.D - org[0]; ck.H = t.H - org[1]; ck.W = t.W - org[2];
      return ck;
    };
    auto put = [&](size_t lane_base, int j, float v) { wimg[lane_base + j / 2] |= (uint32_t)host_f32_to_bf16(v) << (16 * (j & 1)); };
    // FULL chunks: 16 channels x 27 taps, 14 K-steps of (2 taps x 16 channels)
    for (const Src& sr : full) {
      const TDesc& t = slots[sr.slot];
      for (int c0 = 0; c0 < t.Cpad; c0 += 16) {
        chunks.push_back(chunk_of(t, so[sr.slot], c0));
        const size_t w0 = wimg.size();
        wimg.resize(w0 + (size_t)14 * NB * 64 * 4, 0u);
        for (int s = 0; s < 14; ++s)
          for (int b = 0; b < NB; ++b)
            for (int l = 0; l < 64; ++l) {
              const int m = b * 16 + (l & 15), q = l >> 4, tap = 2 * s + (q >> 1);
              if (m >= p.cout || tap >= 27) continue;
              for (int j = 0; j < 8; ++j) {
                const int c = c0 + (q & 1) * 8 + j;
                if (c >= sr.creal) break;
                put(w0 + (((size_t)s * NB + b) * 64 + l) * 4, j, wm.data[((size_t)m * cin_m + (sr.cin_base + c)) * 27 + tap]);
              }
            }
      }
    }
    const int n_full = (int)chunks.size();
    // CENTER chunks: 32 channels of the box's own voxels in the residual source (cropped by half the pass's crop)
    for (const Src& sr : center) {
      const TDesc& t = slots[sr.slot];
      const int org[3] = {so[sr.slot][0] + crop[0] / 2, so[sr.slot][1] + crop[1] / 2, so[sr.slot][2] + crop[2] / 2};
      for (int c0 = 0; c0 < t.Cpad; c0 += 32) {
        chunks.push_back(chunk_of(t, org, c0));
        const size_t w0 = wimg.size();
        wimg.resize(w0 + (size_t)NB * 64 * 4, 0u);
        for (int b = 0; b < NB; ++b)
          for (int l = 0; l < 64; ++l) {
            const int m = b * 16 + (l & 15), q = l >> 4;
            if (m >= p.cout) continue;
            for (int j = 0; j < 8; ++j) {
              const int c = c0 + q * 8 + j;
              if (c >= sr.creal) break;
              put(w0 + ((size_t)b * 64 + l) * 4, j, wr.data[(size_t)m * cin_r + (sr.cin_base + c)]);
            }
          }
      }
    }
    std::vector<float> bias((size_t)16 * NB, 0.f);
    for (int m = 0; m < p.cout; ++m) bias[m] = bm.data[m] + (last ? br.data[m] : 0.f);
    BoxChunk* dchunks = nullptr;
    uint32_t* dw = nullptr;
    float* dbias = nullptr;
    BSMI_HIP(hipMalloc((void**)&dchunks, chunks.size() * sizeof(BoxChunk)));
    plan->allocs.push_back(dchunks);
    BSMI_HIP(hipMalloc((void**)&dw, wimg.size() * 4));
    plan->allocs.push_back(dw);
    BSMI_HIP(hipMalloc((void**)&dbias, bias.size() * 4));
    plan->allocs.push_back(dbias);
    BSMI_HIP(hipMemcpy(dchunks, chunks.data(), chunks.size() * sizeof(BoxChunk), hipMemcpyHostToDevice));
    BSMI_HIP(hipMemcpy(dw, wimg.data(), wimg.size() * 4, hipMemcpyHostToDevice));
    BSMI_HIP(hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    st.box.chunks = dchunks;
    st.box.n_full = n_full;
    st.box.n_center = (int)chunks.size() - n_full;
    st.box.w = dw;
    st.box.bias = dbias;
    st.box.out = (uint16_t*)o.ptr;
    st.box.Do = o.D; st.box.Ho = o.H; st.box.Wo = o.W; st.box.Co = o.Cpad;
    st.use_box = true;
    (void)nsl;
    return BSMI_OK;
  }

  // Winograd form of one 3x3x3 stage (wino.hip): the layer's sources are transformed into V (16 batches), one batched launch
  // of the fused split-bf16 kernel multiplies them with the transformed weights into raw f32 sums, the residual branch of a
  // last stage runs as a short launch of its own, and the output transform finishes (bias, residual, ReLU, (hi, lo) pairs).
  int plan_wino(PassSite& p, int ci, const PackedConv& pc, const TDesc* slots, const int (*so)[3], int nsl, const TDesc& o, PlanStep& st) {
    st.use_wino = false;
    const PackedWino& pw = p.wino[ci];
    const bool wants_fused = fuse_up && (ci == 0 || ci == p.nconv - 1);
    const int wm = pw.m, nbatch = wino_batches(wm);
    if (prec != BSMI_PREC_BF16X3 || !pw.ready || (wm == 2 && ((o.H & 1) || (o.W & 1))) || st.use_box || st.use_rh) {
      if (wants_fused) BSMI_FAIL(BSMI_ERR_STATE, "%s conv %d: the upsampling was fused into this stage, which cannot take the Winograd form", p.prefix.c_str(), ci);
      return BSMI_OK;
    }
    const int nsrc = ci == 0 ? p.nslots : 1;
    const int Dv = o.D + 2, Ty = (o.H + wm - 1) / wm, Tx = (o.W + wm - 1) / wm, Cv = pw.Cv;   // F(4x4): the last tiles may overhang
    const size_t vbatch = (size_t)Dv * Ty * Tx * Cv * 4;  // bytes of one batch of V: (hi, lo) pairs
    if (vbatch >= ((size_t)1 << 31)) {                     // 31-bit byte offsets inside a batch
      if (wants_fused) BSMI_FAIL(BSMI_ERR_STATE, "%s conv %d: fused upsampling, but the transformed input is too large", p.prefix.c_str(), ci);
      return BSMI_OK;
    }
    const size_t Mrows = (size_t)o.D * Ty * Tx;
    void *V = nullptr, *Mbuf = nullptr, *addend = nullptr;
    BSMI_HIP(hipMalloc(&V, nbatch * vbatch + 4096));
    plan->allocs.push_back(V);
    BSMI_HIP(hipMalloc(&Mbuf, nbatch * Mrows * o.Cpad * sizeof(float)));
    plan->allocs.push_back(Mbuf);
    plan->bytes += nbatch * (vbatch + Mrows * o.Cpad * sizeof(float));
    WinoInArgs& wi = st.wino_in;
    memset(&wi, 0, sizeof wi);
    int cv0 = 0;
    for (int q = 0; q < nsrc; ++q) {
      const TDesc& t = slots[q];
      wi.src[q] = t.ptr;
      wi.H[q] = t.H; wi.W[q] = t.W; wi.Cpad[q] = t.Cpad;
      wi.oz[q] = so[q][0]; wi.oy[q] = so[q][1]; wi.ox[q] = so[q][2];
      wi.cv0[q] = cv0;
      cv0 += t.Cpad;
      if (fuse_up && ci == 0 && q == 1) {  // the upsampled map: read below the upsampling
        if (fuse_up->low.Cpad != t.Cpad) BSMI_FAIL(BSMI_ERR_STATE, "%s: fused upsampling with a different channel padding", p.prefix.c_str());
        wi.src[q] = fuse_up->low.ptr;
        wi.H[q] = fuse_up->low.H; wi.W[q] = fuse_up->low.W;
        wi.oz[q] += fuse_up->o[0]; wi.oy[q] += fuse_up->o[1]; wi.ox[q] += fuse_up->o[2];
        wi.upf[q] = fuse_up->f[1];
      }
    }
    if (cv0 != Cv) BSMI_FAIL(BSMI_ERR_STATE, "%s conv %d: winograd channel layout %d != %d", p.prefix.c_str(), ci, cv0, Cv);
    wi.nsrc = nsrc;
    wi.V = V;
    wi.Dv = Dv; wi.Ty = Ty; wi.Tx = Tx; wi.Cv = Cv;
    wi.m = wm;
    // the batched GEMMs: a (3,1,1) convolution of V[b] with U[b]
    ConvArgs& g = st.wino_gemm;
    memset(&g, 0, sizeof g);
    for (int sl = 0; sl < kMaxConvTensors; ++sl) {
      g.t[sl].base = (uint64_t)(uintptr_t)V;
      g.t[sl].sz = (int32_t)((int64_t)Ty * Tx * Cv * 4);
      g.t[sl].sy = (int32_t)((int64_t)Tx * Cv * 4);
      g.t[sl].sx = (int32_t)((int64_t)Cv * 4);
    }
    {
      std::vector<KStep> ks(pw.units.size() / kUnitsPerStep);
      for (size_t s = 0; s < ks.size(); ++s) {
        KStep k;
        memset(&k, 0, sizeof k);
        for (int j = 0; j < kUnitsPerStep; ++j) {
          const WinoUnit& u = pw.units[kUnitsPerStep * s + j];
          if (!u.dummy) k.delta[j] = (int32_t)(((int64_t)u.kz * Ty * Tx * Cv + u.vc0) * 4);
        }
        ks[s] = k;
      }
      KStep* dks = nullptr;
      BSMI_HIP(hipMalloc((void**)&dks, ks.size() * sizeof(KStep)));
      plan->allocs.push_back(dks);
      BSMI_HIP(hipMemcpy(dks, ks.data(), ks.size() * sizeof(KStep), hipMemcpyHostToDevice));
      g.steps = dks;
      g.nsteps = (int)ks.size();
    }
    g.w = pw.w;
    g.w_lo = (const char*)pw.w + pw.lo_image_bytes;
    g.bias = pc.bias;
    g.out = Mbuf;
    g.Do = o.D; g.Ho = Ty; g.Wo = Tx; g.Co = o.Cpad;
    g.M = (int)Mrows;
    g.Npad = pw.Npad;
    g.relu = 0;
    g.nbatch = nbatch;
    g.raw = 1;
    g.a_batch = (int64_t)vbatch;
    g.w_batch = (int64_t)pw.batch_bytes;
    // residual branch
    const bool last = ci == p.nconv - 1;
    const bool cut = fuse_up && last;  // the branch in two parts: the skip connection here, the upsampled map below the upsampling
    if (cut && (pw.res_part[0].empty() || pw.res_part[1].empty())) BSMI_FAIL(BSMI_ERR_STATE, "%s: fused upsampling without a cut residual branch", p.prefix.c_str());
    const std::vector<PackEntry>& res_entries = cut ? pw.res_part[0] : pw.res_entries;
    st.wino_has_res = !res_entries.empty();
    st.wino_has_res_low = false;
    void* low = nullptr;
    int low_off[3] = {0, 0, 0};
    if (cut) {
      const TDesc& g = fuse_up->low;
      const size_t Mlow = (size_t)g.D * g.H * g.W;
      BSMI_HIP(hipMalloc(&low, Mlow * o.Cpad * sizeof(float)));
      plan->allocs.push_back(low);
      plan->bytes += Mlow * o.Cpad * sizeof(float);
      ConvArgs& r = st.wino_res_low;
      memset(&r, 0, sizeof r);
      for (int sl = 0; sl < kMaxConvTensors; ++sl) {
        r.t[sl].base = (uint64_t)(uintptr_t)g.ptr;
        r.t[sl].sz = (int32_t)((int64_t)g.H * g.W * g.Cpad * 4);
        r.t[sl].sy = (int32_t)((int64_t)g.W * g.Cpad * 4);
        r.t[sl].sx = (int32_t)((int64_t)g.Cpad * 4);
      }
      const std::vector<PackEntry>& ents = pw.res_part[1];
      std::vector<KStep> ks(ents.size() / kUnitsPerStep);
      int up_slot = -1;
      for (size_t s = 0; s < ks.size(); ++s) {
        KStep k;
        memset(&k, 0, sizeof k);
        for (int j = 0; j < kUnitsPerStep; ++j) {
          const PackEntry& e = ents[kUnitsPerStep * s + j];
          if (e.dummy) continue;
          k.delta[j] = (int32_t)((int64_t)e.c0 * 4);  // a 1x1x1 convolution of the low-resolution tensor itself: no spatial offset here
          up_slot = e.slot;
          low_off[0] = e.dz; low_off[1] = e.dy; low_off[2] = e.dx;  // the branch's crop: applied by the output transform
        }
        ks[s] = k;
      }
      if (up_slot < 0) BSMI_FAIL(BSMI_ERR_STATE, "%s: empty low-resolution residual", p.prefix.c_str());
      for (int d = 0; d < 3; ++d) low_off[d] += so[up_slot][d] + fuse_up->o[d];
      KStep* dks = nullptr;
      BSMI_HIP(hipMalloc((void**)&dks, ks.size() * sizeof(KStep)));
      plan->allocs.push_back(dks);
      BSMI_HIP(hipMemcpy(dks, ks.data(), ks.size() * sizeof(KStep), hipMemcpyHostToDevice));
      r.steps = dks;
      r.nsteps = (int)ks.size();
      r.w = pw.res_part_w[1];
      r.w_lo = (const char*)pw.res_part_w[1] + pw.res_part_lo[1];
      r.bias = pc.bias;
      r.out = low;
      r.Do = g.D; r.Ho = g.H; r.Wo = g.W; r.Co = o.Cpad;
      r.M = (int)Mlow;
      r.Npad = pw.Npad;
      r.relu = 0;
      r.raw = 1;
      st.wino_has_res_low = true;
    }
    if (st.wino_has_res) {
      BSMI_HIP(hipMalloc(&addend, (size_t)o.D * o.H * o.W * o.Cpad * sizeof(float)));
      plan->allocs.push_back(addend);
      plan->bytes += (size_t)o.D * o.H * o.W * o.Cpad * sizeof(float);
      ConvArgs& r = st.wino_res;
      r = st.conv;  // the direct launch's source tensors and output geometry
      std::vector<KStep> ks(res_entries.size() / kUnitsPerStep);
      for (size_t s = 0; s < ks.size(); ++s) {
        KStep k;
        memset(&k, 0, sizeof k);
        const PackEntry& e0 = res_entries[kUnitsPerStep * s];
        k.tensor = e0.dummy ? 0 : e0.slot;
        for (int j = 0; j < kUnitsPerStep; ++j) {
          const PackEntry& e = res_entries[kUnitsPerStep * s + j];
          if (e.dummy) continue;
          const TDesc& t = slots[e.slot];
          const int64_t off = ((((int64_t)(e.dz + so[e.slot][0]) * t.H) + (e.dy + so[e.slot][1])) * t.W + (e.dx + so[e.slot][2])) * t.Cpad + e.c0;
          k.delta[j] = (int32_t)(off * 4);
        }
        ks[s] = k;
      }
      KStep* dks = nullptr;
      BSMI_HIP(hipMalloc((void**)&dks, ks.size() * sizeof(KStep)));
      plan->allocs.push_back(dks);
      BSMI_HIP(hipMemcpy(dks, ks.data(), ks.size() * sizeof(KStep), hipMemcpyHostToDevice));
      r.steps = dks;
      r.nsteps = (int)ks.size();
      r.w = cut ? pw.res_part_w[0] : pw.res_w;
      r.w_lo = (const char*)r.w + (cut ? pw.res_part_lo[0] : pw.res_lo_image_bytes);
      r.out = addend;
      r.Npad = pw.Npad;
      r.relu = 0;
      r.raw = 1;
      r.nbatch = 0;
    }
    WinoOutArgs& wo = st.wino_out;
    memset(&wo, 0, sizeof wo);
    wo.M = (const float*)Mbuf;
    wo.addend = (const float*)addend;
    if (cut) {
      wo.low = (const float*)low;
      wo.lD = fuse_up->low.D; wo.lH = fuse_up->low.H; wo.lW = fuse_up->low.W;
      wo.lf = fuse_up->f[1];
      wo.loz = low_off[0]; wo.loy = low_off[1]; wo.lox = low_off[2];
    }
    wo.bias = pc.bias;
    wo.out = o.ptr;
    wo.Do = o.D; wo.Ty = Ty; wo.Tx = Tx; wo.Co = o.Cpad;
    wo.m = wm; wo.Ho = o.H; wo.Wo = o.W;
    wo.relu = 1;
    {
      const double kstep = (double)kUnitsPerStep * sube(prec);
      st.exec_flops = 2.0 * (double)round_up((int)Mrows, 256) * pw.Npad * ((double)g.nsteps * kstep) * nbatch * 3.0;
      if (st.wino_has_res) st.exec_flops += 2.0 * (double)round_up(st.wino_res.M, 256) * pw.Npad * ((double)st.wino_res.nsteps * kstep) * 3.0;
      if (st.wino_has_res_low) st.exec_flops += 2.0 * (double)round_up(st.wino_res_low.M, 256) * pw.Npad * ((double)st.wino_res_low.nsteps * kstep) * 3.0;
    }
    st.tile = pw.tile;
    st.use_wino = true;
    (void)nsl;
    return BSMI_OK;
  }

  // One ConvPass (reference unet.py:63-76).  in[s] with per-slot origin org[s]; `sp` is the
  // logical input extent (the extent of the concatenated, cropped input).
  int pass(PassSite& p, const TDesc* in, const int (*org)[3], const int sp_in[3], TDesc& out) {
    int sp[3] = {sp_in[0], sp_in[1], sp_in[2]};
    int crop[3] = {0, 0, 0};
    for (int i = 0; i < p.nconv; ++i)
      for (int d = 0; d < 3; ++d) crop[d] += p.k[i][d] - 1;
    TDesc cur;
    for (int ci = 0; ci < p.nconv; ++ci) {
      const bool last = ci == p.nconv - 1;
      TDesc o;
      o.C = p.cout;
      o.D = sp[0] - (p.k[ci][0] - 1);
      o.H = sp[1] - (p.k[ci][1] - 1);
      o.W = sp[2] - (p.k[ci][2] - 1);
      if (o.D <= 0 || o.H <= 0 || o.W <= 0)
        BSMI_FAIL(BSMI_ERR_INVALID, "%s: input extent (%d,%d,%d) too small for kernel (%d,%d,%d)",
                  p.prefix.c_str(), sp[0], sp[1], sp[2], p.k[ci][0], p.k[ci][1], p.k[ci][2]);
      int rc = alloc(o);
      if (rc) return rc;

      // tensor slots of this launch and their origins
      TDesc slots[kMaxConvTensors];
      int so[kMaxConvTensors][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      int nsl = 0;
      if (ci == 0) {
        for (int s = 0; s < p.nslots; ++s) {
          slots[nsl] = in[s];
          for (int d = 0; d < 3; ++d) so[nsl][d] = org[s][d];
          ++nsl;
        }
      } else {
        slots[nsl++] = cur;
        if (last)
          for (int s = 0; s < p.nslots; ++s) {
            slots[nsl] = in[s];
            for (int d = 0; d < 3; ++d) so[nsl][d] = org[s][d];
            ++nsl;
          }
      }

      // flops: 2 * M * Cout * K_real
      const double M = (double)o.D * o.H * o.W;
      double kreal = 0;
      {
        std::vector<PackEntry> ents;
        build_entries(p, ci, prec, ents);
        for (auto& e : ents) if (!e.dummy) kreal += std::max(0, std::min(e.creal - e.c0, sube(prec)));
      }
      plan->flops += 2.0 * M * p.cout * kreal;

      if (!dry) {
        PackedConv& pc = p.packed[prec][ci];
        PlanStep st;
        st.type = PlanStep::CONV;
        st.tile = pc.tile;
        ConvArgs& a = st.conv;
        memset(&a, 0, sizeof a);
        const int KS = pc.ks;
        std::vector<KStep> ks(pc.entries.size() / kUnitsPerStep * KS);
        const int64_t es = esize(prec) * (prec == BSMI_PREC_BF16X3 ? 2 : 1);  // bytes per channel of a row: (hi, lo) interleaved
        for (int sl = 0; sl < kMaxConvTensors; ++sl) {
          const TDesc& t = slots[sl < nsl ? sl : 0];
          a.t[sl].base = (uint64_t)(uintptr_t)t.ptr;
          a.t[sl].sz = (int32_t)((int64_t)t.H * t.W * t.Cpad * es);
          a.t[sl].sy = (int32_t)((int64_t)t.W * t.Cpad * es);
          a.t[sl].sx = (int32_t)((int64_t)t.Cpad * es);
        }
        for (size_t s = 0; s < ks.size() / KS; ++s) {
          const int slot = pc.entries[kUnitsPerStep * s].slot;
          const TDesc& t = slots[slot];
          KStep k;
          memset(&k, 0, sizeof k);
          k.tensor = slot;
          for (int j = 0; j < kUnitsPerStep; ++j) {
            const PackEntry& e = pc.entries[kUnitsPerStep * s + j];
            if (e.dummy) continue;
            const int64_t off = ((((int64_t)(e.dz + so[slot][0]) * t.H) + (e.dy + so[slot][1])) * t.W +
                                 (e.dx + so[slot][2])) * t.Cpad + e.c0;
            k.delta[j] = (int32_t)(off * es);
          }
          ks[s * KS] = k;
          if (KS == 3) {
            KStep lo = k;  // the same taps of the lo plane (a padding unit stays at offset 0: it meets zero weights)
            for (int j = 0; j < kUnitsPerStep; ++j)
              if (!pc.entries[kUnitsPerStep * s + j].dummy) lo.delta[j] = (int32_t)(k.delta[j] + (int64_t)t.lo_off);
            ks[s * KS + 1] = lo;
            ks[s * KS + 2] = k;
          }
        }
        KStep* dks = nullptr;
        BSMI_HIP(hipMalloc((void**)&dks, ks.size() * sizeof(KStep)));
        plan->allocs.push_back(dks);
        BSMI_HIP(hipMemcpy(dks, ks.data(), ks.size() * sizeof(KStep), hipMemcpyHostToDevice));
        a.steps = dks;
        a.nsteps = (int)ks.size();
        a.w = pc.w;
        a.w_lo = pc.lo_image_bytes ? (const char*)pc.w + pc.lo_image_bytes : nullptr;
        a.bias = pc.bias;
        a.out = o.ptr;
        a.Do = o.D; a.Ho = o.H; a.Wo = o.W; a.Co = o.Cpad;
        a.M = o.D * o.H * o.W;
        a.Npad = pc.Npad;
        a.relu = 1;  // trunk activation is ReLU (model.py passes activation default "ReLU")
        st.flops = 2.0 * M * p.cout * kreal;
        {
          const double x3 = prec == BSMI_PREC_BF16X3 ? 3.0 : 1.0;
          const double kstep = (double)kUnitsPerStep * sube(prec);   // channels x taps of one K-step
          st.exec_flops = 2.0 * (double)round_up(a.M, 256) * pc.Npad * ((double)(ks.size() / KS) * kstep) * x3;
        }
        st.site = &p;
        st.ci = ci;
        st.nsl = nsl;
        for (int sl = 0; sl < nsl; ++sl) {
          st.slots[sl] = slots[sl];
          for (int d = 0; d < 3; ++d) st.so[sl][d] = so[sl][d];
        }
        st.out = o;
        rc = plan_rh(p, ci, pc, slots, so, nsl, o, st);
        if (rc) return rc;
        rc = plan_box(p, ci, slots, so, nsl, o, st);
        if (rc) return rc;
        rc = plan_wino(p, ci, pc, slots, so, nsl, o, st);
        if (rc) return rc;
        if (!st.use_wino && !st.use_box) {
          rc = plan_h16(p, ci, pc, slots, so, o, st);
          if (rc) return rc;
        }
        if (getenv("BSMI_PLAN_DEBUG"))
          fprintf(stderr, "[bsmi plan] %s conv %d: out (%d,%d,%d)x%d tile BN=%d K-steps %d %s\n", p.prefix.c_str(), ci, o.D, o.H, o.W,
                  p.cout, tile_bn(st.tile), st.use_wino ? st.wino_gemm.nsteps : a.nsteps,
                  st.use_wino ? (st.wino_in.m == 4 ? "winograd F(4x4,3x3)" : "winograd F(2x2,3x3)") : st.use_h16 ? "halo-resident" : st.use_box ? "box-halo" : st.use_rh ? "raster-halo" : "gather");
        plan->steps.push_back(st);
      }
      cur = o;
      sp[0] = o.D; sp[1] = o.H; sp[2] = o.W;
    }
    out = cur;
    (void)crop;
    return BSMI_OK;
  }

  int rec(int level, const TDesc& f_in, TDesc& f_out) {
    const int i = h->nl - level - 1;
    TDesc f_left;
    const int org0[1][3] = {{0, 0, 0}};
    const int sp[3] = {f_in.D, f_in.H, f_in.W};
    int rc = pass(h->l_conv[i], &f_in, org0, sp, f_left);
    if (rc) return rc;
    if (level == 0) {
      f_out = f_left;
      return BSMI_OK;
    }
    const int* f = h->cfg.downsample_factors[i];
    const int dims[3] = {f_left.D, f_left.H, f_left.W};
    for (int d = 2; d >= 0; --d)
      if (dims[d] % f[d] != 0)
        BSMI_FAIL(BSMI_ERR_INVALID,
                  "Can not downsample shape (%d, %d, %d) with factor (%d, %d, %d), mismatch in spatial dimension %d",
                  dims[0], dims[1], dims[2], f[0], f[1], f[2], d);
    TDesc g_in;
    g_in.C = f_left.C; g_in.D = f_left.D / f[0]; g_in.H = f_left.H / f[1]; g_in.W = f_left.W / f[2];
    rc = alloc(g_in);
    if (rc) return rc;
    if (!dry) {
      PlanStep st;
      st.type = PlanStep::POOL;
      st.in = f_left; st.out = g_in;
      for (int d = 0; d < 3; ++d) st.f[d] = f[d];
      plan->steps.push_back(st);
    }
    TDesc g_out;
    rc = rec(level - 1, g_in, g_out);
    if (rc) return rc;

    // Upsample.forward (unet.py:215-223): upsample, crop_to_factor, crop skip, concat
    PassSite& rp = h->r_conv[i];
    const int up[3] = {g_out.D * f[0], g_out.H * f[1], g_out.W * f[2]};
    int conv_crop[3] = {0, 0, 0};
    for (int c = 0; c < rp.nconv; ++c)
      for (int d = 0; d < 3; ++d) conv_crop[d] += rp.k[c][d] - 1;
    int target[3];
    for (int d = 0; d < 3; ++d) {
      const int cf = h->crop_factor[i][d];
      const int n = (int)std::floor((double)(up[d] - conv_crop[d]) / cf);
      target[d] = n * cf + conv_crop[d];
      if (target[d] != up[d] && target[d] <= conv_crop[d])
        BSMI_FAIL(BSMI_ERR_INVALID,
                  "Feature map with shape (%d, %d, %d) is too small to ensure translation equivariance",
                  up[0], up[1], up[2]);
    }
    TDesc g_c;
    g_c.C = g_out.C; g_c.D = target[0]; g_c.H = target[1]; g_c.W = target[2];
    rc = alloc(g_c);
    if (rc) return rc;
    int so[2][3];
    for (int d = 0; d < 3; ++d) {
      so[0][d] = (dims[d] - target[d]) / 2;
      so[1][d] = 0;
      if (dims[d] < target[d])
        BSMI_FAIL(BSMI_ERR_INVALID, "skip connection (%d,%d,%d) smaller than upsampled map (%d,%d,%d)",
                  dims[0], dims[1], dims[2], target[0], target[1], target[2]);
    }
    // Fused upsampling (BSMI_FUSE_UP=0: off): when the first and the last stage of the ConvPass both run in Winograd form, they
    // read the map below the upsampling and interpolate on the fly -- the first stage's input transform, and the last stage's
    // residual branch as a 1x1x1 convolution BELOW the upsampling (it commutes with the per-channel interpolation) whose sums
    // the output transform interpolates.  The upsampled map (1.3 GB for the 1500-channel one of the 128^3 block) is then
    // neither written nor read: 0.54 ms of upsampling, 1 GB of transform reads and four fifths of the residual GEMM saved.
    UpFuse uf;
    bool fuse = false;
    if (!dry && prec == BSMI_PREC_BF16X3 && f[0] == 1 && f[1] == 2 && f[2] == 2 && rp.nslots == 2 && rp.wino[0].ready &&
        rp.wino[rp.nconv - 1].ready && !rp.wino[rp.nconv - 1].res_part[1].empty() && g_out.ptr) {
      static const bool on = [] { const char* e = getenv("BSMI_FUSE_UP"); return !(e && e[0] == '0'); }();
      fuse = on;
      int sp[3] = {target[0], target[1], target[2]};
      for (int c = 0; c < rp.nconv && fuse; ++c) {
        for (int d = 0; d < 3; ++d) sp[d] -= rp.k[c][d] - 1;
        const int wm = rp.wino[c].m;
        if ((c == 0 || c == rp.nconv - 1) && ((wm == 2 && ((sp[1] & 1) || (sp[2] & 1))) || sp[0] <= 0 || sp[1] <= 0 || sp[2] <= 0)) fuse = false;
        if ((c == 0 || c == rp.nconv - 1) && (size_t)(sp[0] + 2) * ((sp[1] + wm - 1) / wm) * ((sp[2] + wm - 1) / wm) * rp.wino[c].Cv * 4 >= ((size_t)1 << 31)) fuse = false;
      }
    }
    size_t up_index = 0;
    if (!dry) {
      PlanStep st;
      st.type = PlanStep::UP;
      st.in = g_out; st.out = g_c;
      for (int d = 0; d < 3; ++d) { st.f[d] = f[d]; st.o[d] = (up[d] - target[d]) / 2; }
      st.skip = fuse;
      up_index = plan->steps.size();
      plan->steps.push_back(st);
    }
    if (fuse) {
      uf.low = g_out;
      for (int d = 0; d < 3; ++d) { uf.f[d] = f[d]; uf.o[d] = (up[d] - target[d]) / 2; }
      fuse_up = &uf;
      if (getenv("BSMI_PLAN_DEBUG")) fprintf(stderr, "[bsmi plan] %s: upsampling fused into the Winograd stages\n", rp.prefix.c_str());
    }
    const TDesc ins[2] = {f_left, g_c};
    rc = pass(rp, ins, so, target, f_out);
    fuse_up = nullptr;
    (void)up_index;
    return rc;
  }

  int run(const int64_t in_shape[3]) {
    TDesc x;
    x.C = h->cfg.in_channels; x.D = (int)in_shape[0]; x.H = (int)in_shape[1]; x.W = (int)in_shape[2];
    int rc = alloc(x);
    if (rc) return rc;
    if (!dry) {
      PlanStep st;
      st.type = PlanStep::INPUT;
      st.out = x;
      plan->steps.push_back(st);
    }
    TDesc z;
    rc = rec(h->nl - 1, x, z);
    if (rc) return rc;
    plan->out_shape[0] = z.D; plan->out_shape[1] = z.H; plan->out_shape[2] = z.W;
    for (size_t hd = 0; hd < h->heads.size(); ++hd) {
      plan->flops += 2.0 * 2.0 * (double)z.D * z.H * z.W * h->heads[hd].cin * h->heads[hd].cout;
      if (!dry) {
        PlanStep st;
        st.type = PlanStep::HEAD;
        st.in = z;
        st.head = (int)hd;
        st.flops = 2.0 * 2.0 * (double)z.D * z.H * z.W * h->heads[hd].cin * h->heads[hd].cout;
        plan->steps.push_back(st);
      }
    }
    return BSMI_OK;
  }
};

static void free_plan(Plan* p) {
  for (void* a : p->allocs) (void)hipFree(a);
  p->allocs.clear();
  for (auto* sets : {&p->inflight, &p->spare})
    for (auto& set : *sets)
      for (hipEvent_t e : set) (void)hipEventDestroy(e);
  p->inflight.clear();
  p->spare.clear();
  p->events.clear();
}

// Wait for the events of the profiled forwards of `plan` that have not been read yet and fold them into the totals.
static int harvest(bsmi_unet* h, Plan* plan) {
  if (!plan || plan->inflight.empty()) return BSMI_OK;
  const size_t n = plan->steps.size();
  for (auto& set : plan->inflight) {
    plan->last_ms.assign(n, 0.f);
    BSMI_HIP(hipEventSynchronize(set[2 * n - 1]));
    for (size_t i = 0; i < n; ++i) {
      float t = 0.f;
      BSMI_HIP(hipEventElapsedTime(&t, set[2 * i], set[2 * i + 1]));
      plan->last_ms[i] = t;
      // with the fused first pass, steps 0 and 1 launch nothing: their work is done (and timed) in step 2
      const int ty = (plan->fused_first && i < 2) ? (int)PlanStep::CONV : (int)plan->steps[i].type;
      h->prof_ms[ty] += t;
      h->prof_flops[ty] += plan->steps[i].flops;
      if (ty == 1) h->prof_exec += plan->steps[i].exec_flops > 0 ? plan->steps[i].exec_flops : plan->steps[i].flops * (plan->prec == BSMI_PREC_BF16X3 ? 3.0 : 1.0);
      h->prof_launches[ty] += (plan->fused_first && i < 2) ? 0 : 1;
    }
    plan->spare.push_back(std::move(set));
  }
  plan->inflight.clear();
  plan->pending = false;
  return BSMI_OK;
}

// Opt-in (BSMI_FORWARD_CHAIN=1): one forward pass in flight per GPU and process.  Round 4 found predictions corrupted when two
// forward passes overlapped on one GPU (two handles on two streams, or two processes): the victim was the head kernel, the only
// launch of a pass with a scratch segment (272 bytes per lane); every other launch's output was intact.  The
// head kernel keeps its channels in registers now and no kernel of the engine has a scratch segment (tests/test_kernel_resources.py),
// so passes may overlap.  The chain stays as a diagnostic: a forward on another stream than the previous one first waits, on the
// device, for that one's last launch (an event recorded at the end of every forward).
namespace {
std::mutex g_chain_mu;
struct ChainState { hipEvent_t done = nullptr; hipStream_t last = nullptr; bool any = false; };
ChainState g_chain[16];
}  // namespace
struct ForwardChain {
  std::unique_lock<std::mutex> lock{g_chain_mu};
};
static bool forward_chain_on() {   // BSMI_FORWARD_CHAIN=1: on.  Off by default since the head kernel lost its scratch segment (the trigger)
  static const bool on = [] { const char* e = getenv("BSMI_FORWARD_CHAIN"); return e && e[0] == '1'; }();
  return on;
}
static int forward_chain_enter(int device, hipStream_t s) {
  if (device < 0 || device >= 16 || !forward_chain_on()) return BSMI_OK;
  ChainState& c = g_chain[device];
  if (c.any && c.last != s) BSMI_HIP(hipStreamWaitEvent(s, c.done, 0));
  return BSMI_OK;
}
static int forward_chain_leave(int device, hipStream_t s) {
  if (device < 0 || device >= 16 || !forward_chain_on()) return BSMI_OK;
  ChainState& c = g_chain[device];
  if (!c.done) BSMI_HIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
  BSMI_HIP(hipEventRecord(c.done, s));
  c.last = s;
  c.any = true;
  return BSMI_OK;
}

// The first ConvPass can run as one launch (first_pass.hip): one raw channel, two 3x3x3 convs, at most 16 feature maps.
static bool first_pass_eligible(const bsmi_unet* h) {
  if (h->cfg.in_channels != 1 || h->l_conv.empty()) return false;
  const PassSite& p = h->l_conv[0];
  if (p.nconv != 2 || p.cout > 16) return false;
  for (int c = 0; c < 2; ++c)
    for (int d = 0; d < 3; ++d)
      if (p.k[c][d] != 3) return false;
  const char* e = getenv("BSMI_FUSED_FIRST");
  return !(e && e[0] == '0');
}

int get_plan(bsmi_unet* h, int precision, const int64_t in_shape[3], Plan** out) {
  const std::vector<int64_t> key = {precision, in_shape[0], in_shape[1], in_shape[2]};
  auto it = h->plans.find(key);
  if (it == h->plans.end()) {
    std::unique_ptr<Plan> plan(new Plan);
    plan->prec = precision;
    Planner pl{h, precision, plan.get(), false};
    const int rc = pl.run(in_shape);
    if (rc) {
      free_plan(plan.get());
      return rc;
    }
    const std::vector<PlanStep>& ps = plan->steps;
    const bool fp_ready = precision == BSMI_PREC_BF16 ? h->first_pass.ready : precision == BSMI_PREC_BF16X3 && h->first_pass_x3.ready;
    plan->fused_first = fp_ready && first_pass_eligible(h) && ps.size() > 3 &&
                        ps[0].type == PlanStep::INPUT && ps[1].type == PlanStep::CONV && ps[2].type == PlanStep::CONV &&
                        ps[1].site == &h->l_conv[0] && ps[2].site == &h->l_conv[0] && ps[2].out.Cpad == 16;
    // the planner's uploads and fills (hipMemcpy / hipMemsetAsync on the null stream) before any launch on a caller's non-blocking stream
    BSMI_HIP(hipDeviceSynchronize());
    it = h->plans.emplace(key, std::move(plan)).first;
  }
  *out = it->second.get();
  return BSMI_OK;
}

static int check_shape_arg(const int64_t s[3]) {
  for (int d = 0; d < 3; ++d)
    if (s[d] <= 0 || s[d] > 4096) BSMI_FAIL(BSMI_ERR_INVALID, "bad input shape (%lld,%lld,%lld)", (long long)s[0], (long long)s[1], (long long)s[2]);
  return BSMI_OK;
}

}  // namespace bsmi

extern "C" {

const char* bsmi_last_error(void) { return g_err.c_str(); }
int bsmi_version(void) { return 1; }

int bsmi_unet_create(const bsmi_unet_config* cfg, int device, bsmi_unet** out) {
  if (!cfg || !out) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (cfg->num_levels < 1 || cfg->num_levels > BSMI_MAX_LEVELS) BSMI_FAIL(BSMI_ERR_INVALID, "num_levels %d out of range", cfg->num_levels);
  if (cfg->num_heads < 1 || cfg->num_heads > BSMI_MAX_HEADS) BSMI_FAIL(BSMI_ERR_INVALID, "num_heads %d out of range", cfg->num_heads);
  if (cfg->in_channels < 1 || cfg->num_fmaps < 1 || cfg->fmap_inc_factor < 1 || cfg->num_fmaps_out < 0) BSMI_FAIL(BSMI_ERR_INVALID, "bad channel configuration");
  // no HIP call here: shape / flop arithmetic must work on a host without a GPU
  std::unique_ptr<bsmi_unet> h(new bsmi_unet);
  h->cfg = *cfg;
  h->device = device;
  h->nl = cfg->num_levels;
  auto fm = [&](int level) {
    int64_t c = cfg->num_fmaps;
    for (int i = 0; i < level; ++i) c *= cfg->fmap_inc_factor;
    return (int)c;
  };
  auto fill_k = [&](PassSite& p, int n, const int32_t (*k)[3]) -> int {
    if (n < 1 || n > BSMI_MAX_CONVS) BSMI_FAIL(BSMI_ERR_INVALID, "%s: %d convolutions per pass unsupported", p.prefix.c_str(), n);
    p.nconv = n;
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < 3; ++d) {
        if (k[i][d] < 1 || k[i][d] > 7) BSMI_FAIL(BSMI_ERR_INVALID, "%s: kernel size %d unsupported", p.prefix.c_str(), k[i][d]);
        p.k[i][d] = k[i][d];
      }
    return BSMI_OK;
  };
  for (int l = 0; l < h->nl; ++l) {
    PassSite p;
    p.prefix = "unet.l_conv." + std::to_string(l);
    p.nslots = 1;
    p.cin[0] = l == 0 ? cfg->in_channels : fm(l - 1);
    p.cout = fm(l);
    int rc = fill_k(p, cfg->n_convs_down[l], cfg->kernel_size_down[l]);
    if (rc) return rc;
    h->l_conv.push_back(p);
  }
  for (int l = 0; l < h->nl - 1; ++l) {
    PassSite p;
    p.prefix = "unet.r_conv.0." + std::to_string(l);
    p.nslots = 2;
    p.cin[0] = fm(l);      // skip connection first (torch.cat([f_cropped, g_cropped]), unet.py:223)
    p.cin[1] = fm(l + 1);
    p.cout = (l == 0 && cfg->num_fmaps_out > 0) ? cfg->num_fmaps_out : fm(l);  // unet.py:426-427
    int rc = fill_k(p, cfg->n_convs_up[l], cfg->kernel_size_up[l]);
    if (rc) return rc;
    h->r_conv.push_back(p);
  }
  // crop factors (unet.py:353-362): running product of the downsample factors from the bottom
  {
    int prod[3] = {1, 1, 1};
    for (int l = h->nl - 2; l >= 0; --l) {
      for (int d = 0; d < 3; ++d) {
        if (cfg->downsample_factors[l][d] < 1) BSMI_FAIL(BSMI_ERR_INVALID, "bad downsample factor");
        prod[d] *= cfg->downsample_factors[l][d];
        h->crop_factor[l][d] = prod[d];
      }
    }
  }
  for (auto& p : h->l_conv) register_pass(h.get(), p);
  for (auto& p : h->r_conv) register_pass(h.get(), p);
  for (int i = 0; i < cfg->num_heads; ++i) {
    HeadSite hs;
    hs.prefix = std::string(cfg->head_name[i], strnlen(cfg->head_name[i], BSMI_NAME_LEN));
    hs.cin = (cfg->num_fmaps_out > 0 && h->nl > 1) ? cfg->num_fmaps_out : cfg->num_fmaps;
    hs.cout = cfg->head_dims[i];
    if (hs.cout < 1 || hs.cout > 64) BSMI_FAIL(BSMI_ERR_INVALID, "head %s: dims %d unsupported", hs.prefix.c_str(), hs.cout);
    expect_weight(h.get(), hs.prefix + ".conv_pass.0.weight", {hs.cout, hs.cin, 1, 1, 1});
    expect_weight(h.get(), hs.prefix + ".conv_pass.0.bias", {hs.cout});
    expect_weight(h.get(), hs.prefix + ".residual.0.weight", {hs.cout, hs.cin, 1, 1, 1});
    expect_weight(h.get(), hs.prefix + ".residual.0.bias", {hs.cout});
    h->heads.push_back(hs);
  }
  *out = h.release();
  return BSMI_OK;
}

int bsmi_stream_create_cu_mask(int device, const uint32_t* cu_mask, int n_words, void** stream_out) {
  if (!stream_out || (cu_mask && n_words < 1)) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  BSMI_HIP(hipSetDevice(device));
  hipStream_t s = nullptr;
  if (!cu_mask) BSMI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));  // an ordinary stream the caller may destroy again
  else BSMI_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask));
  *stream_out = (void*)s;
  return BSMI_OK;
}

int bsmi_stream_destroy(int device, void* stream) {
  if (!stream) return BSMI_OK;
  BSMI_HIP(hipSetDevice(device));
  BSMI_HIP(hipStreamDestroy((hipStream_t)stream));
  return BSMI_OK;
}

int bsmi_unet_set_persistent_grid(bsmi_unet* h, int n_cus) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  if (n_cus < -1) BSMI_FAIL(BSMI_ERR_INVALID, "n_cus must be >= -1");
  BSMI_HIP(hipSetDevice(h->device));
  if (h->sk_ws) {
    BSMI_HIP(hipDeviceSynchronize());
    BSMI_HIP(hipFree(h->sk_ws));
    h->sk_ws = nullptr;
  }
  h->sk_grid = 0;
  h->sk_request = n_cus < 0 ? -1 : n_cus / 8 * 8;
  return BSMI_OK;
}

int bsmi_unet_destroy(bsmi_unet* h) {
  if (!h) return BSMI_OK;
  (void)hipSetDevice(h->device);
  for (auto& kv : h->plans) free_plan(kv.second.get());
  auto free_site = [](PassSite& p) {
    for (int pr = 0; pr < BSMI_NUM_PREC; ++pr)
      for (int c = 0; c < BSMI_MAX_CONVS; ++c) {
        if (p.packed[pr][c].w) (void)hipFree(p.packed[pr][c].w);
        if (p.packed[pr][c].bias) (void)hipFree(p.packed[pr][c].bias);
      }
    for (int c = 0; c < BSMI_MAX_CONVS; ++c) {
      if (p.h16[c].w) (void)hipFree(p.h16[c].w);
      if (p.wino[c].w) (void)hipFree(p.wino[c].w);
      if (p.wino[c].res_w) (void)hipFree(p.wino[c].res_w);
      for (int part = 0; part < 2; ++part)
        if (p.wino[c].res_part_w[part]) (void)hipFree(p.wino[c].res_part_w[part]);
    }
  };
  for (auto& p : h->l_conv) free_site(p);
  for (auto& p : h->r_conv) free_site(p);
  free_train_state(h);
  free_first_pass(h->first_pass);
  free_first_pass(h->first_pass_x3);
  if (h->sk_ws) (void)hipFree(h->sk_ws);
  for (auto& hd : h->heads) {
    if (hd.hw) (void)hipFree(hd.hw);
    if (hd.hb) (void)hipFree(hd.hb);
  }
  delete h;
  return BSMI_OK;
}

int bsmi_unet_load_weight(bsmi_unet* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!h || !key || !data || !shape) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  auto it = h->weights.find(key);
  if (it == h->weights.end()) BSMI_FAIL(BSMI_ERR_MISSING, "Unexpected key(s) in state_dict: \"%s\"", key);
  HostWeight& w = it->second;
  bool ok = (size_t)ndim == w.shape.size();
  for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == w.shape[i];
  if (!ok) {
    std::string got, want;
    for (int i = 0; i < ndim; ++i) got += std::to_string(shape[i]) + (i + 1 < ndim ? ", " : "");
    for (size_t i = 0; i < w.shape.size(); ++i) want += std::to_string(w.shape[i]) + (i + 1 < w.shape.size() ? ", " : "");
    BSMI_FAIL(BSMI_ERR_INVALID, "size mismatch for %s: copying a param with shape (%s), the shape in current model is (%s)",
              key, got.c_str(), want.c_str());
  }
  size_t n = 1;
  for (auto s : w.shape) n *= (size_t)s;
  w.data.assign(data, data + n);
  w.loaded = true;
  // weights changed: packed copies are stale
  for (int pr = 0; pr < BSMI_NUM_PREC; ++pr) h->finalized[pr] = false;
  return BSMI_OK;
}

int bsmi_unet_finalize(bsmi_unet* h, int precision) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  if (precision < 0 || precision >= BSMI_NUM_PREC) BSMI_FAIL(BSMI_ERR_INVALID, "unknown precision %d", precision);
  std::string missing;
  for (auto& kv : h->weights)
    if (!kv.second.loaded) missing += (missing.empty() ? "\"" : ", \"") + kv.first + "\"";
  if (!missing.empty()) BSMI_FAIL(BSMI_ERR_MISSING, "Missing key(s) in state_dict: %s", missing.c_str());
  BSMI_HIP(hipSetDevice(h->device));
  auto repack = [&](PassSite& p) -> int {
    for (int c = 0; c < p.nconv; ++c) {
      PackedConv& pc = p.packed[precision][c];
      if (pc.w) { (void)hipFree(pc.w); pc.w = nullptr; }
      if (pc.bias) { (void)hipFree(pc.bias); pc.bias = nullptr; }
      pc.ready = false;
      int rc = pack_conv(h, p, c, precision);
      if (rc) return rc;
      if (precision == BSMI_PREC_BF16X3 && (rc = pack_wino(h, p, c))) return rc;
      if (precision == BSMI_PREC_BF16X3 && (rc = pack_h16(h, p, c))) return rc;
    }
    return BSMI_OK;
  };
  {
    // the ConvPasses side by side (each packs its stages on host threads of its own and uploads its images): the 1500-channel
    // pass alone is two thirds of the work, the others hide behind it
    std::vector<PassSite*> sites;
    for (auto& p : h->l_conv) sites.push_back(&p);
    for (auto& p : h->r_conv) sites.push_back(&p);
    std::vector<int> rcs(sites.size(), BSMI_OK);
    std::vector<std::string> msgs(sites.size());
    const bool timing = getenv("BSMI_PLAN_DEBUG") != nullptr;
    std::vector<std::thread> th;
    for (size_t i = 0; i < sites.size(); ++i)
      th.emplace_back([&, i] {
        (void)hipSetDevice(h->device);
        const auto t0 = std::chrono::steady_clock::now();
        rcs[i] = repack(*sites[i]);
        if (rcs[i]) msgs[i] = bsmi_last_error();
        if (timing) fprintf(stderr, "[bsmi finalize] %s: %.3f s\n", sites[i]->prefix.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      });
    for (auto& t : th) t.join();
    for (size_t i = 0; i < sites.size(); ++i)
      if (rcs[i]) { bsmi::set_error("%s", msgs[i].c_str()); return rcs[i]; }
  }
  for (auto& hd : h->heads) {
    std::vector<float> hw((size_t)hd.cout * 2 * hd.cin), hb((size_t)hd.cout * 2);
    const HostWeight& w1 = h->weights[hd.prefix + ".conv_pass.0.weight"];
    const HostWeight& b1 = h->weights[hd.prefix + ".conv_pass.0.bias"];
    const HostWeight& w2 = h->weights[hd.prefix + ".residual.0.weight"];
    const HostWeight& b2 = h->weights[hd.prefix + ".residual.0.bias"];
    for (int o = 0; o < hd.cout; ++o) {
      for (int c = 0; c < hd.cin; ++c) {
        hw[((size_t)o * 2 + 0) * hd.cin + c] = w1.data[(size_t)o * hd.cin + c];
        hw[((size_t)o * 2 + 1) * hd.cin + c] = w2.data[(size_t)o * hd.cin + c];
      }
      hb[o * 2] = b1.data[o];
      hb[o * 2 + 1] = b2.data[o];
    }
    if (!hd.hw) BSMI_HIP(hipMalloc((void**)&hd.hw, hw.size() * sizeof(float)));
    if (!hd.hb) BSMI_HIP(hipMalloc((void**)&hd.hb, hb.size() * sizeof(float)));
    BSMI_HIP(hipMemcpy(hd.hw, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice));
    BSMI_HIP(hipMemcpy(hd.hb, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if ((precision == BSMI_PREC_BF16 || precision == BSMI_PREC_BF16X3) && first_pass_eligible(h)) {
    const std::string pre = h->l_conv[0].prefix;
    const bool split = precision == BSMI_PREC_BF16X3;
    int rc = pack_first_pass(split ? h->first_pass_x3 : h->first_pass, h->l_conv[0].cout, h->weights[pre + ".conv_pass.0.weight"].data.data(),
                             h->weights[pre + ".conv_pass.0.bias"].data.data(), h->weights[pre + ".conv_pass.2.weight"].data.data(),
                             h->weights[pre + ".conv_pass.2.bias"].data.data(), h->weights[pre + ".residual.0.weight"].data.data(),
                             h->weights[pre + ".residual.0.bias"].data.data(), split);
    if (rc) return rc;
  }
  // plans hold pointers to packed weights: drop those of this precision
  h->last_plan = nullptr;
  for (auto it = h->plans.begin(); it != h->plans.end();) {
    if (it->first[0] == precision) { free_plan(it->second.get()); it = h->plans.erase(it); }
    else ++it;
  }
  h->finalized[precision] = true;
  return BSMI_OK;
}

static int dry_plan(bsmi_unet* h, const int64_t in_shape[3], Plan& plan) {
  int rc = check_shape_arg(in_shape);
  if (rc) return rc;
  Planner pl{h, BSMI_PREC_BF16, &plan, true};
  return pl.run(in_shape);
}

int bsmi_unet_output_shape(bsmi_unet* h, const int64_t in_shape[3], int64_t out_shape[3]) {
  if (!h || !in_shape || !out_shape) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  Plan plan;
  int rc = dry_plan(h, in_shape, plan);
  if (rc) return rc;
  for (int d = 0; d < 3; ++d) out_shape[d] = plan.out_shape[d];
  return BSMI_OK;
}

int bsmi_unet_flops(bsmi_unet* h, const int64_t in_shape[3], double* flops) {
  if (!h || !in_shape || !flops) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  Plan plan;
  int rc = dry_plan(h, in_shape, plan);
  if (rc) return rc;
  *flops = plan.flops;
  return BSMI_OK;
}

int bsmi_unet_forward(bsmi_unet* h, int precision, const void* raw_dev, int raw_dtype,
                      const int64_t in_shape[3], float* const* out_f32_dev,
                      uint8_t* const* out_u8_dev, void* stream) {
  if (!h || !raw_dev || !in_shape) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  if (precision < 0 || precision >= BSMI_NUM_PREC) BSMI_FAIL(BSMI_ERR_INVALID, "unknown precision %d", precision);
  if (raw_dtype != BSMI_RAW_U8 && raw_dtype != BSMI_RAW_F32 && raw_dtype != BSMI_RAW_U8_UNIT) BSMI_FAIL(BSMI_ERR_INVALID, "unknown raw dtype %d", raw_dtype);
  if (!h->finalized[precision]) BSMI_FAIL(BSMI_ERR_STATE, "bsmi_unet_finalize(precision=%d) has not been called", precision);
  int rc = check_shape_arg(in_shape);
  if (rc) return rc;
  BSMI_HIP(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  ForwardChain chain_guard;   // (holds the process-wide lock of the chain until the launches are queued)
  if ((rc = forward_chain_enter(h->device, s))) return rc;
  Plan* plan_ptr = nullptr;
  rc = get_plan(h, precision, in_shape, &plan_ptr);
  if (rc) return rc;
  Plan& plan = *plan_ptr;
  h->last_plan = &plan;
  // profile_period N > 0: every Nth forward records its per-step events (N = 1: every forward)
  const bool prof = h->profile_period > 0 && (h->profile_count++ % h->profile_period) == 0;
  if (prof || h->profile_period == 0) plan.profiled = prof;
  if (prof) plan.pending = true;
  if (prof) {
    // a fresh set of events for this forward: recording must not wait for an earlier forward to finish (the caller may
    // have many blocks in flight); bsmi_unet_profile_read / _totals read the sets
    const size_t n_ev = 2 * plan.steps.size();
    if (plan.inflight.size() >= 4096) {  // nobody reads: do not grow without bound
      rc = harvest(h, &plan);
      if (rc) return rc;
    }
    if (!plan.spare.empty() && plan.spare.back().size() == n_ev) {
      plan.events = std::move(plan.spare.back());
      plan.spare.pop_back();
    } else {
      plan.events.assign(n_ev, nullptr);
      for (auto& e : plan.events) BSMI_HIP(hipEventCreate(&e));
    }
  }
  if (!h->sk_grid) {
    const char* e = getenv("BSMI_STREAMK");
    if ((e && e[0] == '0') || h->sk_request == 0) {
      h->sk_grid = -1;
    } else {
      hipDeviceProp_t prop;
      BSMI_HIP(hipGetDeviceProperties(&prop, h->device));
      h->sk_grid = h->sk_request >= 0 ? h->sk_request : prop.multiProcessorCount / 8 * 8;
      if (const char* g = getenv("BSMI_SK_GRID")) h->sk_grid = std::max(8, atoi(g) / 8 * 8);  // tests: force cuts on small nets
      BSMI_HIP(hipMalloc((void**)&h->sk_ws, stream_k_ws_bytes(h->sk_grid)));
      // (on the forward's own stream: a fill on the null stream is not ordered before launches on a non-blocking stream)
      BSMI_HIP(hipMemsetAsync((char*)h->sk_ws + stream_k_ws_bytes(h->sk_grid) - 64, 0, 64, s));
    }
  }
  // f32 inference on a handle that is training: the f32 images of the launches that train in their split-bf16 form are
  // brought up to date first (train.hip keeps only what the training step itself reads current)
  if (h->train && precision == BSMI_PREC_F32 && !h->train_forward && (rc = train_refresh_f32_images(h, s))) return rc;
  size_t step_idx = 0;
  for (const PlanStep& st : plan.steps) {
    if (prof) BSMI_HIP(hipEventRecord(plan.events[2 * step_idx], s));
    if (plan.fused_first && step_idx < 3) {
      // l_conv.0 as one launch in place of its second conv; the input-preparation and first-conv steps fall away
      if (step_idx == 2) {
        FirstPassArgs fa;
        fa.raw = raw_dev; fa.raw_dtype = raw_dtype;
        fa.D = plan.steps[0].out.D; fa.H = plan.steps[0].out.H; fa.W = plan.steps[0].out.W;
        fa.out = (uint16_t*)st.out.ptr;
        const FirstPassWeights& fw = precision == BSMI_PREC_BF16X3 ? h->first_pass_x3 : h->first_pass;
        fa.w1a = fw.w1a; fa.w2a = fw.w2a; fa.vec = fw.vec;
        fa.split = fw.split;
        rc = launch_first_pass(fa, h->sk_grid > 0 ? h->sk_grid : 256, s);
        if (rc) return rc;
      }
      if (prof) BSMI_HIP(hipEventRecord(plan.events[2 * step_idx + 1], s));
      ++step_idx;
      continue;
    }
    switch (st.type) {
      case PlanStep::INPUT:
        rc = launch_input_prep(precision, raw_dev, raw_dtype, st.out.ptr, st.out.C, st.out.Cpad,
                               (size_t)st.out.D * st.out.H * st.out.W, s);
        break;
      case PlanStep::CONV:
        if (st.use_wino && !(st.tx3 && h->train_forward)) {
          if ((rc = launch_wino_in(st.wino_in, s))) break;
          if ((rc = launch_conv_igemm(st.wino_gemm, precision, st.tile, s, h->sk_ws, h->sk_grid))) break;
          if (st.wino_has_res && (rc = launch_conv_igemm(st.wino_res, precision, st.tile, s, h->sk_ws, h->sk_grid))) break;
          if (st.wino_has_res_low && (rc = launch_conv_igemm(st.wino_res_low, precision, st.tile, s, h->sk_ws, h->sk_grid))) break;
          rc = launch_wino_out(st.wino_out, s);
          break;
        }
        rc = (st.tx3 && h->train_forward) ? train_forward_conv_x3(h, st, s)
             : st.use_h16 ? launch_conv_h16(st.h16, st.h16_rows, s)
             : st.use_box ? launch_conv_box(st.box, s)
             : st.use_rh ? launch_conv_rh(st.rh, precision, st.tile, s, h->sk_ws, h->sk_grid)
                         : launch_conv_igemm(st.conv, precision, st.tile, s, h->sk_ws, h->sk_grid);
        break;
      case PlanStep::POOL:
        rc = launch_maxpool(precision, st.in.ptr, st.out.ptr, st.in.D, st.in.H, st.in.W, st.in.Cpad,
                            st.f[0], st.f[1], st.f[2], s);
        break;
      case PlanStep::UP:
        if (st.skip) break;  // its readers interpolate on the fly (Planner::rec, fused upsampling)
        rc = launch_upsample_crop(precision, st.in.ptr, st.out.ptr, st.in.D, st.in.H, st.in.W, st.in.Cpad,
                                  st.out.D, st.out.H, st.out.W, st.f[0], st.f[1], st.f[2], st.o[0], st.o[1], st.o[2], s);
        break;
      case PlanStep::HEAD: {
        const HeadSite& hd = h->heads[st.head];
        float* of = out_f32_dev ? out_f32_dev[st.head] : nullptr;
        uint8_t* ou = out_u8_dev ? out_u8_dev[st.head] : nullptr;
        if (of || ou)
          rc = launch_head(precision, st.in.ptr, st.in.Cpad, hd.cin, hd.cout, hd.hw, hd.hb, of, ou,
                           (size_t)st.in.D * st.in.H * st.in.W, s);
        break;
      }
    }
    if (rc) return rc;
    if (prof) BSMI_HIP(hipEventRecord(plan.events[2 * step_idx + 1], s));
    ++step_idx;
  }
  if (prof) plan.inflight.push_back(std::move(plan.events));
  return forward_chain_leave(h->device, s);
}

int bsmi_unet_debug_activation(bsmi_unet* h, int step, int what, int64_t shape_out[4], float* host_out, uint64_t capacity) {
  if (!h || !shape_out) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  Plan* plan = h->last_plan;
  if (!plan) BSMI_FAIL(BSMI_ERR_STATE, "no forward has run");
  if (step < 0 || step >= (int)plan->steps.size()) BSMI_FAIL(BSMI_ERR_INVALID, "step %d out of range (%zu steps)", step, plan->steps.size());
  const PlanStep& st = plan->steps[step];
  if (st.type == PlanStep::HEAD) BSMI_FAIL(BSMI_ERR_INVALID, "head steps write into the caller's buffers");
  const TDesc& t = st.out;
  shape_out[0] = t.D; shape_out[1] = t.H; shape_out[2] = t.W; shape_out[3] = t.C;
  if (!host_out) return BSMI_OK;
  const size_t nvox = (size_t)t.D * t.H * t.W;
  if (capacity < nvox * t.C) BSMI_FAIL(BSMI_ERR_INVALID, "buffer of %llu floats too small", (unsigned long long)capacity);
  BSMI_HIP(hipSetDevice(h->device));
  BSMI_HIP(hipDeviceSynchronize());
  const int es = esize(plan->prec);
  const bool split = t.lo_off != 0;
  const size_t bytes = nvox * t.Cpad * es * (split ? 2 : 1);
  std::vector<uint8_t> raw(bytes);
  BSMI_HIP(hipMemcpy(raw.data(), t.ptr, raw.size(), hipMemcpyDeviceToHost));
  for (size_t v = 0; v < nvox; ++v)
    for (int c = 0; c < t.C; ++c) {
      float x;
      if (plan->prec == BSMI_PREC_F32) {
        x = ((const float*)raw.data())[v * t.Cpad + c];
      } else if (!split) {
        x = host_bf16_to_f32(((const uint16_t*)raw.data())[v * t.Cpad + c]);
      } else {  // (hi, lo) vectors of 8 interleaved
        const size_t i = 2 * v * t.Cpad + (size_t)((c >> 3) << 4) + (c & 7);
        const float hi = host_bf16_to_f32(((const uint16_t*)raw.data())[i]);
        const float lo = host_bf16_to_f32(((const uint16_t*)raw.data())[i + 8]);
        x = what == 1 ? hi : (what == 2 ? lo : hi + lo);
      }
      host_out[v * t.C + c] = x;
    }
  return BSMI_OK;
}

int bsmi_unet_profile_enable(bsmi_unet* h, int on) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  h->profile_period = on > 0 ? on : 0;
  h->profile_count = 0;
  return BSMI_OK;
}

int bsmi_unet_profile_read(bsmi_unet* h, int max_n, int* n, int32_t* types, double* ms, double* flops) {
  if (!h || !n) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  Plan* plan = h->last_plan;
  if (!plan || !plan->profiled) BSMI_FAIL(BSMI_ERR_STATE, "no profiled forward to read");
  int rc = harvest(h, plan);
  if (rc) return rc;
  const int cnt = (int)plan->steps.size();
  *n = cnt;
  for (int i = 0; i < cnt && i < max_n; ++i) {
    if (types) types[i] = (int32_t)plan->steps[i].type;
    if (ms) ms[i] = plan->last_ms[i];
    if (flops) flops[i] = plan->steps[i].flops;
  }
  return BSMI_OK;
}

int bsmi_unet_profile_totals(bsmi_unet* h, double ms_by_type[5], double flops_by_type[5],
                             int64_t launches_by_type[5], int reset) {
  if (!h) BSMI_FAIL(BSMI_ERR_INVALID, "null handle");
  int rc = harvest(h, h->last_plan);
  if (rc) return rc;
  for (int i = 0; i < 5; ++i) {
    if (ms_by_type) ms_by_type[i] = h->prof_ms[i];
    if (flops_by_type) flops_by_type[i] = h->prof_flops[i];
    if (launches_by_type) launches_by_type[i] = h->prof_launches[i];
    if (reset) { h->prof_ms[i] = 0; h->prof_flops[i] = 0; h->prof_launches[i] = 0; }
  }
  return BSMI_OK;
}

int bsmi_unet_profile_executed(bsmi_unet* h, double* executed_flops, int reset) {
  if (!h || !executed_flops) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  *executed_flops = h->prof_exec;
  if (reset) h->prof_exec = 0;
  return BSMI_OK;
}

int bsmi_extract_block_reflect_u8(const uint8_t* vol_dev, const int64_t vol_shape[3], const int64_t offset[3],
                                  const int64_t block_shape[3], uint8_t* block_dev, void* stream) {
  if (!vol_dev || !vol_shape || !offset || !block_shape || !block_dev) BSMI_FAIL(BSMI_ERR_INVALID, "null argument");
  for (int d = 0; d < 3; ++d)
    if (vol_shape[d] <= 0 || block_shape[d] <= 0 || vol_shape[d] > (1 << 20) || block_shape[d] > (1 << 20))
      BSMI_FAIL(BSMI_ERR_INVALID, "bad shape");
  return launch_extract_block_reflect(vol_dev, vol_shape, offset, block_shape, block_dev, (hipStream_t)stream);
}

}  // extern "C"
