// Halo-tiled implicit-GEMM 3-D convolution (gfx950).  See conv_halo.hip.
#pragma once
#include "conv_igemm.h"

namespace bsmi {

// One source tensor of a halo launch (channels-last, byte strides) with its extent and the
// origin (voxels) that output voxel (0,0,0) / tap (0,0,0) maps to.
struct HaloSrc {
  uint64_t base;
  int32_t sz, sy, sx;
  int32_t D, H, W;
  int32_t oz, oy, ox;  // origin for the main taps (crop of the skip connection)
  int32_t rz, ry, rx;  // origin for the residual 1x1x1 tap (origin + total crop / 2)
};

// A phase = one 32-channel chunk (64-byte halo rows) of one source tensor, staged once into an
// LDS halo buffer and consumed by `nsteps` K-steps.  LONG: the input halo of the output box, read
// by all kernel taps; SHORT: the box's own voxels at the residual crop offset (1x1x1 branch).
struct HaloPhase {
  int32_t tensor;
  int32_t c0;       // channel offset in bytes
  int32_t kind;     // 0 LONG, 1 SHORT
  int32_t bufbase;  // LDS byte offset of the halo buffer this phase is staged into
};
static_assert(sizeof(HaloPhase) == 16, "HaloPhase layout");

// One K-step = 2 units of 32 bytes of K, each read at halo row (lane row + trow[u]) and 16-byte
// chunk (cb[u] + lane half): one tap x 32 channels (trow equal, cb = 0,2) or, for a 16-channel
// tensor, two taps (cb = 0,0).
struct HaloStep {
  int32_t trow[2];
  int32_t cb;       // cb0 | cb1 << 8 | kind << 16 (kind 1 = SHORT: rows are the box rows, no tap offset)
  int32_t bufbase;  // LDS byte offset of the halo buffer read by this step
  int32_t wait;     // vmcnt variant of the boundary that follows this step (see conv_halo.hip)
  int32_t issue;    // phase whose halo is issued at that boundary, or -1
  int32_t pad[2];
};
static_assert(sizeof(HaloStep) == 32, "HaloStep layout");

constexpr int kHaloBufBytes = 48 * 1024;             // per halo buffer (two of them)
constexpr int kHaloRows = kHaloBufBytes / 64;        // 768 rows of 64 bytes
constexpr int kHaloLongInstr = kHaloRows / 16 / 4;   // 12 LDS-DMA instructions per wave stage a LONG halo
constexpr int kHaloShortInstr = 4;                   // 256 rows x 64 B

struct HaloArgs {
  HaloSrc t[kMaxConvTensors];
  const HaloStep* steps;    // device
  const HaloPhase* phases;  // device
  int nsteps, nphases;
  const void* w;      // packed [nsteps][Npad][64 B] (+ kWeightRowSlack rows)
  const float* bias;  // [Npad]
  void* out;          // [Do][Ho][Wo][Co]
  int Do, Ho, Wo, Co;
  int TZ, TY, TX;     // output box (TZ*TY*TX <= 256 rows of the M tile)
  int NBZ, NBY, NBX;  // boxes per axis
  int HZ, HY, HX;     // LONG halo extent (box + kernel - 1)
  int PZ, PY;         // LDS row pitches of the halo: row(jz,jy,jx) = jz*PZ + jy*PY + jx, HZ*PZ <= kHaloRows
  uint32_t mPZ, mPY, mTX, mTYX;  // ceil(2^32 / d) reciprocals for exact small-integer division
  int Npad;
  int relu;
};

// Pick the output box and the LDS pitches for a stage with kernel k and a WM x WN wave grid, or
// return false if no halo of at most kHaloRows rows exists (the gather kernel is used then).
// The pitches are chosen so that the 16 lanes of every ds_read_b128 lane group read halo rows
// that are distinct mod 16 (bank-conflict free with the (row>>2)&3 chunk swizzle) where possible.
bool halo_choose_geometry(int Do, int Ho, int Wo, const int k[3], int wm, int box[3], int pitch[2]);

int halo_ring_slots(TileCfg cfg);  // weight-ring depth used for this tile (LDS budget)
int launch_conv_halo(const HaloArgs& a, int precision, TileCfg cfg, hipStream_t stream);

}  // namespace bsmi
