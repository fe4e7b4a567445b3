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

// A phase = one 16-channel (LONG: all kernel taps, 32-byte halo rows) or 32-channel (SHORT:
// one tap, 64-byte rows, used for the 1x1x1 residual) slice of one source tensor, staged once
// into an LDS halo buffer and consumed by `nsteps` K-steps.
struct HaloPhase {
  int32_t tensor;
  int32_t c0;       // channel offset in bytes
  int32_t kind;     // 0 LONG, 1 SHORT
  int32_t bufbase;  // LDS byte offset of the halo buffer this phase is staged into
};
static_assert(sizeof(HaloPhase) == 16, "HaloPhase layout");

// One K-step = 2 units of 32 bytes of K.  LONG: two kernel taps of the phase's 16 channels;
// SHORT: the two 16-channel halves of the phase's 32 channels.
struct HaloStep {
  int32_t trow[2];  // halo row offset of each unit's tap
  int32_t fmt;      // cb0 | cb1 << 8 | rsh << 16 | ksh << 20 | kmask << 24
  int32_t bufbase;  // LDS byte offset of the halo buffer read by this step
  int32_t wait;     // vmcnt variant of the boundary that follows this step (see conv_halo.hip)
  int32_t issue;    // phase whose halo is issued at that boundary, or -1
  int32_t pad[2];
};
static_assert(sizeof(HaloStep) == 32, "HaloStep layout");

constexpr int kHaloBufBytes = 32 * 1024;  // per halo buffer (two of them)
constexpr int kHaloLongRows = 1024;       // 32-byte rows
constexpr int kHaloShortRows = 256;       // 64-byte rows (= rows of the output box)
constexpr int kHaloLongInstr = 8;         // LDS-DMA instructions per wave to stage a LONG halo (1024 rows)
constexpr int kHaloShortInstr = 4;        // ... a SHORT halo (256 rows x 64 B)

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
  int HY, HX;         // LONG halo extent in y, x (box + kernel - 1); rows = z-major
  int hv_long;        // rows of the LONG halo actually needed
  int Npad;
  int relu;
};

// can this stage run on the halo kernel?  (all main kernels <= 3 per axis handled; otherwise the
// generic kernel of conv_igemm.hip is used)
bool halo_choose_box(int Do, int Ho, int Wo, const int k[3], int box[3]);

int launch_conv_halo(const HaloArgs& a, int precision, TileCfg cfg, hipStream_t stream);

}  // namespace bsmi
