// Implicit-GEMM 3-D convolution on MFMA (gfx950).  See conv_igemm.hip.
#pragma once
#include "common.h"

namespace bsmi {

// A channels-last activation tensor: [D][H][W][C], C = padded channel count
// (multiple of kChanPad), element type given by the launch precision.
struct ConvTensor {
  const void* ptr;
  int D, H, W, C;
};

// One K-step of the implicit GEMM = 4 "units" of 32 bytes of K (16 bf16 / 8 f32 channels
// each) taken from ONE source tensor.  Unit j of the row of output voxel (z,y,x) is read at
//   base + z*sz + y*sy + x*sx + delta[j]          (all in bytes)
// where delta[j] folds the unit's kernel tap, crop origin and channel offset (host side).
// A many-channel layer has delta = {c, c+32, c+64, c+96}; a 16-channel layer packs four
// kernel taps into one K-step.  Unused units point at delta 0 and meet all-zero weights.
struct KStep {
  uint64_t base;
  int32_t sz, sy, sx;
  int32_t delta[4];
  int32_t pad;
};
static_assert(sizeof(KStep) == 40, "KStep layout");

constexpr int kMaxConvTensors = 3;

struct ConvArgs {
  const KStep* steps;  // device
  int nsteps;
  const void* w;      // device, packed [nsteps][Npad][128 bytes]
  const float* bias;  // device [Npad]
  void* out;          // device [Do][Ho][Wo][Co]
  int Do, Ho, Wo, Co;
  int M;     // Do*Ho*Wo
  int Npad;  // multiple of the tile's BN
  int relu;
};

enum TileCfg { TILE_256x32 = 0, TILE_256x64, TILE_256x160, TILE_256x320, TILE_256x256, TILE_COUNT };

int tile_bm(TileCfg c);
int tile_bn(TileCfg c);
// pick the tile for `cout` real output channels
TileCfg choose_tile(int cout);

int launch_conv_igemm(const ConvArgs& a, int precision, TileCfg cfg, hipStream_t stream);

}  // namespace bsmi
