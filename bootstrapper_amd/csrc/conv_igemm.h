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

// One K-step of the implicit GEMM: `nsub` 32-byte sub-steps (16 bf16 / 8 f32
// channels each) of tensor `tensor`, starting `a_off` elements after the row's
// base voxel (tap offset + crop origin + channel-chunk start, folded on the host).
struct KStep {
  int32_t tensor;
  int32_t a_off;
  int32_t nsub;
  int32_t pad;
};

constexpr int kMaxConvTensors = 3;

struct ConvArgs {
  ConvTensor t[kMaxConvTensors];
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
