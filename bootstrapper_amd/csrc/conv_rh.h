// Raster-halo implicit-GEMM 3-D convolution (gfx950).  See conv_rh.hip.
#pragma once
#include "conv_igemm.h"

namespace bsmi {

// A phase = the activation rows one (source tensor, 32-channel chunk, z-tap) contributes: the
// M tile's rows of the conv-input raster plus the (ky-1)*Win + (kx-1) rows that follow them, 64
// bytes each, staged ONCE into an LDS halo buffer and read by all ky*kx in-plane taps at a row
// offset.  `delta` (bytes) folds the z-tap plane, the crop origin of the tensor and the chunk.
struct RhPhase {
  int32_t tensor;
  int32_t delta;
  int32_t buf;         // halo buffer 0 / 1
  int32_t issue_step;  // K-step whose staging slot carries this halo (-1: the prologue)
};
static_assert(sizeof(RhPhase) == 16, "RhPhase layout");

// One K-step = one in-plane tap x 32 channels (or two x-adjacent taps x 16 channels of a
// 16-channel tensor): A fragments from halo row (tile row + rowoff) of buffer `buf`, weights
// from the ring.  `issue` = phase whose halo is staged together with the weights of K-step
// h + 4 (or -1); `wait` = a * 3 + b: a weight groups and b halos may still be in flight at the
// barrier of this K-step (host-simulated in-order vmcnt queue).
struct RhStep {
  int32_t rowoff;
  int32_t buf_phase;  // buf | phase << 8
  int32_t wait;
  int32_t issue;
};
static_assert(sizeof(RhStep) == 16, "RhStep layout");

struct RhArgs {
  ConvSrc t[kMaxConvTensors];
  const RhStep* steps;    // device
  const RhPhase* phases;  // device
  int nsteps, nphases;
  const void* w;      // packed [nsteps][Npad][64 B]
  const float* bias;  // [Npad]
  void* out;          // [Do][Ho][Wo][Co]
  int Do, Ho, Wo, Co;
  int Hin, Win;  // in-plane extent of the conv-input raster (Ho + ky - 1, Wo + kx - 1)
  int Q;         // Do * Hin * Win rows; M tiles = ceil(Q / 256)
  int Npad;
  int relu;
};

// halo pieces (16 rows x 64 B) per wave of the 8-wave kernels, by tile
int rh_halo_pieces(TileCfg cfg);
// true if the tile's halo buffer holds 256 + (ky-1)*Win + (kx-1) rows
bool rh_supported(TileCfg cfg, int Win, int ky, int kx);

int launch_conv_rh(const RhArgs& a, int precision, TileCfg cfg, hipStream_t stream, float* sk_ws = nullptr, int sk_grid = 0);

// (A fused split-bf16 form of this kernel, conv_rh_x3_kernel -- 32 x 64 wave tiles, one barrier per K-step of 24 MFMAs -- existed in
// rounds 2 and 3 as an opt-in and lost to the gather kernel; conv_h16.hip is the halo form of the split mode.)

}  // namespace bsmi
