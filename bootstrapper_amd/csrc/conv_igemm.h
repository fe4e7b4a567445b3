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

// One K-step of the implicit GEMM = 2 "units" of 32 bytes of K (16 bf16 / 8 f32 channels
// each) taken from ONE source tensor (`tensor` indexes ConvArgs::t).  Unit j of the row of
// output voxel (z,y,x) is read at
//   t.base + z*t.sz + y*t.sy + x*t.sx + delta[j]          (all in bytes)
// where delta[j] folds the unit's kernel tap, crop origin and channel offset (host side).
// A many-channel layer walks its channels 32 at a time; a 16-channel layer packs two kernel
// taps into one K-step.  An unused unit points at delta 0 and meets all-zero weights.
struct KStep {
  int32_t tensor;
  int32_t delta[2];
  int32_t pad;
};
static_assert(sizeof(KStep) == 16, "KStep layout");

struct ConvSrc {
  uint64_t base;       // device pointer of the channels-last source tensor
  int32_t sz, sy, sx;  // byte strides of one step in z, y, x
  int32_t pad;
};

constexpr int kUnitsPerStep = 2;
constexpr int kStepRowBytes = 64;   // bytes of K per tile row per K-step
constexpr int kWeightRowSlack = 64; // weight rows readable past Npad (tile loads are padded to 64 rows)

constexpr int kMaxConvTensors = 3;

struct ConvArgs {
  ConvSrc t[kMaxConvTensors];
  const KStep* steps;  // device
  int nsteps;
  const void* w;      // device, packed [nsteps][Npad][64 bytes] (+ kWeightRowSlack rows)
  const void* w_lo;   // fused split-bf16 launches: the lo image of the weights, same layout (w is the hi image)
  const float* bias;  // device [Npad]
  void* out;          // device [Do][Ho][Wo][Co]
  int64_t pad;        // (BSMI_PREC_BF16X3 tensors carry (hi, lo) vectors interleaved: conv_dev.h act_index)
  int Do, Ho, Wo, Co;
  int M;     // Do*Ho*Wo
  int Npad;  // multiple of the tile's BN
  int relu;
  // Batched launches (wino.hip: the 16 transform-domain GEMMs of a Winograd layer), fused split-bf16 kernels only.  A tile
  // index decodes to (batch, tile of that batch); batch b reads its sources at base + b * a_batch bytes and its weights at
  // w + b * w_batch bytes (both images) and writes rows [b * M, (b + 1) * M) of `out`.
  int nbatch;  // 0 or 1: one GEMM
  int raw;     // 1: the epilogue stores the f32 sums as they are -- no bias, no ReLU, no (hi, lo) split: out = float [rows][Co]
  int64_t a_batch, w_batch;
};

enum TileCfg { TILE_256x32 = 0, TILE_256x64, TILE_256x160, TILE_256x320, TILE_256x256, TILE_COUNT };

int tile_bm(TileCfg c);
int tile_bn(TileCfg c);
// pick the tile for `cout` real output channels
TileCfg choose_tile(int cout);
// 8-wave kernels enabled (BSMI_WAVES8 != 0)
bool two_waves_per_simd();

// Split-K tail workspace: one partial tile of at most kStreamKTileElems floats per persistent workgroup.
constexpr size_t kStreamKTileElems = 256 * 320;
// followed by the 8 per-XCD work-queue counters (zero between launches)
inline size_t stream_k_ws_bytes(int sk_grid) { return (size_t)sk_grid * kStreamKTileElems * sizeof(float) + 64; }

// sk_ws / sk_grid: optional split-K-tail workspace and persistent grid (a multiple of 8, normally the CU
// count); with them, big-tile launches whose tile count is not a multiple of the grid run persistently.
int launch_conv_igemm(const ConvArgs& a, int precision, TileCfg cfg, hipStream_t stream, float* sk_ws = nullptr,
                      int sk_grid = 0);

}  // namespace bsmi
