// Box-halo 3x3x3 convolution for layers with at most 16 output channels (conv_box.hip).
#pragma once
#include "common.h"

namespace bsmi {

// One K chunk: 16 channels of one source tensor with all 27 taps (FULL), or 32 channels of the centre tap
// only (CENTER: the cropped 1x1x1 residual branch).  `base` folds the tensor's crop origin and channel offset.
struct BoxChunk {
  uint64_t base;
  int32_t sz, sy, sx;  // byte strides of the source tensor
  int32_t D, H, W;     // voxels available from `base` on (staging addresses are clamped to them)
};
static_assert(sizeof(BoxChunk) == 32, "BoxChunk layout");

struct BoxArgs {
  const BoxChunk* chunks;  // device: n_full FULL chunks, then n_center CENTER chunks
  int n_full, n_center;
  const uint32_t* w;       // device: A fragments [FULL chunk][14 steps][64 lanes][4], then [CENTER chunk][64 lanes][4]
  const float* bias;       // [16]
  uint16_t* out;           // [Do][Ho][Wo][Co] bf16
  int Do, Ho, Wo, Co;
};

inline bool box_supported(int cout) { return cout <= 16; }
int launch_conv_box(const BoxArgs& a, hipStream_t s);

}  // namespace bsmi
