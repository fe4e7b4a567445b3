// The first ConvPass of a raw-input U-Net as ONE launch (first_pass.hip): input normalisation,
// conv 3x3x3 (1 -> C), ReLU, conv 3x3x3 (C -> C) + cropped 1x1x1 residual, ReLU, C <= 16, bf16 and split-bf16 modes.
#pragma once
#include "common.h"

namespace bsmi {

struct FirstPassWeights {   // device images, packed by pack_first_pass
  uint32_t* w1a = nullptr;  // [64 lanes][4]: A fragment (rows = output channels) of the 27-tap (K = 32) conv
  uint32_t* w2a = nullptr;  // [14 steps][64 lanes][4]: A fragments of the second conv, K-step = 2 taps x 16 channels
  float* vec = nullptr;     // [3][16]: bias of conv 1; bias of conv 2 + residual bias; residual weights
  bool ready = false;
  bool split = false;       // BSMI_PREC_BF16X3: lo images behind the hi images, f32 residual weights
};

struct FirstPassArgs {
  const void* raw;  // [D][H][W] u8 or f32
  int raw_dtype;    // BSMI_RAW_*
  int D, H, W;
  uint16_t* out;    // [D-4][H-4][W-4][16] bf16 (split: [..][32], (hi, lo) vectors of 8 interleaved)
  bool split;
  const uint32_t* w1a;
  const uint32_t* w2a;
  const float* vec;
};

// w1: [C][1][3][3][3], w2: [C][C][3][3][3], wres: [C][1] (host, f32); C <= 16
int pack_first_pass(FirstPassWeights& fw, int C, const float* w1, const float* b1, const float* w2, const float* b2,
                    const float* wres, const float* bres, bool split = false);
void free_first_pass(FirstPassWeights& fw);
int launch_first_pass(const FirstPassArgs& a, int n_cus, hipStream_t s);

}  // namespace bsmi
