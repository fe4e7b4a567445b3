// Launchers of the memory-bound U-Net operators (unet_ops.hip).
#pragma once
#include "common.h"

namespace bsmi {

// BSMI_PREC_BF16X3 tensors hold (hi, lo) 16-byte vectors interleaved (conv_dev.h act_index)
int launch_input_prep(int precision, const void* raw, int raw_dtype, void* out, int cin, int cpad,
                      size_t nvox, hipStream_t s);
int launch_maxpool(int precision, const void* in, void* out, int D, int H, int W, int C, int fz,
                   int fy, int fx, hipStream_t s);
int launch_upsample_crop(int precision, const void* in, void* out, int D, int H, int W, int C, int Do,
                         int Ho, int Wo, int fz, int fy, int fx, int oz, int oy, int ox, hipStream_t s);
int launch_head(int precision, const void* z, int cpad, int cin, int cout, const float* hw,
                const float* hb, float* out_f32, uint8_t* out_u8, size_t nvox, hipStream_t s);
int launch_extract_block_reflect(const uint8_t* vol, const int64_t vs[3], const int64_t off[3],
                                 const int64_t bs3[3], uint8_t* block, hipStream_t s);

}  // namespace bsmi
