// Halo-resident implicit-GEMM convolution for the stages with few output channels (split-bf16 mode).  See conv_h16.hip.
#pragma once
#include "conv_igemm.h"

namespace bsmi {

// A phase = the rows one (source tensor, 16-channel chunk, z tap) contributes: the tile's 512 rows of the conv-input raster and
// the (ky - 1) * Win + (kx - 1) rows behind them, 64 bytes each in the tensor ((hi, lo) vectors of 8 channels interleaved),
// staged ONCE into two LDS planes (hi, lo: 32 bytes per row) and read by all in-plane taps at a row offset.
struct H16Phase {
  int32_t tensor;      // index into H16Args::t
  int32_t delta;       // bytes: z tap, crop origin and channel chunk folded together
  int32_t rows;        // halo rows its taps reach (a multiple of 16 is staged): 512 + (ky - 1) * Win + (kx - 1), 512 for a single tap
  int32_t nsteps;      // its K-steps follow those of the phase before
};
static_assert(sizeof(H16Phase) == 16, "H16Phase layout");

// A K-step = TWO in-plane taps x 16 channels (K = 32 of v_mfma_f32_16x16x32_bf16: k 0-15 tap a, k 16-31 tap b), read from the
// phase's halo at row offsets offa / offb = dy * Win + dx.  A phase with an odd number of taps ends in a K-step whose second
// half meets zero weights (offb = offa).
struct H16Step {
  int32_t offa, offb, pad0, pad1;
};
static_assert(sizeof(H16Step) == 16, "H16Step layout");

struct H16Args {
  ConvSrc t[kMaxConvTensors];
  const H16Phase* phases;  // device
  const H16Step* steps;    // device
  int nphases, nsteps;
  const void* w;     // hi weight image [nsteps][Npad][64 B] (rows of 32 bf16: tap a's 16 channels, tap b's), + kWeightRowSlack rows
  const void* w_lo;  // lo image, same layout
  const float* bias; // [Npad]
  void* out;         // (hi, lo)-interleaved [Do][Ho][Wo][Co]
  int Do, Ho, Wo, Co;
  int Hin, Win;      // in-plane extent of the conv-input raster (Ho + ky - 1, Wo + kx - 1)
  int Q;             // Do * Hin * Win raster rows; tiles = ceil(Q / 512)
  int Npad;          // 16 or 64
  int relu;
  int ntiles, n_big, r_small;  // h16_tiling: tiles 0 .. n_big - 1 have 512 rows, the others r_small x 64
};

constexpr int kH16TileRows = 512;
// rows of the halo buffer a launch needs (768 or 1024) and the row blocks per wave its tiles may have (8, or 7 where that lets
// the smaller buffer do); 0 if the layer does not fit the kernel
int h16_halo_rows(int Win, int ky, int kx, int npad, int* max_r);
// tiling of Q raster rows for a device of n_cus CUs: whole rounds of the resident workgroups where that costs little
void h16_tiling(int64_t Q, int npad, int halo_rows, int max_r, int n_cus, int* ntiles, int* n_big, int* r_small);
int launch_conv_h16(const H16Args& a, int halo_rows, hipStream_t stream);

}  // namespace bsmi
