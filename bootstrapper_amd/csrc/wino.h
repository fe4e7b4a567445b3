// Winograd F(2x2, 3x3) -- and, round 4, F(4x4, 3x3) for the widest stages -- in (y, x) for the 3x3x3 layers of the split-bf16
// mode whose K is large (wino.hip).  `m` = output tile edge (2 or 4), T = m + 2 = input tile edge, T * T batches.
#pragma once
#include <vector>

#include "conv_igemm.h"

namespace bsmi {

constexpr int kWinoBatch = 16;  // F(2x2): transform positions (xi, nu) of a 4 x 4 input tile, b = 4 * xi + nu
constexpr int kWinoMaxBatch = 36;  // F(4x4): 6 x 6 input tile, b = 6 * xi + nu
inline int wino_batches(int m) { return (m + 2) * (m + 2); }
constexpr int kWinoMaxSrc = 2;  // source tensors of the input transform (skip connection + upsampled map)

// Input transform: V[b][z][ty][tx][c] = (B^T d B)[xi][nu] of the 4 x 4 in-plane tile d whose first voxel is
// (z, 2 ty, 2 tx) of the layer's (cropped, concatenated) input; every source's channels land at its own channel offset.
struct WinoInArgs {
  const void* src[kWinoMaxSrc];  // split-bf16 tensors [D][H][W][Cpad]
  int H[kWinoMaxSrc], W[kWinoMaxSrc], Cpad[kWinoMaxSrc];
  int oz[kWinoMaxSrc], oy[kWinoMaxSrc], ox[kWinoMaxSrc];  // origin of the layer's input inside the source (the skip connection's crop)
  int cv0[kWinoMaxSrc];                                   // first channel of the source inside V
  // upf[q] = f > 0: source q is read through an in-plane linear upsampling by f (torch's Upsample(scale (1, f, f), "trilinear",
  // align_corners = False), reference unet.py:143): the layer's input is upsample(src)[oz + z][oy + y][ox + x] and H, W are
  // the dimensions of the LOW-resolution tensor -- the upsampled map is never written (round 3: the 1500-channel map of the
  // 128^3 block is 1.3 GB that the stage's transform was the only reader of besides a residual branch, see WinoOutArgs)
  int upf[kWinoMaxSrc];
  int nsrc;
  void* V;  // split-bf16 [T * T][Dv][Ty][Tx][Cv]
  int Dv, Ty, Tx, Cv;
  int m;    // 2 (0 reads as 2) or 4: tile (ty, tx) starts at voxel (m ty, m tx); F(4x4) tiles may overhang the layer's input (the
            // last tile row / column of an extent that is no multiple of 4): reads are clamped to the source, the outputs
            // that depend on them are never stored
};

// Output transform: out[z][2 ty + p][2 tx + q][c] = relu( (A^T M A)[p][q] + bias[c] + addend[...] ) as (hi, lo) pairs, where
// M[b] = row (z * Ty + ty) * Tx + tx of batch b of the GEMM's f32 sums.
struct WinoOutArgs {
  const float* M;       // [16][Do * Ty * Tx][Co]
  const float* addend;  // optional raw sums of the cropped 1x1x1 residual branch, [Do * Ho * Wo][Co]
  // optional: the part of the residual branch that reads an upsampled map, computed BELOW the upsampling (a 1x1x1 convolution
  // commutes with the per-channel interpolation): raw sums [lD * lH * lW][Co] at low resolution, added as
  // upsample_lf(low)[loz + z][loy + y][lox + x]
  const float* low;
  int lD, lH, lW, lf, loz, loy, lox;
  const float* bias;    // [>= Co]
  void* out;            // split-bf16 [Do][Ho][Wo][Co]
  int Do, Ty, Tx, Co;
  int relu;
  int m, Ho, Wo;        // m = 2 (0 reads as 2): Ho = 2 Ty, Wo = 2 Tx; m = 4: Ho <= 4 Ty, Wo <= 4 Tx (overhanging tiles are cut)
};

int launch_wino_in(const WinoInArgs& a, hipStream_t s);
int launch_wino_out(const WinoOutArgs& a, hipStream_t s);

// One 32-byte unit of K of the transform-domain GEMMs: 16 channels of V at z tap kz (dummy: zero weights, delta 0).
struct WinoUnit {
  int kz, vc0;
  bool dummy;
};
// K-step list shared by the 16 batches: 32-channel chunks of V, the three z taps inside a chunk; even number of K-steps.
void wino_units(int Cv, std::vector<WinoUnit>& out);

// Host side: transformed weights U[b][kz][c][n] = sum_{ky,kx} G[xi][ky] G[nu][kx] w[n][cin(c)][kz][ky][kx], packed as the GEMM's
// B operand [b][K-step][Npad][64 B], hi image then lo image (each `image_elems` bf16 values, slack rows included).
// `cin_of_v[c]`: input channel of V channel c, or -1 for a pad channel.  `w`: OIDHW f32 with `cin` input channels.
void wino_pack_weights(const float* w, int cout, int cin, const std::vector<int>& cin_of_v, int Npad, const std::vector<WinoUnit>& units,
                       std::vector<uint16_t>& packed, size_t& image_elems, size_t& batch_elems, int m = 2);

// The same images made on the device from the stage's raw weights (uploaded for the call): *image_dev = hi image then lo image
// (hipMalloc'ed, the caller frees it); bit for bit what wino_pack_weights makes.
int wino_pack_weights_dev(const float* w, int cout, int cin, const std::vector<int>& cin_of_v, int Npad, const std::vector<WinoUnit>& units, int m,
                          void** image_dev, size_t& image_elems, size_t& batch_elems);

}  // namespace bsmi
