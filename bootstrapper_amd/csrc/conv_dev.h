// Device-side pieces shared by the convolution kernels (conv_igemm.hip, conv_rh.hip).
#pragma once
#include "conv_igemm.h"

namespace bsmi {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// chunk-swizzle key of the 16 x 16 x 32 fragment reads: a ds_read_b128 lane group covers rows 0-3 and
// 12-15 at one 16-byte chunk and rows 4-11 at the next; XOR keys {0, 2, 3, 1} per 4-row group keep
// the 16 lanes of every group on 16 different bank quads
__device__ __forceinline__ int swz16(int g) { return (0x78 >> (2 * g)) & 3; }

struct bf16_elem {
  uint16_t v;
};

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) {
    // 4 x v_mfma_f32_32x32x2_f32: lane half h holds k = 4h..4h+3 of this 8-wide sub-step;
    // instruction t contracts k in {t, 4+t}.  Same permutation on A and B.
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    return acc;
  }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_elem> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void store(bf16_elem* p, float v) {
    __bf16 h = (__bf16)v;
    p->v = __builtin_bit_cast(uint16_t, h);
  }
};

// Element of a split-bf16 tensor (BSMI_PREC_BF16X3): the kernels multiply bf16 planes exactly as in the bf16 mode --
// the host lists every logical K-step three times (hi x hi, lo x hi, hi x lo: unet_api.hip) -- and the epilogue
// writes the f32 result as a (hi, lo) pair of planes: hi = bf16(v), lo = bf16(v - hi).
struct bf16s_elem {
  uint16_t v;
};
template <>
struct Elem<bf16s_elem> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) { return Elem<bf16_elem>::mfma(a, b, acc); }
  static __device__ __forceinline__ void store(bf16s_elem* p, float v) {
    __bf16 h = (__bf16)v;
    p->v = __builtin_bit_cast(uint16_t, h);
  }
};
// The same tensors and results, but the kernel itself walks LOGICAL K-steps: it stages the hi and lo planes of both
// operands once per K-step and multiplies hi x hi, lo x hi and hi x lo from them (conv_x3_body): two thirds of the
// LDS-DMA traffic of the K-step-list form for the same MFMAs.
struct bf16f_elem {
  uint16_t v;
};
template <>
struct Elem<bf16f_elem> {
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t acc) { return Elem<bf16_elem>::mfma(a, b, acc); }
  static __device__ __forceinline__ void store(bf16f_elem* p, float v) {
    __bf16 h = (__bf16)v;
    p->v = __builtin_bit_cast(uint16_t, h);
  }
};
template <typename T>
struct IsFused { static constexpr bool value = false; };
template <>
struct IsFused<bf16f_elem> { static constexpr bool value = true; };
template <typename T>
struct IsSplit { static constexpr bool value = false; };
template <>
struct IsSplit<bf16f_elem> { static constexpr bool value = true; };
template <>
struct IsSplit<bf16s_elem> { static constexpr bool value = true; };
// Activation layout of the split mode: every 16-byte vector of 8 hi values is followed by the 16-byte vector of their 8
// lo values, so the 32 channels of a K-step are ONE 128-byte line per voxel row (hi and lo in separate planes cost a
// half-used line each per K-step: the LDS-DMA gather missed the vector L1 on every row and pulled twice the bytes
// out of L2).  Element index of channel n in the row that starts at element `row` (= voxel * Cpad) of a plain tensor:
template <typename T>
__device__ __forceinline__ size_t act_index(size_t row, int n) {
  if constexpr (IsSplit<T>::value) return 2 * row + (size_t)(((n >> 3) << 4) + (n & 7));
  else return row + (size_t)n;
}
constexpr int kSplitLoElems = 8;  // the lo vector follows its hi vector
// the lo value: what the bf16 rounding of v (the hi value) left over
__device__ __forceinline__ float split_lo(float v) {
  const __bf16 h = (__bf16)v;
  return v - (float)h;
}

// 16-byte store of finished output rows.  Streaming (nt): the 0.1-0.2 GB a layer writes should not
// compete for L2 with the panels the K-loops of the other workgroups are re-reading; measured
// on the 128^3 block, the two largest layers 5.39 -> 5.15 ms and 4.82 -> 4.63 ms, sc1 and sc0 sc1 the
// same within 1 %.  BSMI_STORE_POLICY (dev builds): 0 plain, 1 sc1, 2 nt, 3 sc0 sc1.
#ifndef BSMI_STORE_POLICY
#define BSMI_STORE_POLICY 2
#endif
// The inline-asm forms end in s_nop 1: the compiler does not see an asm statement as a VMEM store, so it does not keep
// the two wait states gfx950 needs between a store of more than 64 bits and a VALU write of its data registers
// (observed: the address arithmetic of the next store landed in the upper half of the previous store's data).
__device__ __forceinline__ void store_stream16(void* p, u32x4_t v) {
#if BSMI_STORE_POLICY == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif BSMI_STORE_POLICY == 2
  __builtin_nontemporal_store(v, (u32x4_t*)p);  // global_store_dwordx4 ... nt
#elif BSMI_STORE_POLICY == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *(u32x4_t*)p = v;
#endif
}

typedef const __attribute__((address_space(1))) char* gptr_t;
typedef __attribute__((address_space(3))) char* lptr_t;
typedef const __attribute__((address_space(4))) int32_t* cint_ptr_t;  // constant AS: scalar loads

}  // namespace bsmi
