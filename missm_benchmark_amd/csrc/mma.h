// 16x16 MFMA building blocks shared by the GEMM and attention kernels (gfx950).
//
// One "step" multiplies a 16 x KS tile of A with a KS x 16 tile of B into a 16x16 fp32 accumulator:
//   bf16 : KS = 32, ONE  v_mfma_f32_16x16x32_bf16 ; lane (i = l&15, g = l>>4) holds 8 k-elements
//   f32  : KS = 16, FOUR v_mfma_f32_16x16x4_f32   ; lane holds 4 k-elements, element j feeds MFMA j
// In both cases a lane's fragment is 16 contiguous bytes: A[row i][k0 + g*KPL .. +KPL) and
// B[k0 + g*KPL ..][col i]  (KPL = 8 / 4).  Any permutation of k is legal as long as A and B agree,
// which is what makes the f32 form (MFMA j sums k = 4g'+j over g') equivalent.
//
// Accumulator (C/D) layout, both types: lane holds rows 4g + r (r = 0..3) of column i.
// "C-as-operand": a 16x16 accumulator tile X can feed the next product as the B operand summing over
// X's ROW index without any lane movement.  For f32 one tile is exactly one step (k = 4g + r).  For
// bf16 two stacked tiles (rows 0-15, 16-31) make one 32-deep step with the k order
//   element j<4 -> row 4g + j of tile 0 ;  element j>=4 -> row 4g + (j-4) of tile 1,
// and the A operand of that product is fetched in the same order (two 4-row transposed LDS reads).
#pragma once
#include "common.h"

namespace missm {

template <typename T> struct Mma;

template <> struct Mma<bf16> {
  using Frag = bf16x8;
  static constexpr int KS = 32;          // k elements per step
  static constexpr int KPL = 8;          // k elements per lane per step
  static constexpr int CTILES = 2;       // accumulator tiles forming one C-as-operand step
  __device__ static __forceinline__ f32x4 step(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ Frag zero() {
    Frag z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
    return z;
  }
  // pack accumulator tiles (t0 rows 0-15, t1 rows 16-31 of the k range) into a B/A operand
  __device__ static __forceinline__ Frag from_acc(f32x4 t0, f32x4 t1) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = (bf16)t0[j]; f[4 + j] = (bf16)t1[j]; }
    return f;
  }
};

template <> struct Mma<float> {
  using Frag = f32x4;
  static constexpr int KS = 16;
  static constexpr int KPL = 4;
  static constexpr int CTILES = 1;
  __device__ static __forceinline__ f32x4 step(Frag a, Frag b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
  }
  __device__ static __forceinline__ Frag zero() { Frag z = {0.f, 0.f, 0.f, 0.f}; return z; }
  __device__ static __forceinline__ Frag from_acc(f32x4 t0, f32x4 /*unused*/) { return t0; }
};

// 16-byte LDS fragment read (ds_read_b128) at a byte offset into a char LDS array
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag lds_frag(const char* lds, int byte_off) {
  return *reinterpret_cast<const typename Mma<T>::Frag*>(lds + byte_off);
}

// Transposed fragment: lane (i, g) receives M[k][c0 + i] for the C-as-operand k order of one step
// starting at row k0, from a row-major swizzled LDS tile with RB-byte rows.
//   bf16: two ds_read_b64_tr_b16 (rows k0+4g.., k0+16+4g..); every lane must be active (EXEC all ones).
//   f32 : four scalar reads of rows k0+4g+j.
template <typename T, int RB> struct TrFrag;

template <int RB> struct TrFrag<bf16, RB> {
  __device__ static __forceinline__ bf16x8 load(const char* lds, int k0, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    // within the 16-lane group, lane i = 4q+p supplies the address of row q, columns 4p..4p+3
    const int q = i >> 2, p = i & 3;
    const int r0 = k0 + 4 * g + q;
    const int cb = (c0 + 4 * p) * 2;  // byte offset inside the row
    const char* a0 = lds + swz<RB>(r0, cb);
    const char* a1 = lds + swz<RB>(r0 + 16, cb);
    i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(a0));
    i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(a1));
    using i16x8 = __attribute__((ext_vector_type(8))) short;
    i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
};

template <int RB> struct TrFrag<float, RB> {
  __device__ static __forceinline__ f32x4 load(const char* lds, int k0, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float*>(lds + swz<RB>(k0 + 4 * g + j, (c0 + i) * 4));
    return v;
  }
};

}  // namespace missm
