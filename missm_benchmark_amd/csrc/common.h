// Shared device helpers for the MI355X (gfx950 / CDNA4) kernels of the MissM hot path.
// Wavefront = 64 lanes everywhere; LDS rows are XOR-swizzled at 16-byte granularity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace missm {

typedef __bf16 bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int kWave = 64;

enum DType : int { kF32 = 0, kBF16 = 1 };

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> { static constexpr int code = kF32; };
template <> struct TypeInfo<bf16> { static constexpr int code = kBF16; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// ---- wave-level reductions (64 lanes) ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- 16-byte vector load/store of 4 floats or 8 bf16, converted through fp32 -------------
template <typename T> struct Vec16;  // elements per 16 bytes
template <> struct Vec16<float> { static constexpr int n = 4; };
template <> struct Vec16<bf16> { static constexpr int n = 8; };

// load 4 consecutive elements as fp32
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
  bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = r;
}

// streaming (non-temporal) forms: for bytes this kernel touches once and nobody reads again soon
__device__ __forceinline__ f32x4 load4_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void store4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

// ---- swizzled LDS addressing --------------------------------------------------------------
// A tile row holds RB bytes (64, 128 or 256); 16-byte chunk c of row r lives at chunk
// c ^ (r & min(RB/16-1, 7)).  For RB == 128 this makes both the ds_read_b128 fragment reads
// (16 rows x one chunk per 16-lane group) and the ds_read_b64_tr_b16 reads (8 rows x 32 B per
// 32-lane half) bank-conflict free (bank = (addr/4) % 64).
template <int RB> __device__ __forceinline__ int swz(int row, int byte_in_row) {
  constexpr int NC = RB / 16;
  constexpr int M = (NC - 1) & 7;
  return row * RB + ((((byte_in_row >> 4) ^ (row & M))) << 4) + (byte_in_row & 15);
}

// quick_gelu(x) = x * sigmoid(1.702 x) and its derivative
// sigmoid through ONE v_exp_f32 and ONE v_rcp_f32 (1 ulp each): an IEEE `1.0f / y` is a 10-instruction div_scale / rcp / fma /
// div_fmas / div_fixup sequence, which made the activation epilogues of the 256x256 GEMM tile VALU-bound (128 values per lane)
__device__ __forceinline__ float sigmoid_1702(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.4426950408889634f)));
}
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoid_1702(x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
  const float s = sigmoid_1702(x);
  return s + 1.702f * x * s * (1.0f - s);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// XCD-aware, bijective remap of a linear workgroup id: workgroups are dealt round-robin over the
// 8 XCDs (id % 8 labels the XCD group), so give each group a contiguous range of logical ids;
// neighbouring logical tiles (which share an operand panel) then share one XCD's L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, local = id >> 3;
  const int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + local;
}

}  // namespace missm
