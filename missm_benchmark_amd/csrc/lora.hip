// LoRA adapters of the vision encoders (reference languagebind/image/modeling_image.py:775-793: get_peft_model(vision_model.encoder,
// LoraConfig(r, lora_alpha, target_modules = q/k/v/out_proj [time branch: temporal_attn.* and temporal_mlp.fc1/fc2]))).
//
// peft's forward of a wrapped linear is  y = x W^T + b + (alpha / r) * ((dropout(x) A^T) B^T);  with lora_dropout = 0 (the reference
// default, configuration_image.py:202) this is the linear with the weight  W' = W + (alpha / r) B A.  The towers therefore keep their
// GEMM path: the compute-dtype weight copies are cast from W' (missm_lora_merge below, once per optimizer step), dX uses W', and the
// full weight gradient G = dY^T X that the dW kernel produces anyway yields the adapter gradients
//     dB = (alpha / r) G A^T      dA = (alpha / r) B^T G
// (missm_lora_grad; r is 2 in the released configs: an [n x k] x [k x r] product is launch- and HBM-bound, plain fp32 VALU code).
// Only A and B are trained; the base weights stay frozen like in the reference.
#include "common.h"
#include "missm_internal.h"

namespace missm {

// W[i, j] += scale * sum_r B[i, r] * A[r, j]  - one workgroup per 16 rows, lanes walk the columns in float4 steps
__global__ __launch_bounds__(256) void lora_merge_kernel(float* __restrict__ W, int ldw, const float* __restrict__ A, const float* __restrict__ B,
                                                        int n_out, int k_in, int r, float scale) {
  const int row0 = blockIdx.x * 16;
  for (int e = threadIdx.x; e < 16 * (k_in / 4); e += 256) {
    const int i = row0 + e / (k_in / 4), j = (e % (k_in / 4)) * 4;
    if (i >= n_out) break;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < r; ++q) acc += B[(size_t)i * r + q] * load4(A + (size_t)q * k_in + j);
    float* w = W + (size_t)i * ldw + j;
    store4(w, load4(w) + scale * acc);
  }
}

// One workgroup per (64 rows of G, adapter rank index q):
//   dB[i, q] += scale * sum_j G[i, j] A[q, j]        (a wave per row, lanes over the columns; one owner per element: plain +=)
//   dA[q, j] += scale * sum_i B[i, q] G[i, j]        (partial over this workgroup's rows, fp32 atomics across the row chunks)
__global__ __launch_bounds__(256) void lora_grad_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ A,
                                                       const float* __restrict__ B, float* __restrict__ dA, float* __restrict__ dB, int n_out,
                                                       int k_in, int r, float scale) {
  const int q = blockIdx.y, row0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* a = A + (size_t)q * k_in;
  for (int i = row0 + wave; i < min(row0 + 64, n_out); i += 4) {
    const float* g = G + (size_t)i * ldg;
    float s = 0.f;
    for (int j = lane * 4; j < k_in; j += 256) {
      const f32x4 gv = load4(g + j), av = load4(a + j);
      s += gv[0] * av[0] + gv[1] * av[1] + gv[2] * av[2] + gv[3] * av[3];
    }
    s = wave_sum(s);
    if (lane == 0) dB[(size_t)i * r + q] += scale * s;
  }
  const int rows = min(64, n_out - row0);
  for (int j = threadIdx.x * 4; j < k_in; j += 1024) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ii = 0; ii < rows; ++ii) acc += B[(size_t)(row0 + ii) * r + q] * load4(G + (size_t)(row0 + ii) * ldg + j);
#pragma unroll
    for (int c = 0; c < 4; ++c) atomicAdd(dA + (size_t)q * k_in + j + c, scale * acc[c]);
  }
}

}  // namespace missm

using namespace missm;

extern "C" int missm_lora_merge(float* W, int ldw, const float* A, const float* B, int n_out, int k_in, int r, float scale, void* stream) {
  MISSM_CHECK_ARG(W && A && B && n_out > 0 && k_in > 0 && r > 0 && k_in % 4 == 0 && ldw % 4 == 0 && ldw >= k_in, "lora_merge: bad shape");
  MISSM_CHECK_ARG(((uintptr_t)W % 16 == 0) && ((uintptr_t)A % 16 == 0), "lora_merge: W and A must be 16-byte aligned");
  hipLaunchKernelGGL(lora_merge_kernel, dim3((n_out + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), W, ldw, A, B, n_out, k_in, r, scale);
  return missm_check_launch("lora_merge");
}

extern "C" int missm_lora_grad(const float* G, int ldg, const float* A, const float* B, float* dA, float* dB, int n_out, int k_in, int r,
                               float scale, void* stream) {
  MISSM_CHECK_ARG(G && A && B && dA && dB && n_out > 0 && k_in > 0 && r > 0 && k_in % 4 == 0 && ldg % 4 == 0 && ldg >= k_in, "lora_grad: bad shape");
  MISSM_CHECK_ARG(((uintptr_t)G % 16 == 0) && ((uintptr_t)A % 16 == 0), "lora_grad: G and A must be 16-byte aligned");
  hipLaunchKernelGGL(lora_grad_kernel, dim3((n_out + 63) / 64, r), dim3(256), 0, static_cast<hipStream_t>(stream), G, ldg, A, B, dA, dB, n_out, k_in,
                     r, scale);
  return missm_check_launch("lora_grad");
}
