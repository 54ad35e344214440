// NT GEMM with fused epilogues for the tower linears:  C[M,N] = alpha * A[M,K] . B[N,K]^T  (+bias, act, residual)
//
// Replaces the vendor GEMMs PyTorch dispatches for nn.Linear / conv patch-embed on the reference hot path
// (CLIPAttention q/k/v/out_proj, CLIPMLP fc1/fc2: languagebind/image/modeling_image.py:69,71;
//  patch_embedding: languagebind/video/modeling_video.py:29-35).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles),
// K step = 128 bytes per row (64 bf16 / 32 f32), double-buffered LDS (64 KiB -> 2 workgroups per CU),
// register-staged prefetch of the next K tile issued before the MFMA block (loads in flight under compute),
// XOR-swizzled 16-byte chunks (conflict-free ds_read_b128), XCD-aware tile order.
// Rows of B are placed in LDS so that a lane owns 4 CONSECUTIVE output columns: global stores, bias and
// residual accesses are 8/16-byte vectors forming 128/256-byte contiguous runs per row.
#include "common.h"
#include "mma.h"
#include "missm_internal.h"
#include <stdlib.h>
#include <math.h>
#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace missm {

constexpr int BM = 128, BN = 128, RB = 128;  // tile rows / cols / bytes per LDS row (k-contiguous operands)
constexpr int GEMM_THREADS = 256;
#define MISSM_MAX_GROUPS 8

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K, lda, ldb, ldc;
  float alpha;
  const float* bias;      // [N] or null
  const float* resid;     // fp32 [M,N] (ld = ldc) or null ; only with out_f32
  const void* aux_in;     // T [M,N] (ld = ldaux): pre-activation for act = ACT_D*
  void* aux_out;          // T [M,N] (ld = ldaux): pre-activation saved by act = ACT_QGELU/GELU
  int ldaux;
  int act;                // MISSM_ACT_*
  int out_f32;            // C is fp32 (else T)
  int accumulate;         // out_f32 only: C += result
  int tiles_m, tiles_n;
  int vec_ok;             // ldc/ldaux/pointers allow 16-byte (fp32) / 8-byte (bf16) vector epilogue accesses
  int splitk, k_per_split;  // splitk > 1: K slices; partial tiles go to `ws`, splitk_reduce_kernel sums them into C
  float* ws;                // split-K workspace: [splitk][tiles][16][256] float4 partial accumulators (register order)
  unsigned long long* dbg;  // diagnostic builds only: per-workgroup {start, loop start, loop end, end, hw id} stamps (100 MHz clock)
  int group_m;              // tile order: groups of group_m tile rows are swept column by column (L2 locality)
  float* colsum_a;          // TA only: colsum_a[m] += sum_k A[k][m] (bias gradient riding in the dW GEMM as a ones-column)
  int desync_phases, desync_step;   // experiment (MISSM_GEMM_DESYNC=<phases>,<ticks of 10 ns>): staggered first tiles of the persistent grid
  unsigned* sched;          // persistent 8-phase NT grid: eight per-XCD tile-queue counters (device, zero between launches); null: static order
  // Grouped launch (8-phase kernels only): `ngroups` problems of ONE shape - the same linear of several shape-identical towers -
  // share a grid, so that B x 197-row towers fill the chip like one long tower does.  NT: tile rows [gi * group_tiles_m, ...)
  // belong to group gi; TN: workgroups [gi * tiles * splitk, ...).  A group's pointers replace the ones above.
  int ngroups, group_tiles_m;
  // TN (weight-gradient) workgroup order.  1: logical id = xcd_remap(blockIdx.x, whole grid) walks (group, K slice, tile row, tile column)
  // with the column fastest, so the ~32 workgroups resident on one XCD are (almost) all tiles of ONE K slice: the slice's rows of dY and X
  // are fetched into that XCD's L2 once instead of by every XCD (measured round 3, [2304 x 768] over 50432 rows: 968 MB fetched per launch
  // against 310 MB of operands with the old order, which dealt the 27 tiles of a slice round-robin over the 8 XCDs).  0: the old order.
  int tn_order;
  struct Group { const void* A; const void* B; void* C; const float* bias; const float* resid; const void* aux_in; void* aux_out; float* colsum_a; };
  Group grp[MISSM_MAX_GROUPS];
};
__device__ __forceinline__ void select_group(GemmArgs& g, const GemmArgs& all, int gi) {
  const GemmArgs::Group& p = all.grp[gi];
  g.A = p.A; g.B = p.B; g.C = p.C; g.bias = p.bias; g.resid = p.resid; g.aux_in = p.aux_in; g.aux_out = p.aux_out; g.colsum_a = p.colsum_a;
}

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];   // source of zero-filled LDS chunks

// swizzle key of a k-major ("transposed") tile row: the 8 rows touched by one half-wave of a transposed fragment read
// ({8g+q} and {8g+8+q}, q = 0..3) get 8 distinct keys -> 8 distinct 32-byte slots of the 256-byte bank row.
__device__ __forceinline__ int tkey(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
template <int RBT> __device__ __forceinline__ int tswz(int row, int chunk16) { return row * RBT + ((chunk16 ^ (tkey(row) << 1)) << 4); }

// The same read issued through inline assembly: the compiler then neither knows it is an LDS read (it would otherwise put
// s_waitcnt vmcnt(0) in front of it whenever an LDS-DMA is in flight - it cannot prove the intrinsic does not alias the DMA's
// LDS writes) nor waits for its result: frag_fence() must stand between these reads and their first use.
__device__ __forceinline__ bf16x8 kmajor_load_untracked(const char* a0, const char* a1) {
  i16x4 lo, hi;
  const unsigned p0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)a0;
  const unsigned p1 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)a1;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(p0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(p1));
  using i16x8 = __attribute__((ext_vector_type(8))) short;
  i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
template <typename F> __device__ __forceinline__ void frag_fence(F (&a)[4], F (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}

// fragment (one MFMA step) of a k-major tile [BK rows of k][COLS columns]: lane (i, g) gets column c0 + i, its KPL k values
template <typename T, int COLS> struct KMajorFrag;
template <int COLS> struct KMajorFrag<bf16, COLS> {
  static constexpr int RBT = COLS * 2;
  __device__ static __forceinline__ bf16x8 load(const char* lds, int ks, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
    const int r1 = ks * 32 + 8 * g + q, col = c0 + 4 * p;
    const char* a0 = lds + tswz<RBT>(r1, col >> 3) + ((col & 7) << 1);
    const char* a1 = lds + tswz<RBT>(r1 + 4, col >> 3) + ((col & 7) << 1);
    i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(a0));
    i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(a1));
    using i16x8 = __attribute__((ext_vector_type(8))) short;
    i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
  static constexpr bool has_untracked = true;
  __device__ static __forceinline__ bf16x8 load_untracked(const char* lds, int ks, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
    const int r1 = ks * 32 + 8 * g + q, col = c0 + 4 * p;
    return kmajor_load_untracked(lds + tswz<RBT>(r1, col >> 3) + ((col & 7) << 1), lds + tswz<RBT>(r1 + 4, col >> 3) + ((col & 7) << 1));
  }
};
template <int COLS> struct KMajorFrag<float, COLS> {
  static constexpr int RBT = COLS * 4;
  static constexpr bool has_untracked = false;
  __device__ static __forceinline__ f32x4 load(const char* lds, int ks, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4, col = c0 + i;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float*>(lds + tswz<RBT>(ks * 16 + 4 * g + j, col >> 2) + ((col & 3) << 2));
    return v;
  }
  __device__ static __forceinline__ f32x4 load_untracked(const char* lds, int ks, int c0, int lane) { return load(lds, ks, c0, lane); }
};

// sum of the k values one lane holds in an operand fragment (bias gradient riding in the dW GEMM): packed bf16 dot
// products against (1, 1) on the otherwise idle VALU instead of extra MFMAs against a ones-column
__device__ __forceinline__ float frag_sum(bf16x8 f, float s) {
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
  for (int p = 0; p < 4; ++p) s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{f[2 * p], f[2 * p + 1]}, one, s, false);
  return s;
}
__device__ __forceinline__ float frag_sum(f32x4 f, float s) { return s + ((f[0] + f[1]) + (f[2] + f[3])); }

// Grouped tile order: consecutive logical ids walk DOWN a group of `gm` tile rows before moving to the next tile column,
// so the ~64 workgroups resident on one XCD cover a gm x (64/gm) patch of tiles and share gm A panels and 64/gm B panels
// in that XCD's 4 MiB L2, instead of one A panel and every B panel (fc1's 4.7 MB weight does not fit next to A).
__device__ __forceinline__ void tile_of(int logical, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
  const int per_group = gm * tiles_n;
  const int grp = logical / per_group;
  const int first = grp * gm;
  const int rows = min(gm, tiles_m - first);
  const int in = logical - grp * per_group;
  tm = first + in % rows;
  tn = in / rows;
}

// Epilogue of one wave's 64x64 accumulator block whose top-left element is (mw, nw).
// !TB: lane owns rows 4*lg + r of each 16-row tile i and the 4 CONSECUTIVE columns 4*li + j  (vector accesses)
//  TB: lane owns column 16*j + li of each n-tile j                                          (scalar accesses)
// Code size matters here: the epilogue runs once per tile but a 12-K-tile GEMM spends a third of its time around it, and
// a fully unrolled nest of run-time `act` branches was ~15k instructions (instruction-cache misses on every tile).
// So: ONE run-time switch, then a tight unrolled store loop per case; everything unusual takes the rolled generic path.
template <typename T>
__device__ __forceinline__ void epi_finish(const GemmArgs& g, float x, size_t off, size_t aoff) {
  if (g.act == MISSM_ACT_QGELU || g.act == MISSM_ACT_GELU) {
    if (g.aux_out) static_cast<T*>(g.aux_out)[aoff] = from_f32<T>(x);
    x = (g.act == MISSM_ACT_QGELU) ? quick_gelu(x) : gelu_erf(x);
  } else if (g.act == MISSM_ACT_DQGELU || g.act == MISSM_ACT_DGELU) {
    const float u = to_f32(static_cast<const T*>(g.aux_in)[aoff]);
    x *= (g.act == MISSM_ACT_DQGELU) ? quick_gelu_grad(u) : gelu_erf_grad(u);
  } else if (g.act == MISSM_ACT_RELU) {
    x = fmaxf(x, 0.f);
  }
  if (g.out_f32) {
    float* c = static_cast<float*>(g.C) + off;
    if (g.resid) x += g.resid[off];
    if (g.accumulate) x += *c;
    *c = x;
  } else {
    static_cast<T*>(g.C)[off] = from_f32<T>(x);
  }
}

// bias of the 4 consecutive output columns a lane owns, fetched at kernel start so its latency hides under the main loop
__device__ __forceinline__ f32x4 prefetch_bias(const GemmArgs& g, int nw, int split, int lane) {
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  const int col = nw + (lane & 15) * 4;
  if (g.bias && split == 0 && col < g.N) {
    if (col + 3 < g.N) b = load4(g.bias + col);
    else
      for (int j = 0; j < 4; ++j) if (col + j < g.N) b[j] = g.bias[col + j];
  }
  return b;
}

// Rare epilogue cases (ragged edges, relu / erf-gelu, accumulate, NN with an activation): the wave parks its 64x64 fp32
// block in its own 16 KiB slice of the (now idle) LDS tile buffers and walks it with ONE rolled loop, so the general
// code exists once per kernel instead of 64 times.  element (row, col) of the block lives at wave_lds[row * 64 + col].
template <typename T, bool TB>
__device__ __forceinline__ void epilogue_generic(const GemmArgs& g, f32x4 (&acc)[4][4], int mw, int nw, int split, int lane,
                                                 float* wave_lds) {
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i * 16 + lg * 4 + r;
        const int col = TB ? j * 16 + li : li * 4 + j;
        wave_lds[row * 64 + col] = acc[i][j][r];
      }
  // same-wave LDS traffic is ordered; no barrier needed (each wave reads back only its own slice)
#pragma unroll 1
  for (int e = lane; e < 64 * 64; e += 64) {
    const int row = e >> 6, col = e & 63;
    const int grow = mw + row, gcol = nw + col;
    if (grow < g.M && gcol < g.N) {
      const float bv = (g.bias && split == 0) ? g.bias[gcol] : 0.f;
      epi_finish<T>(g, wave_lds[e] * g.alpha + bv, (size_t)grow * g.ldc + gcol, (size_t)grow * g.ldaux + gcol);
    }
  }
}

// Split-K without atomics.  fp32 atomics retire one element per L2 channel per clock: a 128x128 tile of them cost 21-30 us,
// more than the main loop of a K = 6304 weight gradient.  (Also measured: the slices meeting inside the GEMM kernel, last
// arriver sums - an agent-scope fence flushes / invalidates the whole XCD L2, 37-85 us per tile; with sc1 stores / loads
// instead the one finishing workgroup reads splitk x 64 KiB serially at memory latency, 15-39 us.)  So every K slice parks
// its accumulators in the per-stream workspace in REGISTER order - thread t, vector v -> float4 #(v*256 + t): coalesced, and
// identical in every slice - and a second, chip-wide kernel sums the slices in slice order (bit-reproducible) and applies
// alpha / bias / accumulate.  One workgroup per (tile, i): thread t owns the same (row, col) set the GEMM thread t owned.
template <bool TB, int WG>
__global__ __launch_bounds__(64 * WG * WG) void splitk_reduce_kernel(GemmArgs g) {
  constexpr int NWAVES = WG * WG, NT = 64 * NWAVES;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int tile = blockIdx.x >> 2, i = blockIdx.x & 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  int tm, tn;
  tile_of(xcd_remap(tile, ntiles), g.tiles_m, g.tiles_n, g.group_m, tm, tn);
  const int mw = tm * (64 * WG) + (wave / WG) * 64, nw = tn * (64 * WG) + (wave % WG) * 64;
  const f32x4* p = reinterpret_cast<const f32x4*>(g.ws) + (size_t)tile * (NWAVES * 1024) + (i * 4) * NT + tid;
  f32x4 a[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int sl = 0; sl < g.splitk; ++sl, p += (size_t)ntiles * (NWAVES * 1024)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += __builtin_nontemporal_load(p + j * NT);
  }
  float* C = static_cast<float*>(g.C);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = mw + i * 16 + lg * 4 + r;
    if (row >= g.M) continue;
    if constexpr (TB) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = nw + j * 16 + li;
        if (col >= g.N) continue;
        float x = a[j][r] * g.alpha + (g.bias ? g.bias[col] : 0.f);
        float* c = C + (size_t)row * g.ldc + col;
        if (g.accumulate) x += *c;
        *c = x;
      }
    } else {
      const int col = nw + li * 4;
      float* c = C + (size_t)row * g.ldc + col;
      if (col + 3 < g.N && g.vec_ok) {
        f32x4 x = {a[0][r], a[1][r], a[2][r], a[3][r]};
        x *= g.alpha;
        if (g.bias) x += load4(g.bias + col);
        if (g.accumulate) x += load4(c);
        store4(c, x);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j < g.N) {
            float x = a[j][r] * g.alpha + (g.bias ? g.bias[col + j] : 0.f);
            if (g.accumulate) x += c[j];
            c[j] = x;
          }
      }
    }
  }
}

// raw pre-activation values (4 consecutive columns of one row) fetched before the main loop for the backward-through-
// activation epilogue: 16 dependent 8-byte loads per lane otherwise sit, latency exposed, between the last MFMA and the stores
template <typename T> struct AuxPre { static constexpr bool ok = false; using V = int; };
template <> struct AuxPre<bf16> { static constexpr bool ok = true; using V = bf16x4; };

template <typename T, bool TB, bool GENERIC_OK = true>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[4][4], int mw, int nw, int split, int lane,
                                              f32x4 bias4, float* wave_lds, const typename AuxPre<T>::V (&upre)[4][4],
                                              bool have_upre) {
  const int li = lane & 15, lg = lane >> 4;
  const bool first_split = split == 0;
  if constexpr (TB) {
    // (register arrays must only ever be indexed with compile-time constants: a rolled loop here sends `acc` to scratch)
    const bool plain_f = g.out_f32 && g.act == MISSM_ACT_NONE && !g.resid && !g.accumulate;   // weight gradients
    const bool plain_t = !g.out_f32 && g.act == MISSM_ACT_NONE;
    if (!plain_f && !plain_t) {
      if constexpr (GENERIC_OK) epilogue_generic<T, TB>(g, acc, mw, nw, split, lane, wave_lds);
      else __builtin_trap();
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = nw + j * 16 + li;
      if (col >= g.N) continue;
      const float bv = (g.bias && first_split) ? g.bias[col] : 0.f;
      if (plain_f) {
        float* C = static_cast<float*>(g.C);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mw + i * 16 + lg * 4 + r;
            if (row < g.M) C[(size_t)row * g.ldc + col] = acc[i][j][r] * g.alpha + bv;
          }
      } else if (plain_t) {
        T* C = static_cast<T*>(g.C);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mw + i * 16 + lg * 4 + r;
            if (row < g.M) C[(size_t)row * g.ldc + col] = from_f32<T>(acc[i][j][r] * g.alpha + bv);
          }
      }
    }
    return;
  } else {
    const int col = nw + li * 4;
    if (nw >= g.N) return;
    const float alpha = g.alpha;
    // wave-uniform decisions only: the generic path exchanges data between the lanes of the wave through LDS
    const bool vec = (nw + 64 <= g.N) && g.vec_ok && g.act != MISSM_ACT_RELU && g.act != MISSM_ACT_GELU &&
                     g.act != MISSM_ACT_DGELU && !g.accumulate;
    // value of (tile i, register r): 4 consecutive columns
#define MISSM_EPI_LOOP(BODY)                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                               \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                             \
        const int row = mw + i * 16 + lg * 4 + r;                                 \
        if (row < g.M) {                                                          \
          f32x4 v = {acc[i][0][r] * alpha + bias4[0], acc[i][1][r] * alpha + bias4[1], acc[i][2][r] * alpha + bias4[2], \
                     acc[i][3][r] * alpha + bias4[3]};                            \
          const size_t off = (size_t)row * g.ldc + col;                           \
          BODY                                                                    \
        }                                                                         \
      }                                                                           \
    }
    if (vec && !g.out_f32 && g.act == MISSM_ACT_NONE) {                 // T store (+bias): QKV projection, dX
      T* C = static_cast<T*>(g.C);
      MISSM_EPI_LOOP(store4(C + off, v);)
    } else if (vec && !g.out_f32 && g.act == MISSM_ACT_QGELU) {         // fc1: activation + saved pre-activation
      T* C = static_cast<T*>(g.C);
      T* U = static_cast<T*>(g.aux_out);
      MISSM_EPI_LOOP(if (U) store4(U + (size_t)row * g.ldaux + col, v);
                     v[0] = quick_gelu(v[0]); v[1] = quick_gelu(v[1]); v[2] = quick_gelu(v[2]); v[3] = quick_gelu(v[3]);
                     store4(C + off, v);)
    } else if (vec && !g.out_f32 && g.act == MISSM_ACT_DQGELU) {        // backward through the activation
      T* C = static_cast<T*>(g.C);
      const T* U = static_cast<const T*>(g.aux_in);
      if constexpr (AuxPre<T>::ok) {
        if (have_upre) {
          MISSM_EPI_LOOP(v[0] *= quick_gelu_grad((float)upre[i][r][0]); v[1] *= quick_gelu_grad((float)upre[i][r][1]);
                         v[2] *= quick_gelu_grad((float)upre[i][r][2]); v[3] *= quick_gelu_grad((float)upre[i][r][3]);
                         store4(C + off, v);)
          return;
        }
      }
      MISSM_EPI_LOOP(const f32x4 u = load4(U + (size_t)row * g.ldaux + col);
                     v[0] *= quick_gelu_grad(u[0]); v[1] *= quick_gelu_grad(u[1]); v[2] *= quick_gelu_grad(u[2]);
                     v[3] *= quick_gelu_grad(u[3]); store4(C + off, v);)
    } else if (vec && g.out_f32 && g.act == MISSM_ACT_NONE) {           // fp32 residual stream / fp32 outputs
      float* C = static_cast<float*>(g.C);
      const float* R = g.resid;
      MISSM_EPI_LOOP(if (R) { const f32x4 q = load4(R + off); v += q; } store4(C + off, v);)
    } else {                                                            // ragged edges, rare activations, accumulate
      if constexpr (GENERIC_OK) epilogue_generic<T, TB>(g, acc, mw, nw, split, lane, wave_lds);
      else __builtin_trap();                                            // host dispatch never sends these to the big tile
    }
#undef MISSM_EPI_LOOP
  }
}

// TA: A is stored [K, M] (reduction index on rows); TB: B is stored [K, N].  TA = TB = false is the NT form.
// k-contiguous LDS rows hold RBK bytes: 128 (K tile 64 bf16, 64 KiB of LDS, 2 workgroups per CU) or 64 (K tile 32 bf16,
// 32 KiB, 4 workgroups per CU: the fill/drain and epilogue of one workgroup hide under three others - short-K GEMMs).
template <int RBK> __device__ __forceinline__ int kswz(int row, int chunk) {
  if constexpr (RBK == 128) return row * 128 + ((chunk ^ (row & 7)) << 4);
  else return row * 64 + ((chunk ^ ((0x78 >> (((row >> 2) & 3) << 1)) & 3)) << 4);   // g = [0,2,3,1]: conflict-free b128 reads
}

// WG = waves per side of the workgroup's wave grid: 2 -> 128x128 tile, 256 threads, two workgroups per CU;
//                                                    4 -> 256x256 tile, 1024 threads, one workgroup per CU (NT form only).
// The per-wave code is the same (64x64 accumulators); the big tile halves the L2 -> LDS bytes staged per flop, which is what
// bounds the small tile on long-M shapes (the main loop runs 31 % faster with staging switched off; 512 workgroups x 32 KiB
// per K tile is ~19 TB/s of L2 traffic).  It needs full 64-column wave blocks and a vector epilogue (no LDS-backed generic
// path: 16 waves x 16 KiB would not fit), which the host dispatch guarantees.
template <typename T, bool TA, bool TB, int RBK, int VAR = 0, int WG = 2>
__global__ __launch_bounds__(64 * WG * WG, WG == 4 ? 4 : (RBK == 128 ? 2 : 4)) void gemm_kernel(GemmArgs g) {
  static_assert(WG == 2 || (WG == 4 && !TA && !TB), "the 256x256 tile is instantiated for the NT form only");
  constexpr int BMt = 64 * WG, BNt = 64 * WG, NWAVES = WG * WG;
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int EPC = 16 / sizeof(T);     // elements per 16-byte chunk
  constexpr int BK = RBK / sizeof(T);     // k elements per tile (64 or 32 bf16 / 32 or 16 f32)
  constexpr int KSTEPS = BK / M_::KS;     // 2 or 1
  constexpr int RBT = KMajorFrag<T, BMt>::RBT; // bytes per row of a k-major tile (128 elements)
  constexpr int NCT = RBT / 16;           // 16-byte chunks per k-major row
  constexpr int TILE_BYTES = BMt * RBK;   // 16 or 8 KiB (32 KiB for the 256-row tile), either layout
  constexpr int NCK = RBK / 16;           // chunks per k-contiguous row
  constexpr int PPW = TILE_BYTES / (NWAVES * 1024);  // 1-KiB pieces per wave per operand tile (4 or 2)
  constexpr int RPK = 1024 / RBK;         // k-contiguous rows per piece
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 4 * TILE_BYTES
  char* ldsA = lds;                       // [2][TILE_BYTES]
  char* ldsB = lds + 2 * TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave / WG, wn = wave % WG;

  const int ntiles = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int split = bid / ntiles;
  const int logical = xcd_remap(bid - split * ntiles, ntiles);
  int tm, tn;
  tile_of(logical, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BMt, n0 = tn * BNt;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const T* __restrict__ A = static_cast<const T*>(g.A);
  const T* __restrict__ B = static_cast<const T*>(g.B);

  // ---- staging: global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write).  One wave
  // instruction fills 1 KiB of LDS linearly (wave-uniform base + 16 B * lane); the swizzle is applied by choosing which
  // global chunk each lane fetches.  4 pieces of A and 4 of B per wave per K tile.  Out-of-range chunks (K edge, ragged
  // M/N of a k-major operand) are fetched from a 16-byte zero block instead.
  const T* a_src[PPW];
  const T* b_src[PPW];
  bool a_ok[PPW], b_ok[PPW];
  int a_kofs[PPW], b_kofs[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pi = wave * PPW + i;
    if constexpr (!TA) {
      const int row = RPK * pi + lane / NCK, c = (kswz<RBK>(row, lane % NCK) - row * RBK) >> 4;
      int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;
      a_src[i] = A + (size_t)gr * g.lda + c * EPC;
      a_ok[i] = true; a_kofs[i] = c * EPC;
    } else {
      constexpr int RPP = 1024 / RBT;
      const int row = RPP * pi + lane / NCT, c = (lane % NCT) ^ (tkey(row) << 1);
      a_src[i] = A + (size_t)row * g.lda + m0 + c * EPC;
      a_ok[i] = (m0 + c * EPC) < g.M; a_kofs[i] = row;
    }
    if constexpr (!TB) {
      const int lr = RPK * pi + lane / NCK, c = (kswz<RBK>(lr, lane % NCK) - lr * RBK) >> 4;
      // LDS row lr holds tile row r (output column n0 + r) with lr = 64*(r/64) + 16*(r%4) + (r%64)/4
      const int r = (lr & ~63) + ((lr & 15) << 2) + ((lr >> 4) & 3);
      int gn = n0 + r; gn = gn < g.N ? gn : g.N - 1;
      b_src[i] = B + (size_t)gn * g.ldb + c * EPC;
      b_ok[i] = true; b_kofs[i] = c * EPC;
    } else {
      constexpr int RPP = 1024 / RBT;
      const int row = RPP * pi + lane / NCT, c = (lane % NCT) ^ (tkey(row) << 1);
      b_src[i] = B + (size_t)row * g.ldb + n0 + c * EPC;
      b_ok[i] = (n0 + c * EPC) < g.N; b_kofs[i] = row;
    }
  }
  using gptr = const __attribute__((address_space(1))) void*;
  using lptr = __attribute__((address_space(3))) void*;
  auto stage = [&](int buf, int k0) {
    char* la = ldsA + buf * TILE_BYTES + wave * (PPW * 1024);
    char* lb = ldsB + buf * TILE_BYTES + wave * (PPW * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const T* pa = (a_ok[i] && (k0 + a_kofs[i]) < kend) ? (TA ? a_src[i] + (size_t)k0 * g.lda : a_src[i] + k0)
                                                          : reinterpret_cast<const T*>(g_zero16);
      const T* pb = (b_ok[i] && (k0 + b_kofs[i]) < kend) ? (TB ? b_src[i] + (size_t)k0 * g.ldb : b_src[i] + k0)
                                                          : reinterpret_cast<const T*>(g_zero16);
      __builtin_amdgcn_global_load_lds((gptr)pa, (lptr)(la + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)pb, (lptr)(lb + i * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient: waves also add up A fragments they hold anyway (packed dot products on the VALU).  A wave row's 4 x KSTEPS
  // fragments per K tile are dealt out to the waves of the first four tile columns that see this A panel (unit = 2 tn + wn),
  // one fragment each.  Measured alternatives: everything on (tn, wn) = (0, 0) - MFMAs against a ones-column or the same dot
  // products - makes those workgroups 20-25 % slower and they set the kernel's span; interleaving the dot products with the
  // MFMAs slows the MFMAs; dealing the K tiles out to ALL tile columns multiplies the atomics on the same addresses.
  constexpr int CS_ITEMS = 4 * KSTEPS;
  const int cs_units = min(CS_ITEMS, WG * g.tiles_n), cs_unit = WG * tn + wn;
  unsigned cs_mask = 0;                  // bit w: this wave sums fragment (ks, i) = (w % KSTEPS, w / KSTEPS)
  if (TA && g.colsum_a != nullptr && cs_unit < cs_units)
    for (int w = 0; w < CS_ITEMS; ++w) cs_mask |= (w % cs_units == cs_unit) ? (1u << w) : 0u;
  cs_mask = __builtin_amdgcn_readfirstlane(cs_mask);   // wave-uniform by construction: make the tests scalar branches
  const bool do_cs = cs_mask != 0;
  float accb[4] = {0.f, 0.f, 0.f, 0.f};   // lane (li, lg): partial sum of row 16 i + li over the k values of lane group lg

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;
  unsigned long long t_start = 0, t_loop = 0, t_loop_end = 0;
  if (g.dbg) t_start = __builtin_amdgcn_s_memrealtime();
  const f32x4 bias4 = prefetch_bias(g, n0 + wn * 64, 0, lane);
  typename AuxPre<T>::V upre[4][4];
  bool have_upre = false;
  if constexpr (AuxPre<T>::ok && !TB && WG == 2) {
    const int nw_ = n0 + wn * 64;
    have_upre = g.act == MISSM_ACT_DQGELU && !g.out_f32 && g.vec_ok && !g.accumulate && nw_ + 64 <= g.N && g.splitk == 1;
    if (have_upre) {
      const T* U = static_cast<const T*>(g.aux_in) + nw_ + li * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(m0 + wm * 64 + i * 16 + lg * 4 + r, g.M - 1);
          upre[i][r] = *reinterpret_cast<const bf16x4*>(U + (size_t)row * g.ldaux);
        }
    }
  }
  stage(0, kbeg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (g.dbg) t_loop = __builtin_amdgcn_s_memrealtime();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // The next tile lands in the other buffer under this tile's MFMAs.  With a k-major operand the LDS-DMA is issued AFTER
    // the fragment reads: the compiler cannot see that ds_read_b64_tr_b16 (an intrinsic without memory operands) does not
    // alias the DMA's LDS writes and would put s_waitcnt vmcnt(0) - the whole global-load latency - in front of the reads.
    constexpr bool STAGE_LATE = VAR == 2 && (TA || TB);
    if (!STAGE_LATE && kt + 1 < nk) stage(buf ^ 1, kbeg + (kt + 1) * BK);
    const char* la = ldsA + buf * TILE_BYTES;
    const char* lb = ldsB + buf * TILE_BYTES;
    if constexpr (VAR == 2) {
      // all fragments of the K tile are requested up front; MFMAs of step 0 start as soon as ITS fragments are back
      Frag fa[KSTEPS][4], fb[KSTEPS][4];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (!TA) fa[ks][i] = lds_frag<T>(la, kswz<RBK>(wm * 64 + i * 16 + li, ks * 4 + lg));
          else fa[ks][i] = KMajorFrag<T, BMt>::load(la, ks, wm * 64 + i * 16, lane);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (!TB) fb[ks][j] = lds_frag<T>(lb, kswz<RBK>(wn * 64 + j * 16 + li, ks * 4 + lg));
          else fb[ks][j] = KMajorFrag<T, BMt>::load(lb, ks, wn * 64 + j * 16, lane);
        }
      }
      if (STAGE_LATE && kt + 1 < nk) stage(buf ^ 1, kbeg + (kt + 1) * BK);
      if constexpr (TA) {
        if (do_cs) {
#pragma unroll
          for (int w = 0; w < CS_ITEMS; ++w)
            if ((cs_mask >> w) & 1u) {
              asm volatile("");   // keeps this a real (scalar) branch: if-converted, every wave would sum all fragments
              accb[w / KSTEPS] = frag_sum(fa[w % KSTEPS][w / KSTEPS], accb[w / KSTEPS]);
            }
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = M_::step(fa[ks][i], fb[ks][j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      Frag fa[4], fb[4];
      const int cchunk = ks * 4 + lg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (!TA) fa[i] = lds_frag<T>(la, kswz<RBK>(wm * 64 + i * 16 + li, cchunk));
        else fa[i] = KMajorFrag<T, BMt>::load_untracked(la, ks, wm * 64 + i * 16, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (!TB) fb[j] = lds_frag<T>(lb, kswz<RBK>(wn * 64 + j * 16 + li, cchunk));
        else fb[j] = KMajorFrag<T, BMt>::load_untracked(lb, ks, wn * 64 + j * 16, lane);
      }
      // k-major fragments came through untracked inline-assembly reads (no s_waitcnt vmcnt(0) in front of them while the
      // next tile's LDS-DMA is in flight): this is the wait for them
      if constexpr ((TA || TB) && KMajorFrag<T, BMt>::has_untracked) frag_fence(fa, fb);
      if constexpr (TA) {
        if (do_cs) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if ((cs_mask >> (i * KSTEPS + ks)) & 1u) {
              asm volatile("");
              accb[i] = frag_sum(fa[i], accb[i]);
            }
        }
      }
      if constexpr (VAR == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = M_::step(fa[i], fb[j], acc[i][j]);
      if constexpr (VAR == 1) __builtin_amdgcn_s_setprio(0);
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  unsigned long long t_pre_cs = 0;
  if (g.dbg) t_pre_cs = __builtin_amdgcn_s_memrealtime();
  if constexpr (TA) {
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bool mine = false;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) mine |= ((cs_mask >> (i * KSTEPS + ks)) & 1u) != 0;
        if (!mine) continue;            // wave-uniform
        float v = accb[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int m = m0 + wm * 64 + i * 16 + li;
        if (lg == 0 && m < g.M) atomicAdd(g.colsum_a + m, v);
      }
    }
  }

  if (g.dbg) t_loop_end = __builtin_amdgcn_s_memrealtime();
  if (g.splitk > 1) {   // K slice: park the partial tile for splitk_reduce_kernel
    f32x4* mine = reinterpret_cast<f32x4*>(g.ws) + ((size_t)split * ntiles + (bid - split * ntiles)) * (NWAVES * 1024) + tid;
#pragma unroll
    for (int v = 0; v < 16; ++v) mine[v * (64 * NWAVES)] = acc[v >> 2][v & 3];
  } else {
    gemm_epilogue<T, TB, WG == 2>(g, acc, m0 + wm * 64, n0 + wn * 64, 0, lane, bias4, reinterpret_cast<float*>(lds + wave * 16384), upre, have_upre);
  }
  if (g.dbg && tid == 0) {
    const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.dbg[(size_t)blockIdx.x * 8 + 5] = t_issued;
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    unsigned long long* d = g.dbg + (size_t)blockIdx.x * 8;
    d[0] = t_start; d[1] = t_loop; d[2] = t_loop_end; d[3] = __builtin_amdgcn_s_memrealtime(); d[4] = hw; d[6] = t_pre_cs;
  }
}

}  // namespace missm
#include "gemm8p.h"
#include "gemm4w.h"
namespace missm {

// Measured and removed again (git history has them; all correct, none faster on the hot-path shapes):
//   * 256x128 / 8 waves / 3-stage ring with counted vmcnt, one workgroup per CU : 845 TFLOP/s at 4096^3, 570-750 on the
//     K = 768 video shapes (every tile's fill/drain is exposed with a single resident workgroup);
//   * 256x256 / 8 waves / K tile 32 / 4-stage ring                              : 862 at 4096^3, 400-670 on the video shapes;
//   * 128x128 / K tile 32 / 4-stage ring, two workgroups per CU                 : 768 at 4096^3, 615-650 on the video shapes;
//   * the same ring kept running across output tiles (persistent workgroups)    : 665 at 4096^3 (the counted vmcnt then also
//     waits for the previous tile's epilogue stores);
//   * staggering co-resident workgroups by 6-25 us                               : -2..-5 %;
//   * a 256x256 tile as 8 waves of 128x64 with a 32-deep K tile (64 KiB of LDS, two workgroups per CU, 198 VGPRs): 755-810
//     TFLOP/s on the video NT shapes against 843-1006 for the 16-wave version below (twice the barriers per MFMA);
//   * the 256x256 / 16-wave tile for the weight-gradient (TN) form, split-K over one round of 256 workgroups: 575-647 TFLOP/s
//     on the video dW shapes against 643-690 for the 128x128 kernel (its 128-VGPR budget has no room to preload both K steps'
//     fragments, so the k-major reads go through untracked inline-assembly loads behind an early LDS-DMA - load_untracked,
//     frag_fence, which the 128x128 kernel's VARIANT 1 still uses - and the 62 MB of partial tiles cost 20 us to reduce).
// What did pay: global_load_lds staging, s_setprio around the MFMA cluster (+4-10 % at K = 768), requesting all fragments
// of a K tile up front (+10 % at long K), a compact epilogue (instruction cache; +16 % at K = 768), bias prefetch, split-K
// sized to ONE resident wave of workgroups, grouped tile order.  This kernel: 1088 TFLOP/s at 4096^3, 740-830 on the video
// tower's K = 768 shapes (random data, one MI355X).

// ---------------------------------------------------------------------------------------------------
// 64x64 tiled transpose with zero padding of the new inner dimension and an optional fused column sum
// (used to feed the NT GEMM with dY^T / X^T for weight gradients; the column sum is the bias gradient).
// in [R, C] (ld) -> out [C, Rp] (ldo >= Rp), Rp = R rounded up to a multiple of 64: out[c][r] = r < R ? in[r][c] : 0
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int C,
                                                           int ld, int ldo, float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // ty 0..3
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < R && c < C) ? to_f32(in[(size_t)r * ld + c]) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < ldo) out[(size_t)c * ldo + r] = from_f32<T>(tile[tx][ty + 4 * i]);
  }
  if (colsum && ty == 0) {
    const int c = c0 + tx;
    if (c < C) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) s += tile[r][tx];
      atomicAdd(colsum + c, s);
    }
  }
}

// grouped column sum: out[group(row)][c] += sum over rows ; group(row) = (row / div) % mod
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, float* __restrict__ out, int R, int C, int ld,
                                                    int div, int mod, int rows_per_block) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int rbeg = blockIdx.x * rows_per_block;
  const int rend = min(R, rbeg + rows_per_block);
  int cur = -1; float s = 0.f;
  for (int r = rbeg; r < rend; ++r) {
    const int gidx = (r / div) % mod;
    if (gidx != cur) { if (cur >= 0) atomicAdd(out + (size_t)cur * C + c, s); cur = gidx; s = 0.f; }
    s += to_f32(in[(size_t)r * ld + c]);
  }
  if (cur >= 0) atomicAdd(out + (size_t)cur * C + c, s);
}

// column sum (bias / temporal-embedding gradients), HBM-bound: 16-byte loads, 8 row-lanes x 32 column-groups per workgroup,
// each workgroup reduces a rpb-row x (32 * VEC)-column panel through LDS and issues one atomic per column.  Grouped form
// (mod > 1): the rows of one workgroup are one run of `div` rows, all in group (first row / div) % mod -> out[group][c].
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ in, float* __restrict__ out, int R, int C, int ld,
                                                        int rpb, int div, int mod) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[8][32 * VEC + 1];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = (blockIdx.y * 32 + cg) * VEC;
  const int rbeg = blockIdx.x * rpb;
  const int rend = min(R, rbeg + rpb);
  out += (size_t)((rbeg / div) % mod) * C;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (c0 < C) {
    for (int r = rbeg + rl; r < rend; r += 8) {
      const T* p = in + (size_t)r * ld + c0;
#pragma unroll
      for (int j = 0; j < VEC; j += 4) {
        const f32x4 v = load4(p + j);
        acc[j] += v[0]; acc[j + 1] += v[1]; acc[j + 2] += v[2]; acc[j + 3] += v[3];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[rl][cg * VEC + j] = acc[j];
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * VEC; i += 256) {
    const int c = blockIdx.y * 32 * VEC + i;
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += red[k][i];
      atomicAdd(out + c, s);
    }
  }
}

// fp32 master weight -> T shadow copy, optionally transposed: dst[c][r] (ldd) or dst[r][c]
template <typename T>
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dst_t,
                                                         int R, int C) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    float v = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    tile[ty + 4 * i][tx] = v;
    if (dst && r < R && c < C) dst[(size_t)r * C + c] = from_f32<T>(v);
  }
  if (!dst_t) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < R) dst_t[(size_t)c * R + r] = from_f32<T>(tile[tx][ty + 4 * i]);
  }
}

// all weight blocks of a tower in ONE launch: one 64x64 tile per workgroup, described by a table built once on the host
struct CastTile { const float* src; void* dst; void* dst_t; int R, C, r0, c0; };
template <typename T>
__global__ __launch_bounds__(256) void cast_weights_batched_kernel(const CastTile* __restrict__ tiles) {
  __shared__ float tile[64][65];
  const CastTile t = tiles[blockIdx.x];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  T* dst = static_cast<T*>(t.dst);
  T* dst_t = static_cast<T*>(t.dst_t);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = t.r0 + ty + 4 * i, c = t.c0 + tx;
    const float v = (r < t.R && c < t.C) ? t.src[(size_t)r * t.C + c] : 0.f;
    tile[ty + 4 * i][tx] = v;
    if (dst && r < t.R && c < t.C) dst[(size_t)r * t.C + c] = from_f32<T>(v);
  }
  if (!dst_t) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = t.c0 + ty + 4 * i, r = t.r0 + tx;
    if (c < t.C && r < t.R) dst_t[(size_t)c * t.R + r] = from_f32<T>(tile[tx][ty + 4 * i]);
  }
}

// The optimizer step of a tower's weight matrices fused with the refresh of their compute-dtype copies: one 64x64 tile per
// workgroup reads p, g, m, v, applies Adam (same arithmetic as adam_kernel, misc.hip), writes p, m, v and the bf16 W / W^T
// tiles - the separate refresh would read the 4 bytes/parameter of p once more.  g, m, v live in flat buffers parallel to
// the master buffer: the same element offset, at a fixed distance (in floats) from it.
struct AdamTileArgs { long g_off, m_off, v_off; float lr_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, gscale; };
template <typename T>
__global__ __launch_bounds__(256) void adam_cast_batched_kernel(const CastTile* __restrict__ tiles, AdamTileArgs a) {
  __shared__ float tile[64][65];
  const CastTile t = tiles[blockIdx.x];
  float* src = const_cast<float*>(t.src);
  T* dst = static_cast<T*>(t.dst);
  T* dst_t = static_cast<T*>(t.dst_t);
  if ((t.C & 3) == 0 && (t.R & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((a.g_off | a.m_off | a.v_off) & 3) == 0) {
    // [r3] 16-byte accesses: a thread owns 4 consecutive columns of a row (16 threads per row, 16 rows per pass) - the scalar version
    // (4 bytes per lane: 256 bytes per wave instruction) moved the 32 bytes per parameter at 4.1 TB/s.  Same arithmetic per element.
    const int tc = (threadIdx.x & 15) * 4, tr = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = tr + 16 * i, r = t.r0 + rl, c = t.c0 + tc;
      f32x4 pv = {0.f, 0.f, 0.f, 0.f};
      if (r < t.R && c < t.C) {
        float* p = src + (size_t)r * t.C + c;
#ifdef MISSM_ADAM_NT              // (measured neutral: 64.75 / 64.97 / 64.41 vs 64.81 / 64.63 / 64.55 ms per step)
        pv = load4_nt(p);
        const f32x4 gv = load4_nt(p + a.g_off);
        f32x4 mv = load4_nt(p + a.m_off), vv = load4_nt(p + a.v_off);
#else
        pv = load4(p);
        const f32x4 gv = load4(p + a.g_off);
        f32x4 mv = load4(p + a.m_off), vv = load4(p + a.v_off);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gg = gv[j] * a.gscale + a.wd * pv[j];
          mv[j] = a.beta1 * mv[j] + (1.f - a.beta1) * gg;
          vv[j] = a.beta2 * vv[j] + (1.f - a.beta2) * gg * gg;
          pv[j] -= a.lr_bc1 * mv[j] / (sqrtf(vv[j]) * a.inv_sqrt_bc2 + a.eps);
        }
#ifdef MISSM_ADAM_NT
        store4_nt(p, pv); store4_nt(p + a.m_off, mv); store4_nt(p + a.v_off, vv);
#else
        store4(p, pv); store4(p + a.m_off, mv); store4(p + a.v_off, vv);
#endif
        if (dst) store4(dst + (size_t)r * t.C + c, pv);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[rl][tc + j] = pv[j];
    }
    if (!dst_t) return;
    __syncthreads();
    // transposed copy: a thread owns 4 consecutive ROWS (= 4 consecutive elements of a W^T row) of one column
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cl = tr + 16 * i, c = t.c0 + cl, r = t.r0 + tc;
      if (c < t.C && r < t.R) store4(dst_t + (size_t)c * t.R + r, f32x4{tile[tc][cl], tile[tc + 1][cl], tile[tc + 2][cl], tile[tc + 3][cl]});
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = t.r0 + ty + 4 * i, c = t.c0 + tx;
    float pv = 0.f;
    if (r < t.R && c < t.C) {
      float* p = src + (size_t)r * t.C + c;
      pv = *p;
      const float gg = p[a.g_off] * a.gscale + a.wd * pv;
      const float mv = a.beta1 * p[a.m_off] + (1.f - a.beta1) * gg;
      const float vv = a.beta2 * p[a.v_off] + (1.f - a.beta2) * gg * gg;
      pv -= a.lr_bc1 * mv / (sqrtf(vv) * a.inv_sqrt_bc2 + a.eps);
      *p = pv; p[a.m_off] = mv; p[a.v_off] = vv;
      if (dst) dst[(size_t)r * t.C + c] = from_f32<T>(pv);
    }
    tile[ty + 4 * i][tx] = pv;
  }
  if (!dst_t) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = t.c0 + ty + 4 * i, r = t.r0 + tx;
    if (c < t.C && r < t.R) dst_t[(size_t)c * t.R + r] = from_f32<T>(tile[tx][ty + 4 * i]);
  }
}

}  // namespace missm

using namespace missm;

static unsigned long long* missm_gemm_debug_buffer = nullptr;
extern "C" void missm_gemm_set_debug_buffer(void* p) { missm_gemm_debug_buffer = static_cast<unsigned long long*>(p); }

// Split-K workspaces: one per stream (launches on a stream are ordered, so consecutive split-K GEMMs may share one; GEMMs on
// different streams run concurrently and must not).  Grown on demand - growth synchronises that stream once, during warm-up.
namespace {
struct SplitKWs { float* ws = nullptr; size_t bytes = 0; unsigned* sched = nullptr; };
std::mutex g_ws_mu;
std::unordered_map<void*, SplitKWs> g_ws;

int splitk_workspace(void* stream, size_t bytes, float** ws) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  SplitKWs& w = g_ws[stream];
  if (w.bytes < bytes) {
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 1;
    if (w.ws) (void)hipFree(w.ws);
    w = SplitKWs();
    const size_t nb = bytes < (size_t(32) << 20) ? (size_t(32) << 20) : bytes;
    if (hipMalloc(reinterpret_cast<void**>(&w.ws), nb) != hipSuccess) { w = SplitKWs(); return 1; }
    w.bytes = nb;
  }
  *ws = w.ws;
  return 0;
}

// the tile-queue counters of the dynamically scheduled persistent grid: one zero-filled 64-byte block per stream (launches on one
// stream are ordered; every launch leaves its counters at zero)
int tile_queue_counters(void* stream, unsigned** sched) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  SplitKWs& w = g_ws[stream];
  if (!w.sched) {
    if (hipMalloc(reinterpret_cast<void**>(&w.sched), 64) != hipSuccess) { w.sched = nullptr; return 1; }
    if (hipMemset(w.sched, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(w.sched); w.sched = nullptr; return 1; }
  }
  *sched = w.sched;
  return 0;
}
}  // namespace

int missm_stream_workspace(void* stream, size_t bytes, float** ws) { return splitk_workspace(stream, bytes, ws); }

extern "C" void missm_gemm_release_workspaces(void) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  (void)hipDeviceSynchronize();
  for (auto& kv : g_ws) { if (kv.second.ws) (void)hipFree(kv.second.ws); if (kv.second.sched) (void)hipFree(kv.second.sched); }
  g_ws.clear();
}

// ngroups > 1: `groups` holds the operands of `ngroups` problems of this one shape (A .. colsum_a above are group 0's).  Returns
// MISSM_GROUPED_UNAVAILABLE (without launching anything) when no grouped kernel covers the call: the caller loops instead.
#define MISSM_GROUPED_UNAVAILABLE 1000
// set around the row-remainder launch of a split NT product (see the 256x256 dispatch): that piece may take the 256x128 kernel
static thread_local int g_row_remainder = 0;

static int gemm_core(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                     int trans_b, float alpha, const float* bias, const float* resid, const void* aux_in, void* aux_out,
                     int ldaux, int act, int out_f32, int accumulate, int splitk, float* colsum_a, int dtype, void* stream,
                     int ngroups, const GemmArgs::Group* groups) {
  MISSM_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem");
  MISSM_CHECK_ARG(!colsum_a || trans_a, "gemm: colsum_a rides only in the A^T (weight-gradient) form");
  MISSM_CHECK_ARG(dtype == kBF16 || dtype == kF32, "gemm: dtype must be 0 (f32) or 1 (bf16)");
  const int epc = dtype == kBF16 ? 8 : 4;
  MISSM_CHECK_ARG(lda % epc == 0 && ldb % epc == 0, "gemm: lda, ldb must be multiples of 16 bytes");
  MISSM_CHECK_ARG(trans_a ? (M % epc == 0) : (K % epc == 0), "gemm: the contiguous extent of A must be a multiple of 16 bytes");
  MISSM_CHECK_ARG(trans_b ? (N % epc == 0) : (K % epc == 0), "gemm: the contiguous extent of B must be a multiple of 16 bytes");
  MISSM_CHECK_ARG(!(trans_a && !trans_b), "gemm: A^T with k-contiguous B is not instantiated");
  MISSM_CHECK_ARG(!(resid && !out_f32) && !(accumulate && !out_f32), "gemm: resid/accumulate need out_f32");
  MISSM_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0), "gemm: unaligned operand");
  MISSM_CHECK_ARG(!bias || ((uintptr_t)bias % 16 == 0), "gemm: bias must be 16-byte aligned");
  // MISSM_GEMM_LOG=<file>: one line per tile-kernel dispatch, in launch order - tools/gemm_insitu.py joins it with a rocprofv3
  // kernel trace for the per-shape time INSIDE the training step.
  static FILE* shape_log = getenv("MISSM_GEMM_LOG") ? fopen(getenv("MISSM_GEMM_LOG"), "w") : nullptr;
  auto log_shape = [&](int m_eff) {
    if (!shape_log) return;
    fprintf(shape_log, "%d %d %d %d %d %d %d %d %d %d %d %d %d\n", m_eff, N, K, trans_a, trans_b, act, out_f32, resid != nullptr,
            aux_in != nullptr, aux_out != nullptr, colsum_a != nullptr, accumulate, ngroups > 1 ? ngroups : 1);
    fflush(shape_log);
  };
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha;
  g.bias = bias; g.resid = resid; g.aux_in = aux_in; g.aux_out = aux_out; g.ldaux = ldaux; g.act = act;
  g.out_f32 = out_f32; g.accumulate = accumulate; g.colsum_a = colsum_a;
  g.ngroups = ngroups > 1 ? ngroups : 1; g.group_tiles_m = 0; g.tn_order = 0; g.sched = nullptr; g.desync_phases = 0; g.desync_step = 0;
  if (ngroups > 1) {
    MISSM_CHECK_ARG(ngroups <= MISSM_MAX_GROUPS && groups, "gemm: too many groups");
    for (int i = 0; i < ngroups; ++i) g.grp[i] = groups[i];
  }
  static const int group_m_env = getenv("MISSM_GEMM_GROUP_M") ? atoi(getenv("MISSM_GEMM_GROUP_M")) : 0;
  g.group_m = 1;   // set once the tile grid is known
  g.dbg = missm_gemm_debug_buffer;
  static const int variant = getenv("MISSM_GEMM_VARIANT") ? atoi(getenv("MISSM_GEMM_VARIANT")) : -1;   // scheduling experiments
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  g.vec_ok = (ldc % 4 == 0) && (ldaux % 4 == 0) && ((uintptr_t)C % 16 == 0) && ((uintptr_t)resid % 16 == 0) &&
             ((uintptr_t)aux_in % 16 == 0) && ((uintptr_t)aux_out % 16 == 0);
  for (int i = 1; i < ngroups; ++i) {
    const GemmArgs::Group& q = groups[i];
    g.vec_ok = g.vec_ok && ((uintptr_t)q.C % 16 == 0) && ((uintptr_t)q.resid % 16 == 0) && ((uintptr_t)q.aux_in % 16 == 0) &&
               ((uintptr_t)q.aux_out % 16 == 0);
    MISSM_CHECK_ARG(((uintptr_t)q.A % 16 == 0) && ((uintptr_t)q.B % 16 == 0) && ((uintptr_t)q.bias % 16 == 0), "gemm: unaligned group operand");
    MISSM_CHECK_ARG(!q.bias == !bias && !q.resid == !resid && !q.aux_in == !aux_in && !q.aux_out == !aux_out && !q.colsum_a == !colsum_a,
                    "gemm: the groups of one launch must use the same optional operands");
  }
  // measured on the video tower (GROUP_M 1 / 8 / 16): QKV 668 / 732 / 756, fc1 646 / 692 / 703, fc2 (6 tile columns) 795 / 772 / 729
  g.group_m = group_m_env > 0 ? group_m_env : (g.tiles_n >= 12 ? 16 : (g.tiles_n >= 8 ? 8 : 1));
  const int bk = dtype == kBF16 ? 64 : 32;
  static const int big_env = getenv("MISSM_GEMM_BIG") ? atoi(getenv("MISSM_GEMM_BIG")) : -1;   // 0 never, 1 whenever legal
  const int tiles = g.tiles_m * g.tiles_n;
  static const int fill_env = getenv("MISSM_GEMM_FILL") ? atoi(getenv("MISSM_GEMM_FILL")) : 512;
  const int fill = fill_env;   // workgroups that fill the chip once (2 per CU)
  if (splitk <= 0) {  // auto: fill the chip exactly ONCE (2 workgroups x 256 CUs) - one resident wave of blocks, no tail.
    // Measured (dW shapes, tiles x splits): 144x3 = 432 -> 427-596 TFLOP/s, 144x4 = 576 -> 280-420 (a second, nearly empty
    // round), 36x12 -> 506 vs 36x8 -> 379; fewer splits also means fewer fp32 atomics (1.3 TB/s chip-wide).
    splitk = 1;
    if (out_f32 && !resid && act == MISSM_ACT_NONE && tiles < fill) {
      splitk = fill / tiles;
      const int maxs = K / (4 * bk);
      if (splitk > maxs) splitk = maxs;
      if (splitk < 1) splitk = 1;
    }
  }
  MISSM_CHECK_ARG(splitk == 1 || (out_f32 && !resid && act == MISSM_ACT_NONE), "gemm: split-K needs an fp32 output and no activation / residual");
  int kps = (K + splitk - 1) / splitk;
  kps = (kps + bk - 1) / bk * bk;
  splitk = (K + kps - 1) / kps;
  g.splitk = splitk; g.k_per_split = kps;
  g.ws = nullptr;
  if (splitk > 1) {
    if (splitk_workspace(stream, (size_t)splitk * tiles * (BM * BN * sizeof(float)), &g.ws)) {
      missm_set_error("gemm: cannot allocate the split-K workspace");
      return MISSM_ERR_LAUNCH;
    }
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // ---- weight gradients of long activations (both operands k-major): 8-phase 256x256 kernel over K slices + ordered reduce
  static const int use8p_tn = getenv("MISSM_GEMM_8P_TN") ? atoi(getenv("MISSM_GEMM_8P_TN")) : 1;
  // (measured, rows = 6304: 2304x768 502 vs 455 TFLOP/s for the 128x128 split-K kernel, 3072x768 577 vs 498, 768x768 218 vs 236;
  //  768x768 over 16384 rows 410 vs 441: the big tile needs >= 18 output tiles or a long reduction)
  static const int tn_min_k = getenv("MISSM_GEMM_8P_TN_MINK") ? atoi(getenv("MISSM_GEMM_8P_TN_MINK")) : 4096;
  if (use8p_tn && dtype == kBF16 && trans_a && trans_b && out_f32 && !resid && act == MISSM_ACT_NONE && !bias && M % 128 == 0 &&
      N % 128 == 0 && K >= tn_min_k && (K >= 20000 || ((M + 255) / 256) * ((N + 255) / 256) * (ngroups > 1 ? ngroups : 1) >= 18) && (size_t)K * lda * 2 < (size_t(1) << 32) && (size_t)K * ldb * 2 < (size_t(1) << 32)) {
    const int tm2 = (M + 255) / 256, tn2 = (N + 255) / 256, t2 = tm2 * tn2;
    // K slices: minimise  rounds of 256 workgroups x (K tiles per slice x ~1.4 us + ~9 us of prologue / park)  + the reduce.
    // One slice per 256 / tiles is the optimum when it fills a round exactly (video tower: 27 tiles x 9 slices), but four grouped
    // [3072 x 768] gradients are 144 tiles: one slice leaves 44 % of the CUs idle for the whole reduction, three slices are two
    // full rounds of a third of the length.
    int sp = 1, kps8 = (K + 127) / 128 * 128;
    {
      const int wgs1 = t2 * g.ngroups;
      float best = 1e30f;
      for (int c = 1; c <= 64; ++c) {
        int kc = ((K + c - 1) / c + 127) / 128 * 128;
        if (kc < 1024) kc = 1024;
        const int cs = (K + kc - 1) / kc;                 // slices actually needed at this slice length
        if (cs != c && c > 1) continue;
        const int rounds = (wgs1 * cs + 255) / 256;
        // (the reduce re-reads one 256 KiB fp32 tile per workgroup: ~0.055 us each at 5 TB/s)
        const float cost = rounds * ((kc / 64) * 1.4f + 9.0f) + (cs > 1 ? 3.0f + 0.055f * (wgs1 * cs) : 0.0f);
        if (cost < best) { best = cost; sp = cs; kps8 = kc; }
      }
    }
    g.tiles_m = tm2; g.tiles_n = tn2; g.splitk = sp; g.k_per_split = kps8;
    g.group_m = group_m_env > 0 ? group_m_env : (tn2 >= 4 ? 8 : 1);
    static const int tn_order_env = getenv("MISSM_GEMM_TN_ORDER") ? atoi(getenv("MISSM_GEMM_TN_ORDER")) : 1;
    g.tn_order = tn_order_env;
    if (splitk_workspace(stream, (size_t)g.ngroups * sp * t2 * (256 * 256 * sizeof(float)), &g.ws)) {
      missm_set_error("gemm: cannot allocate the split-K workspace");
      return MISSM_ERR_LAUNCH;
    }
    auto kt = use8p_tn == 2 ? gemm8p_tn_kernel<false> : gemm8p_tn_kernel<true>;
    static bool attr_tn[2] = {false, false};
    if (!attr_tn[use8p_tn == 2]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
        missm_set_error("gemm: cannot raise dynamic LDS to 128 KiB");
        return MISSM_ERR_LAUNCH;
      }
      attr_tn[use8p_tn == 2] = true;
    }
    log_shape(M);
    hipLaunchKernelGGL(kt, dim3(g.ngroups * t2 * sp), dim3(512), 128 * 1024, s, g);
    hipLaunchKernelGGL(splitk_reduce8p_kernel, dim3(g.ngroups * t2 * 16), dim3(512), 0, s, g);
    return missm_check_launch("gemm8p_tn");
  }
  if (ngroups > 1 && (trans_a || trans_b)) return MISSM_GROUPED_UNAVAILABLE;
  // ---- 256x128 tile, 4 waves, two workgroups per CU (gemm4w.h) - OPT-IN.  Measured inside the step against the 8-phase kernel
  // (profiles/r02_gemm_insitu_8p_vs_4w.txt, one stream): it wins where the 256x256 grid leaves partial rounds or the epilogue
  // weighs most - every grouped launch of the four 197-token towers (N = 768: -8..-18 %, N = 2304 / 3072: 0..-6 %) and the video
  // tower's out-projection shapes (N = 768, K = 768: -7..-10 %); it ties on the K = 768 / N >= 2304 video shapes and loses 2-4 %
  // at K = 2304 / 3072 (8 % at 4096^3), where the main loop dominates and the big tile halves the L2 -> LDS bytes per flop.
  // That rule (MISSM_GEMM_4W=3) takes 1.2 ms out of the 52 ms of serial GEMM time - and makes the real two-stream step 0.75 %
  // SLOWER (461.9 vs 465.2 samples/s, three alternating runs on one box): what it gains alone - filled partial rounds, hidden
  // epilogues - the second stream already provides, and 80 KiB workgroups of one lane fragment the CUs that the other lane's
  // 128 KiB workgroups need whole.  Default therefore 0.  MISSM_GEMM_4W: 0 never, 1 wherever legal, 2 whenever K <= 1024,
  // 3 the single-stream rule.
  static const int use4w = getenv("MISSM_GEMM_4W") ? atoi(getenv("MISSM_GEMM_4W")) : 0;
  // The row remainder of a split product (27 of the video tower's 197 tile rows at N = 768) is a single partial round whatever runs it:
  // there the 256x128 kernel replaces the 128x128 one (MISSM_GEMM_TAIL4W=0 restores it).
  static const int tail4w = getenv("MISSM_GEMM_TAIL4W") ? atoi(getenv("MISSM_GEMM_TAIL4W")) : 1;
  // ... and so does an ungrouped product that the 256x256 kernel would not take because its grid fills the chip badly (a single
  // 197-token tower: 75 big tiles at N = 768) but that still has >= 48 tiles of 256x128: it replaces the 128x128 kernel there too.
  bool small4w = false;
  if (tail4w && ngroups <= 1 && !g_row_remainder) {
    const int t2b = ((M + 255) / 256) * ((N + 255) / 256), rb = (t2b + 255) / 256;
    static const int eff4 = getenv("MISSM_GEMM_BIG_EFF") ? atoi(getenv("MISSM_GEMM_BIG_EFF")) : 75;
    small4w = !(t2b >= 192 && t2b * 100 >= rb * 256 * eff4) && ((M + 255) / 256) * ((N + 127) / 128) >= 48;
  }
  // ... and the UNGROUPED products with K <= 1024 and N <= 1024 (the video tower's out-projections and their dX: three tile columns, an
  // epilogue-heavy K = 768 main loop): on the big lane alone the 256x128 kernel pays inside the two-stream step too, +0.6 %
  // (456.8 vs 454.1 samples/s, three alternating runs) - it was putting the small towers' grouped launches on it that cost.
  static const int out4w_env = getenv("MISSM_GEMM_OUT4W") ? atoi(getenv("MISSM_GEMM_OUT4W")) : 1;
  const bool out4w = out4w_env && tail4w && ngroups <= 1 && !g_row_remainder && K <= 1024 && N <= 1024;
  const bool remainder4w = (tail4w && g_row_remainder && ngroups <= 1) || small4w || out4w;
  const bool rule4w = remainder4w || use4w == 1 || (use4w == 2 && K <= 1024) || (use4w == 3 && (ngroups > 1 || (K <= 1024 && N <= 1024)));
  if (rule4w && dtype == kBF16 && !trans_a && !trans_b && splitk == 1 && g.vec_ok && N % 64 == 0 && !accumulate && K % 64 == 0 && K >= 128 &&
      (act == MISSM_ACT_NONE || act == MISSM_ACT_QGELU || act == MISSM_ACT_DQGELU) &&
      (size_t)lda * 2 * 128 < (size_t(1) << 31) && (size_t)ldb * 2 * 128 < (size_t(1) << 31) &&
      ((M + 255) / 256) * ((N + 127) / 128) * g.ngroups >= (out4w && !small4w ? 256 : (remainder4w ? 32 : 256))) {
    g.group_tiles_m = (M + 255) / 256;
    g.tiles_m = g.group_tiles_m * g.ngroups; g.tiles_n = (N + 127) / 128;
    g.group_m = group_m_env > 0 ? group_m_env : (g.tiles_n >= 8 ? 8 : 1);
    static bool attr4 = false;
    if (!attr4) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
        missm_set_error("gemm: cannot raise dynamic LDS to 80 KiB");
        return MISSM_ERR_LAUNCH;
      }
      attr4 = true;
    }
    // MISSM_GEMM_4WC=1: the resident, K-continuous form of this kernel (gemm4w.h: gemm4wc_kernel) where the grid has more than 512 tiles.
    // (A first resident form that requested the next tile's five prologue half tiles in front of the epilogue and drew tiles from the
    //  per-XCD queues was parity-green and SLOWER than one workgroup per tile - QKV 182 vs 174 us, fc1 + QuickGELU 311 vs 301, same box:
    //  twenty requests per wave queued in front of the sixteen stores, which took 6.5 instead of 4.5 us to issue.  Removed.)
    static const int cont4w = getenv("MISSM_GEMM_4WC") ? atoi(getenv("MISSM_GEMM_4WC")) : 0;
    const int nwg4 = g.tiles_m * g.tiles_n;
    log_shape(M);
    if (cont4w && nwg4 > 512 && K >= 192) {    // resident grid, K loop continuous across tiles (gemm4wc_kernel)
      static bool attr4c = false;
      if (!attr4c) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4wc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
          missm_set_error("gemm: cannot raise dynamic LDS to 80 KiB");
          return MISSM_ERR_LAUNCH;
        }
        attr4c = true;
      }
      hipLaunchKernelGGL(gemm4wc_kernel, dim3(512), dim3(256), 80 * 1024, s, g);
      return missm_check_launch("gemm4wc");
    }
    hipLaunchKernelGGL(gemm4w_kernel, dim3(nwg4), dim3(256), 80 * 1024, s, g);
    return missm_check_launch("gemm4w");
  }
  dim3 grid(tiles * splitk), block(GEMM_THREADS);
  // ---- 256x256 tile (16 waves, one workgroup per CU) for long-M NT products whose tile grid fills whole rounds of 256 CUs
  if (dtype == kBF16 && !trans_a && !trans_b && splitk == 1 && g.vec_ok && N % 64 == 0 && !accumulate && big_env != 0 &&
      (act == MISSM_ACT_NONE || act == MISSM_ACT_QGELU || act == MISSM_ACT_DQGELU)) {
    const int tm2 = (M + 255) / 256, tn2 = (N + 255) / 256, t2 = tm2 * tn2 * g.ngroups;
    const int rounds = (t2 + 255) / 256;
    static const int eff_env = getenv("MISSM_GEMM_BIG_EFF") ? atoi(getenv("MISSM_GEMM_BIG_EFF")) : 75;
    // >= 75 % of the last-round-padded grid is real work; a grouped launch replaces `ngroups` under-filled 128x128 grids (300 small
    // tiles on 512 slots each for N = 768), so it pays from 55 % on (N = 768: 300 big tiles = 1.17 rounds, twice as fast)
    const bool fills = t2 >= 192 && t2 * 100 >= rounds * 256 * (g.ngroups > 1 ? 55 : eff_env);
    if (big_env == 1 || fills) {
      // A last round that would be mostly idle goes to the 128x128 kernel instead: the big tiles take the tile rows that fill
      // whole rounds of 256 CUs, the remaining rows are a second, small launch (e.g. N = 768: 591 big tiles = 2.3 rounds ->
      // 510 big tiles + 324 small ones).
      const int full = t2 / 256, rem = t2 - full * 256;
      int m_big = M;
      if (big_env != 1 && full >= 2 && rem > 0 && rem < 160 && g.ngroups == 1) m_big = (full * 256 / tn2) * 256;
      if (m_big < M) {
        const size_t esz = 2, csz = out_f32 ? 4 : 2;
        g_row_remainder = 1;
        int rc = missm_gemm(static_cast<const char*>(A) + (size_t)m_big * lda * esz, B, static_cast<char*>(C) + (size_t)m_big * ldc * csz,
                            M - m_big, N, K, lda, ldb, ldc, 0, 0, alpha, bias, resid ? resid + (size_t)m_big * ldc : nullptr,
                            aux_in ? static_cast<const char*>(aux_in) + (size_t)m_big * ldaux * esz : nullptr,
                            aux_out ? static_cast<char*>(aux_out) + (size_t)m_big * ldaux * esz : nullptr, ldaux, act, out_f32,
                            accumulate, 1, nullptr, dtype, stream);
        g_row_remainder = 0;
        if (rc) return rc;
        g.M = m_big;
      }
      const int tmb = ((g.M + 255) / 256) * g.ngroups;
      g.group_tiles_m = (g.M + 255) / 256;
      g.tiles_m = tmb; g.tiles_n = tn2;
      g.group_m = group_m_env > 0 ? group_m_env : (tn2 >= 4 ? 8 : 1);
      // 8-wave / 8-phase pipeline (gemm8p.h): whole pairs of K tiles, vector epilogue only
      static const int use8p = getenv("MISSM_GEMM_8P") ? atoi(getenv("MISSM_GEMM_8P")) : 1;
      if (use8p && K % 128 == 0 && (size_t)lda * 2 * 128 < (size_t(1) << 31) && (size_t)ldb * 2 * 256 < (size_t(1) << 31)) {
        auto k8 = use8p == 2 ? gemm8p_kernel<false> : gemm8p_kernel<true>;
        static bool attr8[2] = {false, false};
        if (!attr8[use8p == 2]) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 16) != hipSuccess) {
            missm_set_error("gemm: cannot raise dynamic LDS to 128 KiB");
            return MISSM_ERR_LAUNCH;
          }
          attr8[use8p == 2] = true;
        }
        // MISSM_GEMM_PERSIST: 0 one workgroup per tile; 1 one workgroup per CU walks the tiles b, b + 256, ... and requests the next
        // tile's first half tiles before its epilogue; 2 (default, round 3) the same grid DRAWS its tiles from per-XCD queues.
        // Round 2 measured 1 against 0 at +-3 % and left it off: its tile-top wait (vmcnt(6)) sat behind every store of the epilogue.
        // With the epilogue's operations counted into that wait and the dynamic order (in-kernel stamps, tools/gemm_timeline2.py:
        // the relaunch gap of 2.4 - 4.8 us per tile becomes 0.9 us, a slow epilogue costs its CU a tile instead of holding the
        // launch): QKV 174 - 181 -> 167 - 169 us, fc1 (no activation) 242 -> 228 - 235, four-tower QKV 93 -> 88, dX of fc2 167 -> 154,
        // everything else within +-2 %; the two-stream step 67.24 / 67.03 -> 66.35 / 66.54 ms (alternating runs on one box).
        static const int persist = getenv("MISSM_GEMM_PERSIST") ? atoi(getenv("MISSM_GEMM_PERSIST")) : 2;
        static int ncu = 0;
        if (ncu == 0) {
          int dev = 0; hipDeviceProp_t prop;
          if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { missm_set_error("gemm: cannot query the device"); return MISSM_ERR_LAUNCH; }
          ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        const int nwg = persist ? (tmb * tn2 < ncu ? tmb * tn2 : ncu) : tmb * tn2;
        if (persist == 2 && tmb * tn2 > ncu && ncu % 8 == 0 && tile_queue_counters(stream, &g.sched)) {
          missm_set_error("gemm: cannot allocate the tile-queue counters");
          return MISSM_ERR_LAUNCH;
        }
        static const char* desync_env = getenv("MISSM_GEMM_DESYNC");
        if (desync_env && g.sched) {
          int ph = 0, st = 0;
          if (sscanf(desync_env, "%d,%d", &ph, &st) == 2) { g.desync_phases = ph; g.desync_step = st; }
        }
        log_shape(g.M);
        hipLaunchKernelGGL(k8, dim3(nwg), dim3(512), 128 * 1024 + 16, s, g);
        return missm_check_launch("gemm8p");
      }
      if (ngroups > 1) return MISSM_GROUPED_UNAVAILABLE;
      log_shape(g.M);
      auto k = gemm_kernel<bf16, false, false, 128, 1, 4>;
      static bool attr_set = false;
      if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
          missm_set_error("gemm: cannot raise dynamic LDS to 128 KiB");
          return MISSM_ERR_LAUNCH;
        }
        attr_set = true;
      }
      // (s_setprio around the MFMA cluster: +4 % at K = 2304 / 3072 on this tile, -4..5 % at K = 768 - four waves per SIMD
      //  already interleave; the opposite of the 128x128 kernel, where it helps the short-K shapes)
      if (variant == 0 || (variant < 0 && K <= 1024)) {
        auto k0 = gemm_kernel<bf16, false, false, 128, 0, 4>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipLaunchKernelGGL(k0, dim3(tmb * tn2), dim3(1024), 128 * 1024, s, g);
        return missm_check_launch("gemm256");
      }
      hipLaunchKernelGGL(k, dim3(tmb * tn2), dim3(1024), 128 * 1024, s, g);
      return missm_check_launch("gemm256");
    }
  }
  if (ngroups > 1) return MISSM_GROUPED_UNAVAILABLE;
  // (a 32-deep K tile with 4 workgroups per CU was measured too: -3..20 % once the epilogue was compact; removed)
  // scheduling variant of the 128x128 kernel (measured, random data): s_setprio around the MFMA cluster is worth +4..10 % on
  // the K = 768 shapes; requesting all fragments of the K tile up front is worth +10 % at long K (968 vs 878 TFLOP/s at 4096^3)
  // k-major (dW / NN) forms: variant 1 (per-K-step fragments through untracked reads behind an early LDS-DMA, 147 VGPRs) is
  // 3-6 % slower than variant 2 (all fragments first, LDS-DMA after them, 173 VGPRs) when the GEMM has the chip to itself,
  // but the whole step is 0.85 % faster with it when two streams share the chip (385.7 vs 382.5 samples/s, same box)
  static const int var_tn = getenv("MISSM_GEMM_VAR_TN") ? atoi(getenv("MISSM_GEMM_VAR_TN")) : 1;
  const int var = variant >= 0 ? variant : ((trans_a || trans_b) ? var_tn : (K > 1024 ? 2 : 1));
#define MISSM_GEMM_LAUNCH(T, TA, TB)                                                                       \
  do {                                                                                                     \
    if (var == 1) hipLaunchKernelGGL((gemm_kernel<T, TA, TB, 128, 1>), grid, block, 64 * 1024, s, g);  \
    else if (var == 2) hipLaunchKernelGGL((gemm_kernel<T, TA, TB, 128, 2>), grid, block, 64 * 1024, s, g);  \
    else hipLaunchKernelGGL((gemm_kernel<T, TA, TB, 128>), grid, block, 64 * 1024, s, g);                   \
  } while (0)
  log_shape(M);
  if (dtype == kBF16) {
    if (!trans_a && !trans_b) MISSM_GEMM_LAUNCH(bf16, false, false);
    else if (!trans_a && trans_b) MISSM_GEMM_LAUNCH(bf16, false, true);
    else MISSM_GEMM_LAUNCH(bf16, true, true);
  } else {
    if (!trans_a && !trans_b) MISSM_GEMM_LAUNCH(float, false, false);
    else if (!trans_a && trans_b) MISSM_GEMM_LAUNCH(float, false, true);
    else MISSM_GEMM_LAUNCH(float, true, true);
  }
#undef MISSM_GEMM_LAUNCH
  if (splitk > 1) {
    if (trans_b) hipLaunchKernelGGL((splitk_reduce_kernel<true, 2>), dim3(tiles * 4), block, 0, s, g);
    else hipLaunchKernelGGL((splitk_reduce_kernel<false, 2>), dim3(tiles * 4), block, 0, s, g);
  }
  return missm_check_launch("gemm");
}

extern "C" int missm_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                          int trans_b, float alpha, const float* bias, const float* resid, const void* aux_in, void* aux_out,
                          int ldaux, int act, int out_f32, int accumulate, int splitk, float* colsum_a, int dtype, void* stream) {
  return gemm_core(A, B, C, M, N, K, lda, ldb, ldc, trans_a, trans_b, alpha, bias, resid, aux_in, aux_out, ldaux, act, out_f32, accumulate,
                   splitk, colsum_a, dtype, stream, 1, nullptr);
}

extern "C" int missm_gemm_grouped(int ngroups, const void* const* A, const void* const* B, void* const* C, int M, int N, int K, int lda,
                                  int ldb, int ldc, int trans_a, int trans_b, float alpha, const float* const* bias,
                                  const float* const* resid, const void* const* aux_in, void* const* aux_out, int ldaux, int act,
                                  int out_f32, int accumulate, int splitk, float* const* colsum_a, int dtype, void* stream) {
  MISSM_CHECK_ARG(ngroups >= 1 && ngroups <= MISSM_MAX_GROUPS && A && B && C, "gemm_grouped: 1..8 groups");
  GemmArgs::Group grp[MISSM_MAX_GROUPS];
  for (int i = 0; i < ngroups; ++i)
    grp[i] = GemmArgs::Group{A[i], B[i], C[i], bias ? bias[i] : nullptr, resid ? resid[i] : nullptr, aux_in ? aux_in[i] : nullptr,
                             aux_out ? aux_out[i] : nullptr, colsum_a ? colsum_a[i] : nullptr};
  if (ngroups > 1) {
    const int rc = gemm_core(grp[0].A, grp[0].B, grp[0].C, M, N, K, lda, ldb, ldc, trans_a, trans_b, alpha, grp[0].bias, grp[0].resid,
                             grp[0].aux_in, grp[0].aux_out, ldaux, act, out_f32, accumulate, splitk, grp[0].colsum_a, dtype, stream, ngroups, grp);
    if (rc != MISSM_GROUPED_UNAVAILABLE) return rc;
  }
  for (int i = 0; i < ngroups; ++i) {      // no grouped kernel for this shape / layout / dtype: one launch per problem
    const int rc = gemm_core(grp[i].A, grp[i].B, grp[i].C, M, N, K, lda, ldb, ldc, trans_a, trans_b, alpha, grp[i].bias, grp[i].resid,
                             grp[i].aux_in, grp[i].aux_out, ldaux, act, out_f32, accumulate, splitk, grp[i].colsum_a, dtype, stream, 1, nullptr);
    if (rc) return rc;
  }
  return MISSM_OK;
}

extern "C" int missm_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, float alpha,
                             const float* bias, const float* resid, const void* aux_in, void* aux_out, int ldaux, int act,
                             int out_f32, int accumulate, int dtype, void* stream) {
  return missm_gemm(A, B, C, M, N, K, lda, ldb, ldc, 0, 0, alpha, bias, resid, aux_in, aux_out, ldaux, act, out_f32, accumulate, 1,
                    nullptr, dtype, stream);
}

extern "C" int missm_transpose_pad(const void* in, void* out, int R, int C, int ld, int ldo, float* colsum, int dtype,
                                   void* stream) {
  MISSM_CHECK_ARG(R > 0 && C > 0 && ldo >= R, "transpose: bad shape");
  const int rp = (ldo + 63) / 64;
  dim3 grid(rp, (C + 63) / 64), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == kBF16)
    hipLaunchKernelGGL(transpose_pad_kernel<bf16>, grid, block, 0, s, (const bf16*)in, (bf16*)out, R, C, ld, ldo, colsum);
  else
    hipLaunchKernelGGL(transpose_pad_kernel<float>, grid, block, 0, s, (const float*)in, (float*)out, R, C, ld, ldo, colsum);
  return missm_check_launch("transpose_pad");
}

extern "C" int missm_colsum(const void* in, float* out, int R, int C, int ld, int div, int mod, int dtype, void* stream) {
  MISSM_CHECK_ARG(R > 0 && C > 0 && div > 0 && mod > 0, "colsum: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int vec = dtype == kBF16 ? 8 : 4;
  if ((mod == 1 || div >= 32) && C % vec == 0 && ld % vec == 0 && ((uintptr_t)in % 16 == 0)) {
    const int rpb_v = mod == 1 ? 256 : div;        // grouped: one run of `div` rows (one group) per workgroup
    dim3 grid((R + rpb_v - 1) / rpb_v, (C + 32 * vec - 1) / (32 * vec)), block(256);
    if (dtype == kBF16) hipLaunchKernelGGL(colsum_vec_kernel<bf16>, grid, block, 0, s, (const bf16*)in, out, R, C, ld, rpb_v, mod == 1 ? 1 : div, mod);
    else hipLaunchKernelGGL(colsum_vec_kernel<float>, grid, block, 0, s, (const float*)in, out, R, C, ld, rpb_v, mod == 1 ? 1 : div, mod);
    return missm_check_launch("colsum_vec");
  }
  int rpb = 128;
  if (mod > 1 && div > 1) rpb = div;            // one group per block when groups are row runs
  dim3 grid((R + rpb - 1) / rpb, (C + 255) / 256), block(256);
  if (dtype == kBF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, s, (const bf16*)in, out, R, C, ld, div, mod, rpb);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)in, out, R, C, ld, div, mod, rpb);
  return missm_check_launch("colsum");
}

extern "C" int missm_cast_weights_batched(const void* tiles, int ntiles, int dtype, void* stream) {
  MISSM_CHECK_ARG(tiles && ntiles > 0, "cast_weights_batched: empty table");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == kBF16) hipLaunchKernelGGL(cast_weights_batched_kernel<bf16>, dim3(ntiles), dim3(256), 0, s, (const CastTile*)tiles);
  else hipLaunchKernelGGL(cast_weights_batched_kernel<float>, dim3(ntiles), dim3(256), 0, s, (const CastTile*)tiles);
  return missm_check_launch("cast_weights_batched");
}

extern "C" int missm_adam_cast_batched(const void* tiles, int ntiles, long g_off, long m_off, long v_off, int step, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, float grad_scale, int dtype, void* stream) {
  MISSM_CHECK_ARG(tiles && ntiles > 0 && step >= 1, "adam_cast_batched: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  AdamTileArgs a{g_off, m_off, v_off, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == kBF16) hipLaunchKernelGGL(adam_cast_batched_kernel<bf16>, dim3(ntiles), dim3(256), 0, s, (const CastTile*)tiles, a);
  else hipLaunchKernelGGL(adam_cast_batched_kernel<float>, dim3(ntiles), dim3(256), 0, s, (const CastTile*)tiles, a);
  return missm_check_launch("adam_cast_batched");
}

extern "C" int missm_cast_weight(const float* src, void* dst, void* dst_t, int R, int C, int dtype, void* stream) {
  MISSM_CHECK_ARG(R > 0 && C > 0, "cast_weight: bad shape");
  dim3 grid((R + 63) / 64, (C + 63) / 64), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == kBF16) hipLaunchKernelGGL(cast_weight_kernel<bf16>, grid, block, 0, s, src, (bf16*)dst, (bf16*)dst_t, R, C);
  else hipLaunchKernelGGL(cast_weight_kernel<float>, grid, block, 0, s, src, (float*)dst, (float*)dst_t, R, C);
  return missm_check_launch("cast_weight");
}
