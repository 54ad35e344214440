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

namespace missm {

constexpr int BM = 128, BN = 128, RB = 128;  // tile rows / cols / bytes per LDS row (k-contiguous operands)
constexpr int GEMM_THREADS = 256;

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
  int splitk, k_per_split;  // splitk > 1: each K slice atomically adds its partial into the (zeroed) fp32 C
};

// swizzle key of a k-major ("transposed") tile row: the 8 rows touched by one half-wave of a transposed fragment read
// ({8g+q} and {8g+8+q}, q = 0..3) get 8 distinct keys -> 8 distinct 32-byte slots of the 256-byte bank row.
__device__ __forceinline__ int tkey(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
template <int RBT> __device__ __forceinline__ int tswz(int row, int chunk16) { return row * RBT + ((chunk16 ^ (tkey(row) << 1)) << 4); }

// fragment (one MFMA step) of a k-major tile [BK rows of k][128 columns]: lane (i, g) gets column c0 + i, its KPL k values
template <typename T> struct KMajorFrag;
template <> struct KMajorFrag<bf16> {
  static constexpr int RBT = 256;
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
};
template <> struct KMajorFrag<float> {
  static constexpr int RBT = 512;
  __device__ static __forceinline__ f32x4 load(const char* lds, int ks, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4, col = c0 + i;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float*>(lds + tswz<RBT>(ks * 16 + 4 * g + j, col >> 2) + ((col & 3) << 2));
    return v;
  }
};

// TA: A is stored [K, M] (reduction index on rows); TB: B is stored [K, N].  TA = TB = false is the NT form.
template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(GemmArgs g) {
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int EPC = 16 / sizeof(T);     // elements per 16-byte chunk
  constexpr int BK = RB / sizeof(T);      // k elements per tile (64 bf16 / 32 f32)
  constexpr int KSTEPS = BK / M_::KS;     // 2
  constexpr int RBT = KMajorFrag<T>::RBT; // bytes per row of a k-major tile (128 elements)
  constexpr int NCT = RBT / 16;           // 16-byte chunks per k-major row
  constexpr int TILE_BYTES = BM * RB;     // 16 KiB either layout
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];
  char* ldsA = lds;                       // [2][TILE_BYTES]
  char* ldsB = lds + 2 * TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntiles = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int split = bid / ntiles;
  const int logical = xcd_remap(bid - split * ntiles, ntiles);
  const int tm = logical / g.tiles_n, tn = logical % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const T* __restrict__ A = static_cast<const T*>(g.A);
  const T* __restrict__ B = static_cast<const T*>(g.B);

  // ---- staging: 4 x 16-byte chunks of A and of B per thread per K tile ----
  int a_lds[4], b_lds[4];
  const T* a_ptr[4];
  const T* b_ptr[4];
  bool a_ok[4], b_ok[4];
  int a_kofs[4], b_kofs[4];   // k offset (inside the tile) this chunk covers, for the K-edge predicate
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (!TA) {
      const int r = (tid >> 3) + 32 * i, c = tid & 7;
      int gr = m0 + r; gr = gr < g.M ? gr : g.M - 1;
      a_ptr[i] = A + (size_t)gr * g.lda + c * EPC;
      a_lds[i] = swz<RB>(r, c * 16);
      a_ok[i] = true; a_kofs[i] = c * EPC;
    } else {
      const int idx = tid + 256 * i, r = idx / NCT, c = idx % NCT;
      a_ptr[i] = A + (size_t)r * g.lda + m0 + c * EPC;
      a_lds[i] = tswz<RBT>(r, c);
      a_ok[i] = (m0 + c * EPC) < g.M; a_kofs[i] = r;
    }
    if constexpr (!TB) {
      const int r = (tid >> 3) + 32 * i, c = tid & 7;
      // B tile row r (output column n0 + r) goes to LDS row pi(r) = 64*(r/64) + 16*(r%4) + (r%64)/4
      const int lr = (r & 64) + ((r & 3) << 4) + ((r & 63) >> 2);
      int gn = n0 + r; gn = gn < g.N ? gn : g.N - 1;
      b_ptr[i] = B + (size_t)gn * g.ldb + c * EPC;
      b_lds[i] = swz<RB>(lr, c * 16);
      b_ok[i] = true; b_kofs[i] = c * EPC;
    } else {
      const int idx = tid + 256 * i, r = idx / NCT, c = idx % NCT;
      b_ptr[i] = B + (size_t)r * g.ldb + n0 + c * EPC;
      b_lds[i] = tswz<RBT>(r, c);
      b_ok[i] = (n0 + c * EPC) < g.N; b_kofs[i] = r;
    }
  }

  u32x4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ain = a_ok[i] && (k0 + a_kofs[i]) < kend;
      const bool bin = b_ok[i] && (k0 + b_kofs[i]) < kend;
      ra[i] = u32x4{0, 0, 0, 0};
      rb[i] = u32x4{0, 0, 0, 0};
      if (ain) ra[i] = *reinterpret_cast<const u32x4*>(TA ? a_ptr[i] + (size_t)k0 * g.lda : a_ptr[i] + k0);
      if (bin) rb[i] = *reinterpret_cast<const u32x4*>(TB ? b_ptr[i] + (size_t)k0 * g.ldb : b_ptr[i] + k0);
    }
  };
  auto lstore = [&](int buf) {
    char* la = ldsA + buf * TILE_BYTES;
    char* lb = ldsB + buf * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(la + a_lds[i]) = ra[i];
      *reinterpret_cast<u32x4*>(lb + b_lds[i]) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;
  gload(kbeg);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);   // next tile's loads fly under this tile's MFMAs
    const char* la = ldsA + buf * TILE_BYTES;
    const char* lb = ldsB + buf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      Frag fa[4], fb[4];
      const int cbyte = (ks * 4 + lg) * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (!TA) fa[i] = lds_frag<T>(la, swz<RB>(wm * 64 + i * 16 + li, cbyte));
        else fa[i] = KMajorFrag<T>::load(la, ks, wm * 64 + i * 16, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (!TB) fb[j] = lds_frag<T>(lb, swz<RB>(wn * 64 + j * 16 + li, cbyte));
        else fb[j] = KMajorFrag<T>::load(lb, ks, wn * 64 + j * 16, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = M_::step(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  // !TB: lane owns rows 4*lg + r of each 16-row tile i and the 4 CONSECUTIVE columns 4*li + j  (vector accesses)
  //  TB: lane owns column 16*j + li of each n-tile j                                          (scalar accesses)
  const bool first_split = split == 0;
  auto finish = [&](float x, size_t off, size_t aoff) -> void {   // scalar tail of the epilogue for one element
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
      if (g.splitk > 1) { atomicAdd(c, x); return; }
      if (g.resid) x += g.resid[off];
      if (g.accumulate) x += *c;
      *c = x;
    } else {
      static_cast<T*>(g.C)[off] = from_f32<T>(x);
    }
  };

  if constexpr (TB) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + li;
      if (col >= g.N) continue;
      const float bv = (g.bias && first_split) ? g.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * 64 + i * 16 + lg * 4 + r;
          if (row < g.M) finish(acc[i][j][r] * g.alpha + bv, (size_t)row * g.ldc + col, (size_t)row * g.ldaux + col);
        }
    }
    return;
  } else {
    const int col = n0 + wn * 64 + li * 4;
    if (col >= g.N) return;
    const bool full4 = (col + 3 < g.N) && g.vec_ok && g.splitk == 1;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && first_split) {
      if (col + 3 < g.N) bias4 = load4(g.bias + col);
      else
        for (int j = 0; j < 4; ++j) if (col + j < g.N) bias4[j] = g.bias[col + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + lg * 4 + r;
        if (row >= g.M) continue;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[i][j][r] * g.alpha + bias4[j];
        const size_t off = (size_t)row * g.ldc + col;
        const size_t aoff = (size_t)row * g.ldaux + col;
        if (full4) {
          if (g.act == MISSM_ACT_QGELU || g.act == MISSM_ACT_GELU) {
            if (g.aux_out) store4(static_cast<T*>(g.aux_out) + aoff, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (g.act == MISSM_ACT_QGELU) ? quick_gelu(v[j]) : gelu_erf(v[j]);
          } else if (g.act == MISSM_ACT_DQGELU || g.act == MISSM_ACT_DGELU) {
            f32x4 u = load4(static_cast<const T*>(g.aux_in) + aoff);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= (g.act == MISSM_ACT_DQGELU) ? quick_gelu_grad(u[j]) : gelu_erf_grad(u[j]);
          } else if (g.act == MISSM_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if (g.out_f32) {
            float* c = static_cast<float*>(g.C) + off;
            if (g.resid) { f32x4 q = load4(g.resid + off); v += q; }
            if (g.accumulate) { f32x4 q = load4(c); v += q; }
            store4(c, v);
          } else {
            store4(static_cast<T*>(g.C) + off, v);
          }
        } else {
          for (int j = 0; j < 4; ++j)
            if (col + j < g.N) finish(v[j], off + j, aoff + j);
        }
      }
    }
  }
}

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
  if (mod == 1) {
    float s = 0.f;
    for (int r = rbeg; r < rend; ++r) s += to_f32(in[(size_t)r * ld + c]);
    atomicAdd(out + c, s);
  } else {
    int cur = -1; float s = 0.f;
    for (int r = rbeg; r < rend; ++r) {
      const int gidx = (r / div) % mod;
      if (gidx != cur) { if (cur >= 0) atomicAdd(out + (size_t)cur * C + c, s); cur = gidx; s = 0.f; }
      s += to_f32(in[(size_t)r * ld + c]);
    }
    if (cur >= 0) atomicAdd(out + (size_t)cur * C + c, s);
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

}  // namespace missm

using namespace missm;

extern "C" int missm_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                          int trans_b, float alpha, const float* bias, const float* resid, const void* aux_in, void* aux_out,
                          int ldaux, int act, int out_f32, int accumulate, int splitk, int dtype, void* stream) {
  MISSM_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem");
  MISSM_CHECK_ARG(dtype == kBF16 || dtype == kF32, "gemm: dtype must be 0 (f32) or 1 (bf16)");
  const int epc = dtype == kBF16 ? 8 : 4;
  MISSM_CHECK_ARG(lda % epc == 0 && ldb % epc == 0, "gemm: lda, ldb must be multiples of 16 bytes");
  MISSM_CHECK_ARG(trans_a ? (M % epc == 0) : (K % epc == 0), "gemm: the contiguous extent of A must be a multiple of 16 bytes");
  MISSM_CHECK_ARG(trans_b ? (N % epc == 0) : (K % epc == 0), "gemm: the contiguous extent of B must be a multiple of 16 bytes");
  MISSM_CHECK_ARG(!(trans_a && !trans_b), "gemm: A^T with k-contiguous B is not instantiated");
  MISSM_CHECK_ARG(!(resid && !out_f32) && !(accumulate && !out_f32), "gemm: resid/accumulate need out_f32");
  MISSM_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0), "gemm: unaligned operand");
  MISSM_CHECK_ARG(!bias || ((uintptr_t)bias % 16 == 0), "gemm: bias must be 16-byte aligned");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha;
  g.bias = bias; g.resid = resid; g.aux_in = aux_in; g.aux_out = aux_out; g.ldaux = ldaux; g.act = act;
  g.out_f32 = out_f32; g.accumulate = accumulate;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  g.vec_ok = (ldc % 4 == 0) && (ldaux % 4 == 0) && ((uintptr_t)C % 16 == 0) && ((uintptr_t)resid % 16 == 0) &&
             ((uintptr_t)aux_in % 16 == 0) && ((uintptr_t)aux_out % 16 == 0);
  const int bk = dtype == kBF16 ? 64 : 32;
  const int tiles = g.tiles_m * g.tiles_n;
  if (splitk <= 0) {  // auto: only worth it when the tile grid cannot fill the 256 CUs at 2 workgroups each
    splitk = 1;
    if (out_f32 && !resid && !accumulate && act == MISSM_ACT_NONE && tiles < 256) {
      splitk = 512 / tiles;
      const int maxs = K / (8 * bk);
      if (splitk > maxs) splitk = maxs;
      if (splitk < 1) splitk = 1;
    }
  }
  MISSM_CHECK_ARG(splitk == 1 || (out_f32 && !resid && !accumulate && act == MISSM_ACT_NONE),
                  "gemm: split-K needs a zero-initialised fp32 output and no epilogue");
  int kps = (K + splitk - 1) / splitk;
  kps = (kps + bk - 1) / bk * bk;
  splitk = (K + kps - 1) / kps;
  g.splitk = splitk; g.k_per_split = kps;
  dim3 grid(tiles * splitk), block(GEMM_THREADS);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MISSM_GEMM_LAUNCH(T, TA, TB) hipLaunchKernelGGL((gemm_kernel<T, TA, TB>), grid, block, 0, s, g)
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
  return missm_check_launch("gemm");
}

extern "C" int missm_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, float alpha,
                             const float* bias, const float* resid, const void* aux_in, void* aux_out, int ldaux, int act,
                             int out_f32, int accumulate, int dtype, void* stream) {
  return missm_gemm(A, B, C, M, N, K, lda, ldb, ldc, 0, 0, alpha, bias, resid, aux_in, aux_out, ldaux, act, out_f32, accumulate, 1,
                    dtype, stream);
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
  int rpb = 128;
  if (mod > 1 && div > 1) rpb = div;            // one group per block when groups are row runs
  dim3 grid((R + rpb - 1) / rpb, (C + 255) / 256), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == kBF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, s, (const bf16*)in, out, R, C, ld, div, mod, rpb);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)in, out, R, C, ld, div, mod, rpb);
  return missm_check_launch("colsum");
}

extern "C" int missm_cast_weight(const float* src, void* dst, void* dst_t, int R, int C, int dtype, void* stream) {
  MISSM_CHECK_ARG(R > 0 && C > 0, "cast_weight: bad shape");
  dim3 grid((R + 63) / 64, (C + 63) / 64), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == kBF16) hipLaunchKernelGGL(cast_weight_kernel<bf16>, grid, block, 0, s, src, (bf16*)dst, (bf16*)dst_t, R, C);
  else hipLaunchKernelGGL(cast_weight_kernel<float>, grid, block, 0, s, src, (float*)dst, (float*)dst_t, R, C);
  return missm_check_launch("cast_weight");
}
