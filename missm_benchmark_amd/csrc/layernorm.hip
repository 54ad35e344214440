// LayerNorm forward/backward (eps 1e-5, affine) on the fp32 residual stream, one wavefront per row.
// Replaces torch's ATen LayerNorm at the reference sites pre_layrnorm / layer_norm1/2 / temporal_layer_norm1 /
// post_layernorm / final_layer_norm (languagebind/image/modeling_image.py:70,72,82,465,604,606) and the fusion
// LayerNorm (src/model/baseline.py:48).  HBM-bound: the row lives in registers (one 16-byte load per lane per
// 256 columns), statistics by xor-shuffles across the 64 lanes, two-pass variance like the reference.
//
// Fusions:  * optional per-row additive vector written back to the stream (temporal_embedding add,
//             image/modeling_image.py:110-114, without materialising the '(b t) n d <-> (b n) t d' shuffles)
//           * optional row gather on input (CLS / EOT pooling, :658-662, :519-522)
//           * backward accumulates straight into the fp32 residual-gradient stream and reduces
//             dgamma/dbeta per workgroup before one 256-byte-contiguous atomic per wave.
#include "common.h"
#include <stdlib.h>
#include "missm_internal.h"
#include <type_traits>

namespace missm {

struct LnFwdArgs {
  const float* x;        // [*, cols] fp32 input rows
  float* x_wb;           // if add != null: x + add is written back here (may alias x)
  const float* add;      // [mod, cols] or null ; index (row / div) % mod
  int div, mod;
  int in_mul;            // input row = row * in_mul + (in_off ? in_off[row] : 0)
  const int* in_off;
  const float* gamma; const float* beta;
  void* y;               // [rows, cols] Tout
  float* mean; float* rstd;  // [rows] (nullable)
  int rows, cols; float eps;
};

// [r3] The rows can be grid-strided with a one-row look-ahead like the backward (MISSM_LN_FWD_BLOCKS caps the grid).  Measured inside
// the step (alternating runs on one box): one row per wave - 12 608 four-row workgroups per video launch, the default - 67.18 / 67.20 ms,
// capped at 1024 / 2048 / 4096 workgroups 67.31 / 67.40 / 67.28 ms: the forward is not latency-bound the way the backward was (8 waves
// per SIMD already keep 96 KB per CU in flight), so the cap stays off.  gamma / beta are loaded once per wave.  The look-ahead is guarded
// (row + step < rows): every index tensor (in_off) is read for valid rows only - the lesson of the round-2 abort.
template <typename Tout, int CH>
__device__ __forceinline__ void ln_fwd_body(const LnFwdArgs& a, int bx, int gx) {
  const int lane = threadIdx.x & 63;
  const int row0 = bx * 4 + (threadIdx.x >> 6), rstep = gx * 4;
  if (row0 >= a.rows) return;
  f32x4 gm[CH], bt[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (c * 64 + lane) * 4;
    gm[c] = col < a.cols ? load4(a.gamma + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    bt[c] = col < a.cols ? load4(a.beta + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 nv[CH];
  auto issue = [&](int row) {
    const size_t irow = (size_t)row * a.in_mul + (a.in_off ? a.in_off[row] : 0);
    const float* xr = a.x + irow * a.cols;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * 64 + lane) * 4;
      // (the residual stream is read once here and next by this LayerNorm's backward, a whole step later: streaming load -
      //  64.56 / 64.38 / 64.39 vs 64.81 / 64.63 / 64.55 ms per step with both LayerNorm kernels on it, alternating runs)
      nv[c] = col < a.cols ? load4_nt(xr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  issue(row0);
  for (int row = row0; row < a.rows; row += rstep) {
    const size_t irow = (size_t)row * a.in_mul + (a.in_off ? a.in_off[row] : 0);
    f32x4 v[CH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * 64 + lane) * 4;
      v[c] = nv[c];
      if (col < a.cols) {
        if (a.add) {
          f32x4 t = load4(a.add + (size_t)((row / a.div) % a.mod) * a.cols + col);
          v[c] += t;
          store4(a.x_wb + irow * a.cols + col, v[c]);
        }
        s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
      }
    }
    if (row + rstep < a.rows) issue(row + rstep);
    const float mean = wave_sum(s) / a.cols;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if (col < a.cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[c][j] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / a.cols + a.eps);
    if (lane == 0) {
      if (a.mean) a.mean[row] = mean;
      if (a.rstd) a.rstd[row] = rstd;
    }
    Tout* yr = static_cast<Tout*>(a.y) + (size_t)row * a.cols;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if (col < a.cols) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mean) * rstd * gm[c][j] + bt[c][j];
        store4(yr + col, o);
      }
    }
  }
}

template <typename Tout, int CH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdArgs a) { ln_fwd_body<Tout, CH>(a, blockIdx.x, gridDim.x); }

// Grouped launch: the same LayerNorm site of several shape-identical towers that run in lock-step (towers.forward_lanes) - one grid,
// blockIdx.y selects the tower's pointers (four 6304-row launches of 9-20 us each were latency, not bandwidth).
constexpr int LN_MAX_GROUPS = 8;
struct LnFwdGroups { const float* x[LN_MAX_GROUPS]; const float* gamma[LN_MAX_GROUPS]; const float* beta[LN_MAX_GROUPS]; void* y[LN_MAX_GROUPS];
                     float* mean[LN_MAX_GROUPS]; float* rstd[LN_MAX_GROUPS]; };
template <typename Tout, int CH>
__global__ __launch_bounds__(256) void ln_fwd_grouped_kernel(LnFwdArgs a, LnFwdGroups gs) {
  const int g = blockIdx.y;
  a.x = gs.x[g]; a.gamma = gs.gamma[g]; a.beta = gs.beta[g]; a.y = gs.y[g]; a.mean = gs.mean[g]; a.rstd = gs.rstd[g];
  ln_fwd_body<Tout, CH>(a, blockIdx.x, gridDim.x);
}

struct LnBwdArgs {
  const void* dy;        // Tin rows ; dy row = row / dy_div ; scaled by dy_scale
  int dy_div; float dy_scale;
  const float* x;        // LN input rows (fp32), gathered like the forward (in_mul / in_off)
  int in_mul; const int* in_off;
  const float* mean; const float* rstd; const float* gamma;
  float* dx;             // fp32, same row mapping as x ; accumulate ? += : =
  int accumulate;
  float* dgamma; float* dbeta;  // fp32 [cols], atomically accumulated (caller zeroes)
  int rows, cols;
  void* dx_cast;         // optional Tin copy of the updated dx rows (GEMM operand of the next backward block)
  // optional group sums of the UPDATED dx rows: gsum[(row / gs_div) % gs_mod][col] += dx[row][col] (atomics; caller zeroes) - the
  // temporal-embedding gradient of the video tower (rows of one frame are one run of gs_div = S rows, the frame's time index is its
  // group), which used to be a separate column-sum pass over the 155 MB residual gradient.  Instantiated as GS: a wave then walks the
  // rows of ONE frame (workgroup = 4 waves of one frame, `gridDim.x * 4 / (rows / gs_div)` waves per frame) and carries one sum.
  float* gsum; int gs_div, gs_mod;
  // gs_strided: group = row % gs_mod instead (members gs_mod rows apart): the position-embedding gradient behind pre_layrnorm -
  // d position_embedding[s] = sum over frames of dx[n, s, :] (class_embedding's gradient is row 0 of the same sums)
  int gs_strided;
};

// FAST: cols == CH * 256, accumulate, dx_cast and no row gather - the residual-stream LayerNorms of every tower layer.  With the
// column guards and the uniform branches gone the loop is straight-line code, so the compiler can COUNT the memory operations:
// it waits for the prefetched row with vmcnt(#stores issued after it) instead of vmcnt(0) on the stores' write acknowledgements.
template <typename Tin, int CH, bool FAST, bool GS = false>
__device__ __forceinline__ void ln_bwd_body(const LnBwdArgs& a, int bx, int gx) {
  __shared__ float red[2][4][CH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 ag[CH], ab[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { ag[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 gm[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (c * 64 + lane) * 4;
    gm[c] = (FAST || col < a.cols) ? load4(a.gamma + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv = 1.0f / a.cols;
  // One row per wave per trip, software-pipelined: the NEXT row's x / dy / running-gradient loads are issued before this row's
  // two reductions and stores, so a wave always has a row in flight (one row at a time left the memory pipe idle through every
  // reduce -> store -> address -> load turn-around: 4.3 TB/s; the registers for the second row are free at 3 waves per SIMD).
  using DyRaw = std::conditional_t<std::is_same<Tin, float>::value, f32x4, bf16x4>;   // kept unconverted while in flight
  f32x4 nx[CH], nprev[CH];
  DyRaw ndy[CH];
  float nmean = 0.f, nrstd = 0.f;
  auto issue = [&](int row) {
    const size_t irow = (size_t)row * a.in_mul + ((!FAST && a.in_off) ? a.in_off[row] : 0);
    const float* xr = a.x + irow * a.cols;
    const float* pr = a.dx + irow * a.cols;
    const Tin* dyr = static_cast<const Tin*>(a.dy) + (size_t)(row / a.dy_div) * a.cols;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if ((FAST || col < a.cols)) {
        nx[c] = load4_nt(xr + col);            // (streamed: see the forward)
        ndy[c] = *reinterpret_cast<const DyRaw*>(dyr + col);
        if (FAST || a.accumulate) nprev[c] = load4(pr + col);
      }
    }
    nmean = a.mean[row]; nrstd = a.rstd[row];
  };
  int row0 = bx * 4 + wave, rstep = gx * 4, rend = a.rows, grp = 0;
  f32x4 asum[GS ? CH : 1];
  if constexpr (GS) {
    const int w = bx * 4 + wave;
    if (a.gs_strided) {                                       // group = token index: rows f, f + gs_mod, ... dealt over wpf waves
      const int wpf = (gx * 4) / a.gs_mod;
      const int f = w / wpf;
      row0 = f + (w - f * wpf) * a.gs_mod; rstep = wpf * a.gs_mod; rend = a.rows; grp = f;
    } else {
      const int wpf = (gx * 4) / (a.rows / a.gs_div);         // waves per frame (host: an exact multiple of 4)
      const int f = w / wpf;
      row0 = f * a.gs_div + (w - f * wpf); rstep = wpf; rend = (f + 1) * a.gs_div; grp = f % a.gs_mod;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) asum[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (row0 < rend) issue(row0);
  for (int row = row0; row < rend; row += rstep) {
    const size_t irow = (size_t)row * a.in_mul + ((!FAST && a.in_off) ? a.in_off[row] : 0);
    const float mean = nmean, rstd = nrstd;
    f32x4 xh[CH], g[CH], prev[CH];
    float s1 = 0.f, s2 = 0.f;
    float* dxr = a.dx + irow * a.cols;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * 64 + lane) * 4;
      prev[c] = ((FAST || a.accumulate) && (FAST || col < a.cols)) ? nprev[c] : f32x4{0.f, 0.f, 0.f, 0.f};
      if ((FAST || col < a.cols)) {
        const f32x4 xv = nx[c];
        f32x4 d = {(float)ndy[c][0], (float)ndy[c][1], (float)ndy[c][2], (float)ndy[c][3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d[j] *= a.dy_scale;
          xh[c][j] = (xv[j] - mean) * rstd;
          g[c][j] = d[j] * gm[c][j];
          s1 += g[c][j];
          s2 += g[c][j] * xh[c][j];
          ag[c][j] += d[j] * xh[c][j];
          ab[c][j] += d[j];
        }
      } else {
        xh[c] = f32x4{0.f, 0.f, 0.f, 0.f}; g[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (row + rstep < rend) issue(row + rstep);
    s1 = wave_sum(s1) * inv;
    s2 = wave_sum(s2) * inv;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if ((FAST || col < a.cols)) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (g[c][j] - s1 - xh[c][j] * s2);
        o += prev[c];
        store4(dxr + col, o);
        if (FAST || a.dx_cast) store4(static_cast<Tin*>(a.dx_cast) + irow * a.cols + col, o);
        if constexpr (GS) asum[c] += o;
      }
    }
  }
  // cross-wave reduction of the parameter gradients, then one atomic per column per workgroup
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    store4(&red[0][wave][(c * 64 + lane) * 4], ag[c]);
    store4(&red[1][wave][(c * 64 + lane) * 4], ab[c]);
  }
  __syncthreads();
  for (int col = threadIdx.x; col < a.cols; col += 256) {
    const float sg = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
    const float sb = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
    if (a.dgamma) atomicAdd(a.dgamma + col, sg);
    if (a.dbeta) atomicAdd(a.dbeta + col, sb);
  }
  if constexpr (GS) {                        // the four waves of a workgroup walked one frame: one atomic per column
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; ++c) store4(&red[0][wave][(c * 64 + lane) * 4], asum[c]);
    __syncthreads();
    float* gs = a.gsum + (size_t)grp * a.cols;
    for (int col = threadIdx.x; col < a.cols; col += 256)
      atomicAdd(gs + col, red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col]);
  }
}

template <typename Tin, int CH, bool FAST = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs a) { ln_bwd_body<Tin, CH, FAST>(a, blockIdx.x, gridDim.x); }
template <typename Tin, int CH, bool FAST = false>
__global__ __launch_bounds__(256) void ln_bwd_gs_kernel(LnBwdArgs a) { ln_bwd_body<Tin, CH, FAST, true>(a, blockIdx.x, gridDim.x); }

struct LnBwdGroups { const void* dy[LN_MAX_GROUPS]; const float* x[LN_MAX_GROUPS]; const float* mean[LN_MAX_GROUPS]; const float* rstd[LN_MAX_GROUPS];
                     const float* gamma[LN_MAX_GROUPS]; float* dx[LN_MAX_GROUPS]; float* dgamma[LN_MAX_GROUPS]; float* dbeta[LN_MAX_GROUPS];
                     void* dx_cast[LN_MAX_GROUPS]; };
template <typename Tin, int CH, bool FAST>
__global__ __launch_bounds__(256) void ln_bwd_grouped_kernel(LnBwdArgs a, LnBwdGroups gs) {
  const int g = blockIdx.y;
  a.dy = gs.dy[g]; a.x = gs.x[g]; a.mean = gs.mean[g]; a.rstd = gs.rstd[g]; a.gamma = gs.gamma[g]; a.dx = gs.dx[g];
  a.dgamma = gs.dgamma[g]; a.dbeta = gs.dbeta[g]; a.dx_cast = gs.dx_cast[g];
  ln_bwd_body<Tin, CH, FAST>(a, blockIdx.x, gridDim.x);
}

// mean over groups of T consecutive rows: out[b] = mean_t in[b*T + t]  (pooled.reshape(B,T,-1).mean(1), :662)
__global__ void mean_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int T, int cols) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * cols) return;
  const int b = idx / cols, c = idx % cols;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += in[((size_t)b * T + t) * cols + c];
  out[idx] = s / T;
}

template <typename Tout, typename Fn> static int dispatch_ch(int cols, Fn&& fn) {
  if (cols <= 256) return fn(std::integral_constant<int, 1>{});
  if (cols <= 512) return fn(std::integral_constant<int, 2>{});
  if (cols <= 768) return fn(std::integral_constant<int, 3>{});
  if (cols <= 1024) return fn(std::integral_constant<int, 4>{});
  if (cols <= 2048) return fn(std::integral_constant<int, 8>{});
  missm_set_error("layernorm: cols %d > 2048 unsupported", cols);
  return MISSM_ERR_INVALID;
}

}  // namespace missm

using namespace missm;

extern "C" int missm_layernorm_fwd(const float* x, float* x_wb, const float* add, int add_div, int add_mod, int in_mul,
                                   const int* in_off, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                   int rows, int cols, float eps, int out_dtype, void* stream) {
  MISSM_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "layernorm_fwd: cols must be a positive multiple of 4");
  MISSM_CHECK_ARG(!add || (x_wb && add_div > 0 && add_mod > 0), "layernorm_fwd: add needs x_wb, div, mod");
  LnFwdArgs a{x, x_wb, add, add_div > 0 ? add_div : 1, add_mod > 0 ? add_mod : 1, in_mul > 0 ? in_mul : 1, in_off, gamma, beta,
              y, mean, rstd, rows, cols, eps};
  static const int fcap = getenv("MISSM_LN_FWD_BLOCKS") ? atoi(getenv("MISSM_LN_FWD_BLOCKS")) : (1 << 30);   // default: no cap (one row per wave)
  dim3 grid((rows + 3) / 4 < fcap ? (rows + 3) / 4 : fcap), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_ch<void>(cols, [&](auto ch) {
    constexpr int CH = decltype(ch)::value;
    if (out_dtype == kBF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16, CH>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ln_fwd_kernel<float, CH>), grid, block, 0, s, a);
    return MISSM_OK;
  });
  if (rc != MISSM_OK) return rc;
  return missm_check_launch("layernorm_fwd");
}

static int layernorm_bwd_core(const void* dy, int dy_div, float dy_scale, const float* x, int in_mul, const int* in_off,
                              const float* mean, const float* rstd, const float* gamma, float* dx, int accumulate,
                              float* dgamma, float* dbeta, void* dx_cast, float* gsum, int gs_div, int gs_mod, int rows, int cols,
                              int dy_dtype, void* stream) {
  MISSM_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "layernorm_bwd: cols must be a positive multiple of 4");
  LnBwdArgs a{dy, dy_div > 0 ? dy_div : 1, dy_scale, x, in_mul > 0 ? in_mul : 1, in_off, mean, rstd, gamma, dx, accumulate,
              dgamma, dbeta, rows, cols, dx_cast, gsum, gs_div > 0 ? gs_div : 1, gs_mod, gs_div == 0 ? 1 : 0};
  if (gsum) {
    const bool strided = gs_div == 0;              // gs_div = 0: group = row % gs_mod
    MISSM_CHECK_ARG(gs_div >= 0 && gs_mod > 0 && (strided ? rows % gs_mod == 0 : rows % gs_div == 0) && !in_off && a.in_mul == 1,
                    "layernorm_bwd: group sums need rows = frames x gs_div (or a multiple of gs_mod), no row gather");
    const int frames = strided ? gs_mod : rows / gs_div;
    const int wpf = frames <= 1024 ? 8 : 4;        // waves per frame (video tower at B = 32: 256 frames x 8 = the 512-workgroup grid)
    dim3 grid(frames * wpf / 4), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = dispatch_ch<void>(cols, [&](auto ch) {
      constexpr int CH = decltype(ch)::value;
      const bool fast = cols == CH * 256 && accumulate && dx_cast;
      if (dy_dtype == kBF16 && fast) hipLaunchKernelGGL((ln_bwd_gs_kernel<bf16, CH, true>), grid, block, 0, s, a);
      else if (dy_dtype == kBF16) hipLaunchKernelGGL((ln_bwd_gs_kernel<bf16, CH>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((ln_bwd_gs_kernel<float, CH>), grid, block, 0, s, a);
      return MISSM_OK;
    });
    if (rc != MISSM_OK) return rc;
    return missm_check_launch("layernorm_bwd_groupsum");
  }
  // grid cap: round 1's kernel (one row in flight per wave) wanted 768-1024 workgroups; with the rows software-pipelined two
  // workgroups per CU already keep the memory pipe full, and fewer blocks mean fewer dgamma / dbeta atomics and less pressure on
  // the second stream: inside the two-stream step 512 measures 463.5 samples/s against 460.4 at 768 and 462.9 at 384
  static const int cap = getenv("MISSM_LN_BLOCKS") ? atoi(getenv("MISSM_LN_BLOCKS")) : 512;
  int blocks = (rows + 3) / 4;
  if (blocks > cap) blocks = cap;
  dim3 grid(blocks), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_ch<void>(cols, [&](auto ch) {
    constexpr int CH = decltype(ch)::value;
    const bool fast = cols == CH * 256 && accumulate && dx_cast && !in_off;
    if (dy_dtype == kBF16 && fast) hipLaunchKernelGGL((ln_bwd_kernel<bf16, CH, true>), grid, block, 0, s, a);
    else if (dy_dtype == kBF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16, CH>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ln_bwd_kernel<float, CH>), grid, block, 0, s, a);
    return MISSM_OK;
  });
  if (rc != MISSM_OK) return rc;
  return missm_check_launch("layernorm_bwd");
}

extern "C" int missm_layernorm_bwd(const void* dy, int dy_div, float dy_scale, const float* x, int in_mul, const int* in_off,
                                   const float* mean, const float* rstd, const float* gamma, float* dx, int accumulate,
                                   float* dgamma, float* dbeta, void* dx_cast, int rows, int cols, int dy_dtype, void* stream) {
  return layernorm_bwd_core(dy, dy_div, dy_scale, x, in_mul, in_off, mean, rstd, gamma, dx, accumulate, dgamma, dbeta, dx_cast, nullptr, 1, 1,
                            rows, cols, dy_dtype, stream);
}

extern "C" int missm_layernorm_bwd_groupsum(const void* dy, float dy_scale, const float* x, const float* mean, const float* rstd,
                                            const float* gamma, float* dx, int accumulate, float* dgamma, float* dbeta, void* dx_cast,
                                            float* gsum, int gs_div, int gs_mod, int rows, int cols, int dy_dtype, void* stream) {
  MISSM_CHECK_ARG(gsum != nullptr, "layernorm_bwd_groupsum: gsum is required");
  return layernorm_bwd_core(dy, 1, dy_scale, x, 1, nullptr, mean, rstd, gamma, dx, accumulate, dgamma, dbeta, dx_cast, gsum, gs_div, gs_mod,
                            rows, cols, dy_dtype, stream);
}

extern "C" int missm_layernorm_fwd_grouped(int ngroups, const float* const* x, const float* const* gamma, const float* const* beta, void* const* y,
                                           float* const* mean, float* const* rstd, int rows, int cols, float eps, int out_dtype, void* stream) {
  MISSM_CHECK_ARG(ngroups >= 1 && ngroups <= LN_MAX_GROUPS && x && gamma && beta && y && mean && rstd, "layernorm_fwd_grouped: 1..8 groups");
  MISSM_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "layernorm_fwd_grouped: cols must be a positive multiple of 4");
  LnFwdArgs a{nullptr, nullptr, nullptr, 1, 1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rows, cols, eps};
  LnFwdGroups gs;
  for (int g = 0; g < ngroups; ++g) { gs.x[g] = x[g]; gs.gamma[g] = gamma[g]; gs.beta[g] = beta[g]; gs.y[g] = y[g]; gs.mean[g] = mean[g]; gs.rstd[g] = rstd[g]; }
  static const int fcap = getenv("MISSM_LN_FWD_BLOCKS") ? atoi(getenv("MISSM_LN_FWD_BLOCKS")) : (1 << 30);
  const int per = fcap / ngroups > 0 ? fcap / ngroups : 1;
  dim3 grid((rows + 3) / 4 < per ? (rows + 3) / 4 : per, ngroups), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_ch<void>(cols, [&](auto ch) {
    constexpr int CH = decltype(ch)::value;
    if (out_dtype == kBF16) hipLaunchKernelGGL((ln_fwd_grouped_kernel<bf16, CH>), grid, block, 0, s, a, gs);
    else hipLaunchKernelGGL((ln_fwd_grouped_kernel<float, CH>), grid, block, 0, s, a, gs);
    return MISSM_OK;
  });
  if (rc != MISSM_OK) return rc;
  return missm_check_launch("layernorm_fwd_grouped");
}

extern "C" int missm_layernorm_bwd_grouped(int ngroups, const void* const* dy, const float* const* x, const float* const* mean,
                                           const float* const* rstd, const float* const* gamma, float* const* dx, float* const* dgamma,
                                           float* const* dbeta, void* const* dx_cast, int rows, int cols, int dy_dtype, void* stream) {
  MISSM_CHECK_ARG(ngroups >= 1 && ngroups <= LN_MAX_GROUPS && dy && x && mean && rstd && gamma && dx && dx_cast, "layernorm_bwd_grouped: 1..8 groups");
  MISSM_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "layernorm_bwd_grouped: cols must be a positive multiple of 4");
  LnBwdArgs a{nullptr, 1, 1.0f, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, rows, cols, nullptr, nullptr, 1, 1, 0};
  LnBwdGroups gs;
  for (int g = 0; g < ngroups; ++g) {
    gs.dy[g] = dy[g]; gs.x[g] = x[g]; gs.mean[g] = mean[g]; gs.rstd[g] = rstd[g]; gs.gamma[g] = gamma[g]; gs.dx[g] = dx[g];
    gs.dgamma[g] = dgamma ? dgamma[g] : nullptr; gs.dbeta[g] = dbeta ? dbeta[g] : nullptr; gs.dx_cast[g] = dx_cast[g];
    MISSM_CHECK_ARG(gs.dx_cast[g] != nullptr, "layernorm_bwd_grouped: every group needs dx_cast (the residual-stream form)");
  }
  static const int cap = getenv("MISSM_LN_BLOCKS") ? atoi(getenv("MISSM_LN_BLOCKS")) : 512;
  int blocks = (rows + 3) / 4;
  const int per = cap / ngroups > 0 ? cap / ngroups : 1;        // the cap is for the whole launch
  if (blocks > per) blocks = per;
  dim3 grid(blocks, ngroups), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = dispatch_ch<void>(cols, [&](auto ch) {
    constexpr int CH = decltype(ch)::value;
    const bool fast = cols == CH * 256;
    if (dy_dtype == kBF16 && fast) hipLaunchKernelGGL((ln_bwd_grouped_kernel<bf16, CH, true>), grid, block, 0, s, a, gs);
    else if (dy_dtype == kBF16) hipLaunchKernelGGL((ln_bwd_grouped_kernel<bf16, CH, false>), grid, block, 0, s, a, gs);
    else hipLaunchKernelGGL((ln_bwd_grouped_kernel<float, CH, false>), grid, block, 0, s, a, gs);
    return MISSM_OK;
  });
  if (rc != MISSM_OK) return rc;
  return missm_check_launch("layernorm_bwd_grouped");
}

extern "C" int missm_mean_rows(const float* in, float* out, int B, int T, int cols, void* stream) {
  MISSM_CHECK_ARG(B > 0 && T > 0 && cols > 0, "mean_rows: bad shape");
  const int n = B * cols;
  hipLaunchKernelGGL(mean_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, B, T, cols);
  return missm_check_launch("mean_rows");
}
