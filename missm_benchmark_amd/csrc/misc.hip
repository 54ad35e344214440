// Embedding, pooling-tail, fusion-projection, loss and optimizer kernels (HBM- or launch-bound, all fp32 math).
#include "common.h"
#include "missm_internal.h"

namespace missm {

// ---- patch unfold: the k = stride = ps conv (video/modeling_video.py:29-35) becomes a GEMM over these rows ----
// Output rows hold Kp = K rounded up to a multiple of 8 elements (16-byte GEMM operand rows; K = C * ps * ps): Kp == K whenever
// ps % 4 == 0, a 14-pixel patch (K = 588) gets 4 zero columns.  VEC: ps % 4 == 0 - four consecutive kx share a pixel row.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void unfold_kernel(const float* __restrict__ px, T* __restrict__ out, int B, int Tn, int C, int H,
                                                    int W, int ps, long sb, long st, long sc) {
  const int gw = W / ps, gh = H / ps, P = gw * gh, K = C * ps * ps, Kp = (K + 7) / 8 * 8;
  const int quads = Kp / 4;
  const long total = (long)B * Tn * P * quads;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int qd = idx % quads;
    const long rowi = idx / quads;
    const int p = rowi % P;
    const long n = rowi / P;
    const int b = n / Tn, t = n % Tn;
    const int py = p / gw, pxx = p % gw;
    const float* img = px + b * sb + t * st;
    if constexpr (VEC) {
      const int k = qd * 4, c = k / (ps * ps), rem = k % (ps * ps), ky = rem / ps, kx = rem % ps;
      store4(out + rowi * Kp + k, load4(img + c * sc + (long)(py * ps + ky) * W + pxx * ps + kx));
    } else {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = qd * 4 + j;
        if (k < K) {
          const int c = k / (ps * ps), rem = k % (ps * ps), ky = rem / ps, kx = rem % ps;
          v[j] = img[c * sc + (long)(py * ps + ky) * W + pxx * ps + kx];
        }
      }
      store4(out + rowi * Kp + qd * 4, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_assemble_kernel(const T* __restrict__ patches, const float* __restrict__ cls,
                                                            const float* __restrict__ pos, float* __restrict__ x, int N, int S, int d) {
  const int q4 = d / 4;
  const long total = (long)N * S * q4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (idx % q4) * 4;
    const long row = idx / q4;
    const int s = row % S;
    const long n = row / S;
    f32x4 v = (s == 0) ? load4(cls + c) : load4(patches + (n * (S - 1) + s - 1) * d + c);
    v += load4(pos + (long)s * d + c);
    store4(x + row * d + c, v);
  }
}

__global__ __launch_bounds__(256) void token_embed_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ tok,
                                                             const float* __restrict__ pos, float* __restrict__ h, int B, int S, int d,
                                                             int vocab) {
  const int q4 = d / 4;
  const long total = (long)B * S * q4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (idx % q4) * 4;
    const long row = idx / q4;
    const int s = row % S;
    const long id = ids[row];
    f32x4 v;
    if (id >= 0 && id < vocab) {
      v = load4(tok + id * d + c);
      v += load4(pos + (long)s * d + c);
    } else {
      const float nan = __builtin_nanf("");       // out-of-range id: poison the row instead of reading out of bounds
      v = f32x4{nan, nan, nan, nan};
    }
    store4(h + row * d + c, v);
  }
}

__global__ __launch_bounds__(256) void token_embed_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ dh,
                                                             float* __restrict__ dtok, float* __restrict__ dpos, int B, int S, int d,
                                                             int vocab) {
  const long total = (long)B * S * d;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = idx % d;
    const long row = idx / d;
    const int s = row % S;
    const float g = dh[idx];
    const long id = ids[row];
    if (id < 0 || id >= vocab) continue;
    atomicAdd(dtok + id * d + c, g);
    atomicAdd(dpos + (long)s * d + c, g);
  }
}

__global__ void argmax_rows_kernel(const long* __restrict__ ids, int* __restrict__ out, int B, int S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long best = ids[(long)b * S]; int bi = 0;
  for (int s = 1; s < S; ++s) { const long v = ids[(long)b * S + s]; if (v > best) { best = v; bi = s; } }
  out[b] = bi;
}

// fp32 rows -> T rows with an optional row gather: in_row = r + r / rdiv + roff  (rdiv > 0), else r
template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ in, T* __restrict__ out, long R, int C, int rdiv, int roff) {
  const int q4 = C / 4;
  const long total = R * q4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (idx % q4) * 4;
    const long r = idx / q4;
    const long ir = rdiv > 0 ? r + r / rdiv + roff : r;
    store4(out + r * C + c, load4(in + ir * C + c));
  }
}

// ---- small fp32 linears of the projection / fusion tail: one wavefront per output element ----
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y, int B, int I, int O,
                                                              int ldy, int relu, const long* __restrict__ row_code, long code,
                                                              const float* __restrict__ x_sub, int select, float alpha, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= (long)B * O) return;
  const int b = e / O, o = e % O;
  float acc = 0.f;
  bool masked = row_code && row_code[b] == code;
  if (select) {                 // select mode: ONLY the rows carrying the code are computed and written (dedicated networks)
    if (!masked) return;
    masked = false;
  }
  if (!masked || x_sub) {       // a masked row is either zeroed or computed from the substitute input row
    const float* xr = masked ? x_sub : x + (long)b * I;
    const float* wr = w + (long)o * I;
    for (int i = lane * 4; i < I; i += 256) {
      const f32x4 a = load4(xr + i), c = load4(wr + i);
      acc += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
    }
    acc = wave_sum(acc);
    if (bias) acc += bias[o];
    acc *= alpha;
    if (relu) acc = fmaxf(acc, 0.f);
  }
  if (lane == 0) { float* yp = y + (long)b * ldy + o; *yp = accumulate ? *yp + acc : acc; }
}

// effective dy (masked by row code and by relu) helper
__device__ __forceinline__ float eff_dy(const float* dy, int lddy, const float* relu_y, const long* row_code, long code, int b, int o, int O,
                                        int select = 0) {
  if (row_code && ((row_code[b] == code) != (select != 0))) return 0.f;   // masked rows (or, in select mode, all the others)
  const float g = dy[(long)b * lddy + o];
  if (relu_y && relu_y[(long)b * O + o] <= 0.f) return 0.f;
  return g;
}

// dx[b, i] (+)= sum_o dyeff[b, o] w[o, i].  One workgroup per (sample, 64 inputs): its 8 waves take an eighth of the outputs
// each (lane = input column: the weight rows are read coalesced), four interleaved partial sums per lane, and the 8 partials
// meet in LDS in a FIXED order - the result is bit-reproducible from run to run.  (The first version split o over blockIdx.y
// and met in fp32 atomics: the arrival order changed the last bits of the gradient that enters the towers, and their bf16
// backward amplifies a 1-ulp difference into 1e-3-level differences of the weight gradients.  One thread per element with the
// whole reduction in a loop is reproducible too, but latency-bound: 153 us per call at O = 768.)
__global__ __launch_bounds__(512) void small_linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             const float* __restrict__ relu_y, float* __restrict__ dx, int B, int I, int O,
                                                             int lddy, const long* __restrict__ row_code, long code, int select, float alpha,
                                                             int accumulate) {
  __shared__ float part[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, i = blockIdx.x * 64 + lane;
  const int per = (O + 7) / 8, o0 = wave * per, o1 = min(O, o0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < I) {
    int o = o0;
    for (; o + 3 < o1; o += 4) {
      a0 += eff_dy(dy, lddy, relu_y, row_code, code, b, o, O, select) * w[(long)o * I + i];
      a1 += eff_dy(dy, lddy, relu_y, row_code, code, b, o + 1, O, select) * w[(long)(o + 1) * I + i];
      a2 += eff_dy(dy, lddy, relu_y, row_code, code, b, o + 2, O, select) * w[(long)(o + 2) * I + i];
      a3 += eff_dy(dy, lddy, relu_y, row_code, code, b, o + 3, O, select) * w[(long)(o + 3) * I + i];
    }
    for (; o < o1; ++o) a0 += eff_dy(dy, lddy, relu_y, row_code, code, b, o, O, select) * w[(long)o * I + i];
  }
  part[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && i < I) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += part[k][lane];
    acc *= alpha;
    const long idx = (long)b * I + i;
    dx[idx] = accumulate ? dx[idx] + acc : acc;
  }
}

// dw[o, i] = sum_b dyeff[b, o] x[b, i] ; dbias[o] = sum_b dyeff[b, o]
__global__ __launch_bounds__(256) void small_linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ relu_y, float* __restrict__ dw,
                                                             float* __restrict__ dbias, int B, int I, int O, int lddy,
                                                             const long* __restrict__ row_code, long code, const float* __restrict__ x_sub,
                                                             int select, float alpha, int accumulate) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)O * I) return;
  const int o = idx / I, i = idx % I;
  float acc = 0.f, accb = 0.f;
  for (int b = 0; b < B; ++b) {
    // with a substitute row the masked samples still feed the weights (their input is x_sub), only dx is cut
    const bool subst = x_sub && row_code && row_code[b] == code;
    const float g = eff_dy(dy, lddy, relu_y, subst ? nullptr : row_code, code, b, o, O, select);
    acc += g * (subst ? x_sub[i] : x[(long)b * I + i]);
    accb += g;
  }
  acc *= alpha; accb *= alpha;
  dw[idx] = accumulate ? dw[idx] + acc : acc;          // accumulate: a layer shared by several modalities (channel attention)
  if (dbias && i == 0) dbias[o] = accumulate ? dbias[o] + accb : accb;
}

// channel-attention gate of the intra-modality attention head (src/model/baseline.py:198-201):
//   y[b, f] (+)= row b missing ? 0 : d[b, f] * sigmoid(pre[b, f])          d has row stride ldd (a slice of [d | fusion_repr])
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ d, int ldd, const float* __restrict__ pre,
                                                      float* __restrict__ y, int B, int F, const long* __restrict__ row_code, long code,
                                                      int accumulate) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * F) return;
  const int b = idx / F, f = idx % F;
  float v = 0.f;
  if (!(row_code && row_code[b] == code)) v = d[(long)b * ldd + f] / (1.f + __expf(-pre[idx]));
  y[idx] = accumulate ? y[idx] + v : v;
}
// dd[b, f] (+)= dy * g ; dpre[b, f] = dy * d * g * (1 - g) ; both 0 for a missing row.  dd has row stride lddd.
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ d, int ldd,
                                                      const float* __restrict__ pre, float* __restrict__ dd, int lddd,
                                                      float* __restrict__ dpre, int B, int F, const long* __restrict__ row_code,
                                                      long code, int accumulate_dd) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * F) return;
  const int b = idx / F, f = idx % F;
  float gd = 0.f, gp = 0.f;
  if (!(row_code && row_code[b] == code)) {
    const float g = 1.f / (1.f + __expf(-pre[idx])), g_y = dy[idx];
    gd = g_y * g;
    gp = g_y * d[(long)b * ldd + f] * g * (1.f - g);
  }
  float* o = dd + (long)b * lddd + f;
  *o = accumulate_dd ? *o + gd : gd;
  dpre[idx] = gp;
}

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int D, float scale) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float ss = 0.f;
  for (int i = lane; i < D; i += 64) { const float v = x[(long)b * D + i]; ss += v * v; }
  const float inv = scale / sqrtf(wave_sum(ss));
  for (int i = lane; i < D; i += 64) y[(long)b * D + i] = x[(long)b * D + i] * inv;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                                        int B, int D, float scale) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float ss = 0.f, dot = 0.f;
  for (int i = lane; i < D; i += 64) { const float v = x[(long)b * D + i]; ss += v * v; dot += v * dy[(long)b * D + i]; }
  ss = wave_sum(ss); dot = wave_sum(dot);
  const float rn = 1.0f / sqrtf(ss);
  for (int i = lane; i < D; i += 64) {
    const float v = x[(long)b * D + i];
    dx[(long)b * D + i] = scale * rn * (dy[(long)b * D + i] - v * dot / ss);
  }
}

__global__ __launch_bounds__(64) void cross_entropy_kernel(const float* __restrict__ logits, const long* __restrict__ labels,
                                                          float* __restrict__ loss, float* __restrict__ dlogits, int B, int C) {
  // single wavefront: lane strides over samples, then one shuffle reduction (B, C are small)
  const int lane = threadIdx.x;
  float part = 0.f;
  for (int b = lane; b < B; b += 64) {
    const float* lr = logits + (long)b * C;
    float mx = lr[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, lr[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(lr[c] - mx);
    const float lse = mx + logf(s);
    const long y = labels[b];
    part += (y >= 0 && y < C) ? lse - lr[y] : __builtin_nanf("");   // out-of-range label: NaN loss, no out-of-bounds read
    if (dlogits)
      for (int c = 0; c < C; ++c) dlogits[(long)b * C + c] = (expf(lr[c] - lse) - (c == y ? 1.f : 0.f)) / B;
  }
  part = wave_sum(part);
  if (lane == 0) *loss = part / B;
}


// ---- distillation losses of the student training modes (reference train_ddp.py:70-88,232-244) ---------------------------------
// KL_loss: F.kl_div(log_softmax(s / T), softmax(t / T), reduction='batchmean') over the rows whose mask byte is non-zero (all rows
// when mask == nullptr; the reference gathers them with boolean indexing, train_ddp.py:238-240).  One wavefront, lanes stride over
// rows (B, C are small).  loss = sum_rows sum_c p_t (log p_t - log p_s) / n_rows ; ds = (p_s - p_t) / (T n_rows) on those rows.
__global__ __launch_bounds__(64) void kl_loss_kernel(const float* __restrict__ s, const float* __restrict__ t, const unsigned char* __restrict__ mask,
                                                    float* __restrict__ loss, float* __restrict__ ds, int B, int C, float inv_temp) {
  const int lane = threadIdx.x;
  float part = 0.f, cnt = 0.f;
  for (int b = lane; b < B; b += 64) cnt += (!mask || mask[b]) ? 1.f : 0.f;
  cnt = wave_sum(cnt);                       // 0 selected rows: 0 / 0 = NaN, like the reference's batchmean over an empty selection
  for (int b = lane; b < B; b += 64) {
    const bool on = !mask || mask[b];
    const float* sr = s + (long)b * C;
    const float* tr = t + (long)b * C;
    if (on) {
      float ms = sr[0] * inv_temp, mt = tr[0] * inv_temp;
      for (int c = 1; c < C; ++c) { ms = fmaxf(ms, sr[c] * inv_temp); mt = fmaxf(mt, tr[c] * inv_temp); }
      float zs = 0.f, zt = 0.f;
      for (int c = 0; c < C; ++c) { zs += expf(sr[c] * inv_temp - ms); zt += expf(tr[c] * inv_temp - mt); }
      const float ls = ms + logf(zs), lt = mt + logf(zt);
      for (int c = 0; c < C; ++c) {
        const float lps = sr[c] * inv_temp - ls, lpt = tr[c] * inv_temp - lt;
        const float pt = expf(lpt);
        part += pt > 0.f ? pt * (lpt - lps) : 0.f;
        if (ds) ds[(long)b * C + c] = (expf(lps) - pt) * inv_temp / cnt;
      }
    } else if (ds) {
      for (int c = 0; c < C; ++c) ds[(long)b * C + c] = 0.f;
    }
  }
  part = wave_sum(part);
  if (lane == 0) *loss = part / cnt;
}

// nn.MSELoss(): mean over all elements of (a - b)^2 ; da = 2 (a - b) / n   (b = the detached teacher features)
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ loss,
                                                      float* __restrict__ da, long n) {
  __shared__ float red[4];
  float part = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {            // ONE workgroup: a fixed summation order (bit-reproducible)
    const float d = a[i] - b[i];
    part += d * d;
    if (da) da[i] = 2.f * d / (float)n;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) *loss = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

// ---- the same two losses for LARGE inputs (per-token features, big batches): the one-workgroup kernels above walk every row / element
// from a single CU.  Rows / chunks are spread over the chip, partial sums go through the stream's scratch and are added up
// in a FIXED order by one workgroup: still bit-reproducible.
__global__ __launch_bounds__(256) void mask_count_kernel(const unsigned char* __restrict__ mask, int B, float* __restrict__ cnt) {
  __shared__ float red[4];
  float c = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) c += (!mask || mask[b]) ? 1.f : 0.f;
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) *cnt = (red[0] + red[1]) + (red[2] + red[3]);
}

// one wave per row: lanes stride over the row's C columns
__global__ __launch_bounds__(256) void kl_rows_kernel(const float* __restrict__ s, const float* __restrict__ t, const unsigned char* __restrict__ mask,
                                                     const float* __restrict__ cnt, float* __restrict__ part, float* __restrict__ ds, int B, int C,
                                                     float inv_temp) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const bool on = !mask || mask[b];
  const float* sr = s + (long)b * C;
  const float* tr = t + (long)b * C;
  float p = 0.f;
  if (on) {
    float ms = -__builtin_huge_valf(), mt = ms;
    for (int c = lane; c < C; c += 64) { ms = fmaxf(ms, sr[c] * inv_temp); mt = fmaxf(mt, tr[c] * inv_temp); }
    ms = wave_max(ms); mt = wave_max(mt);
    float zs = 0.f, zt = 0.f;
    for (int c = lane; c < C; c += 64) { zs += expf(sr[c] * inv_temp - ms); zt += expf(tr[c] * inv_temp - mt); }
    const float ls = ms + logf(wave_sum(zs)), lt = mt + logf(wave_sum(zt));
    const float n = *cnt;
    for (int c = lane; c < C; c += 64) {
      const float lps = sr[c] * inv_temp - ls, lpt = tr[c] * inv_temp - lt;
      const float pt = expf(lpt);
      p += pt > 0.f ? pt * (lpt - lps) : 0.f;
      if (ds) ds[(long)b * C + c] = (expf(lps) - pt) * inv_temp / n;
    }
    p = wave_sum(p);
  } else if (ds) {
    for (int c = lane; c < C; c += 64) ds[(long)b * C + c] = 0.f;
  }
  if (lane == 0) part[b] = p;
}

// loss = (sum of n partials, fixed order) * scale / (*div or 1)
__global__ __launch_bounds__(256) void ordered_sum_kernel(const float* __restrict__ part, long n, const float* __restrict__ div, float scale,
                                                         float* __restrict__ loss) {
  __shared__ float red[256];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) a += part[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] * scale / (div ? *div : 1.f);
}

// chunk c of `per` elements: partial sum of squares in a fixed order, da written on the way
__global__ __launch_bounds__(256) void mse_chunks_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                                        float* __restrict__ da, long n, long per) {
  __shared__ float red[4];
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float p = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float d = a[i] - b[i];
    p += d * d;
    if (da) da[i] = 2.f * d / (float)n;
  }
  p = wave_sum(p);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = p;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// teacher EMA of the MTD student mode (train_ddp.py:256-259): tea = decay * tea + (1 - decay) * stu
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ tea, const float* __restrict__ stu, long n, float decay) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) tea[i] = tea[i] * decay + stu[i] * (1.f - decay);
}


// ---- GPU-side image / depth preprocessing (reference processing_image.py:18-28, processing_thermal.py:18-28, processing_depth.py:21-55) ----
// One launch turns a decoded image into the tower's pixel_values: ToTensor (u8 -> [0,1]) or DepthNorm (/1000, clip, / max_depth),
// Resize(S, bicubic) of the SHORTER edge, CenterCrop(S), Normalize(mean, std) - only the S x S pixels that survive the crop are
// ever computed.  The resampling is torchvision's tensor path = ATen's antialiased bicubic (separable cubic with a = -0.5, the
// support stretched by the down-scaling factor, weights normalised to 1): restated from the published algorithm
// (aten/src/ATen/native/UpSampleBicubic2d / "_upsample_bicubic2d_aa"); torchvision is absent here, so parity is unpinned.
struct PrepArgs {
  const void* src; float* dst;
  int src_u8, chw, H, W, C;          // source: uint8 or float32; [C,H,W] or [H,W,C]; C = 1 (replicated) or 3
  int new_h, new_w, top, left, S;    // resized size, crop origin, output edge
  float pre_scale, pre_min, pre_max, pre_div;   // v = clip(v * pre_scale, pre_min, pre_max) / pre_div   (pre_max <= 0: no upper clip)
  float mean[3], inv_std[3];
};
__device__ __forceinline__ float cubic_aa(float x) {   // a = -0.5 (the antialiased path's filter)
  x = fabsf(x);
  if (x < 1.f) return ((1.5f * x - 2.5f) * x) * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * -0.5f;
  return 0.f;
}
__device__ __forceinline__ void aa_window(int i, float scale, int in_size, int& xmin, int& xsize, float& center, float& invscale) {
  const float support = scale >= 1.f ? 2.f * scale : 2.f;
  invscale = scale >= 1.f ? 1.f / scale : 1.f;
  center = scale * (i + 0.5f);
  xmin = max(0, (int)(center - support + 0.5f));
  xsize = min(in_size, (int)(center + support + 0.5f)) - xmin;
}
__global__ __launch_bounds__(256) void preprocess_image_kernel(PrepArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.S * a.S) return;
  const int oy = idx / a.S, ox = idx % a.S;
  const float sh = (float)a.H / (float)a.new_h, sw = (float)a.W / (float)a.new_w;
  int ymin, ysize, xmin, xsize; float cy, cx, isy, isx;
  aa_window(oy + a.top, sh, a.H, ymin, ysize, cy, isy);
  aa_window(ox + a.left, sw, a.W, xmin, xsize, cx, isx);
  float wys = 0.f, wxs = 0.f;
  for (int j = 0; j < ysize; ++j) wys += cubic_aa((j + ymin - cy + 0.5f) * isy);
  for (int i = 0; i < xsize; ++i) wxs += cubic_aa((i + xmin - cx + 0.5f) * isx);
  float acc[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < ysize; ++j) {
    const float wy = cubic_aa((j + ymin - cy + 0.5f) * isy) / wys;
    const int y = ymin + j;
    float row[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < xsize; ++i) {
      const float wx = cubic_aa((i + xmin - cx + 0.5f) * isx) / wxs;
      const int x = xmin + i;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c >= a.C) break;
        const size_t off = a.chw ? ((size_t)c * a.H + y) * a.W + x : ((size_t)y * a.W + x) * a.C + c;
        float v = a.src_u8 ? (float)static_cast<const unsigned char*>(a.src)[off] : static_cast<const float*>(a.src)[off];
        v *= a.pre_scale;
        v = fmaxf(v, a.pre_min);
        if (a.pre_max > 0.f) v = fminf(v, a.pre_max);
        row[c] += wx * (v / a.pre_div);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += wy * row[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = a.C == 1 ? acc[0] : acc[c];
    a.dst[((size_t)c * a.S + oy) * a.S + ox] = (v - a.mean[c]) * a.inv_std[c];
  }
}


// ---- SuperGAT attention over the per-sample modality graphs of the graph-fusion heads (reference src/model/baseline.py:11-24,
// 240-331: torch_geometric.nn.SuperGATConv, attention_type 'MX', self loops added, negative_slope 0.2; the self-supervised
// link-prediction loss it also computes in training never reaches the output and is not used by the reference).
// torch_geometric is absent and unpinned upstream: restated from the published algorithm (Kim & Oh, ICLR 2021) - PARITY UNPINNED.
// One wavefront per (sample, head).  xp = lin(x) [B, M, H, C] (M <= 8 nodes).  Edge j -> i exists iff i == j or both nodes are
// present (node_ok).  e_ij = leaky_relu((xp_j . att_l + xp_i . att_r) * sigmoid(xp_i . xp_j)); alpha_i. = softmax_j over the edges;
// out_i = sum_j alpha_ij xp_j.
constexpr int kMaxNodes = 8;
struct SgatArgs { const float* xp; const float* att_l; const float* att_r; const unsigned char* node_ok; float* out; float* alpha;
                  const float* dout; float* dxp; float* datt_l_part; float* datt_r_part; int B, M, H, C;
                  const float* bias; float* out_act; const float* pre; };   // out = conv + bias; out_act = gelu(out) (optional); pre: saved out

__device__ __forceinline__ void sgat_scores(const SgatArgs& a, int b, int h, int lane, float (&sl)[kMaxNodes], float (&sr)[kMaxNodes],
                                            float (&lg)[kMaxNodes][kMaxNodes]) {
  const float* x = a.xp + ((size_t)b * a.M * a.H + h) * a.C;     // node j at x + j * H * C
  const size_t ns = (size_t)a.H * a.C;
  for (int j = 0; j < a.M; ++j) {
    float pl = 0.f, pr = 0.f;
    for (int c = lane; c < a.C; c += 64) { const float v = x[j * ns + c]; pl += v * a.att_l[h * a.C + c]; pr += v * a.att_r[h * a.C + c]; }
    sl[j] = wave_sum(pl); sr[j] = wave_sum(pr);
  }
  for (int i = 0; i < a.M; ++i)
    for (int j = i; j < a.M; ++j) {
      float pd = 0.f;
      for (int c = lane; c < a.C; c += 64) pd += x[i * ns + c] * x[j * ns + c];
      lg[i][j] = lg[j][i] = wave_sum(pd);
    }
}

__global__ __launch_bounds__(64) void sgat_fwd_kernel(SgatArgs a) {
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H, lane = threadIdx.x;
  float sl[kMaxNodes], sr[kMaxNodes], lg[kMaxNodes][kMaxNodes];
  sgat_scores(a, b, h, lane, sl, sr, lg);
  const float* x = a.xp + ((size_t)b * a.M * a.H + h) * a.C;
  const size_t ns = (size_t)a.H * a.C;
  for (int i = 0; i < a.M; ++i) {
    float e[kMaxNodes], mx = -__builtin_huge_valf();
    for (int j = 0; j < a.M; ++j) {
      const bool edge = i == j || (a.node_ok[b * a.M + i] && a.node_ok[b * a.M + j]);
      float v = (sl[j] + sr[i]) / (1.f + expf(-lg[i][j]));
      v = v > 0.f ? v : 0.2f * v;
      e[j] = edge ? v : -__builtin_huge_valf();
      mx = fmaxf(mx, e[j]);
    }
    float z = 0.f;
    for (int j = 0; j < a.M; ++j) { e[j] = expf(e[j] - mx); z += e[j]; }
    for (int j = 0; j < a.M; ++j) {
      e[j] /= z;
      if (lane == 0) a.alpha[(((size_t)b * a.H + h) * a.M + i) * a.M + j] = e[j];
    }
    for (int c = lane; c < a.C; c += 64) {
      float o = a.bias ? a.bias[h * a.C + c] : 0.f;
      for (int j = 0; j < a.M; ++j) o += e[j] * x[j * ns + c];
      const size_t oi = ((size_t)(b * a.M + i) * a.H + h) * a.C + c;
      a.out[oi] = o;
      if (a.out_act) a.out_act[oi] = gelu_erf(o);
    }
  }
}

// backward of the above from d out [B, M, H, C]: d xp, and per-sample partial sums of d att_l / d att_r [B, H, C] (summed over the
// batch by a column-sum launch afterwards: a fixed order, bit-reproducible)
__global__ __launch_bounds__(64) void sgat_bwd_kernel(SgatArgs a) {
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H, lane = threadIdx.x;
  float sl[kMaxNodes], sr[kMaxNodes], lg[kMaxNodes][kMaxNodes];
  sgat_scores(a, b, h, lane, sl, sr, lg);
  const float* x = a.xp + ((size_t)b * a.M * a.H + h) * a.C;
  const float* go = a.dout + ((size_t)b * a.M * a.H + h) * a.C;          // (already multiplied by gelu'(pre) by the caller kernel below)
  const float* al = a.alpha + ((size_t)b * a.H + h) * a.M * a.M;
  const size_t ns = (size_t)a.H * a.C;
  float dsl[kMaxNodes], dsr[kMaxNodes], dlg[kMaxNodes][kMaxNodes];
  for (int j = 0; j < a.M; ++j) { dsl[j] = 0.f; dsr[j] = 0.f; }
  for (int i = 0; i < a.M; ++i) {
    float da[kMaxNodes], dot = 0.f;
    for (int j = 0; j < a.M; ++j) {
      float pd = 0.f;
      for (int c = lane; c < a.C; c += 64) pd += go[i * ns + c] * x[j * ns + c];
      da[j] = wave_sum(pd);                              // d alpha_ij
      dot += al[i * a.M + j] * da[j];
    }
    for (int j = 0; j < a.M; ++j) {
      const float p = al[i * a.M + j];
      float de = p * (da[j] - dot);                      // softmax backward (p = 0 on non-edges)
      const float sg = 1.f / (1.f + expf(-lg[i][j])), t = sl[j] + sr[i], pre = t * sg;
      de *= pre > 0.f ? 1.f : 0.2f;                      // leaky_relu backward
      dsl[j] += de * sg; dsr[i] += de * sg;
      dlg[i][j] = de * t * sg * (1.f - sg);
    }
  }
  for (int c = lane; c < a.C; c += 64) {
    const float wl = a.att_l[h * a.C + c], wr = a.att_r[h * a.C + c];
    float pl = 0.f, pr = 0.f;
    for (int n = 0; n < a.M; ++n) {
      float d = dsl[n] * wl + dsr[n] * wr;
      for (int i = 0; i < a.M; ++i) d += al[i * a.M + n] * go[i * ns + c];                 // out_i = sum_j alpha_ij xp_j
      for (int m = 0; m < a.M; ++m) d += (dlg[n][m] + dlg[m][n]) * x[m * ns + c];          // logit_ij = xp_i . xp_j
      a.dxp[((size_t)(b * a.M + n) * a.H + h) * a.C + c] = d;
      pl += dsl[n] * x[n * ns + c]; pr += dsr[n] * x[n * ns + c];
    }
    a.datt_l_part[((size_t)b * a.H + h) * a.C + c] = pl;
    a.datt_r_part[((size_t)b * a.H + h) * a.C + c] = pr;
  }
}

__device__ __forceinline__ uint32_t hash_u64(unsigned long long x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((x ^ (x >> 31)) >> 32);
}

__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ mask,
                                                         long n, float p, unsigned long long seed) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float u = hash_u64(seed * 0x100000001B3ull + (unsigned long long)i) * (1.0f / 4294967296.0f);
  const unsigned char keep = u >= p;
  mask[i] = keep;
  y[i] = keep ? x[i] / (1.0f - p) : 0.f;
}

__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                         float* __restrict__ dx, long n, float p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  dx[i] = mask[i] ? dy[i] / (1.0f - p) : 0.f;
}

// Fused Adam: 16 B read of p, g, m, v + 12 B write of p, m, v per parameter = 28 B/param of HBM traffic.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, long n4, long n, float lr_bc1, float inv_sqrt_bc2, float beta1,
                                                  float beta2, float eps, float wd, float gscale) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
    const long i = q * 4;
    if (i + 3 < n) {
      f32x4 pv = load4(p + i), gv = load4(g + i), mv = load4(m + i), vv = load4(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gg = gv[j] * gscale + wd * pv[j];
        mv[j] = beta1 * mv[j] + (1.f - beta1) * gg;
        vv[j] = beta2 * vv[j] + (1.f - beta2) * gg * gg;
        pv[j] -= lr_bc1 * mv[j] / (sqrtf(vv[j]) * inv_sqrt_bc2 + eps);
      }
      store4(p + i, pv); store4(m + i, mv); store4(v + i, vv);
    } else {
      for (long k = i; k < n; ++k) {
        float gg = g[k] * gscale + wd * p[k];
        m[k] = beta1 * m[k] + (1.f - beta1) * gg;
        v[k] = beta2 * v[k] + (1.f - beta2) * gg * gg;
        p[k] -= lr_bc1 * m[k] / (sqrtf(v[k]) * inv_sqrt_bc2 + eps);
      }
    }
  }
}

static inline int grid_for(long total, int cap = 4096) {
  long g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace missm

using namespace missm;
#define S_(x) static_cast<hipStream_t>(x)

extern "C" int missm_unfold_patches(const float* pixels, void* out, int B, int T, int C, int H, int W, int ps, long stride_b,
                                    long stride_t, long stride_c, int dtype, void* stream) {
  MISSM_CHECK_ARG(B > 0 && T > 0 && C > 0 && ps > 0 && H % ps == 0 && W % ps == 0, "unfold: bad shape (the patch size must divide the image)");
  const bool vec = ps % 4 == 0 && W % 4 == 0 && stride_b % 4 == 0 && stride_t % 4 == 0 && stride_c % 4 == 0;
  const long total = (long)B * T * (H / ps) * (W / ps) * ((C * ps * ps + 7) / 8 * 2);
  dim3 grid(grid_for(total, 8192)), block(256);
#define MISSM_UNFOLD(TT, VV) hipLaunchKernelGGL((unfold_kernel<TT, VV>), grid, block, 0, S_(stream), pixels, (TT*)out, B, T, C, H, W, ps, stride_b, stride_t, stride_c)
  if (dtype == kBF16) { if (vec) MISSM_UNFOLD(bf16, true); else MISSM_UNFOLD(bf16, false); }
  else { if (vec) MISSM_UNFOLD(float, true); else MISSM_UNFOLD(float, false); }
#undef MISSM_UNFOLD
  return missm_check_launch("unfold_patches");
}

extern "C" int missm_cast_rows(const float* in, void* out, long R, int C, int rdiv, int roff, int dtype, void* stream) {
  MISSM_CHECK_ARG(R > 0 && C > 0 && C % 4 == 0, "cast_rows: bad shape");
  dim3 grid(grid_for(R * (C / 4), 8192)), block(256);
  if (dtype == kBF16) hipLaunchKernelGGL(cast_rows_kernel<bf16>, grid, block, 0, S_(stream), in, (bf16*)out, R, C, rdiv, roff);
  else hipLaunchKernelGGL(cast_rows_kernel<float>, grid, block, 0, S_(stream), in, (float*)out, R, C, rdiv, roff);
  return missm_check_launch("cast_rows");
}

extern "C" int missm_embed_assemble(const void* patches, const float* cls, const float* pos, float* x, int N, int S, int d, int dtype,
                                    void* stream) {
  MISSM_CHECK_ARG(N > 0 && S > 1 && d > 0 && d % 4 == 0, "embed_assemble: bad shape");
  dim3 grid(grid_for((long)N * S * (d / 4), 8192)), block(256);
  if (dtype == kBF16) hipLaunchKernelGGL(embed_assemble_kernel<bf16>, grid, block, 0, S_(stream), (const bf16*)patches, cls, pos, x, N, S, d);
  else hipLaunchKernelGGL(embed_assemble_kernel<float>, grid, block, 0, S_(stream), (const float*)patches, cls, pos, x, N, S, d);
  return missm_check_launch("embed_assemble");
}

extern "C" int missm_token_embed_fwd(const long* ids, const float* tok, const float* pos, float* h, int B, int S, int d, int vocab,
                                     void* stream) {
  MISSM_CHECK_ARG(B > 0 && S > 0 && d > 0 && d % 4 == 0 && vocab > 0, "token_embed_fwd: bad shape");
  hipLaunchKernelGGL(token_embed_fwd_kernel, dim3(grid_for((long)B * S * (d / 4))), dim3(256), 0, S_(stream), ids, tok, pos, h, B, S, d, vocab);
  return missm_check_launch("token_embed_fwd");
}

extern "C" int missm_token_embed_bwd(const long* ids, const float* dh, float* dtok, float* dpos, int B, int S, int d, int vocab,
                                     void* stream) {
  MISSM_CHECK_ARG(B > 0 && S > 0 && d > 0 && vocab > 0, "token_embed_bwd: bad shape");
  hipLaunchKernelGGL(token_embed_bwd_kernel, dim3(grid_for((long)B * S * d)), dim3(256), 0, S_(stream), ids, dh, dtok, dpos, B, S, d, vocab);
  return missm_check_launch("token_embed_bwd");
}

extern "C" int missm_argmax_rows(const long* ids, int* out, int B, int S, void* stream) {
  MISSM_CHECK_ARG(B > 0 && S > 0, "argmax_rows: bad shape");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, S_(stream), ids, out, B, S);
  return missm_check_launch("argmax_rows");
}

extern "C" int missm_small_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int ldy, int relu,
                                      const long* row_code, long code, const float* x_sub, int select, float alpha, int accumulate,
                                      void* stream) {
  MISSM_CHECK_ARG(B > 0 && I > 0 && O > 0 && I % 4 == 0 && ldy >= O, "small_linear_fwd: bad shape (I must be a multiple of 4, ldy >= O)");
  MISSM_CHECK_ARG(!x_sub || row_code, "small_linear_fwd: a substitute row needs row codes");
  MISSM_CHECK_ARG(!select || (row_code && !x_sub), "small_linear_fwd: select mode needs row codes and no substitute row");
  const long e = (long)B * O;
  hipLaunchKernelGGL(small_linear_fwd_kernel, dim3((e + 3) / 4), dim3(256), 0, S_(stream), x, w, bias, y, B, I, O, ldy, relu, row_code, code, x_sub, select, alpha, accumulate);
  return missm_check_launch("small_linear_fwd");
}

extern "C" int missm_small_linear_bwd(const float* dy, int lddy, const float* x, const float* w, const float* relu_y, float* dx, float* dw,
                                      float* dbias, int B, int I, int O, const long* row_code, long code, const float* x_sub,
                                      int select, float alpha, int accumulate_dx, int accumulate_dw, void* stream) {
  MISSM_CHECK_ARG(B > 0 && I > 0 && O > 0 && lddy >= O, "small_linear_bwd: bad shape");
  MISSM_CHECK_ARG(!relu_y || lddy == O, "small_linear_bwd: the relu mask is dense, dy must be too");
  MISSM_CHECK_ARG(!relu_y || alpha == 1.0f, "small_linear_bwd: alpha with a fused ReLU is not supported");
  if (dx) {
    hipLaunchKernelGGL(small_linear_dx_kernel, dim3((I + 63) / 64, B), dim3(512), 0, S_(stream), dy, w, relu_y, dx, B, I, O, lddy, row_code, code, select, alpha, accumulate_dx);
  }
  if (dw) hipLaunchKernelGGL(small_linear_dw_kernel, dim3(((long)O * I + 255) / 256), dim3(256), 0, S_(stream), dy, x, relu_y, dw, dbias, B, I, O, lddy, row_code, code, x_sub, select, alpha, accumulate_dw);
  return missm_check_launch("small_linear_bwd");
}

// dst[b, 0:W] += src[b, 0:W] for two row-strided blocks (gradient slices of a concatenated feature row meeting again)
__global__ __launch_bounds__(256) void add_block_kernel(float* __restrict__ dst, int lddst, const float* __restrict__ src, int ldsrc, int B, int W) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * W) return;
  const int b = idx / W, c = idx % W;
  dst[(long)b * lddst + c] += src[(long)b * ldsrc + c];
}

// dst[b, 0:W] = row b carries the code ? 0 : src[b, 0:W]   (a modality's embedding block of the concatenated feature row with
// its missing rows zeroed, src/model/baseline.py:370-374; the same kernel maps the gradient back)
__global__ __launch_bounds__(256) void masked_copy_block_kernel(float* __restrict__ dst, int lddst, const float* __restrict__ src, int ldsrc,
                                                               int B, int W, const long* __restrict__ row_code, long code, int keep_matching) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * W) return;
  const int b = idx / W, c = idx % W;
  const bool hit = row_code && row_code[b] == code;
  dst[(long)b * lddst + c] = (hit != (keep_matching != 0)) ? 0.f : src[(long)b * ldsrc + c];   // (no codes: keep_matching must be 0)
}

extern "C" int missm_masked_copy_block(float* dst, int lddst, const float* src, int ldsrc, int B, int W, const long* row_code, long code,
                                       int keep_matching, void* stream) {
  MISSM_CHECK_ARG(B > 0 && W > 0 && lddst >= W && ldsrc >= W && (!keep_matching || row_code), "masked_copy_block: bad shape");
  hipLaunchKernelGGL(masked_copy_block_kernel, dim3(((long)B * W + 255) / 256), dim3(256), 0, S_(stream), dst, lddst, src, ldsrc, B, W, row_code, code, keep_matching);
  return missm_check_launch("masked_copy_block");
}

extern "C" int missm_add_block(float* dst, int lddst, const float* src, int ldsrc, int B, int W, void* stream) {
  MISSM_CHECK_ARG(B > 0 && W > 0 && lddst >= W && ldsrc >= W, "add_block: bad shape");
  hipLaunchKernelGGL(add_block_kernel, dim3(((long)B * W + 255) / 256), dim3(256), 0, S_(stream), dst, lddst, src, ldsrc, B, W);
  return missm_check_launch("add_block");
}

extern "C" int missm_gate_fwd(const float* d, int ldd, const float* pre, float* y, int B, int F, const long* row_code, long code,
                              int accumulate, void* stream) {
  MISSM_CHECK_ARG(B > 0 && F > 0 && ldd >= F, "gate_fwd: bad shape");
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(((long)B * F + 255) / 256), dim3(256), 0, S_(stream), d, ldd, pre, y, B, F, row_code, code, accumulate);
  return missm_check_launch("gate_fwd");
}

extern "C" int missm_gate_bwd(const float* dy, const float* d, int ldd, const float* pre, float* dd, int lddd, float* dpre, int B, int F,
                              const long* row_code, long code, int accumulate_dd, void* stream) {
  MISSM_CHECK_ARG(B > 0 && F > 0 && ldd >= F && lddd >= F, "gate_bwd: bad shape");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(((long)B * F + 255) / 256), dim3(256), 0, S_(stream), dy, d, ldd, pre, dd, lddd, dpre, B, F, row_code,
                     code, accumulate_dd);
  return missm_check_launch("gate_bwd");
}

extern "C" int missm_l2norm_scale_fwd(const float* x, float* y, int B, int D, float scale, void* stream) {
  MISSM_CHECK_ARG(B > 0 && D > 0, "l2norm_fwd: bad shape");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, S_(stream), x, y, B, D, scale);
  return missm_check_launch("l2norm_fwd");
}

extern "C" int missm_l2norm_scale_bwd(const float* dy, const float* x, float* dx, int B, int D, float scale, void* stream) {
  MISSM_CHECK_ARG(B > 0 && D > 0, "l2norm_bwd: bad shape");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, S_(stream), dy, x, dx, B, D, scale);
  return missm_check_launch("l2norm_bwd");
}

extern "C" int missm_cross_entropy(const float* logits, const long* labels, float* loss, float* dlogits, int B, int C, void* stream) {
  MISSM_CHECK_ARG(B > 0 && C > 0, "cross_entropy: bad shape");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(64), 0, S_(stream), logits, labels, loss, dlogits, B, C);
  return missm_check_launch("cross_entropy");
}

extern "C" int missm_kl_loss(const float* student, const float* teacher, const unsigned char* row_mask, float* loss, float* dstudent, int B,
                             int C, float temperature, void* stream) {
  MISSM_CHECK_ARG(B > 0 && C > 0 && temperature > 0.f, "kl_loss: bad shape");
  if ((long)B * C > (1L << 16)) {           // large inputs: one wave per row over the whole chip, ordered second stage
    float* ws = nullptr;
    if (missm_stream_workspace(stream, ((size_t)B + 64) * sizeof(float), &ws)) { missm_set_error("kl_loss: cannot allocate scratch"); return MISSM_ERR_LAUNCH; }
    hipLaunchKernelGGL(mask_count_kernel, dim3(1), dim3(256), 0, S_(stream), row_mask, B, ws);
    hipLaunchKernelGGL(kl_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, S_(stream), student, teacher, row_mask, ws, ws + 64, dstudent, B, C, 1.0f / temperature);
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, S_(stream), ws + 64, (long)B, ws, 1.0f, loss);
    return missm_check_launch("kl_loss");
  }
  hipLaunchKernelGGL(kl_loss_kernel, dim3(1), dim3(64), 0, S_(stream), student, teacher, row_mask, loss, dstudent, B, C, 1.0f / temperature);
  return missm_check_launch("kl_loss");
}

extern "C" int missm_mse_loss(const float* a, const float* b, float* loss, float* da, long n, void* stream) {
  MISSM_CHECK_ARG(n > 0, "mse_loss: empty input");
  if (n > (1L << 16)) {
    const long per = 16384;
    const long chunks = (n + per - 1) / per;
    float* ws = nullptr;
    if (missm_stream_workspace(stream, (size_t)chunks * sizeof(float), &ws)) { missm_set_error("mse_loss: cannot allocate scratch"); return MISSM_ERR_LAUNCH; }
    hipLaunchKernelGGL(mse_chunks_kernel, dim3((unsigned)chunks), dim3(256), 0, S_(stream), a, b, ws, da, n, per);
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, S_(stream), ws, chunks, (const float*)nullptr, 1.0f / (float)n, loss);
    return missm_check_launch("mse_loss");
  }
  hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(256), 0, S_(stream), a, b, loss, da, n);
  return missm_check_launch("mse_loss");
}

extern "C" int missm_ema_update(float* teacher, const float* student, long n, float decay, void* stream) {
  MISSM_CHECK_ARG(n > 0 && decay >= 0.f && decay <= 1.f, "ema_update: bad args");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, S_(stream), teacher, student, n, decay);
  return missm_check_launch("ema_update");
}

extern "C" int missm_preprocess_image(const void* src, int src_u8, int chw, int H, int W, int C, float* dst, int S, float pre_scale,
                                      float pre_min, float pre_max, float pre_div, const float* mean3, const float* std3, void* stream) {
  MISSM_CHECK_ARG(src && dst && H > 0 && W > 0 && (C == 1 || C == 3) && S > 0 && mean3 && std3 && pre_div != 0.f, "preprocess_image: bad args");
  PrepArgs a;
  a.src = src; a.dst = dst; a.src_u8 = src_u8; a.chw = chw; a.H = H; a.W = W; a.C = C; a.S = S;
  // transforms.Resize(S): the shorter edge becomes S, the longer one int(S * long / short); CenterCrop: int(round((n - S) / 2))
  if (H <= W) { a.new_h = S; a.new_w = (int)((long)S * W / H); } else { a.new_w = S; a.new_h = (int)((long)S * H / W); }
  a.top = (int)lrintf((a.new_h - S) / 2.0f); a.left = (int)lrintf((a.new_w - S) / 2.0f);
  a.pre_scale = pre_scale; a.pre_min = pre_min; a.pre_max = pre_max; a.pre_div = pre_div;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.inv_std[c] = 1.0f / std3[c]; }
  hipLaunchKernelGGL(preprocess_image_kernel, dim3((S * S + 255) / 256), dim3(256), 0, S_(stream), a);
  return missm_check_launch("preprocess_image");
}

// dy_eff = dy * gelu'(pre)   (the GELU between the two SuperGAT layers of fusion_gcn, src/model/baseline.py:16,21)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = dy[i] * gelu_erf_grad(pre[i]);
}

extern "C" int missm_gelu_bwd(const float* dy, const float* pre, float* dx, long n, void* stream) {
  MISSM_CHECK_ARG(n > 0 && dy && pre && dx, "gelu_bwd: bad args");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, S_(stream), dy, pre, dx, n);
  return missm_check_launch("gelu_bwd");
}

extern "C" int missm_sgat_fwd(const float* xp, const float* att_l, const float* att_r, const unsigned char* node_ok, const float* bias,
                              float* out, float* out_gelu, float* alpha, int B, int M, int H, int C, void* stream) {
  MISSM_CHECK_ARG(B > 0 && M > 0 && M <= kMaxNodes && H > 0 && C > 0 && xp && att_l && att_r && node_ok && out && alpha, "sgat_fwd: bad args (at most 8 nodes)");
  SgatArgs a{xp, att_l, att_r, node_ok, out, alpha, nullptr, nullptr, nullptr, nullptr, B, M, H, C, bias, out_gelu, nullptr};
  hipLaunchKernelGGL(sgat_fwd_kernel, dim3(B * H), dim3(64), 0, S_(stream), a);
  return missm_check_launch("sgat_fwd");
}

extern "C" int missm_sgat_bwd(const float* xp, const float* att_l, const float* att_r, const unsigned char* node_ok, const float* alpha,
                              const float* dout, float* dxp, float* datt_l_part, float* datt_r_part, int B, int M, int H, int C, void* stream) {
  MISSM_CHECK_ARG(B > 0 && M > 0 && M <= kMaxNodes && H > 0 && C > 0 && xp && alpha && dout && dxp && datt_l_part && datt_r_part, "sgat_bwd: bad args");
  SgatArgs a{xp, att_l, att_r, node_ok, nullptr, const_cast<float*>(alpha), dout, dxp, datt_l_part, datt_r_part, B, M, H, C, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(sgat_bwd_kernel, dim3(B * H), dim3(64), 0, S_(stream), a);
  return missm_check_launch("sgat_bwd");
}

extern "C" int missm_dropout_fwd(const float* x, float* y, unsigned char* mask, long n, float p, unsigned long long seed, void* stream) {
  MISSM_CHECK_ARG(n > 0 && p >= 0.f && p < 1.f, "dropout_fwd: bad args");
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, S_(stream), x, y, mask, n, p, seed);
  return missm_check_launch("dropout_fwd");
}

extern "C" int missm_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long n, float p, void* stream) {
  MISSM_CHECK_ARG(n > 0 && p >= 0.f && p < 1.f, "dropout_bwd: bad args");
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, S_(stream), dy, mask, dx, n, p);
  return missm_check_launch("dropout_bwd");
}

extern "C" int missm_adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float grad_scale, void* stream) {
  MISSM_CHECK_ARG(n > 0 && step >= 1, "adam_step: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const long n4 = (n + 3) / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4, 2048)), dim3(256), 0, S_(stream), p, g, m, v, n4, n, (float)(lr / bc1),
                     (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale);
  return missm_check_launch("adam_step");
}
