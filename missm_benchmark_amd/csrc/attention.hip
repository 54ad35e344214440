// Fused multi-head attention, forward and backward, for the CLIP towers (head_dim 64 MFMA path; any small L path).
// Replaces CLIPAttention's unfused bmm + softmax + bmm (third-party, called from
// languagebind/image/modeling_image.py:121-126,140-145) without ever materialising the S x S scores.
//
// Row addressing covers both attention flavours of the video tower without any '(b t) n d <-> (b n) t d' copy
// (image/modeling_image.py:112-127): token j of sequence q lives at row
//     (q / seq_div) * seq_outer + (q % seq_div) * seq_inner + j * tok_stride
// of the fused [rows, 3d] QKV matrix (spatial: div 1, outer S, stride 1; temporal: div S, outer T*S, inner 1, stride S).
//
// MFMA kernels (17 <= L <= 256, head_dim 64): one 256-thread workgroup per (sequence, head); the whole K and V
// of the head sit in LDS (2 x 28 KiB bf16 at L = 197 -> two workgroups per CU).  Scores are computed TRANSPOSED
// (S^T = K Q^T) so a lane owns one query column: the softmax reduction is register-local plus two xor-shuffles,
// and the probability tile is already the B operand of the PV product (O^T = V^T P^T) - P never touches LDS.
// V^T fragments come from ds_read_b64_tr_b16 on the row-major V tile.  The backward runs two passes inside one
// launch (query-on-lane for dQ, key-on-lane for dK/dV), re-staging LDS in between; D = rowsum(P o dP) is
// formed in registers, so O is not needed and nothing is accumulated with atomics.
#include <stdlib.h>
#include "common.h"
#include "mma.h"
#include "missm_internal.h"

namespace missm {

struct AttnArgs {
  const void* qkv; void* out; float* lse;
  const void* dout; void* dqkv;
  int nseq, L, H, d, ld, ldo;
  int seq_div, seq_outer, seq_inner, tok_stride;
  int causal; const int* key_mask; float scale;
};

__device__ __forceinline__ size_t seq_base(const AttnArgs& a, int q) {
  return (size_t)(q / a.seq_div) * a.seq_outer + (size_t)(q % a.seq_div) * a.seq_inner;
}

constexpr int HD = 64;
constexpr int ANW = 8;            // waves per MFMA-attention workgroup (512 threads): query / key tiles are dealt round-robin
constexpr int ATHREADS = ANW * 64;
// Waves of a whole-workgroup unit.  S = 197 has 13 tiles in 14 slots: 8 waves hold 2,2,2,2,2,1,1,1 tiles, 7 waves would hold
// 2,2,2,2,2,2,1 - measured SLOWER (fwd 78 -> 85 us, bwd 272 -> 294 us at 256 frames x 12 heads): the kernels are bound by
// per-wave latency (47 % of wave cycles parked on s_waitcnt, 32 % on issue dependencies, profiles/r02_attn_sq_counters.txt),
// so the extra resident waves are worth more than the balance.
constexpr int attn_waves(int ntp) { (void)ntp; return ANW; }
constexpr float kNegInf = -__builtin_huge_valf();

__device__ __attribute__((aligned(16))) unsigned int g_attn_zero16[4];   // source of zero-filled LDS chunks
// cache policy of the head-slice stagings (a (sequence, head) slice is read by one workgroup, once per pass): 0 default, 2 = nt
#ifndef MISSM_ATTN_LOAD_AUX
#define MISSM_ATTN_LOAD_AUX 0
#endif

// stage L rows (zero-filled up to LP) of one head's [L, 64] slice into a swizzled LDS tile with global_load_lds:
// every 1-KiB piece is one wave instruction, all pieces of the tile are in flight together (no VGPR round trip);
// the caller waits with vmcnt(0) + barrier.  LDS position (row, chunk c') holds source chunk c' ^ (row & M).
template <typename T, int RBv, int NWV>
__device__ __forceinline__ void stage_head(char* lds, const T* src, size_t base, int tok_stride, int ld, int col0, int L, int LP,
                                           int lane, int wave) {
  constexpr int NC = RBv / 16, EPC = 16 / sizeof(T), RPP = 1024 / RBv, M = (NC - 1) & 7;
  using gptr = const __attribute__((address_space(1))) void*;
  using lptr = __attribute__((address_space(3))) void*;
  const int npieces = LP / RPP;
  for (int pi = wave; pi < npieces; pi += NWV) {
    const int row = pi * RPP + lane / NC, c = (lane % NC) ^ (row & M);
    const T* p = (row < L) ? src + (base + (size_t)row * tok_stride) * ld + col0 + c * EPC
                           : reinterpret_cast<const T*>(g_attn_zero16);
    __builtin_amdgcn_global_load_lds((gptr)p, (lptr)(lds + pi * 1024), 16, 0, MISSM_ATTN_LOAD_AUX);
  }
}
template <bool PW> __device__ __forceinline__ void stage_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (!PW) __syncthreads();   // per-wave units own their LDS slice: the wave's own vmcnt is the only dependency
}

// PW = false: one workgroup (8 waves) per (sequence, head), tiles dealt over the waves  (S = 197 / 77)
// PW = true : one WAVE per (sequence, head) with its own LDS slice, no workgroup barrier (time attention, L = T <= 32):
//             the 8x8 score block rides in one 16x16 MFMA tile; 6 MFMAs per unit instead of ~2000 VALU FMAs per lane.
// MASK: the additive key bias (key_padding_mask / padded keys, -inf) is applied to every key tile; otherwise only to the last two.
// FULL: the host guarantees L > 16 (NTP - 2): every key tile but the last two holds 16 keys, so the bulk of the tile loop carries
//       no run-time condition at all (the conditions cost more than they saved: 145 branches, 150 waits and ~130 register
//       zero-fills per query tile in the first conditional version - the kernel is VALU-issue-bound).
template <typename T, int NTP, bool PW, bool CAUSAL, bool MASK = true, bool FULL = false>
__global__ __launch_bounds__(ATHREADS, sizeof(T) == 2 ? 4 : 2) void attn_fwd_mfma_kernel(AttnArgs a) {
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T);
  constexpr int LP = NTP * 16;
  constexpr int KSQ = HD / M_::KS;            // k-steps over head_dim
  constexpr int NU = NTP / M_::CTILES;        // C-as-operand steps over keys
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  constexpr int SLICE = 2 * LP * RBv + LP * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  char* smem = smem_all + (PW ? wave * SLICE : 0);
  char* ldsK = smem;
  char* ldsV = smem + LP * RBv;
  float* kbias = reinterpret_cast<float*>(smem + 2 * LP * RBv);

  // (units in launch order: sending the 12 heads of a sequence - 128-byte slices of the same QKV rows - to ONE XCD through
  //  xcd_remap was measured slower, forward 78 -> 83 us, backward 264 -> 272 us)
  const int unit = PW ? blockIdx.x * ANW + wave : blockIdx.x;
  if (PW && unit >= a.nseq * a.H) return;
  const int seq = unit / a.H, h = unit % a.H;
  const size_t base = seq_base(a, seq);
  const T* qkv = static_cast<const T*>(a.qkv);
  const int L = a.L;
  const float sl2 = a.scale * 1.4426950408889634f;   // scale * log2(e)
  constexpr int NWV = PW ? 1 : attn_waves(NTP);   // waves cooperating on one unit
  const int wv = PW ? 0 : wave;               // this wave's index among them

  stage_head<T, RBv, NWV>(ldsK, qkv, base, a.tok_stride, a.ld, a.d + h * HD, L, LP, lane, wv);
  stage_head<T, RBv, NWV>(ldsV, qkv, base, a.tok_stride, a.ld, 2 * a.d + h * HD, L, LP, lane, wv);
  for (int k = PW ? lane : tid; k < LP; k += PW ? 64 : NWV * 64)
    kbias[k] = (k < L && (!a.key_mask || a.key_mask[(size_t)seq * L + k] != 0)) ? 0.f : kNegInf;

  // Q fragments of every query tile this wave owns are fetched while the K/V pieces are still in flight
  const int nqt = (L + 15) / 16;
  constexpr int MAXQ = (NTP + NWV - 1) / NWV;
  Frag qfa[MAXQ][KSQ];
#pragma unroll
  for (int t = 0; t < MAXQ; ++t) {
    const int qi = (wv + t * NWV) * 16 + li;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      qfa[t][ks] = M_::zero();
      if (qi < L) qfa[t][ks] = *reinterpret_cast<const Frag*>(qkv + (base + (size_t)qi * a.tok_stride) * a.ld + h * HD + ks * M_::KS + lg * M_::KPL);
    }
  }
  stage_wait<PW>();

#pragma unroll
  for (int t = 0; t < MAXQ; ++t) {
    const int qt = wv + t * NWV;
    if (qt >= nqt) break;
    const int qi = qt * 16 + li;
    const bool qvalid = qi < L;
    const size_t qrow = base + (size_t)(qvalid ? qi : 0) * a.tok_stride;
    Frag (&qf)[KSQ] = qfa[t];
    // The kernel is VALU-issue-bound (the exp2 alone is 2 of ~7.5 issue slots per score in the first version, 56 MFMAs per 16
    // queries against ~450 VALU instructions), so everything that is not the exp2 is kept off the VALU where possible:
    //  * key tiles that hold no key at all (L = 197 pads to 14 tiles of 16, the 14th is empty) are neither multiplied nor
    //    exponentiated: their probabilities are literal zeros;
    //  * the additive key bias (padding / key_padding_mask, -inf) is applied only where a tile can hold a masked key;
    //  * the row maximum runs over the RAW accumulators as v_max3 (two scores per instruction), the softmax scale and the
    //    maximum meet in ONE fused multiply-add in front of the exp2: p = exp2(s * scale * log2e - max * scale * log2e);
    //  * the row sum is one more row of the PV product (a ones-row appended to V^T: 7 MFMAs on the idle matrix pipe instead
    //    of 56 adds), taken from the SAME rounded probabilities that multiply V;
    //  * fragment addresses are lane constants + immediates (the swizzle key of row 16 kt + li is li & 7 for every kt).
    const int nkt = (L + 15) >> 4;            // key tiles that hold at least one key
    constexpr int NFULL = (FULL && !PW) ? NTP - 2 : 0;
    int koff[KSQ];
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) koff[ks] = swz<RBv>(li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T));
    f32x4 p[NTP];
    float mx = kNegInf;
#pragma unroll
    for (int kt = 0; kt < NTP; ++kt) {
      if (kt < NFULL || kt < nkt) {           // (compile-time true for the full tiles)
        f32x4 acc = M_::step(lds_frag<T>(ldsK + kt * 16 * RBv, koff[0]), qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int ks = 1; ks < KSQ; ++ks) acc = M_::step(lds_frag<T>(ldsK + kt * 16 * RBv, koff[ks]), qf[ks], acc);
        if (MASK || kt >= NFULL) acc += *reinterpret_cast<const f32x4*>(kbias + kt * 16 + 4 * lg);
        if constexpr (CAUSAL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (kt * 16 + 4 * lg + r > qi) acc[r] = kNegInf;
        }
        mx = fmaxf(fmaxf(mx, acc[0]), acc[1]);           // v_max3_f32
        mx = fmaxf(fmaxf(mx, acc[2]), acc[3]);
        p[kt] = acc;
      } else {
        p[kt] = f32x4{kNegInf, kNegInf, kNegInf, kNegInf};
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == kNegInf) mx = 0.f;
    const float mxs = mx * sl2;
#pragma unroll
    for (int kt = 0; kt < NTP; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[kt][r] = __builtin_amdgcn_exp2f(p[kt][r] * sl2 - mxs);    // (empty tiles: exp2(-inf) = 0)
    // PV with the key steps OUTSIDE: a probability fragment is packed, used for the four 16-wide slices of the head (+ the
    // ones-row: O^T gets a 65th row from a V^T row of ones, whose A operand is a constant) and dropped - nothing but the five
    // accumulators lives across steps (the fp32 instantiation spilled 888 bytes per lane holding all fragments).
    Frag ones = M_::zero();
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < M_::KPL; ++j) ones[j] = (T)1.0f;
    }
    f32x4 oacc[HD / 16], sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (!(u * M_::CTILES < NFULL || u * M_::CTILES < nkt)) continue;      // (wave-uniform) a step whose key tiles are all empty
      const Frag pf = M_::from_acc(p[u * M_::CTILES], p[u * M_::CTILES + M_::CTILES - 1]);
      sacc = M_::step(ones, pf, sacc);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) oacc[dt] = M_::step(TrFrag<T, RBv>::load(ldsV, u * M_::KS, dt * 16, lane), pf, oacc[dt]);
    }
    const float sum = __shfl(sacc[0], li, 64);   // row 0 of the ones tile lives in register 0 of the lanes with lg == 0
    const float inv = 1.0f / sum;
    if (a.lse && qvalid && lg == 0) a.lse[((size_t)seq * a.H + h) * L + qi] = (mxs + __log2f(sum)) * 0.6931471805599453f;
    T* out = static_cast<T*>(a.out);
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      oacc[dt] *= inv;
      if (qvalid) store4(out + qrow * a.ldo + h * HD + dt * 16 + 4 * lg, oacc[dt]);
    }
  }
}

// dot product of the k values two lanes hold in matching operand fragments (D = rowsum(dO . O))
__device__ __forceinline__ float frag_dot(bf16x8 x, bf16x8 y, float s) {
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
#pragma unroll
  for (int p = 0; p < 4; ++p) s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{x[2 * p], x[2 * p + 1]}, bf16x2{y[2 * p], y[2 * p + 1]}, s, false);
  return s;
}
__device__ __forceinline__ float frag_dot(f32x4 x, f32x4 y, float s) { return s + ((x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3])); }

// Backward.  D[q] = sum_j P[q,j] dP[q,j] is taken from the saved forward output instead (D = dO[q,:] . O[q,:], the same number):
// the probabilities / score gradients of a tile row can then be packed into MFMA operand fragments as they are produced,
// two 16-key tiles at a time, instead of living in 2 x NTP fp32 tiles until the row sum is known - 216 -> <= 128 VGPRs at
// S = 197, i.e. two 8-wave workgroups per CU instead of one.
//
// MASK / FULL as in the forward: with FULL (L > 16 (NTP - 2)) and no key mask the first NTP - 2 tiles of either pass carry no
// run-time condition, so the tile loop is straight-line code the compiler can schedule ACROSS tiles.  The conditional version
// issued every tile as  ds_read -> wait -> 2 dependent MFMAs -> branch -> exp -> branch  with nothing of the next tile in flight:
// 47 % of the wave cycles parked on s_waitcnt, 32 % on issue dependencies (profiles/r02_attn_sq_counters.txt).
template <typename T, int NTP, bool PW, bool CAUSAL, bool MASK = true, bool FULL = false>
__global__ __launch_bounds__(ATHREADS, sizeof(T) == 2 ? 4 : 2) void attn_bwd_mfma_kernel(AttnArgs a) {
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T);
  constexpr int LP = NTP * 16;
  constexpr int KSQ = HD / M_::KS;
  constexpr int CT = M_::CTILES;
  constexpr int NU = NTP / CT;
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  constexpr int SLICE = 2 * LP * RBv + 3 * LP * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  char* smem = smem_all + (PW ? wave * SLICE : 0);
  char* X0 = smem;                 // pass A: K   ; pass B: Q
  char* X1 = smem + LP * RBv;      // pass A: V   ; pass B: dO
  float* kbias = reinterpret_cast<float*>(smem + 2 * LP * RBv);
  float* lseL = kbias + LP;
  float* Dl = lseL + LP;

  const int unit = PW ? blockIdx.x * ANW + wave : blockIdx.x;
  if (PW && unit >= a.nseq * a.H) return;
  const int seq = unit / a.H, h = unit % a.H;
  constexpr int NWV = PW ? 1 : attn_waves(NTP);
  const int wv = PW ? 0 : wave;
  const size_t base = seq_base(a, seq);
  const T* qkv = static_cast<const T*>(a.qkv);
  const T* dout = static_cast<const T*>(a.dout);
  const T* fout = static_cast<const T*>(a.out);
  T* dqkv = static_cast<T*>(a.dqkv);
  const int L = a.L;
  const float sl2 = a.scale * 1.4426950408889634f;   // scale * log2(e)
  const float inv_scale = 1.0f / a.scale;             // lse / scale goes into LDS: it is the score accumulators' initial value
  const float* lse = a.lse + ((size_t)seq * a.H + h) * L;
  constexpr int NFULL = (FULL && !PW) ? NTP - 2 : 0;   // tiles (keys in pass A, queries in pass B) known to be full

  stage_head<T, RBv, NWV>(X0, qkv, base, a.tok_stride, a.ld, a.d + h * HD, L, LP, lane, wv);
  stage_head<T, RBv, NWV>(X1, qkv, base, a.tok_stride, a.ld, 2 * a.d + h * HD, L, LP, lane, wv);
  for (int k = PW ? lane : tid; k < LP; k += PW ? 64 : NWV * 64) {
    kbias[k] = (k < L && (!a.key_mask || a.key_mask[(size_t)seq * L + k] != 0)) ? 0.f : kNegInf;
    lseL[k] = k < L ? lse[k] * inv_scale : __builtin_huge_valf();
  }
  const int nt = (L + 15) / 16;
  constexpr int MAXQ = (NTP + NWV - 1) / NWV;   // query / key tiles per wave
  // Q / dO fragments of this wave's FIRST query tile are fetched while the K / V pieces are in flight (the second tile's at
  // its turn: holding both costs 16 VGPRs the 128-register budget of two workgroups per CU does not have at S = 197);
  // D of every tile is computed up front, pass B needs all of them.
  auto load_q = [&](int qt, Frag (&qf)[KSQ], Frag (&dof)[KSQ]) -> float {
    const int qi = qt * 16 + li;
    const size_t qrow = base + (size_t)(qi < L ? qi : 0) * a.tok_stride;
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      qf[ks] = M_::zero(); dof[ks] = M_::zero();
      if (qi < L) {
        qf[ks] = *reinterpret_cast<const Frag*>(qkv + qrow * a.ld + h * HD + ks * M_::KS + lg * M_::KPL);
        dof[ks] = *reinterpret_cast<const Frag*>(dout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL);
        const Frag of = *reinterpret_cast<const Frag*>(fout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL);
        dsum = frag_dot(dof[ks], of, dsum);
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    return dsum;
  };
  Frag qf0[KSQ], dof0[KSQ];
  float dsa[MAXQ];
  dsa[0] = load_q(wv, qf0, dof0);
  if (lg == 0) Dl[wv * 16 + li] = dsa[0];
#pragma unroll
  for (int t = 1; t < MAXQ; ++t) {          // D of the later tiles only (every query tile belongs to exactly one wave)
    const int qi = (wv + t * NWV) * 16 + li;
    const size_t qrow = base + (size_t)(qi < L ? qi : 0) * a.tok_stride;
    float dsum = 0.f;
    if (qi < L) {
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks)
        dsum = frag_dot(*reinterpret_cast<const Frag*>(dout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL),
                        *reinterpret_cast<const Frag*>(fout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL), dsum);
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    dsa[t] = dsum;
    if (lg == 0 && qi < LP) Dl[qi] = dsum;
  }
  stage_wait<PW>();

  // ---------------- pass A: query on the lane -> dQ ----------------
#pragma unroll
  for (int t = 0; t < MAXQ; ++t) {
    const int qt = wv + t * NWV;
    if (qt >= nt) break;
    const int qi = qt * 16 + li;
    const bool qvalid = qi < L;
    const size_t qrow = base + (size_t)(qvalid ? qi : 0) * a.tok_stride;
    Frag qf[KSQ], dof[KSQ];
    if (t == 0) {
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) { qf[ks] = qf0[ks]; dof[ks] = dof0[ks]; }
    } else {
      (void)load_q(qt, qf, dof);
    }
    // Row constants ride in as the INITIAL accumulators (the kernel is VALU-issue-bound): S' = q.k - lse / scale and
    // dP' = dO.v - D leave the MFMA chains ready, p = exp2(S' * scale * log2e) needs no subtraction, dS = p * dP' no difference.
    // An out-of-range query has lse = +inf: S' = -inf, p = 0.  Key tiles without any key are skipped (dS = 0).
    const float nlq = -lseL[qi];
    const float ndsum = -dsa[t];
    f32x4 dqa[HD / 16];                       // dQ^T of this query tile: accumulated key step by key step (no fragment array)
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dqa[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u * CT >= NFULL && u * CT >= nt) continue;             // (wave-uniform) no key in this step
      f32x4 dsv[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int kt = u * CT + c;
        if (kt >= NFULL && kt >= nt) { dsv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
        f32x4 sc = {nlq, nlq, nlq, nlq}, dp = {ndsum, ndsum, ndsum, ndsum};
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
          const int off = swz<RBv>(kt * 16 + li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T));
          sc = M_::step(lds_frag<T>(X0, off), qf[ks], sc);
          dp = M_::step(lds_frag<T>(X1, off), dof[ks], dp);
        }
        if (MASK || kt >= NFULL) sc += *reinterpret_cast<const f32x4*>(kbias + kt * 16 + 4 * lg);   // (-inf on masked / padded keys)
        // (the softmax scale is applied once to the dQ / dK accumulators instead of to every dS element)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(sc[r] * sl2);
          if constexpr (CAUSAL) { if (kt * 16 + 4 * lg + r > qi) pv = 0.f; }
          dsv[c][r] = pv * dp[r];
        }
      }
      const Frag dsf = M_::from_acc(dsv[0], dsv[CT - 1]);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) dqa[dt] = M_::step(TrFrag<T, RBv>::load(X0, u * M_::KS, dt * 16, lane), dsf, dqa[dt]);
    }
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      dqa[dt] *= a.scale;
      if (qvalid) store4(dqkv + qrow * a.ld + h * HD + dt * 16 + 4 * lg, dqa[dt]);
    }
  }
  if constexpr (!PW) __syncthreads();
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---------------- pass B: key on the lane -> dK, dV ----------------
  stage_head<T, RBv, NWV>(X0, qkv, base, a.tok_stride, a.ld, h * HD, L, LP, lane, wv);
  stage_head<T, RBv, NWV>(X1, dout, base, a.tok_stride, a.ldo, h * HD, L, LP, lane, wv);
  // (this wave's K / V fragments are fetched per key tile: prefetching both tiles costs 16 more live VGPRs, which at
  //  S = 197 is the difference between spilling and not under the 128-VGPR budget of two workgroups per CU)
  Frag kf0[KSQ], vf0[KSQ];
  {
    const int key = wv * 16 + li;
    const size_t krow = base + (size_t)(key < L ? key : 0) * a.tok_stride;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      kf0[ks] = M_::zero(); vf0[ks] = M_::zero();
      if (key < L) {
        kf0[ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + a.d + h * HD + ks * M_::KS + lg * M_::KPL);
        vf0[ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + 2 * a.d + h * HD + ks * M_::KS + lg * M_::KPL);
      }
    }
  }
  stage_wait<PW>();
#pragma unroll
  for (int t = 0; t < MAXQ; ++t) {
    const int kt = wv + t * NWV;
    if (kt >= nt) break;
    const int key = kt * 16 + li;
    const bool kin = key < L;
    const size_t krow = base + (size_t)(kin ? key : 0) * a.tok_stride;
    Frag kf[KSQ], vf[KSQ];
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      if (t == 0) { kf[ks] = kf0[ks]; vf[ks] = vf0[ks]; }
      else {
        kf[ks] = M_::zero(); vf[ks] = M_::zero();
        if (kin) {
          kf[ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + a.d + h * HD + ks * M_::KS + lg * M_::KPL);
          vf[ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + 2 * a.d + h * HD + ks * M_::KS + lg * M_::KPL);
        }
      }
    }
    const float kb = kbias[key];               // 0, or -inf on a masked / padded key: rides in the exp2's multiply-add
    f32x4 dva[HD / 16], dka[HD / 16];          // dV^T / dK^T of this key tile, accumulated query step by query step
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) { dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dka[dt] = dva[dt]; }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u * CT >= NFULL && u * CT >= nt) continue;             // (wave-uniform) no query in this step
      f32x4 pv4[CT], dsv[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int qt = u * CT + c;
        if (qt >= NFULL && qt >= nt) { pv4[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dsv[c] = pv4[c]; continue; }     // query tile without any query
        // initial accumulators: -lse / scale and -D of the tile's 4 query rows this lane holds (see pass A)
        f32x4 sc = -*reinterpret_cast<const f32x4*>(lseL + qt * 16 + 4 * lg);
        f32x4 dp = -*reinterpret_cast<const f32x4*>(Dl + qt * 16 + 4 * lg);
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
          const int off = swz<RBv>(qt * 16 + li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T));
          sc = M_::step(lds_frag<T>(X0, off), kf[ks], sc);
          dp = M_::step(lds_frag<T>(X1, off), vf[ks], dp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(sc[r] * sl2 + kb);
          if constexpr (CAUSAL) { if (key > qt * 16 + 4 * lg + r) pv = 0.f; }
          pv4[c][r] = pv;
          dsv[c][r] = pv * dp[r];
        }
      }
      const Frag pf = M_::from_acc(pv4[0], pv4[CT - 1]);
      const Frag dsf = M_::from_acc(dsv[0], dsv[CT - 1]);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        dva[dt] = M_::step(TrFrag<T, RBv>::load(X1, u * M_::KS, dt * 16, lane), pf, dva[dt]);
        dka[dt] = M_::step(TrFrag<T, RBv>::load(X0, u * M_::KS, dt * 16, lane), dsf, dka[dt]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      dka[dt] *= a.scale;
      if (kin) {
        store4(dqkv + krow * a.ld + a.d + h * HD + dt * 16 + 4 * lg, dka[dt]);
        store4(dqkv + krow * a.ld + 2 * a.d + h * HD + dt * 16 + 4 * lg, dva[dt]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Single-pass backward (bf16, head_dim 64, 97 <= L <= 224, not causal).  The two-pass kernel above recomputes the scores and
// dP twice (query-on-lane for dQ, key-on-lane for dK / dV): 2444 MFMAs, 2 x 169 score tiles of exp2 / VALU work and two LDS
// stagings per (sequence, head).  Here ONE key-on-lane sweep feeds all three gradients:
//   * 16 waves, one workgroup per CU; Q, dO and K sit in LDS (3 x 28 KiB), this wave's K / V fragments in registers;
//   * wave kt < nt ("S-wave") owns key tile kt: per block of 32 queries it forms S^T and dP^T tiles (4 MFMAs), p = exp2(...),
//     dS = p * dP', accumulates dV^T += dO^T P and dK^T += Q^T dS through C-as-operand fragments (8 MFMAs) exactly like
//     pass B above - and drops its bf16 dS tile into a double-buffered LDS block [32 queries][224 keys];
//   * the remaining 16 - nt waves ("dQ-waves") turn the PREVIOUS block's dS into dQ^T = K^T dS^T (8 output tiles x 7 key
//     steps, A = transposed K reads, B = one ds_read_b128 of the dS block) while the S-waves are on the next block:
//     one workgroup barrier per block, nothing is accumulated with atomics, dQ is written once.
//   The dS block stores the keys of each 32-key group in the C-as-operand order (mma.h) so that the dQ-waves' B fragment
//   matches TrFrag's k order with a single 16-byte read.
// Per (sequence, head) at S = 197: 1796 MFMAs, 169 score tiles of VALU work, one staging.
//
// MEASURED SLOWER than the two-pass kernel and therefore OPT-IN (MISSM_ATTN_SP=1; parity-tested either way): 318 us against
// 267 us at 256 frames x 12 heads.  Three 28 KiB operand tiles leave room for ONE workgroup per CU, so nothing hides a unit's
// memory phase: with both compute stages switched off the kernel still takes 158 us (13.2 us per unit: 2.3 launch, the rest
// the 179 KB a unit moves at an effective 3.6 TB/s - a head's slice is 128 bytes of every 4.6 KB QKV row), the stages add
// 86 us (S-waves) + 81 us (dQ-waves) on top, and starting the CUs 2-6 us apart only added the delay.  The two-pass kernel
// runs two workgroups per CU and overlaps one's staging with the other's MFMAs.  What would make this design pay: Q / dO
// streamed per 32-query block (LDS 76 KB -> the next unit's K load under the last blocks), or a head-major QKV layout.
// ---------------------------------------------------------------------------------------------------
constexpr int SPW = 16;                    // waves of the single-pass workgroup
constexpr int DSP = 464;                   // bytes per query row of the dS block (224 keys x 2 B, padded: 4 rows apart = +16 banks)

template <int NTP>
__global__ __launch_bounds__(SPW * 64) void attn_bwd_sp_kernel(AttnArgs a) {
  using T = bf16;
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T);      // 128
  constexpr int LP = NTP * 16;
  constexpr int KSQ = HD / M_::KS;         // 2
  constexpr int NU = NTP / 2;              // blocks of 32 queries / steps of 32 keys
  static_assert(NTP % 2 == 0 && NTP <= SPW - 2 && LP * 2 <= DSP, "single-pass backward: tile count");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* XQ = smem;                         // Q   [LP][64]
  char* XO = smem + LP * RBv;              // dO
  char* XK = smem + 2 * LP * RBv;          // K
  char* dsb = smem + 3 * LP * RBv;         // [2][32][DSP]
  float* kbias = reinterpret_cast<float*>(dsb + 2 * 32 * DSP);
  float* nlse = kbias + LP;                // -lse / scale (-inf past L): initial score accumulators
  float* nD = nlse + LP;                   // -rowsum(dO . O): initial dP accumulators

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  const int unit = blockIdx.x;
  const int seq = unit / a.H, h = unit % a.H;
  const size_t base = seq_base(a, seq);
  const T* qkv = static_cast<const T*>(a.qkv);
  const T* dout = static_cast<const T*>(a.dout);
  const T* fout = static_cast<const T*>(a.out);
  T* dqkv = static_cast<T*>(a.dqkv);
  const int L = a.L;
  const float sl2 = a.scale * 1.4426950408889634f;
  const float inv_scale = 1.0f / a.scale;
  const float* lse = a.lse + ((size_t)seq * a.H + h) * L;
  const int nt = (L + 15) >> 4;            // key / query tiles that hold at least one row (host: nt <= NTP <= 14)

  stage_head<T, RBv, SPW>(XQ, qkv, base, a.tok_stride, a.ld, h * HD, L, LP, lane, wave);
  stage_head<T, RBv, SPW>(XO, dout, base, a.tok_stride, a.ldo, h * HD, L, LP, lane, wave);
  stage_head<T, RBv, SPW>(XK, qkv, base, a.tok_stride, a.ld, a.d + h * HD, L, LP, lane, wave);
  for (int k = tid; k < LP; k += SPW * 64) {
    kbias[k] = (k < L && (!a.key_mask || a.key_mask[(size_t)seq * L + k] != 0)) ? 0.f : kNegInf;
    nlse[k] = k < L ? -lse[k] * inv_scale : kNegInf;
  }
  for (int i = tid; i < 2 * 32 * DSP / 16; i += SPW * 64) reinterpret_cast<f32x4*>(dsb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave < NTP) {                        // D of query tile `wave` (dO . O over the head, 16 columns per lane, summed over lg)
    const int qi = wave * 16 + li;
    const size_t qrow = base + (size_t)(qi < L ? qi : 0) * a.tok_stride;
    float dsum = 0.f;
    if (qi < L) {
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks)
        dsum = frag_dot(*reinterpret_cast<const Frag*>(dout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL),
                        *reinterpret_cast<const Frag*>(fout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL), dsum);
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    if (lg == 0) nD[qi] = -dsum;
  }
  const bool s_wave = wave < nt;           // wave-uniform role
  const int key = wave * 16 + li;          // (S-waves) this lane's key column
  const bool kin = s_wave && key < L;
  const size_t krow = base + (size_t)(kin ? key : 0) * a.tok_stride;
  Frag kf[KSQ], vf[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) {
    kf[ks] = M_::zero(); vf[ks] = M_::zero();
    if (kin) {
      kf[ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + a.d + h * HD + ks * M_::KS + lg * M_::KPL);
      vf[ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + 2 * a.d + h * HD + ks * M_::KS + lg * M_::KPL);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float kb = s_wave ? kbias[key] : 0.f;
  f32x4 dva[HD / 16], dka[HD / 16];        // (S-waves) dV^T / dK^T of the key tile
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) { dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dka[dt] = dva[dt]; }
  // dS block addressing.  Writer: query row c * 16 + 4 lg + r, key slot inside the 32-key group = 8 (li >> 2) + 4 (kt & 1) + (li & 3)
  const int ds_wr = (4 * lg) * DSP + ((wave >> 1) * 32 + 8 * (li >> 2) + 4 * (wave & 1) + (li & 3)) * 2;
  const int ndq = SPW - nt, dq_idx = wave - nt;   // (dQ-waves)

  // Block qb: S-waves fill dS buffer qb & 1, barrier, dQ-waves consume it while the S-waves fill the other buffer for block
  // qb + 1; the next barrier is passed only when the dQ-waves are done with buffer qb & 1, which the S-waves refill in qb + 2.
#pragma unroll 1
  for (int qb = 0; qb < NU; ++qb) {
    if (2 * qb >= nt) break;               // (uniform) no query left
    if (s_wave) {
      char* buf = dsb + (qb & 1) * (32 * DSP);
      f32x4 pv4[2], dsv[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int qt = 2 * qb + c;
        if (qt >= nt) { pv4[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dsv[c] = pv4[c]; continue; }   // (uniform) last block, odd tile count
        f32x4 sc = *reinterpret_cast<const f32x4*>(nlse + qt * 16 + 4 * lg);
        f32x4 dp = *reinterpret_cast<const f32x4*>(nD + qt * 16 + 4 * lg);
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
          const int off = swz<RBv>(qt * 16 + li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T));
          sc = M_::step(lds_frag<T>(XQ, off), kf[ks], sc);
          dp = M_::step(lds_frag<T>(XO, off), vf[ks], dp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(sc[r] * sl2 + kb);
          pv4[c][r] = pv;
          dsv[c][r] = pv * dp[r];
        }
      }
      const Frag pf = M_::from_acc(pv4[0], pv4[1]);
      const Frag dsf = M_::from_acc(dsv[0], dsv[1]);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<T*>(buf + ds_wr + (c * 16 + r) * DSP) = dsf[4 * c + r];
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        dva[dt] = M_::step(TrFrag<T, RBv>::load(XO, qb * M_::KS, dt * 16, lane), pf, dva[dt]);
        dka[dt] = M_::step(TrFrag<T, RBv>::load(XQ, qb * M_::KS, dt * 16, lane), dsf, dka[dt]);
      }
    }
    __syncthreads();
    if (!s_wave) {
      const char* buf = dsb + (qb & 1) * (32 * DSP);
      for (int t = dq_idx; t < 8; t += ndq) {          // output tile (query tile t >> 2 of the block, head slice t & 3)
        const int qt = 2 * qb + (t >> 2), dt = t & 3;
        if (qt >= nt) continue;
        const char* brow = buf + ((t >> 2) * 16 + li) * DSP + lg * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (2 * u >= nt) continue;                     // (uniform) no key in this step
          acc = M_::step(TrFrag<T, RBv>::load(XK, u * M_::KS, dt * 16, lane), *reinterpret_cast<const Frag*>(brow + u * 64), acc);
        }
        acc *= a.scale;
        const int qi = qt * 16 + li;
        if (qi < L) store4(dqkv + (base + (size_t)qi * a.tok_stride) * a.ld + h * HD + dt * 16 + 4 * lg, acc);
      }
    }
  }
  if (s_wave) {
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      dka[dt] *= a.scale;
      if (kin) {
        store4(dqkv + krow * a.ld + a.d + h * HD + dt * 16 + 4 * lg, dka[dt]);
        store4(dqkv + krow * a.ld + 2 * a.d + h * HD + dt * 16 + 4 * lg, dva[dt]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Long sequences (L > 256, head_dim 64, not causal): the released audio checkpoint's 8 x 74 spectrogram grid is 593 tokens per
// frame - more keys than fit in LDS at once.  K / V travel through LDS in chunks of 14 key tiles (224 keys; 8 with fp32); a workgroup
// owns a block of 16 query tiles (two per wave; 8 / one per wave with fp32 operands) of one (sequence, head) and walks the chunks:
//   forward : online softmax - running row maximum and row sum per query, the O^T accumulators are rescaled when the maximum moves
//             (the row sum is the ones-row accumulator of the PV product, so it is rescaled by the same multiply);
//   backward: no rescaling at all (p = exp2(s c - lse) from the saved log-sum-exp): the dQ kernel walks key chunks for a query
//             block, the dK / dV kernel walks QUERY chunks (Q, dO, lse, D staged per chunk) for a block of 16 key tiles.
// Same fragment conventions as the kernels above (scores transposed, C-as-operand products); this path trades speed for reach:
// runtime chunk loops, up to 256 VGPRs, one workgroup per CU.
// ---------------------------------------------------------------------------------------------------
// key (query) tiles per LDS chunk: 14 (224 keys) with bf16 operands, 8 with fp32 (the fully unrolled chunk body otherwise spills)
template <typename T> constexpr int long_ck() { return sizeof(T) == 2 ? 14 : 8; }
// query (key) tiles per wave of a long-sequence workgroup: two with bf16 operands, one with fp32 (twice the fragment registers)
template <typename T> constexpr int long_tpw() { return sizeof(T) == 2 ? 2 : 1; }

template <typename T>
__global__ __launch_bounds__(ATHREADS, 2) void attn_fwd_long_kernel(AttnArgs a) {
  constexpr int TPW = long_tpw<T>(), LQB = TPW * ANW, LCK = long_ck<T>();
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T);
  constexpr int LP = LCK * 16;
  constexpr int KSQ = HD / M_::KS;
  constexpr int NU = LCK / M_::CTILES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsK = smem;
  char* ldsV = smem + LP * RBv;
  float* kbias = reinterpret_cast<float*>(smem + 2 * LP * RBv);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  const int nqb = (((a.L + 15) >> 4) + LQB - 1) / LQB;
  const int unit = blockIdx.x / nqb, qb = blockIdx.x % nqb;
  const int seq = unit / a.H, h = unit % a.H;
  const size_t base = seq_base(a, seq);
  const T* qkv = static_cast<const T*>(a.qkv);
  const int L = a.L;
  const float sl2 = a.scale * 1.4426950408889634f;
  const int nchunks = (L + LP - 1) / LP;

  Frag qf[TPW][KSQ];
  float mrun[TPW];
  f32x4 oacc[TPW][HD / 16], sacc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int qi = (qb * LQB + wave + t * ANW) * 16 + li;
    sacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mrun[t] = kNegInf;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) oacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      qf[t][ks] = M_::zero();
      if (qi < L) qf[t][ks] = *reinterpret_cast<const Frag*>(qkv + (base + (size_t)qi * a.tok_stride) * a.ld + h * HD + ks * M_::KS + lg * M_::KPL);
    }
  }
  Frag ones = M_::zero();
  if (li == 0) {
#pragma unroll
    for (int j = 0; j < M_::KPL; ++j) ones[j] = (T)1.0f;
  }
  for (int c = 0; c < nchunks; ++c) {
    const int k0 = c * LP, kn = min(L - k0, LP);            // keys of this chunk
    __syncthreads();                                         // the previous chunk's fragments are all read
    stage_head<T, RBv, ANW>(ldsK, qkv, base + (size_t)k0 * a.tok_stride, a.tok_stride, a.ld, a.d + h * HD, kn, LP, lane, wave);
    stage_head<T, RBv, ANW>(ldsV, qkv, base + (size_t)k0 * a.tok_stride, a.tok_stride, a.ld, 2 * a.d + h * HD, kn, LP, lane, wave);
    for (int k = tid; k < LP; k += ATHREADS)
      kbias[k] = (k < kn && (!a.key_mask || a.key_mask[(size_t)seq * L + k0 + k] != 0)) ? 0.f : kNegInf;
    stage_wait<false>();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if ((qb * LQB + wave + t * ANW) * 16 >= L) continue;   // (wave-uniform) no query in this tile
      f32x4 p[LCK];
      float mx = kNegInf;
#pragma unroll
      for (int kt = 0; kt < LCK; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks)
          acc = M_::step(lds_frag<T>(ldsK + kt * 16 * RBv, swz<RBv>(li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T))), qf[t][ks], acc);
        acc += *reinterpret_cast<const f32x4*>(kbias + kt * 16 + 4 * lg);
        mx = fmaxf(fmaxf(mx, acc[0]), acc[1]);
        mx = fmaxf(fmaxf(mx, acc[2]), acc[3]);
        p[kt] = acc;
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // (no branch on the per-lane maximum: the transposed LDS reads below need every lane.  All keys so far masked: mnew = -inf,
      //  the probabilities are exp2(-inf) = 0 and the zero accumulators stay zero.)
      const float mnew = fmaxf(mrun[t], mx);
      const float alpha = mrun[t] == kNegInf ? 0.f : __builtin_amdgcn_exp2f((mrun[t] - mnew) * sl2);
      mrun[t] = mnew;
      const float mxs = mnew == kNegInf ? 0.f : mnew * sl2;
      sacc[t] *= alpha;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) oacc[t][dt] *= alpha;
#pragma unroll
      for (int kt = 0; kt < LCK; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[kt][r] = __builtin_amdgcn_exp2f(p[kt][r] * sl2 - mxs);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const Frag pf = M_::from_acc(p[u * M_::CTILES], p[u * M_::CTILES + M_::CTILES - 1]);
        sacc[t] = M_::step(ones, pf, sacc[t]);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) oacc[t][dt] = M_::step(TrFrag<T, RBv>::load(ldsV, u * M_::KS, dt * 16, lane), pf, oacc[t][dt]);
      }
    }
  }
  T* out = static_cast<T*>(a.out);
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int qi = (qb * LQB + wave + t * ANW) * 16 + li;
    if ((qb * LQB + wave + t * ANW) * 16 >= L) continue;
    const float sum = __shfl(sacc[t][0], li, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    const size_t qrow = base + (size_t)(qi < L ? qi : 0) * a.tok_stride;
    if (a.lse && qi < L && lg == 0)
      a.lse[((size_t)seq * a.H + h) * L + qi] = sum > 0.f ? (mrun[t] * sl2 + __log2f(sum)) * 0.6931471805599453f : kNegInf;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      oacc[t][dt] *= inv;
      if (qi < L) store4(out + qrow * a.ldo + h * HD + dt * 16 + 4 * lg, oacc[t][dt]);
    }
  }
}


// dQ of a block of 16 query tiles: key chunks through LDS (pass A of attn_bwd_mfma_kernel with a chunk loop)
template <typename T>
__global__ __launch_bounds__(ATHREADS, 2) void attn_bwd_long_dq_kernel(AttnArgs a) {
  constexpr int TPW = long_tpw<T>(), LQB = TPW * ANW, LCK = long_ck<T>();
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T);
  constexpr int LP = LCK * 16;
  constexpr int KSQ = HD / M_::KS;
  constexpr int CT = M_::CTILES;
  constexpr int NU = LCK / CT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* X0 = smem;                 // K chunk
  char* X1 = smem + LP * RBv;      // V chunk
  float* kbias = reinterpret_cast<float*>(smem + 2 * LP * RBv);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  const int nqb = (((a.L + 15) >> 4) + LQB - 1) / LQB;
  const int unit = blockIdx.x / nqb, qb = blockIdx.x % nqb;
  const int seq = unit / a.H, h = unit % a.H;
  const size_t base = seq_base(a, seq);
  const T* qkv = static_cast<const T*>(a.qkv);
  const T* dout = static_cast<const T*>(a.dout);
  const T* fout = static_cast<const T*>(a.out);
  T* dqkv = static_cast<T*>(a.dqkv);
  const int L = a.L;
  const float sl2 = a.scale * 1.4426950408889634f;
  const float inv_scale = 1.0f / a.scale;
  const float* lse = a.lse + ((size_t)seq * a.H + h) * L;
  const int nchunks = (L + LP - 1) / LP;

  Frag qf[TPW][KSQ], dof[TPW][KSQ];
  float nlq[TPW], nds[TPW];
  f32x4 dqa[TPW][HD / 16];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int qi = (qb * LQB + wave + t * ANW) * 16 + li;
    const size_t qrow = base + (size_t)(qi < L ? qi : 0) * a.tok_stride;
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      qf[t][ks] = M_::zero(); dof[t][ks] = M_::zero();
      if (qi < L) {
        qf[t][ks] = *reinterpret_cast<const Frag*>(qkv + qrow * a.ld + h * HD + ks * M_::KS + lg * M_::KPL);
        dof[t][ks] = *reinterpret_cast<const Frag*>(dout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL);
        dsum = frag_dot(dof[t][ks], *reinterpret_cast<const Frag*>(fout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL), dsum);
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    nds[t] = -dsum;
    nlq[t] = qi < L ? -lse[qi] * inv_scale : kNegInf;        // (a query past L: S' = -inf, p = 0)
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dqa[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int c = 0; c < nchunks; ++c) {
    const int k0 = c * LP, kn = min(L - k0, LP);
    __syncthreads();
    stage_head<T, RBv, ANW>(X0, qkv, base + (size_t)k0 * a.tok_stride, a.tok_stride, a.ld, a.d + h * HD, kn, LP, lane, wave);
    stage_head<T, RBv, ANW>(X1, qkv, base + (size_t)k0 * a.tok_stride, a.tok_stride, a.ld, 2 * a.d + h * HD, kn, LP, lane, wave);
    for (int k = tid; k < LP; k += ATHREADS)
      kbias[k] = (k < kn && (!a.key_mask || a.key_mask[(size_t)seq * L + k0 + k] != 0)) ? 0.f : kNegInf;
    stage_wait<false>();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if ((qb * LQB + wave + t * ANW) * 16 >= L) continue;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        f32x4 dsv[CT];
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
          const int kt = u * CT + cc;
          f32x4 sc = {nlq[t], nlq[t], nlq[t], nlq[t]}, dp = {nds[t], nds[t], nds[t], nds[t]};
#pragma unroll
          for (int ks = 0; ks < KSQ; ++ks) {
            const int off = swz<RBv>(kt * 16 + li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T));
            sc = M_::step(lds_frag<T>(X0, off), qf[t][ks], sc);
            dp = M_::step(lds_frag<T>(X1, off), dof[t][ks], dp);
          }
          sc += *reinterpret_cast<const f32x4*>(kbias + kt * 16 + 4 * lg);
#pragma unroll
          for (int r = 0; r < 4; ++r) dsv[cc][r] = __builtin_amdgcn_exp2f(sc[r] * sl2) * dp[r];
        }
        const Frag dsf = M_::from_acc(dsv[0], dsv[CT - 1]);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) dqa[t][dt] = M_::step(TrFrag<T, RBv>::load(X0, u * M_::KS, dt * 16, lane), dsf, dqa[t][dt]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int qi = (qb * LQB + wave + t * ANW) * 16 + li;
    if (qi >= L) continue;
    const size_t qrow = base + (size_t)qi * a.tok_stride;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      dqa[t][dt] *= a.scale;
      store4(dqkv + qrow * a.ld + h * HD + dt * 16 + 4 * lg, dqa[t][dt]);
    }
  }
}

// dK, dV of a block of 16 key tiles: QUERY chunks (Q, dO, -lse / scale, -D) through LDS (pass B with a chunk loop)
template <typename T>
__global__ __launch_bounds__(ATHREADS, 2) void attn_bwd_long_dkv_kernel(AttnArgs a) {
  constexpr int TPW = long_tpw<T>(), LQB = TPW * ANW, LCK = long_ck<T>();
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T);
  constexpr int LP = LCK * 16;
  constexpr int KSQ = HD / M_::KS;
  constexpr int CT = M_::CTILES;
  constexpr int NU = LCK / CT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* X0 = smem;                 // Q chunk
  char* X1 = smem + LP * RBv;      // dO chunk
  float* nlse = reinterpret_cast<float*>(smem + 2 * LP * RBv);
  float* nD = nlse + LP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  const int nkb = (((a.L + 15) >> 4) + LQB - 1) / LQB;
  const int unit = blockIdx.x / nkb, kb_ = blockIdx.x % nkb;
  const int seq = unit / a.H, h = unit % a.H;
  const size_t base = seq_base(a, seq);
  const T* qkv = static_cast<const T*>(a.qkv);
  const T* dout = static_cast<const T*>(a.dout);
  const T* fout = static_cast<const T*>(a.out);
  T* dqkv = static_cast<T*>(a.dqkv);
  const int L = a.L;
  const float sl2 = a.scale * 1.4426950408889634f;
  const float inv_scale = 1.0f / a.scale;
  const float* lse = a.lse + ((size_t)seq * a.H + h) * L;
  const int nchunks = (L + LP - 1) / LP;

  Frag kf[TPW][KSQ], vf[TPW][KSQ];
  float kbv[TPW];
  f32x4 dva[TPW][HD / 16], dka[TPW][HD / 16];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int key = (kb_ * LQB + wave + t * ANW) * 16 + li;
    const size_t krow = base + (size_t)(key < L ? key : 0) * a.tok_stride;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      kf[t][ks] = M_::zero(); vf[t][ks] = M_::zero();
      if (key < L) {
        kf[t][ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + a.d + h * HD + ks * M_::KS + lg * M_::KPL);
        vf[t][ks] = *reinterpret_cast<const Frag*>(qkv + krow * a.ld + 2 * a.d + h * HD + ks * M_::KS + lg * M_::KPL);
      }
    }
    kbv[t] = (key < L && (!a.key_mask || a.key_mask[(size_t)seq * L + key] != 0)) ? 0.f : kNegInf;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) { dva[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dka[t][dt] = dva[t][dt]; }
  }
  for (int c = 0; c < nchunks; ++c) {
    const int q0 = c * LP, qn = min(L - q0, LP);             // queries of this chunk
    __syncthreads();
    stage_head<T, RBv, ANW>(X0, qkv, base + (size_t)q0 * a.tok_stride, a.tok_stride, a.ld, h * HD, qn, LP, lane, wave);
    stage_head<T, RBv, ANW>(X1, dout, base + (size_t)q0 * a.tok_stride, a.tok_stride, a.ldo, h * HD, qn, LP, lane, wave);
    for (int k = tid; k < LP; k += ATHREADS) nlse[k] = k < qn ? -lse[q0 + k] * inv_scale : kNegInf;
    for (int qt = wave; qt < LCK; qt += ANW) {               // -D of the chunk's query tiles
      const int ql = qt * 16 + li;
      const size_t qrow = base + (size_t)(ql < qn ? q0 + ql : 0) * a.tok_stride;
      float dsum = 0.f;
      if (ql < qn) {
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks)
          dsum = frag_dot(*reinterpret_cast<const Frag*>(dout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL),
                          *reinterpret_cast<const Frag*>(fout + qrow * a.ldo + h * HD + ks * M_::KS + lg * M_::KPL), dsum);
      }
      dsum += __shfl_xor(dsum, 16, 64);
      dsum += __shfl_xor(dsum, 32, 64);
      if (lg == 0) nD[ql] = -dsum;
    }
    stage_wait<false>();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if ((kb_ * LQB + wave + t * ANW) * 16 >= L) continue;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        f32x4 pv4[CT], dsv[CT];
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
          const int qt = u * CT + cc;
          f32x4 sc = *reinterpret_cast<const f32x4*>(nlse + qt * 16 + 4 * lg);
          f32x4 dp = *reinterpret_cast<const f32x4*>(nD + qt * 16 + 4 * lg);
#pragma unroll
          for (int ks = 0; ks < KSQ; ++ks) {
            const int off = swz<RBv>(qt * 16 + li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T));
            sc = M_::step(lds_frag<T>(X0, off), kf[t][ks], sc);
            dp = M_::step(lds_frag<T>(X1, off), vf[t][ks], dp);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sc[r] * sl2 + kbv[t]);
            pv4[cc][r] = pv;
            dsv[cc][r] = pv * dp[r];
          }
        }
        const Frag pf = M_::from_acc(pv4[0], pv4[CT - 1]);
        const Frag dsf = M_::from_acc(dsv[0], dsv[CT - 1]);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
          dva[t][dt] = M_::step(TrFrag<T, RBv>::load(X1, u * M_::KS, dt * 16, lane), pf, dva[t][dt]);
          dka[t][dt] = M_::step(TrFrag<T, RBv>::load(X0, u * M_::KS, dt * 16, lane), dsf, dka[t][dt]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int key = (kb_ * LQB + wave + t * ANW) * 16 + li;
    if (key >= L) continue;
    const size_t krow = base + (size_t)key * a.tok_stride;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      dka[t][dt] *= a.scale;
      store4(dqkv + krow * a.ld + a.d + h * HD + dt * 16 + 4 * lg, dka[t][dt]);
      store4(dqkv + krow * a.ld + 2 * a.d + h * HD + dt * 16 + 4 * lg, dva[t][dt]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Small-sequence path (L <= 32, any head_dim <= 128 that is a multiple of 8): the video tower's temporal
// attention (L = T = 8, B*197 sequences x 12 heads) and the tiny parity configs.  No MFMA: one wavefront
// packs 64 / L (sequence, head) pairs, lane = (pair, query); operands sit in LDS as T, math in fp32.
// HBM-bound by construction (3 KB in, 1 KB out per pair at L = 8).
// ---------------------------------------------------------------------------------------------------
// LDS rows of the small path are padded by 16 bytes: a lane reads ITS OWN row (stride = row pitch), and an unpadded
// 128-byte pitch would put all 64 lanes on the same banks.
template <typename T> __device__ __forceinline__ int small_pitch(int hd) { return hd + 16 / (int)sizeof(T); }

template <typename T>
__device__ __forceinline__ void stage_small(T* dst, const T* src, const AttnArgs& a, int pair0, int npairs_total, int PW, int hd,
                                            int ld, int col_base, int lane) {
  constexpr int EPC = 16 / sizeof(T);
  const int NC = hd / EPC, hp = small_pitch<T>(hd);
  const int L = a.L;
  for (int idx = lane; idx < PW * L * NC; idx += 64) {
    const int rowid = idx / NC, c = idx % NC;
    const int pl = rowid / L, j = rowid % L;
    const int pair = pair0 + pl;
    u32x4 v = {0, 0, 0, 0};
    if (pair < npairs_total) {
      const int seq = pair / a.H, h = pair % a.H;
      const size_t row = seq_base(a, seq) + (size_t)j * a.tok_stride;
      v = *reinterpret_cast<const u32x4*>(src + row * ld + col_base + h * hd + c * EPC);
    }
    *reinterpret_cast<u32x4*>(dst + (size_t)rowid * hp + c * EPC) = v;
  }
}

template <typename T, int MAXL>
__global__ __launch_bounds__(64) void attn_small_fwd_kernel(AttnArgs a, int hd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = a.L, PW = 64 / L, lane = threadIdx.x;
  const int total = a.nseq * a.H;
  const int pair0 = blockIdx.x * PW;
  const int hp = small_pitch<T>(hd);
  T* sQ = reinterpret_cast<T*>(smem);
  T* sK = sQ + (size_t)PW * L * hp;
  T* sV = sK + (size_t)PW * L * hp;
  const T* qkv = static_cast<const T*>(a.qkv);
  stage_small<T>(sQ, qkv, a, pair0, total, PW, hd, a.ld, 0, lane);
  stage_small<T>(sK, qkv, a, pair0, total, PW, hd, a.ld, a.d, lane);
  stage_small<T>(sV, qkv, a, pair0, total, PW, hd, a.ld, 2 * a.d, lane);
  __syncthreads();
  const int pl = lane / L, q = lane % L, pair = pair0 + pl;
  if (pl >= PW || pair >= total) return;
  const int seq = pair / a.H, h = pair % a.H;
  const T* Qr = sQ + ((size_t)pl * L + q) * hp;
  const T* Kp = sK + (size_t)pl * L * hp;
  const T* Vp = sV + (size_t)pl * L * hp;
  float s[MAXL];
  float mx = kNegInf;
#pragma unroll
  for (int j = 0; j < MAXL; ++j) {
    s[j] = kNegInf;
    if (j < L) {
      float acc = 0.f;
      for (int d0 = 0; d0 < hd; d0 += 4) {
        const f32x4 qv = load4(Qr + d0), kv = load4(Kp + (size_t)j * hp + d0);
        acc += qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3];
      }
      bool ok = !(a.causal && j > q);
      if (a.key_mask && a.key_mask[(size_t)seq * L + j] == 0) ok = false;
      s[j] = ok ? acc * a.scale : kNegInf;
      mx = fmaxf(mx, s[j]);
    }
  }
  if (mx == kNegInf) mx = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAXL; ++j) { s[j] = (j < L) ? __expf(s[j] - mx) : 0.f; sum += s[j]; }
  const float inv = 1.0f / sum;
  if (a.lse) a.lse[((size_t)seq * a.H + h) * L + q] = mx + __logf(sum);
  T* out = static_cast<T*>(a.out) + (seq_base(a, seq) + (size_t)q * a.tok_stride) * a.ldo + h * hd;
  for (int d0 = 0; d0 < hd; d0 += 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MAXL; ++j)
      if (j < L) { const f32x4 vv = load4(Vp + (size_t)j * hp + d0); acc += vv * s[j]; }
    acc *= inv;
    store4(out + d0, acc);
  }
}

template <typename T, int MAXL>
__global__ __launch_bounds__(64) void attn_small_bwd_kernel(AttnArgs a, int hd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = a.L, PW = 64 / L, lane = threadIdx.x;
  const int total = a.nseq * a.H;
  const int pair0 = blockIdx.x * PW;
  const int hp = small_pitch<T>(hd);
  const size_t mat = (size_t)PW * L * hp;
  T* sQ = reinterpret_cast<T*>(smem);
  T* sK = sQ + mat;
  T* sV = sK + mat;
  T* sdO = sV + mat;
  float* sP = reinterpret_cast<float*>(sdO + mat);   // [PW][L][L]
  float* sdS = sP + (size_t)PW * L * L;
  const T* qkv = static_cast<const T*>(a.qkv);
  stage_small<T>(sQ, qkv, a, pair0, total, PW, hd, a.ld, 0, lane);
  stage_small<T>(sK, qkv, a, pair0, total, PW, hd, a.ld, a.d, lane);
  stage_small<T>(sV, qkv, a, pair0, total, PW, hd, a.ld, 2 * a.d, lane);
  stage_small<T>(sdO, static_cast<const T*>(a.dout), a, pair0, total, PW, hd, a.ldo, 0, lane);
  __syncthreads();
  const int pl = lane / L, q = lane % L, pair = pair0 + pl;
  const bool active = pl < PW && pair < total;
  const int seq = active ? pair / a.H : 0, h = active ? pair % a.H : 0;
  const T* Qp = sQ + (size_t)pl * L * hp;
  const T* Kp = sK + (size_t)pl * L * hp;
  const T* Vp = sV + (size_t)pl * L * hp;
  const T* dOp = sdO + (size_t)pl * L * hp;
  T* dqkv = static_cast<T*>(a.dqkv);
  if (active) {
    float p[MAXL], dp[MAXL];
    float mx = kNegInf;
#pragma unroll
    for (int j = 0; j < MAXL; ++j) {
      p[j] = kNegInf; dp[j] = 0.f;
      if (j < L) {
        float acc = 0.f, acc2 = 0.f;
        for (int d0 = 0; d0 < hd; d0 += 4) {
          const f32x4 qv = load4(Qp + (size_t)q * hp + d0), kv = load4(Kp + (size_t)j * hp + d0);
          const f32x4 gv = load4(dOp + (size_t)q * hp + d0), vv = load4(Vp + (size_t)j * hp + d0);
          acc += qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3];
          acc2 += gv[0] * vv[0] + gv[1] * vv[1] + gv[2] * vv[2] + gv[3] * vv[3];
        }
        bool ok = !(a.causal && j > q);
        if (a.key_mask && a.key_mask[(size_t)seq * L + j] == 0) ok = false;
        p[j] = ok ? acc * a.scale : kNegInf;
        dp[j] = acc2;
        mx = fmaxf(mx, p[j]);
      }
    }
    if (mx == kNegInf) mx = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXL; ++j) { p[j] = (j < L) ? __expf(p[j] - mx) : 0.f; sum += p[j]; }
    const float inv = 1.0f / sum;
    float D = 0.f;
#pragma unroll
    for (int j = 0; j < MAXL; ++j) { p[j] *= inv; D += p[j] * dp[j]; }
#pragma unroll
    for (int j = 0; j < MAXL; ++j)
      if (j < L) {
        const float dsv = p[j] * (dp[j] - D) * a.scale;
        sP[((size_t)pl * L + q) * L + j] = p[j];
        sdS[((size_t)pl * L + q) * L + j] = dsv;
        dp[j] = dsv;
      }
    T* dq = dqkv + (seq_base(a, seq) + (size_t)q * a.tok_stride) * a.ld + h * hd;
    for (int d0 = 0; d0 < hd; d0 += 4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < MAXL; ++j)
        if (j < L) { const f32x4 kv = load4(Kp + (size_t)j * hp + d0); acc += kv * dp[j]; }
      store4(dq + d0, acc);
    }
  }
  __syncthreads();
  if (active) {  // lane = (pair, key q)
    T* dk = dqkv + (seq_base(a, seq) + (size_t)q * a.tok_stride) * a.ld + a.d + h * hd;
    T* dv = dk + a.d;
    for (int d0 = 0; d0 < hd; d0 += 4) {
      f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < L; ++i) {
        const float dsv = sdS[((size_t)pl * L + i) * L + q], pv = sP[((size_t)pl * L + i) * L + q];
        const f32x4 qv = load4(Qp + (size_t)i * hp + d0), gv = load4(dOp + (size_t)i * hp + d0);
        ak += qv * dsv;
        av += gv * pv;
      }
      store4(dk + d0, ak);
      store4(dv + d0, av);
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Time attention of the video tower (L = T <= 8 tokens per sequence, head_dim 64, not causal): [r3] a dedicated pair of kernels.
// The per-wave instantiation of the kernels above (NTP = 2) spent a 32-row K and a 32-row V tile on 8 real rows (six of every eight
// LDS-DMA pieces fetched zeros), held one (sequence, head) unit per wave and - in the backward - made two dependent memory round trips
// (K / V for the dQ pass, then Q / dO for the dK / dV pass): 3.8 / 3.2 TB/s, bound by the latency of short-lived waves.
// Here a wave owns a PAIR of units: rows 0-7 of every 16-row tile belong to unit 2p, rows 8-15 to unit 2p + 1, and the 16 x 16 score
// tile is block-diagonal (a key and a query of different units never meet: -inf).  Every operand tile holds real rows only (one 1-KiB
// LDS-DMA piece per unit and matrix in bf16), the backward stages K, V, Q and dO up front (ONE round trip), and the second half of the
// 32-deep bf16 MFMA step - keys 16-31, which do not exist - is a literal zero operand instead of zero rows in LDS.  Twice the units
// in flight per CU at a fifth of the LDS.
// ---------------------------------------------------------------------------------------------------
template <typename T, int RB> struct TrHalf;     // C-as-operand A fragment over rows 0-15 only (the step's rows 16-31 are zeros)
template <int RB> struct TrHalf<bf16, RB> {
  __device__ static __forceinline__ bf16x8 load(const char* lds, int c0, int lane) {
    const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
    const char* a0 = lds + swz<RB>(4 * g + q, (c0 + 4 * p) * 2);
    i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(a0));
    using i16x8 = __attribute__((ext_vector_type(8))) short;
    i16x8 v = {lo[0], lo[1], lo[2], lo[3], 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, v);
  }
};
template <int RB> struct TrHalf<float, RB> {
  __device__ static __forceinline__ f32x4 load(const char* lds, int c0, int lane) { return TrFrag<float, RB>::load(lds, 0, c0, lane); }
};
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag acc_lo(f32x4 t) { return Mma<T>::from_acc(t, f32x4{0.f, 0.f, 0.f, 0.f}); }

struct TimeUnit { size_t base; int seq, h; bool ok; };
__device__ __forceinline__ TimeUnit time_unit(const AttnArgs& a, int unit) {
  TimeUnit u;
  u.ok = unit < a.nseq * a.H;
  u.seq = u.ok ? unit / a.H : 0; u.h = u.ok ? unit % a.H : 0;
  u.base = seq_base(a, u.seq);
  return u;
}

template <typename T>
__global__ __launch_bounds__(ATHREADS, sizeof(T) == 2 ? 8 : 4) void attn_time_fwd_kernel(AttnArgs a) {
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T), KSQ = HD / M_::KS;
  constexpr int SLICE = 2 * 16 * RBv;
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  const int pair = blockIdx.x * ANW + wave;
  if (2 * pair >= a.nseq * a.H) return;       // (wave-uniform: the transposed LDS reads below need every lane)
  char* ldsK = smem_all + wave * SLICE;
  char* ldsV = ldsK + 16 * RBv;
  const T* qkv = static_cast<const T*>(a.qkv);
  const int L = a.L;
  const TimeUnit u0 = time_unit(a, 2 * pair), u1 = time_unit(a, 2 * pair + 1);
  stage_head<T, RBv, 1>(ldsK, qkv, u0.base, a.tok_stride, a.ld, a.d + u0.h * HD, L, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsK + 8 * RBv, qkv, u1.base, a.tok_stride, a.ld, a.d + u1.h * HD, u1.ok ? L : 0, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsV, qkv, u0.base, a.tok_stride, a.ld, 2 * a.d + u0.h * HD, L, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsV + 8 * RBv, qkv, u1.base, a.tok_stride, a.ld, 2 * a.d + u1.h * HD, u1.ok ? L : 0, 8, lane, 0);
  // this lane's query: unit li >> 3 of the pair, token li & 7
  const bool second = li >= 8;
  const TimeUnit& uq = second ? u1 : u0;
  const int tq = li & 7;
  const bool qvalid = uq.ok && tq < L;
  const size_t qrow = uq.base + (size_t)(qvalid ? tq : 0) * a.tok_stride;
  Frag qf[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) {
    qf[ks] = M_::zero();
    if (qvalid) qf[ks] = *reinterpret_cast<const Frag*>(qkv + qrow * a.ld + uq.h * HD + ks * M_::KS + lg * M_::KPL);
  }
  // validity of the four keys 4 lg + r this lane meets: same unit as the query, a real token, not masked
  bool kok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tk = 4 * (lg & 1) + r;
    kok[r] = qvalid && (lg >> 1) == (li >> 3) && tk < L && (!a.key_mask || a.key_mask[(size_t)uq.seq * L + tk] != 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float sl2 = a.scale * 1.4426950408889634f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) acc = M_::step(lds_frag<T>(ldsK, swz<RBv>(li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T))), qf[ks], acc);
  float mx = kNegInf;
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r] = kok[r] ? acc[r] : kNegInf; mx = fmaxf(mx, acc[r]); }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (mx == kNegInf) mx = 0.f;
  const float mxs = mx * sl2;
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r] = __builtin_amdgcn_exp2f(acc[r] * sl2 - mxs); sum += acc[r]; }
  const Frag pf = acc_lo<T>(acc);
  // (the row sum is taken from the ROUNDED probabilities that multiply V, like the ones-row of the big kernel)
  if constexpr (sizeof(T) == 2) { sum = 0.f; _Pragma("unroll") for (int r = 0; r < 4; ++r) sum += (float)pf[r]; }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (a.lse && qvalid && lg == 0) a.lse[((size_t)uq.seq * a.H + uq.h) * L + tq] = (mxs + __log2f(sum)) * 0.6931471805599453f;
  T* out = static_cast<T*>(a.out);
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    f32x4 o = M_::step(TrHalf<T, RBv>::load(ldsV, dt * 16, lane), pf, f32x4{0.f, 0.f, 0.f, 0.f});
    o *= inv;
    if (qvalid) store4(out + qrow * a.ldo + uq.h * HD + dt * 16 + 4 * lg, o);
  }
}

template <typename T>
__global__ __launch_bounds__(ATHREADS, sizeof(T) == 2 ? 4 : 2) void attn_time_bwd_kernel(AttnArgs a) {
  using M_ = Mma<T>;
  using Frag = typename M_::Frag;
  constexpr int RBv = HD * sizeof(T), KSQ = HD / M_::KS;
  constexpr int SLICE = 4 * 16 * RBv + 2 * 16 * 4;
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  const int pair = blockIdx.x * ANW + wave;
  if (2 * pair >= a.nseq * a.H) return;
  char* ldsK = smem_all + wave * SLICE;
  char* ldsV = ldsK + 16 * RBv;
  char* ldsQ = ldsV + 16 * RBv;
  char* ldsG = ldsQ + 16 * RBv;               // dO
  float* lseL = reinterpret_cast<float*>(ldsG + 16 * RBv);
  float* Dl = lseL + 16;
  const T* qkv = static_cast<const T*>(a.qkv);
  const T* dout = static_cast<const T*>(a.dout);
  const T* fout = static_cast<const T*>(a.out);
  T* dqkv = static_cast<T*>(a.dqkv);
  const int L = a.L;
  const TimeUnit u0 = time_unit(a, 2 * pair), u1 = time_unit(a, 2 * pair + 1);
  const int L1 = u1.ok ? L : 0;
  stage_head<T, RBv, 1>(ldsK, qkv, u0.base, a.tok_stride, a.ld, a.d + u0.h * HD, L, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsK + 8 * RBv, qkv, u1.base, a.tok_stride, a.ld, a.d + u1.h * HD, L1, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsV, qkv, u0.base, a.tok_stride, a.ld, 2 * a.d + u0.h * HD, L, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsV + 8 * RBv, qkv, u1.base, a.tok_stride, a.ld, 2 * a.d + u1.h * HD, L1, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsQ, qkv, u0.base, a.tok_stride, a.ld, u0.h * HD, L, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsQ + 8 * RBv, qkv, u1.base, a.tok_stride, a.ld, u1.h * HD, L1, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsG, dout, u0.base, a.tok_stride, a.ldo, u0.h * HD, L, 8, lane, 0);
  stage_head<T, RBv, 1>(ldsG + 8 * RBv, dout, u1.base, a.tok_stride, a.ldo, u1.h * HD, L1, 8, lane, 0);
  // this lane's row li of the pair tile (a query in pass A, a key in pass B): unit li >> 3, token li & 7
  const bool second = li >= 8;
  const TimeUnit& um = second ? u1 : u0;
  const int tm = li & 7;
  const bool mvalid = um.ok && tm < L;
  const size_t mrow = um.base + (size_t)(mvalid ? tm : 0) * a.tok_stride;
  // D = rowsum(dO . O) of this lane's query from the saved forward output (the same number as sum_j P dP), lse / scale alongside
  float dsum = 0.f;
  if (mvalid) {
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks)
      dsum = frag_dot(*reinterpret_cast<const Frag*>(dout + mrow * a.ldo + um.h * HD + ks * M_::KS + lg * M_::KPL),
                      *reinterpret_cast<const Frag*>(fout + mrow * a.ldo + um.h * HD + ks * M_::KS + lg * M_::KPL), dsum);
  }
  dsum += __shfl_xor(dsum, 16, 64);
  dsum += __shfl_xor(dsum, 32, 64);
  const float inv_scale = 1.0f / a.scale;
  const float lq = mvalid ? a.lse[((size_t)um.seq * a.H + um.h) * L + tm] * inv_scale : __builtin_huge_valf();
  if (lg == 0) { lseL[li] = lq; Dl[li] = dsum; }
  const bool kmask_ok = mvalid && (!a.key_mask || a.key_mask[(size_t)um.seq * L + tm] != 0);   // pass B: this lane's key
  bool kokA[4];                                 // pass A: the four keys 4 lg + r against this lane's query
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tk = 4 * (lg & 1) + r;
    kokA[r] = mvalid && (lg >> 1) == (li >> 3) && tk < L && (!a.key_mask || a.key_mask[(size_t)um.seq * L + tk] != 0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const float sl2 = a.scale * 1.4426950408889634f;
  int off[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) off[ks] = swz<RBv>(li, (ks * M_::KS + lg * M_::KPL) * (int)sizeof(T));

  // ---------------- pass A: query on the lane -> dQ ----------------
  {
    f32x4 sc = {-lq, -lq, -lq, -lq}, dp = {-dsum, -dsum, -dsum, -dsum};
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      sc = M_::step(lds_frag<T>(ldsK, off[ks]), lds_frag<T>(ldsQ, off[ks]), sc);
      dp = M_::step(lds_frag<T>(ldsV, off[ks]), lds_frag<T>(ldsG, off[ks]), dp);
    }
    f32x4 ds;
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[r] = kokA[r] ? __builtin_amdgcn_exp2f(sc[r] * sl2) * dp[r] : 0.f;
    const Frag dsf = acc_lo<T>(ds);
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      f32x4 dq = M_::step(TrHalf<T, RBv>::load(ldsK, dt * 16, lane), dsf, f32x4{0.f, 0.f, 0.f, 0.f});
      dq *= a.scale;
      if (mvalid) store4(dqkv + mrow * a.ld + um.h * HD + dt * 16 + 4 * lg, dq);
    }
  }
  // ---------------- pass B: key on the lane -> dK, dV ----------------
  {
    f32x4 sc = -*reinterpret_cast<const f32x4*>(lseL + 4 * lg);
    f32x4 dp = -*reinterpret_cast<const f32x4*>(Dl + 4 * lg);
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      sc = M_::step(lds_frag<T>(ldsQ, off[ks]), lds_frag<T>(ldsK, off[ks]), sc);
      dp = M_::step(lds_frag<T>(ldsG, off[ks]), lds_frag<T>(ldsV, off[ks]), dp);
    }
    f32x4 pv, ds;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // query 4 lg + r of the tile: same unit as this lane's key (lse = +inf already zeroes queries that do not exist)
      const bool ok = kmask_ok && (lg >> 1) == (li >> 3);
      pv[r] = ok ? __builtin_amdgcn_exp2f(sc[r] * sl2) : 0.f;
      ds[r] = pv[r] * dp[r];
    }
    const Frag pf = acc_lo<T>(pv), dsf = acc_lo<T>(ds);
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      f32x4 dv = M_::step(TrHalf<T, RBv>::load(ldsG, dt * 16, lane), pf, f32x4{0.f, 0.f, 0.f, 0.f});
      f32x4 dk = M_::step(TrHalf<T, RBv>::load(ldsQ, dt * 16, lane), dsf, f32x4{0.f, 0.f, 0.f, 0.f});
      dk *= a.scale;
      if (mvalid) {
        store4(dqkv + mrow * a.ld + a.d + um.h * HD + dt * 16 + 4 * lg, dk);
        store4(dqkv + mrow * a.ld + 2 * a.d + um.h * HD + dt * 16 + 4 * lg, dv);
      }
    }
  }
}

}  // namespace missm

using namespace missm;

namespace {
template <typename K> int launch_dyn(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t s, const char* what) {
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) { missm_set_error("%s: cannot raise dynamic LDS to %zu: %s", what, shmem, hipGetErrorString(e)); return MISSM_ERR_LAUNCH; }
  }
  return MISSM_OK;
}

int fill_args(AttnArgs& a, const void* qkv, void* out, float* lse, const void* dout, void* dqkv, int nseq, int L, int H, int hd,
              int ld, int ldo, int seq_div, int seq_outer, int seq_inner, int tok_stride, int causal, const int* key_mask,
              float scale) {
  a.qkv = qkv; a.out = out; a.lse = lse; a.dout = dout; a.dqkv = dqkv; a.nseq = nseq; a.L = L; a.H = H; a.d = H * hd;
  a.ld = ld; a.ldo = ldo; a.seq_div = seq_div > 0 ? seq_div : 1; a.seq_outer = seq_outer; a.seq_inner = seq_inner;
  a.tok_stride = tok_stride; a.causal = causal; a.key_mask = key_mask; a.scale = scale;
  return MISSM_OK;
}

template <typename T, bool BWD> int launch_attn(const AttnArgs& a, int hd, hipStream_t s) {
  const int L = a.L;
  static const int use_time = getenv("MISSM_ATTN_TIME") ? atoi(getenv("MISSM_ATTN_TIME")) : 1;
  if (use_time && L <= 8 && hd == HD && !a.causal) {   // the video tower's time attention: two (sequence, head) units per wave
    const int pairs = (a.nseq * a.H + 1) / 2;
    dim3 grid((pairs + ANW - 1) / ANW), block(ATHREADS);
    const size_t shmem = (size_t)ANW * (BWD ? 4 * 16 * HD * sizeof(T) + 2 * 16 * 4 : 2 * 16 * HD * sizeof(T));
    if constexpr (BWD) {
      auto k = attn_time_bwd_kernel<T>;
      int rc = launch_dyn(k, grid, block, shmem, s, "attn_time_bwd"); if (rc) return rc;
      hipLaunchKernelGGL(k, grid, block, shmem, s, a);
    } else {
      auto k = attn_time_fwd_kernel<T>;
      int rc = launch_dyn(k, grid, block, shmem, s, "attn_time_fwd"); if (rc) return rc;
      hipLaunchKernelGGL(k, grid, block, shmem, s, a);
    }
    return missm_check_launch("attn_time");
  }
  if (L <= 32 && hd == HD) {   // time attention / short text: one wave per (sequence, head) on the MFMA path
    constexpr int NTP = 2;
    const int units = a.nseq * a.H;
    dim3 grid((units + ANW - 1) / ANW), block(ATHREADS);
    size_t shmem = (size_t)ANW * (2 * NTP * 16 * HD * sizeof(T) + (size_t)NTP * 16 * 4 * (BWD ? 3 : 1));
    auto k = a.causal ? (BWD ? attn_bwd_mfma_kernel<T, NTP, true, true> : attn_fwd_mfma_kernel<T, NTP, true, true>)
                      : (BWD ? attn_bwd_mfma_kernel<T, NTP, true, false> : attn_fwd_mfma_kernel<T, NTP, true, false>);
    int rc = launch_dyn(k, grid, block, shmem, s, "attn_wave"); if (rc) return rc;
    hipLaunchKernelGGL(k, grid, block, shmem, s, a);
    return missm_check_launch("attn_wave");
  }
  if (L <= 32) {
    const int PW = 64 / L;
    const int total = a.nseq * a.H;
    dim3 grid((total + PW - 1) / PW), block(64);
    size_t shmem = (size_t)PW * L * (hd + 16 / sizeof(T)) * sizeof(T) * (BWD ? 4 : 3) + (BWD ? (size_t)PW * L * L * 8 : 0);
#define MISSM_SMALL(MAXL)                                                                                   \
    do {                                                                                                    \
      auto k = BWD ? attn_small_bwd_kernel<T, MAXL> : attn_small_fwd_kernel<T, MAXL>;                         \
      int rc = launch_dyn(k, grid, block, shmem, s, "attn_small"); if (rc) return rc;                        \
      hipLaunchKernelGGL(k, grid, block, shmem, s, a, hd);                                                   \
    } while (0)
    if (L <= 8) MISSM_SMALL(8); else if (L <= 16) MISSM_SMALL(16); else MISSM_SMALL(32);
#undef MISSM_SMALL
    return missm_check_launch("attn_small");
  }
  if (hd != HD) { missm_set_error("attention: L=%d head_dim=%d unsupported (need L<=32, or head_dim 64)", L, hd); return MISSM_ERR_INVALID; }
  // (257 .. 288 tokens with bf16 operands - ViT-L/14 at 224 x 224 is 257 - still fit the LDS-resident kernels: NTP = 18, 72 KiB,
  //  two workgroups per CU; MISSM_ATTN_NTP18=0 sends them to the key-chunked kernels)
  static const int ntp18 = getenv("MISSM_ATTN_NTP18") ? atoi(getenv("MISSM_ATTN_NTP18")) : 1;
  const bool lds18 = ntp18 && sizeof(T) == 2 && L > 256 && L <= 288 && !a.causal;
  if (L > 256 && !lds18) {       // key-chunked kernels (the 593-token spectrogram grid of the released audio checkpoint)
    if (a.causal) { missm_set_error("attention: causal attention over more than 256 tokens is not instantiated"); return MISSM_ERR_INVALID; }
    constexpr int LQBh = long_tpw<T>() * ANW;
    const int nqb = (((L + 15) / 16) + LQBh - 1) / LQBh;
    const dim3 grid_l(a.nseq * a.H * nqb), block_l(ATHREADS);
    if constexpr (BWD) {
      constexpr int LCKh = long_ck<T>();
      const size_t sh_dq = (size_t)2 * LCKh * 16 * HD * sizeof(T) + (size_t)LCKh * 16 * 4, sh_kv = (size_t)2 * LCKh * 16 * HD * sizeof(T) + (size_t)2 * LCKh * 16 * 4;
      auto kq = attn_bwd_long_dq_kernel<T>;
      auto kk = attn_bwd_long_dkv_kernel<T>;
      int rc = launch_dyn(kq, grid_l, block_l, sh_dq, s, "attn_bwd_long_dq"); if (rc) return rc;
      rc = launch_dyn(kk, grid_l, block_l, sh_kv, s, "attn_bwd_long_dkv"); if (rc) return rc;
      hipLaunchKernelGGL(kq, grid_l, block_l, sh_dq, s, a);
      hipLaunchKernelGGL(kk, grid_l, block_l, sh_kv, s, a);
    } else {
      constexpr int LCKh = long_ck<T>();
      const size_t sh = (size_t)2 * LCKh * 16 * HD * sizeof(T) + (size_t)LCKh * 16 * 4;
      auto k = attn_fwd_long_kernel<T>;
      int rc = launch_dyn(k, grid_l, block_l, sh, s, "attn_fwd_long"); if (rc) return rc;
      hipLaunchKernelGGL(k, grid_l, block_l, sh, s, a);
    }
    return missm_check_launch("attn_long");
  }
  dim3 grid(a.nseq * a.H);
  if constexpr (BWD && sizeof(T) == 2) {
    // single-pass backward (attn_bwd_sp_kernel), opt-in: measured slower than the two-pass kernel (see its header)
    static const int use_sp = getenv("MISSM_ATTN_SP") ? atoi(getenv("MISSM_ATTN_SP")) : 0;
    if (use_sp && !a.causal && L > 96 && L <= 224) {
      constexpr int NTP = 14;
      const size_t shmem = (size_t)3 * NTP * 16 * HD * 2 + 2 * 32 * DSP + (size_t)3 * NTP * 16 * 4;
      auto k = attn_bwd_sp_kernel<NTP>;
      int rc = launch_dyn(k, grid, dim3(SPW * 64), shmem, s, "attn_bwd_sp"); if (rc) return rc;
      hipLaunchKernelGGL(k, grid, dim3(SPW * 64), shmem, s, a);
      return missm_check_launch("attn_bwd_sp");
    }
  }
  // forward specialisations: FULL when every key tile but the last two is full (L > 16 (NTP - 2): S = 197 / NTP 14, S = 77 / NTP 6),
  // MASK when a key_padding_mask (or causality, which always comes with one here) can hit any tile
  const bool mask = a.key_mask != nullptr || a.causal;
#define MISSM_MFMA(NTP)                                                                                     \
  do {                                                                                                      \
    size_t shmem = (size_t)2 * NTP * 16 * HD * sizeof(T) + (size_t)NTP * 16 * 4 * (BWD ? 3 : 1);              \
    const bool full = L > 16 * (NTP - 2);                                                                   \
    const dim3 block(attn_waves(NTP) * 64);                                                                 \
    if constexpr (BWD) {                                                                                    \
      /* (the straight-line FULL variants are bf16 only: the fp32 instantiation spills under them) */       \
      constexpr bool FB = sizeof(T) == 2;                                                                   \
      auto k = a.causal ? attn_bwd_mfma_kernel<T, NTP, false, true>                                          \
               : (FB && full && !mask) ? attn_bwd_mfma_kernel<T, NTP, false, false, false, FB>                \
               : (FB && full) ? attn_bwd_mfma_kernel<T, NTP, false, false, true, FB>                          \
                              : attn_bwd_mfma_kernel<T, NTP, false, false>;                                   \
      int rc = launch_dyn(k, grid, block, shmem, s, "attn_mfma"); if (rc) return rc;                          \
      hipLaunchKernelGGL(k, grid, block, shmem, s, a);                                                       \
    } else {                                                                                                \
      auto k = (full && !mask) ? attn_fwd_mfma_kernel<T, NTP, false, false, false, true>                       \
               : (full && a.causal) ? attn_fwd_mfma_kernel<T, NTP, false, true, true, true>                    \
               : a.causal ? attn_fwd_mfma_kernel<T, NTP, false, true, true, false>                             \
                          : attn_fwd_mfma_kernel<T, NTP, false, false, true, false>;                           \
      int rc = launch_dyn(k, grid, block, shmem, s, "attn_mfma"); if (rc) return rc;                          \
      hipLaunchKernelGGL(k, grid, block, shmem, s, a);                                                       \
    }                                                                                                       \
  } while (0)
  if (L <= 96) MISSM_MFMA(6); else if (L <= 224) MISSM_MFMA(14); else if (L <= 256) MISSM_MFMA(16);
  else { if constexpr (sizeof(T) == 2) MISSM_MFMA(18); }
#undef MISSM_MFMA
  return missm_check_launch("attn_mfma");
}
}  // namespace

extern "C" int missm_attention_fwd(const void* qkv, void* out, float* lse, int nseq, int L, int H, int head_dim, int ld, int ldo,
                                   int seq_div, int seq_outer, int seq_inner, int tok_stride, int causal, const int* key_mask,
                                   float scale, int dtype, void* stream) {
  MISSM_CHECK_ARG(nseq > 0 && L > 0 && H > 0 && head_dim > 0 && head_dim % 8 == 0 && head_dim <= 128, "attention_fwd: bad shape");
  MISSM_CHECK_ARG(ld % 8 == 0 && ldo % 8 == 0, "attention_fwd: ld/ldo must be multiples of 8");
  AttnArgs a; fill_args(a, qkv, out, lse, nullptr, nullptr, nseq, L, H, head_dim, ld, ldo, seq_div, seq_outer, seq_inner, tok_stride, causal, key_mask, scale);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == kBF16 ? launch_attn<bf16, false>(a, head_dim, s) : launch_attn<float, false>(a, head_dim, s);
}

extern "C" int missm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int nseq, int L,
                                   int H, int head_dim, int ld, int ldo, int seq_div, int seq_outer, int seq_inner, int tok_stride,
                                   int causal, const int* key_mask, float scale, int dtype, void* stream) {
  MISSM_CHECK_ARG(nseq > 0 && L > 0 && H > 0 && head_dim > 0 && head_dim % 8 == 0 && head_dim <= 128, "attention_bwd: bad shape");
  MISSM_CHECK_ARG(out != nullptr, "attention_bwd: the forward output is required (D = rowsum(dO . O))");
  MISSM_CHECK_ARG(ld % 8 == 0 && ldo % 8 == 0, "attention_bwd: ld/ldo must be multiples of 8");
  AttnArgs a; fill_args(a, qkv, const_cast<void*>(out), const_cast<float*>(lse), dout, dqkv, nseq, L, H, head_dim, ld, ldo, seq_div, seq_outer, seq_inner, tok_stride, causal, key_mask, scale);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == kBF16 ? launch_attn<bf16, true>(a, head_dim, s) : launch_attn<float, true>(a, head_dim, s);
}
