// 256x256 NT GEMM tile as an 8-wave, 8-phase software pipeline (bf16 MFMA 16x16x32) - included by gemm.hip.
//
// Why a second tile kernel: the 16-wave 256x256 kernel in gemm.hip drains its LDS-DMA (s_waitcnt vmcnt(0)) and meets at a
// workgroup barrier once per K tile with every wave in the same phase (54-57 % MFMA-busy main loop).  Here
//   * a wave owns 128 x 64 outputs (128 accumulator VGPRs): half the LDS -> register bytes per flop of a 64 x 64 wave block;
//   * operands travel as eight 16 KiB HALF tiles per two K tiles (A rows 0-127 / 128-255 and two interleaved halves of the B
//     rows, K tile 64): one half tile is requested per phase, SEVEN half tiles ahead of its use, straight into LDS
//     (buffer_load ... lds), and is waited for with a COUNTED s_waitcnt vmcnt(6) once per K tile - never 0 inside the loop;
//   * a phase = {fragment reads of the half tile that became new, one half-tile request} | barrier | 16 MFMAs (one 64 x 32
//     quadrant x K 64) | barrier; waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave's MFMAs cover the
//     other's LDS reads and DMA issue (/opt/skills/guides/cdna_hip_programming.md section 5, "8-phase template").
// Hazards (derivation in DESIGN.md section 4):
//   RAW  a half tile is read at the earliest one phase after the vmcnt wait that retires it (the wait sits before the
//        phase's first barrier, every wave passes it before any wave of either group reads);
//   WAR  half tile s + 8 reuses the slot of s and is requested in phase s + 1: two or more phases after the last read of s for
//        three of the four half tiles; for the first one of a K tile (B half 0, read in the phase before) its four reads are
//        retired by a counted lgkmcnt BEFORE the reading phase's first barrier.
// Fragment reads are inline assembly: hipcc otherwise guards every LDS read behind an in-flight LDS-DMA with vmcnt(0).
#pragma once

namespace missm {

template <int OFF> __device__ __forceinline__ bf16x8 lds_read128(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

struct Gemm8pSrc {
  __amdgpu_buffer_rsrc_t a[2], b[2], none;   // A rows of half 0 / 1, B rows of half 0 / 1, zero-record descriptor (drops the load)
  unsigned va[2], vb[2];                     // per-lane byte offsets of this wave's two 1 KiB pieces inside a half tile
};

// one half tile = 16 pieces of 1 KiB; wave w moves pieces 2w, 2w + 1 (LDS rows 8 * piece .. + 7, 128 bytes of k each)
template <int SLOT, int H>
__device__ __forceinline__ void stage_half(char* lds, const Gemm8pSrc& s, int wave, int kbyte, bool live) {
  using lptr = __attribute__((address_space(3))) void*;
  constexpr bool IS_A = (H == 1 || H == 3);
  constexpr int HALF = H >> 1;
  const __amdgpu_buffer_rsrc_t r = live ? (IS_A ? s.a[HALF] : s.b[HALF]) : s.none;
  char* dst = lds + SLOT * 16384 + wave * 2048;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)dst, 16, IS_A ? s.va[0] : s.vb[0], kbyte, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)(dst + 1024), 16, IS_A ? s.va[1] : s.vb[1], kbyte, 0, 0);
}

// tile `bid` of the launch: its group's operand pointers (grouped launch), its origin and its buffer descriptors
__device__ __forceinline__ void gemm8p_tile(const GemmArgs& gall, int logical, GemmArgs& g, Gemm8pSrc& src, int& m0, int& n0) {
  int tm, tn;
  tile_of(logical, gall.tiles_m, gall.tiles_n, gall.group_m, tm, tn);
  if (gall.ngroups > 1) {                    // grouped launch: this tile row's group supplies the operands
    const int gi = tm / gall.group_tiles_m;
    tm -= gi * gall.group_tiles_m;
    select_group(g, gall, gi);
  }
  m0 = tm * 256; n0 = tn * 256;
  const bf16* A = static_cast<const bf16*>(g.A);
  const bf16* B = static_cast<const bf16*>(g.B);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ra = g.M - (m0 + 128 * h), rb = g.N - (n0 + 2 * h);
    const unsigned na = ra <= 0 ? 0u : (unsigned)min(ra, 128) * (unsigned)g.lda * 2u;
    const unsigned nb = rb <= 0 ? 0u : (unsigned)min(rb, 254) * (unsigned)g.ldb * 2u;
    src.a[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A + (size_t)(m0 + 128 * h) * g.lda), 0, na, 0x00020000);
    src.b[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B + (size_t)(n0 + 2 * h) * g.ldb), 0, nb, 0x00020000);
  }
  src.none = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, 0, 0x00020000);
}

// half tiles 0 .. 6 of a tile (K tile 0 and three quarters of K tile 1)
__device__ __forceinline__ void gemm8p_prologue(char* lds, const Gemm8pSrc& src, int wave, int nt) {
  stage_half<0, 0>(lds, src, wave, 0, true);
  stage_half<1, 1>(lds, src, wave, 0, true);
  stage_half<2, 2>(lds, src, wave, 0, true);
  stage_half<3, 3>(lds, src, wave, 0, true);
  stage_half<4, 0>(lds, src, wave, 128, nt > 1);
  stage_half<5, 1>(lds, src, wave, 128, nt > 1);
  stage_half<6, 2>(lds, src, wave, 128, nt > 1);
}


// ---------------------------------------------------------------------------------------------------------------------
// Branch-free vector epilogue of the 8-phase tile.  gemm_epilogue (gemm.hip) guards every row with `row < M`: sixteen exec
// branches per 64 x 64 block, which also pin every dependent load (saved pre-activation, fp32 residual) directly in front of
// its use - the backward-through-activation epilogue took as long as the main loop (17 us of a 34 us tile, in-kernel stamps).
// Here C / aux / residual are addressed through range-checked buffer descriptors that span exactly the wave's valid rows
// (rows past M: loads return 0, stores are dropped), so the block is straight-line code: all loads of a block are issued
// before the first use, stores follow back to back.  A lane owns rows 4 lg + r of each 16-row tile i and 4 consecutive columns.
// Row (ha, i, r) is a SCALAR offset (128 ha + 16 i + r) * row pitch; the lane offset is one VGPR per matrix.
// (Non-temporal stores, aux = nt, were measured: +6..8 % on the two activation epilogues alone, -2 % on QKV, 465.9 vs 467.4
//  samples/s for the whole step - the outputs are read back by the next kernel; kept at the default policy.)
// ---------------------------------------------------------------------------------------------------------------------
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
// Cache policy of the epilogue's C / aux stores (buffer aux bits: 0 default, 2 = nt, 16 = sc1 = write-through that does not keep the
// line in the XCD's L2).  Round 2 measured nt (worse on the whole step); round 3 measured sc1 against the default in alternating runs on
// one box: 66.93 / 66.98 vs 67.07 / 67.36 ms per step (+0.35 %) - the outputs are far larger than L2 and are next read by another
// kernel, keeping their lines in L2 only evicts operand panels.  (EXTRA=-DMISSM_EPI_AUX=<n> builds another policy for A/B runs.)
// [r3, last part] with the waterfall loops gone (section "Branch-free vector epilogue" above: the stores issue back to back) nt wins: 63.54 /
// 63.26 ms per step against 64.29 / 64.45 / 64.53 for sc1 and 64.41 / 64.57 for the default policy (alternating runs on one box);
// nt + sc1 (18) and nt on the weight-gradient kernel's parked slices (MISSM_PARK_NT) measured no better than nt alone.
#ifndef MISSM_EPI_AUX
#define MISSM_EPI_AUX 2
#endif
// cache policy of the epilogue's once-read operands (saved pre-activation, fp32 residual): nt, 63.38 / 63.68 / 63.69 vs 63.90 / 63.91 / 63.81 ms per step
#ifndef MISSM_EPI_LOAD_AUX
#define MISSM_EPI_LOAD_AUX 2
#endif

struct Epi8p {
  __amdgpu_buffer_rsrc_t c, aux, res;
  unsigned vc, vaux;           // per-lane byte offsets (C / residual share one, the bf16 aux matrix has its own pitch)
  unsigned pitch_c, pitch_aux; // row pitch in bytes
  unsigned wc, waux;           // per-lane byte offsets of the WIDE bf16 stores (below)
  bool odd;                    // lane & 1
};

// WIDE bf16 stores.  A lane owns rows 4 lg + r (r = 0..3) x 4 consecutive columns of a 16-row tile: four 8-byte stores, and the
// epilogue of a K = 768 tile was bound by their ISSUE, not by bandwidth (in-kernel stamps, round 3: 3.6 us for the 128 KiB of a QKV
// tile, 7.7 us for the 256 KiB of an fc1 tile = 34 GB/s per CU, unchanged when the CUs' epilogues were moved out of phase).  Lane
// pairs (li, li ^ 1) therefore trade halves through DPP: the even lane ends up with rows r = 0, 1 x 8 columns, the odd lane with
// rows r = 2, 3 x 8 columns - two 16-byte stores per lane instead of four 8-byte ones, every wave instruction still writing whole
// 128-byte lines (eight of them).
__device__ __forceinline__ unsigned dpp_swap1(unsigned x) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true); }
template <int AUXBITS>
__device__ __forceinline__ void store_wide_bf16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned pitch, unsigned soff, bool odd,
                                                const u32x2 (&P)[4]) {
  const u32x2 s0 = odd ? P[0] : P[2], s1 = odd ? P[1] : P[3];     // what the partner needs
  const u32x2 k0 = odd ? P[2] : P[0], k1 = odd ? P[3] : P[1];     // what stays
  const u32x2 r0 = {dpp_swap1(s0[0]), dpp_swap1(s0[1])}, r1 = {dpp_swap1(s1[0]), dpp_swap1(s1[1])};
  const u32x4 w0 = odd ? u32x4{r0[0], r0[1], k0[0], k0[1]} : u32x4{k0[0], k0[1], r0[0], r0[1]};
  const u32x4 w1 = odd ? u32x4{r1[0], r1[1], k1[0], k1[1]} : u32x4{k1[0], k1[1], r1[0], r1[1]};
  // soffset stays the literal 0 and the tile's scalar row offset goes into the lane offset: with a REGISTER in the soffset field hipcc
  // assumes that a store's data registers may be rewritten by the very next instruction (GCNHazardRecognizer: "this hazard only exists
  // if the instruction is not using a register in the soffset field") - on gfx950 the first data register of a 16-byte store was then
  // overwritten before the last lanes had been read (round 3, once the waterfall loops that used to separate store and overwrite were
  // gone: component 0 of rows r = 0..2, lanes 12-15 of every 16, in tiles here and there; found with an all-rows parity check).
  __builtin_amdgcn_raw_buffer_store_b128(w0, rsrc, voff + soff, 0, AUXBITS);
  __builtin_amdgcn_raw_buffer_store_b128(w1, rsrc, voff + pitch + soff, 0, AUXBITS);
}

__device__ __forceinline__ u32x2 pack_bf16x4(f32x4 v) {
  bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  return __builtin_bit_cast(u32x2, r);
}
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2 w) {
  const bf16x4 r = __builtin_bit_cast(bf16x4, w);
  return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
}

// MODE 0: bf16 = acc * alpha + bias            1: + QuickGELU, pre-activation saved to aux (if any)
//      2: bf16 = (acc * alpha + bias) * QuickGELU'(aux)             3: fp32 = acc * alpha + bias (+ residual)
template <int MODE>
__device__ __forceinline__ void gemm8p_epilogue(const Epi8p& e, bool has_aux, bool has_res, float alpha, f32x4 bias4,
                                                const f32x4 (&acc)[2][2][4][2]) {
#define MISSM_VAL(ha, i, r)                                                                                            \
  f32x4{acc[ha][0][i][0][r] * alpha + bias4[0], acc[ha][0][i][1][r] * alpha + bias4[1],                               \
        acc[ha][1][i][0][r] * alpha + bias4[2], acc[ha][1][i][1][r] * alpha + bias4[3]}
  // address = lane offset of row r (4 VGPRs per matrix) + SCALAR offset of the 16-row tile (ha, i): 8 scalars per matrix
  unsigned vcr[4], var[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { vcr[r] = e.vc + r * e.pitch_c; var[r] = e.vaux + r * e.pitch_aux; }
#define MISSM_SC(ha, i) ((128 * (ha) + 16 * (i)) * e.pitch_c)
#define MISSM_SA(ha, i) ((128 * (ha) + 16 * (i)) * e.pitch_aux)
  if constexpr (MODE == 2) {
    // pre-activation rows in chunks of two 16-row tiles (8 loads, 16 VGPRs), two chunks in flight: the next chunk is requested
    // before this one is used, so only the first request's latency is exposed
    u32x2 u[2][8];
    auto request = [&](int ch, u32x2 (&dst)[8]) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ha = ch >> 1, i = 2 * (ch & 1) + (k >> 2), r = k & 3;
        dst[k] = __builtin_amdgcn_raw_buffer_load_b64(e.aux, var[r], MISSM_SA(ha, i), MISSM_EPI_LOAD_AUX);
      }
    };
    request(0, u[0]);
    u32x2 pk[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (ch + 1 < 4) request(ch + 1, u[(ch + 1) & 1]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ha = ch >> 1, i = 2 * (ch & 1) + (k >> 2), r = k & 3;
        f32x4 v = MISSM_VAL(ha, i, r);
        const f32x4 uu = unpack_bf16x4(u[ch & 1][k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= quick_gelu_grad(uu[j]);
        pk[r] = pack_bf16x4(v);
        if (r == 3) store_wide_bf16<MISSM_EPI_AUX>(e.c, e.wc, e.pitch_c, MISSM_SC(ha, i), e.odd, pk);
      }
    }
  } else if constexpr (MODE == 3) {
    if (has_res) {
      // residual rows in chunks of one 16-row tile (4 loads, 16 VGPRs), two chunks in flight
      u32x4 q[2][4];
      auto request = [&](int ch, u32x4 (&dst)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r] = __builtin_amdgcn_raw_buffer_load_b128(e.res, vcr[r], MISSM_SC(ch >> 2, ch & 3), MISSM_EPI_LOAD_AUX);
      };
      request(0, q[0]);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        if (ch + 1 < 8) request(ch + 1, q[(ch + 1) & 1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x4 v = MISSM_VAL(ch >> 2, ch & 3, r);
          v += __builtin_bit_cast(f32x4, q[ch & 1][r]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), e.c, vcr[r] + MISSM_SC(ch >> 2, ch & 3), 0, MISSM_EPI_AUX);   // (soffset 0: see store_wide_bf16)
        }
      }
    } else {
#pragma unroll
      for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 v = MISSM_VAL(ha, i, r);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), e.c, vcr[r] + MISSM_SC(ha, i), 0, MISSM_EPI_AUX);
          }
    }
  } else {
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x2 pu[4], pa[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x4 v = MISSM_VAL(ha, i, r);
          if constexpr (MODE == 1) {
            pu[r] = pack_bf16x4(v);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = quick_gelu(v[j]);
          }
          pa[r] = pack_bf16x4(v);
        }
        if constexpr (MODE == 1) {
          if (has_aux) store_wide_bf16<MISSM_EPI_AUX>(e.aux, e.waux, e.pitch_aux, MISSM_SA(ha, i), e.odd, pu);
        }
        store_wide_bf16<MISSM_EPI_AUX>(e.c, e.wc, e.pitch_c, MISSM_SC(ha, i), e.odd, pa);
      }
  }
#undef MISSM_VAL
#undef MISSM_SC
#undef MISSM_SA
}

// The wave's 128 x 64 outputs (rows mwb + 128 ha + 16 i + 4 lg + r, columns nw + 4 li ..+3) through the branch-free epilogue;
// returns the number of vector-memory operations it issued (16 .. 64: the persistent grid counts them into the next tile's first
// vmcnt wait), or -1 if the case is not covered (the caller takes the guarded epilogue of gemm.hip): buffer offsets are 32-bit, so
// matrices of 4 GiB and more, ragged column blocks, accumulate and the rare activations stay there.
__device__ __forceinline__ int gemm8p_store_tile(const GemmArgs& g, const f32x4 (&acc)[2][2][4][2], int mwb, int nw, int lane,
                                                  f32x4 bias4) {
  const int li = lane & 15, lg = lane >> 4;
  const int esz = g.out_f32 ? 4 : 2;
  const int mode = g.out_f32 ? (g.act == MISSM_ACT_NONE ? 3 : -1)
                             : (g.act == MISSM_ACT_NONE ? 0 : (g.act == MISSM_ACT_QGELU ? 1 : (g.act == MISSM_ACT_DQGELU ? 2 : -1)));
  const void* auxp = mode == 1 ? g.aux_out : (mode == 2 ? g.aux_in : nullptr);
  const bool has_aux = auxp != nullptr;
  const bool fast = mode >= 0 && g.vec_ok && !g.accumulate && nw + 64 <= g.N && (mode != 2 || has_aux) && (mode == 3 || !g.resid) &&
                    (size_t)g.M * g.ldc * esz < (size_t(1) << 32) && (size_t)g.M * g.ldaux * 2 < (size_t(1) << 32);
  if (!fast) return -1;
  // Every descriptor input goes through readfirstlane: in the resident kernels the tile's pointers and origin are loop-carried, hipcc
  // then keeps them in VECTOR registers and wraps EVERY buffer operation of the epilogue in a waterfall loop (readfirstlane x 4, compare,
  // saveexec, the operation, loop: 195 of them in the 8-phase kernel since round 2 made its grid resident - found in the .s in round 3).
  auto uptr = [](const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<char*>(((unsigned long long)hi << 32) | lo);
  };
  mwb = __builtin_amdgcn_readfirstlane(mwb);
  Epi8p e;
  e.pitch_c = (unsigned)__builtin_amdgcn_readfirstlane(g.ldc * esz); e.pitch_aux = (unsigned)__builtin_amdgcn_readfirstlane(g.ldaux * 2);
  const unsigned rows = (unsigned)__builtin_amdgcn_readfirstlane(min(max(g.M - mwb, 0), 192));
  char* cb = uptr(g.C) + (size_t)mwb * e.pitch_c;
  e.c = __builtin_amdgcn_make_buffer_rsrc(cb, 0, rows * e.pitch_c, 0x00020000);
  e.res = g.resid ? __builtin_amdgcn_make_buffer_rsrc(uptr(g.resid) + (size_t)mwb * e.pitch_c, 0, rows * e.pitch_c, 0x00020000)
                  : __builtin_amdgcn_make_buffer_rsrc(cb, 0, 0, 0x00020000);
  e.aux = has_aux ? __builtin_amdgcn_make_buffer_rsrc(uptr(auxp) + (size_t)mwb * e.pitch_aux, 0, rows * e.pitch_aux, 0x00020000)
                  : __builtin_amdgcn_make_buffer_rsrc(cb, 0, 0, 0x00020000);
  e.vc = (unsigned)(lg * 4) * e.pitch_c + (unsigned)(nw + li * 4) * esz;
  e.vaux = (unsigned)(lg * 4) * e.pitch_aux + (unsigned)(nw + li * 4) * 2u;
  e.odd = (li & 1) != 0;
  e.wc = (unsigned)(lg * 4 + 2 * (li & 1)) * e.pitch_c + (unsigned)(nw + (li >> 1) * 8) * 2u;      // (bf16 C only)
  e.waux = (unsigned)(lg * 4 + 2 * (li & 1)) * e.pitch_aux + (unsigned)(nw + (li >> 1) * 8) * 2u;
  if (mode == 0) { gemm8p_epilogue<0>(e, false, false, g.alpha, bias4, acc); return 16; }
  if (mode == 1) { gemm8p_epilogue<1>(e, has_aux, false, g.alpha, bias4, acc); return has_aux ? 32 : 16; }
  if (mode == 2) { gemm8p_epilogue<2>(e, true, false, g.alpha, bias4, acc); return 48; }
  gemm8p_epilogue<3>(e, false, g.resid != nullptr, g.alpha, bias4, acc);
  return g.resid != nullptr ? 64 : 32;
}

// STAGGER: waves 4-7 run one barrier behind waves 0-3.
// The grid can be RESIDENT (one workgroup per CU walks tiles: statically b, b + gridDim.x, ... or drawn from the per-XCD queues below):
// once a tile's main loop has ended every LDS slot is free, so the next tile's first seven half tiles are requested BEFORE this tile's
// epilogue - their 1.5-2.4 us of latency (in-kernel stamps) and the workgroup relaunch hide under the stores.  vmcnt is in issue order:
// the epilogue's own loads / stores are younger than those requests and are COUNTED into the wait at the top of the next tile
// (vmcnt(6 + operations): the stores stay in flight across it).
template <bool STAGGER>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(GemmArgs gall) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 8 slots x 16 KiB; slot = (4 * (K tile & 1) + h), h: B0 A0 B1 A1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 15, lg = lane >> 4;
  const int ntiles = gall.tiles_m * gall.tiles_n;
  const int nt = gall.K >> 6;                // K tiles (host guarantees K % 128 == 0: whole pairs)

  // ---- global -> LDS addressing.  LDS row r of a half tile holds 128 bytes of k, chunk c stored at c ^ (r & 7).
  //  A half ha : LDS row r <-> C row   m0 + 128 ha + r                       (wave wr reads rows 64 wr + 16 i + li)
  //  B half hb : LDS row r <-> C column n0 + 64 (r >> 5) + 4 (r & 15) + 2 hb + ((r >> 4) & 1)
  //              (wave wc reads rows 32 wc + 16 j + li: with both halves a lane owns 4 CONSECUTIVE columns 64 wc + 4 li + 0..3)
  // Tile order.  Static: workgroup b walks the hardware-order ids b, b + gridDim.x, ... (logical id = xcd_remap(id): workgroups that
  // share b & 7 share an XCD under round-robin placement and get one contiguous range of logical tiles - L2 locality, speed only).
  // DYNAMIC (gall.sched, persistent grids only): that XCD range is a QUEUE - the first tile of workgroup b is entry b >> 3, every
  // further one is drawn with one returning atomic on the queue's counter at the TOP of the tile before (a microsecond of latency
  // under a 16 us main loop).  A slow tile (epilogue stores colliding: p90 8 us against a median of 3.7) then costs its CU one
  // tile less instead of making the launch wait for it.  Every workgroup ends on exactly one failing draw, so queue x sees
  // cnt_x draws in all: the draw that returns cnt_x - 1 is the last one and puts the counter back to zero for the next launch.
  const bool dyn = gall.sched != nullptr;    // scalar
  const int qx = blockIdx.x & 7;
  const int q_cnt = (ntiles >> 3) + (qx < (ntiles & 7) ? 1 : 0);             // tiles in this workgroup's queue
  const int q_start = xcd_remap(qx, ntiles);                                 // its first logical id
  const int q_wgs = ((int)gridDim.x >> 3) + (qx < ((int)gridDim.x & 7) ? 1 : 0);   // workgroups that serve it (each starts on a static entry)
  int bid = blockIdx.x;                      // hardware-order id (static order; also the diagnostic stamp slot)
  int logical = xcd_remap(bid, ntiles);
  if (dyn) bid = logical;
  GemmArgs g = gall;                         // (scalar fields only are ever read through this copy)
  Gemm8pSrc src;
  int m0, n0;
  gemm8p_tile(gall, logical, g, src, m0, n0);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 16 * wave + 8 * q + (lane >> 3);           // LDS row inside the half tile
    const int c = (lane & 7) ^ (r & 7);                      // the k chunk that lands at stored position lane & 7
    src.va[q] = (unsigned)r * (unsigned)gall.lda * 2u + (unsigned)c * 16u;
    const int col = 64 * (r >> 5) + 4 * (r & 15) + ((r >> 4) & 1);
    src.vb[q] = (unsigned)col * (unsigned)gall.ldb * 2u + (unsigned)c * 16u;
  }

  // ---- fragment read addresses (bytes): row * 128 + ((4 ks + lg) ^ (row & 7)) * 16; + slot and 16-row-tile immediates
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  unsigned aaddr[2][2], baddr[2][2];         // [K tile parity][k step]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned sw = (unsigned)(((ks * 4 + lg) ^ (li & 7)) << 4);
    aaddr[0][ks] = lds0 + (unsigned)(wr * 64 + li) * 128u + sw;
    baddr[0][ks] = lds0 + (unsigned)(wc * 32 + li) * 128u + sw;
    aaddr[1][ks] = aaddr[0][ks] + 65536u;
    baddr[1][ks] = baddr[0][ks] + 65536u;
  }
  // (measured and removed: starting the first round of workgroups (b / 8) % 8 x 2.5 us apart, so that the CUs' epilogues do
  //  not all hit HBM in the same phase: -1..-20 % on every video shape)
  gemm8p_prologue(lds, src, wave, nt);
  if (gall.desync_phases > 1) {               // experiment: start the workgroups of an XCD in `desync_phases` groups, desync_step ticks apart
    const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)(((int)blockIdx.x >> 3) % gall.desync_phases) * (unsigned long long)gall.desync_step;
    while (__builtin_amdgcn_s_memrealtime() - t_in < wait) __builtin_amdgcn_s_sleep(4);
  }
  int epi_ops = 0;                           // vector-memory operations of the previous tile's epilogue (0: first tile / guarded path)

  for (;;) {                                 // ---- one output tile per iteration
  unsigned long long t_start = 0, t_loop = 0, t_loop_end = 0;       // diagnostic runs only (tools/gemm_timeline.py)
  if (g.dbg) t_start = __builtin_amdgcn_s_memrealtime();
  // The draw is a BUFFER atomic that every lane executes with only thread 0 in range (a 4-byte descriptor: the others are dropped by
  // the range check and return 0): no branch, no phi - the compiler keeps the result pending and waits for it at its first use, behind
  // the main loop.  (The plain builtin atomic under `if (tid == 0)` is rewritten by hipcc's atomic optimizer into a wave-aggregated one
  // whose result it needs on the spot: s_waitcnt vmcnt(0) at the top of the tile, behind every store of the previous epilogue and this
  // tile's fourteen requests.  An inline-assembly atomic is no way out either: the compiler copies its output register while the
  // value is still in flight - tiles were drawn twice and skipped.)
  unsigned drawn = 0;
  if (dyn) {
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(gall.sched + qx, 0, 4, 0x00020000);
    drawn = (unsigned)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, qr, tid == 0 ? 0 : 64, 0, 0);
  }
  f32x4 bias4 = prefetch_bias(g, n0 + wc * 64, 0, lane);
  f32x4 acc[2][2][4][2];                     // [A half][B half][16-row tile][16-column tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // K tile 0 has landed (this wave's pieces).  Requests in issue order: [7 half tiles | the previous tile's epilogue (epi_ops loads /
  // stores) | bias (0-4 loads)]: K tile 0 is complete once at most the youngest 6 + epi_ops are pending - K tile 1's three half
  // tiles AND the epilogue's stores stay in flight (round 2 waited vmcnt(6) here, i.e. for every store of the epilogue to be
  // acknowledged: that wait was the persistent grid's whole loss against a relaunch).  The counter holds 6 bits: 63 is still a
  // lower bound on what is younger than K tile 0 when epi_ops = 64.
  if (epi_ops == 16) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
  else if (epi_ops == 32) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
  else if (epi_ops == 48) asm volatile("s_waitcnt vmcnt(54)" ::: "memory");
  else if (epi_ops == 64) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                // ... everybody's
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 fall one barrier behind (re-joined after the loop)
  if (g.dbg) t_loop = __builtin_amdgcn_s_memrealtime();

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];     // A half in use [i][ks]; B half 0 / 1 [j][ks]

#define MISSM_8P_MFMA(HA, HB, FB)                                                             \
  __builtin_amdgcn_s_setprio(1);                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
        acc[HA][HB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], FB[j][ks], acc[HA][HB][i][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);

#define MISSM_8P_READ_A(PAR, SLOT_H)                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
    fa[0][ks] = lds_read128<(SLOT_H) * 16384 + 0 * 2048>(aaddr[PAR][ks]);                    \
    fa[1][ks] = lds_read128<(SLOT_H) * 16384 + 1 * 2048>(aaddr[PAR][ks]);                    \
    fa[2][ks] = lds_read128<(SLOT_H) * 16384 + 2 * 2048>(aaddr[PAR][ks]);                    \
    fa[3][ks] = lds_read128<(SLOT_H) * 16384 + 3 * 2048>(aaddr[PAR][ks]);                    \
  }
#define MISSM_8P_READ_B(PAR, SLOT_H, FB)                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
    FB[0][ks] = lds_read128<(SLOT_H) * 16384 + 0 * 2048>(baddr[PAR][ks]);                    \
    FB[1][ks] = lds_read128<(SLOT_H) * 16384 + 1 * 2048>(baddr[PAR][ks]);                    \
  }
#define MISSM_8P_FENCE_ALL()                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                         \
               : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fa[0][1]), "+v"(fa[1][1]), "+v"(fa[2][1]), \
                 "+v"(fa[3][1]), "+v"(fb0[0][0]), "+v"(fb0[1][0]), "+v"(fb0[0][1]), "+v"(fb0[1][1]), "+v"(fb1[0][0]),            \
                 "+v"(fb1[1][0]), "+v"(fb1[0][1]), "+v"(fb1[1][1]));                          \
  __builtin_amdgcn_sched_barrier(0);

  // One K tile = 4 phases.  PAR = parity of the K tile (its slots are 4 PAR + h).  Phase p requests half tile 4 t + p + 7:
  //   p = 0 -> (t + 1, A1) slot 4 (1 - PAR) + 3 ; p = 1, 2, 3 -> (t + 2, B0 / A0 / B1) slots 4 PAR + 0 / 1 / 2
  auto ktile = [&](auto par_tag, int t) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int S = 4 * PAR;               // first slot of this K tile (in units of 16 KiB relative to the parity base)
    const int kb1 = (t + 1) * 128, kb2 = (t + 2) * 128;
    const bool live1 = t + 1 < nt, live2 = t + 2 < nt;
    // ---- phase 0: A half 0 x B half 0
    MISSM_8P_READ_B(PAR, 0, fb0)
    MISSM_8P_READ_A(PAR, 1)
    stage_half<4 * (1 - PAR) + 3, 3>(lds, src, wave, kb1, live1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");         // the four B-half-0 reads are back: its slot is re-filled next phase
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(0, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: A half 0 x B half 1
    MISSM_8P_READ_B(PAR, 2, fb1)
    stage_half<S + 0, 0>(lds, src, wave, kb2, live2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(0, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: A half 1 x B half 1
    MISSM_8P_READ_A(PAR, 3)
    stage_half<S + 1, 1>(lds, src, wave, kb2, live2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(1, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A half 1 x B half 0 (fragments kept since phase 0); the next K tile must have landed before its reads
    stage_half<S + 2, 2>(lds, src, wave, kb2, live2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_MFMA(1, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  // note on slot arithmetic: READ/stage offsets above are relative to the parity base (aaddr[PAR] carries + 64 KiB for PAR = 1),
  // stage_half takes ABSOLUTE slots.
  for (int t = 0; t < nt; t += 2) {
    ktile(std::integral_constant<int, 0>{}, t);
    ktile(std::integral_constant<int, 1>{}, t + 1);
  }
#undef MISSM_8P_MFMA
#undef MISSM_8P_READ_A
#undef MISSM_8P_READ_B
#undef MISSM_8P_FENCE_ALL
  // (the compiler waits for the bias vector at its first use; in the epilogue that wait was vmcnt(0) - behind the next tile's fourteen
  //  requests.  Using it here, where nothing but dropped requests is pending, leaves the epilogue without that wait.)
  asm volatile("" : "+v"(bias4));
  if (dyn) {                                 // the draw went out a whole main loop ago: hand it to the other waves through LDS
    using ldsw = volatile __attribute__((address_space(3))) unsigned*;
    if (tid == 0) *(ldsw)(lds + 131072) = drawn;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!STAGGER) __builtin_amdgcn_s_barrier();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();        // waves 0-3 wait for the lagging group
  if (g.dbg) t_loop_end = __builtin_amdgcn_s_memrealtime();
  // every fragment read of this tile is complete (phase 3 reads nothing and everybody is past its last barrier): all slots are
  // free.  Request the NEXT tile's first half tiles now; the (dropped) requests past the last K tile are older and harmless.
  int nbid = bid + (int)gridDim.x, nlogical;
  bool more;                                 // wave-uniform
  if (dyn) {
    using ldsw = volatile __attribute__((address_space(3))) unsigned*;
    const unsigned d = __builtin_amdgcn_readfirstlane(*(ldsw)(lds + 131072));
    const int k = q_wgs + (int)d;            // queue entry (entries 0 .. q_wgs - 1 were the static first tiles)
    more = k < q_cnt;
    nlogical = q_start + k;
    nbid = nlogical;                         // (stamp slot of the diagnostic runs: any unique id)
    if ((int)d == q_cnt - 1 && tid == 0) __hip_atomic_store(gall.sched + qx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    more = nbid < ntiles;
    nlogical = xcd_remap(nbid, ntiles);
  }
  // (the next tile's descriptors are KEPT across the epilogue - round 2 rebuilt them behind it to save scalar registers, which put
  //  ~300 dependent scalar instructions, three integer divisions among them, between two tiles: 0.9 us of a 21 us K = 768 tile)
  GemmArgs gn = gall;
  Gemm8pSrc srcn = src;
  int m0n = 0, n0n = 0;
  if (more) {
    gemm8p_tile(gall, nlogical, gn, srcn, m0n, n0n);
    gemm8p_prologue(lds, srcn, wave, nt);
  }

  // ---- epilogue: two 64 x 64 blocks per wave (A half 0 / 1), a lane owns rows 4 lg + r of each 16-row tile and the four
  // consecutive columns 64 wc + 4 li + {0, 1 (B half 0), 2, 3 (B half 1)}: the same register picture as gemm_kernel's.
  epi_ops = gemm8p_store_tile(g, acc, m0 + 64 * wr, n0 + 64 * wc, lane, bias4);
  if (epi_ops < 0) {
    // (guarded epilogue of gemm.hip, dependent loads issued at their use: matrices of 4 GiB and more, rare activations)
#pragma unroll
    for (int ha = 0; ha < 2; ++ha) {
      typename AuxPre<bf16>::V upre[4][4];
      f32x4 blk[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        blk[i][0] = acc[ha][0][i][0]; blk[i][1] = acc[ha][0][i][1];
        blk[i][2] = acc[ha][1][i][0]; blk[i][3] = acc[ha][1][i][1];
      }
      gemm_epilogue<bf16, false, false>(g, blk, m0 + 128 * ha + 64 * wr, n0 + 64 * wc, 0, lane, bias4, nullptr, upre, false);
    }
  }
  if (g.dbg && tid == 0) {
    const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = g.dbg + (size_t)bid * 8;
    d[0] = t_start; d[1] = t_loop; d[2] = t_loop_end; d[3] = t_issued;
    d[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); d[5] = t_issued; d[6] = t_loop_end;
    d[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15;    // XCC id
  }
  if (!more) break;
  bid = nbid; logical = nlogical;
#if defined(MISSM_8P_REBUILD_DESC)           // (A/B builds: round 2's rebuild behind the epilogue)
  g = gall;
  gemm8p_tile(gall, logical, g, src, m0, n0);
  continue;
#endif
  g.A = gn.A; g.B = gn.B; g.C = gn.C; g.bias = gn.bias; g.resid = gn.resid; g.aux_in = gn.aux_in; g.aux_out = gn.aux_out;
  src.a[0] = srcn.a[0]; src.a[1] = srcn.a[1]; src.b[0] = srcn.b[0]; src.b[1] = srcn.b[1];
  m0 = m0n; n0 = n0n;
  }                                          // ---- next tile
}

// =====================================================================================================================
// The same pipeline for the weight gradient  dW[M, N] = sum_k A[k, M]^T B[k, N]  (both operands K-MAJOR: the activations dY
// [rows, n_out] and X [rows, k_in] are read where they lie, the reduction index is the row).  A half tile is 64 k-rows x 128
// columns (256-byte rows, 16-byte chunk c of row r stored at c ^ (tkey(r) << 1) - gemm.hip's k-major image), fragments come
// through ds_read_b64_tr_b16 (two per fragment).  K is split over workgroups; every slice parks its 256 x 256 fp32 tile in the
// workspace in register order and splitk_reduce8p_kernel sums the slices in slice order (bit-reproducible).  The bias
// gradient (column sums of A) rides along: the 4 x tiles_n waves that see one A panel (wc = 0..3 of every tile column) take
// turns by K tile - wave (tn, wc) adds up its eight A fragments on the K tiles t = 4 tn + wc (mod 4 tiles_n) - a few per cent of
// VALU work on every wave instead of a third more on the workgroups of one tile column (measured: 169 -> 240 us that way).
// Host guarantees: M % 128 == 0, N % 128 == 0, k_per_split % 128 == 0 (K itself may be ragged).
// =====================================================================================================================
template <int OFF> __device__ __forceinline__ i16x4 lds_read_tr64(unsigned addr) {
  i16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
__device__ __forceinline__ bf16x8 join_tr(i16x4 lo, i16x4 hi) {
  using i16x8 = __attribute__((ext_vector_type(8))) short;
  i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int SLOT, int H>
__device__ __forceinline__ void stage_half_k(char* lds, const Gemm8pSrc& s, int wave, unsigned kbyte_a, unsigned kbyte_b, bool live) {
  using lptr = __attribute__((address_space(3))) void*;
  constexpr bool IS_A = (H == 1 || H == 3);
  constexpr int HALF = H >> 1;
  const __amdgpu_buffer_rsrc_t r = live ? (IS_A ? s.a[HALF] : s.b[HALF]) : s.none;
  char* dst = lds + SLOT * 16384 + wave * 2048;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)dst, 16, IS_A ? s.va[0] : s.vb[0], IS_A ? kbyte_a : kbyte_b, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)(dst + 1024), 16, IS_A ? s.va[1] : s.vb[1], IS_A ? kbyte_a : kbyte_b, 0, 0);
}

template <bool STAGGER>
__global__ __launch_bounds__(512, 2) void gemm8p_tn_kernel(GemmArgs gall) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 15, lg = lane >> 4;

  const int ntiles = gall.tiles_m * gall.tiles_n;
  const int per_group = ntiles * gall.splitk;
  int gi, split, tix, tm, tn;
  GemmArgs g = gall;
  if (gall.tn_order == 1) {                   // XCD-contiguous (group, K slice, tile row, tile column): see GemmArgs::tn_order
    const int logical = xcd_remap(blockIdx.x, per_group * gall.ngroups);
    gi = logical / per_group;
    const int rem = logical - gi * per_group;
    split = rem / ntiles; tix = rem - split * ntiles;
    tm = tix / gall.tiles_n; tn = tix - tm * gall.tiles_n;
    if (gall.ngroups > 1) select_group(g, gall, gi);
  } else {
    gi = blockIdx.x / per_group;
    const int bid = blockIdx.x - gi * per_group;   // grouped launch: group-major workgroup ids
    if (gall.ngroups > 1) select_group(g, gall, gi);
    split = bid / ntiles; tix = bid - split * ntiles;
    tile_of(xcd_remap(tix, ntiles), g.tiles_m, g.tiles_n, g.group_m, tm, tn);
  }
  const int m0 = tm * 256, n0 = tn * 256;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nt = ((kend - kbeg + 127) >> 7) << 1;   // whole pairs of K tiles; rows past K read as zeros (descriptor range)
  if (nt <= 0) return;                        // (uniform; cannot happen with the host's slicing)

  Gemm8pSrc src;
  {
    const bf16* A = static_cast<const bf16*>(g.A);
    const bf16* B = static_cast<const bf16*>(g.B);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // the range check sees voffset + soffset (measured: a 64-row window made every K tile but the first read zeros), so the
      // descriptor spans rows 0 .. K-1 of this half's 128 columns: rows >= K (ragged end of the reduction) come back as zeros
      const unsigned na = (m0 + 128 * h < g.M) ? (unsigned)(((size_t)(g.K - 1) * g.lda + 128) * 2) : 0u;
      const unsigned nb = (n0 + 128 * h < g.N) ? (unsigned)(((size_t)(g.K - 1) * g.ldb + 128) * 2) : 0u;
      src.a[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A + m0 + 128 * h), 0, na, 0x00020000);
      src.b[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B + n0 + 128 * h), 0, nb, 0x00020000);
    }
    src.none = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 8 * wave + 4 * q + (lane >> 4);           // k row inside the half tile
      const int c = (lane & 15) ^ (tkey(r) << 1);             // the chunk that lands at stored position lane & 15
      src.va[q] = (unsigned)r * (unsigned)g.lda * 2u + (unsigned)c * 16u;
      src.vb[q] = (unsigned)r * (unsigned)g.ldb * 2u + (unsigned)c * 16u;
    }
  }
  const unsigned ka0 = (unsigned)kbeg * (unsigned)g.lda * 2u, kas = 64u * (unsigned)g.lda * 2u;   // byte offset of K tile 0, per-tile step
  const unsigned kb0 = (unsigned)kbeg * (unsigned)g.ldb * 2u, kbs = 64u * (unsigned)g.ldb * 2u;

  // ---- transposed fragment reads: lane (i = 4 q + p, g) addresses row 8 g + q (+ 4 for the second read), columns c0 + 4 p ..
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  unsigned aaddr[2][4], baddr[2][2];         // [K tile parity][16-column tile]
  {
    const int q = li >> 2, p = li & 3;
    const int r1 = 8 * lg + q, tk = q | ((lg & 1) << 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = wr * 64 + 16 * i + 4 * p;
      aaddr[0][i] = lds0 + (unsigned)(r1 * 256 + (((col >> 3) ^ (tk << 1)) << 4) + ((col & 7) << 1));
      aaddr[1][i] = aaddr[0][i] + 65536u;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wc * 32 + 16 * j + 4 * p;
      baddr[0][j] = lds0 + (unsigned)(r1 * 256 + (((col >> 3) ^ (tk << 1)) << 4) + ((col & 7) << 1));
      baddr[1][j] = baddr[0][j] + 65536u;
    }
  }

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = g.colsum_a != nullptr;                  // wave-uniform
  const int cs_every = 4 * g.tiles_n;
  int cs_next = 4 * tn + wc;                                 // the next K tile whose A fragments this wave sums
  float accb[2][4];                                          // [A half][16-row tile]: column sums over this lane's k values
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[a][i] = 0.f;

  stage_half_k<0, 0>(lds, src, wave, ka0, kb0, true);
  stage_half_k<1, 1>(lds, src, wave, ka0, kb0, true);
  stage_half_k<2, 2>(lds, src, wave, ka0, kb0, true);
  stage_half_k<3, 3>(lds, src, wave, ka0, kb0, true);
  stage_half_k<4, 0>(lds, src, wave, ka0 + kas, kb0 + kbs, nt > 1);
  stage_half_k<5, 1>(lds, src, wave, ka0 + kas, kb0 + kbs, nt > 1);
  stage_half_k<6, 2>(lds, src, wave, ka0 + kas, kb0 + kbs, nt > 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { fa[i][0] = Mma<bf16>::zero(); fa[i][1] = Mma<bf16>::zero(); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { fb0[j][0] = fb0[j][1] = fb1[j][0] = fb1[j][1] = Mma<bf16>::zero(); }

#define MISSM_8P_MFMA(HA, HB, FB)                                                             \
  __builtin_amdgcn_s_setprio(1);                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
        acc[HA][HB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], FB[j][ks], acc[HA][HB][i][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
#define MISSM_8P_TR(ADDR, BASE) join_tr(lds_read_tr64<(BASE)>(ADDR), lds_read_tr64<(BASE) + 1024>(ADDR))
#define MISSM_8P_READ_A(PAR, SLOT_H)                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
    fa[i][0] = MISSM_8P_TR(aaddr[PAR][i], (SLOT_H) * 16384);                                  \
    fa[i][1] = MISSM_8P_TR(aaddr[PAR][i], (SLOT_H) * 16384 + 8192);                           \
  }
#define MISSM_8P_READ_B(PAR, SLOT_H, FB)                                                      \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
    FB[j][0] = MISSM_8P_TR(baddr[PAR][j], (SLOT_H) * 16384);                                  \
    FB[j][1] = MISSM_8P_TR(baddr[PAR][j], (SLOT_H) * 16384 + 8192);                           \
  }
#define MISSM_8P_FENCE_ALL()                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                         \
               : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fa[0][1]), "+v"(fa[1][1]), "+v"(fa[2][1]), \
                 "+v"(fa[3][1]), "+v"(fb0[0][0]), "+v"(fb0[1][0]), "+v"(fb0[0][1]), "+v"(fb0[1][1]), "+v"(fb1[0][0]),            \
                 "+v"(fb1[1][0]), "+v"(fb1[0][1]), "+v"(fb1[1][1]));                          \
  __builtin_amdgcn_sched_barrier(0);
#define MISSM_8P_COLSUM(HA)                                                                   \
  if (cs_now) {                                                                               \
    asm volatile("");      /* keeps this a real (scalar) branch */                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) accb[HA][i] = frag_sum(fa[i][1], frag_sum(fa[i][0], accb[HA][i]));  \
  }

  auto ktile = [&](auto par_tag, int t) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int S = 4 * PAR;
    const unsigned ka1 = ka0 + (unsigned)(t + 1) * kas, kb1 = kb0 + (unsigned)(t + 1) * kbs;
    const unsigned ka2 = ka1 + kas, kb2 = kb1 + kbs;
    const bool live1 = t + 1 < nt, live2 = t + 2 < nt;
    const bool cs_now = do_cs && t == cs_next;
    if (cs_now) cs_next += cs_every;
    // ---- phase 0: A half 0 x B half 0 (8 + 16 transposed reads; the first nine are back at lgkmcnt(15): all of B half 0)
    MISSM_8P_READ_B(PAR, 0, fb0)
    MISSM_8P_READ_A(PAR, 1)
    stage_half_k<4 * (1 - PAR) + 3, 3>(lds, src, wave, ka1, kb1, live1);
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(0, 0, fb0)
    MISSM_8P_COLSUM(0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: A half 0 x B half 1
    MISSM_8P_READ_B(PAR, 2, fb1)
    stage_half_k<S + 0, 0>(lds, src, wave, ka2, kb2, live2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(0, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: A half 1 x B half 1
    MISSM_8P_READ_A(PAR, 3)
    stage_half_k<S + 1, 1>(lds, src, wave, ka2, kb2, live2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(1, 1, fb1)
    MISSM_8P_COLSUM(1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A half 1 x B half 0
    stage_half_k<S + 2, 2>(lds, src, wave, ka2, kb2, live2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_MFMA(1, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  for (int t = 0; t < nt; t += 2) {
    ktile(std::integral_constant<int, 0>{}, t);
    ktile(std::integral_constant<int, 1>{}, t + 1);
  }
#undef MISSM_8P_MFMA
#undef MISSM_8P_TR
#undef MISSM_8P_READ_A
#undef MISSM_8P_READ_B
#undef MISSM_8P_FENCE_ALL
#undef MISSM_8P_COLSUM
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (do_cs) {                                // bias gradient of rows m0 + 128 ha + 64 wr + 16 i + li (this wave's K tiles)
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = accb[ha][i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int m = m0 + 128 * ha + 64 * wr + 16 * i + li;
        if (lg == 0 && m < g.M) atomicAdd(g.colsum_a + m, v);
      }
  }
  // park the partial tile: thread t, vector v = ((ha * 2 + hb) * 4 + i) * 2 + j  ->  float4 #(v * 512 + t)   (coalesced)
  f32x4* mine = reinterpret_cast<f32x4*>(g.ws) + ((size_t)gi * per_group + (size_t)split * ntiles + tix) * (32 * 512) + tid;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#ifdef MISSM_PARK_NT
          __builtin_nontemporal_store(acc[a][b][i][j], &mine[(((a * 2 + b) * 4 + i) * 2 + j) * 512]);
#else
          mine[(((a * 2 + b) * 4 + i) * 2 + j) * 512] = acc[a][b][i][j];
#endif
        }
}

// sums the K slices of gemm8p_tn_kernel in slice order.  One workgroup per (tile, sixteenth): thread t adds up vectors
// v = 2 e, 2 e + 1 of every slice and stores its 8 elements: rows m0 + 128 ha + 64 wr + 16 i + 4 lg + r, column
// n0 + 128 hb + 32 wc + 16 j + li  (16 consecutive columns per lane group: 64-byte runs).
// [r3] 16 workgroups per tile instead of 8 and the slice loop unrolled by four: the first version (27 tiles x 8 = 216 workgroups on
// 256 CUs, four loads in flight per thread) read the parked slices at 3.6 TB/s - latency-, not bandwidth-bound.  The sum over the
// slices still runs in slice order (the unrolled loads are added one after the other): results are bit-identical.
__global__ __launch_bounds__(512) void splitk_reduce8p_kernel(GemmArgs gall) {
  const int ntiles = gall.tiles_m * gall.tiles_n;
  const int gt = blockIdx.x >> 4, e = blockIdx.x & 15;
  const int gi = gt / ntiles, tix = gt - gi * ntiles;
  GemmArgs g = gall;
  if (gall.ngroups > 1) select_group(g, gall, gi);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, li = lane & 15, lg = lane >> 4;
  int tm, tn;
  if (gall.tn_order == 1) { tm = tix / g.tiles_n; tn = tix - tm * g.tiles_n; }
  else tile_of(xcd_remap(tix, ntiles), g.tiles_m, g.tiles_n, g.group_m, tm, tn);
  const size_t stride = (size_t)ntiles * (32 * 512);
  const f32x4* p = reinterpret_cast<const f32x4*>(g.ws) + ((size_t)gi * ntiles * g.splitk + tix) * (32 * 512) + (size_t)(2 * e) * 512 + tid;
  f32x4 a[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  int sl = 0;
  for (; sl + 4 <= g.splitk; sl += 4, p += 4 * stride) {
    f32x4 x[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) x[u][v] = __builtin_nontemporal_load(p + u * stride + v * 512);
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[0] += x[u][0]; a[1] += x[u][1]; }
  }
  for (; sl < g.splitk; ++sl, p += stride) {
#pragma unroll
    for (int v = 0; v < 2; ++v) a[v] += __builtin_nontemporal_load(p + v * 512);
  }
  float* C = static_cast<float*>(g.C);
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int vv = 2 * e + v, j = vv & 1, i = (vv >> 1) & 3, hb = (vv >> 3) & 1, ha = vv >> 4;
    const int col = tn * 256 + 128 * hb + 32 * wc + 16 * j + li;
    if (col >= g.N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 256 + 128 * ha + 64 * wr + 16 * i + 4 * lg + r;
      if (row >= g.M) continue;
      float* c = C + (size_t)row * g.ldc + col;
      float x = a[v][r] * g.alpha;
      if (g.accumulate) x += *c;
      *c = x;
    }
  }
}

}  // namespace missm
